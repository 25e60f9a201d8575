"""what pinning the stream's slots costs on this box: hipHostMalloc of one array, of a slot's eight arrays one after the other, in
parallel threads, as one arena; hipHostRegister of malloc'd memory.  python profiles/microbench/pin_cost.py"""
import ctypes as C, time, threading
hip = C.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostFree.argtypes = [C.c_void_p]
hip.hipSetDevice(0); hip.hipFree(None)
def alloc(nbytes, flags=2):
    p = C.c_void_p(); t = time.perf_counter(); rc = hip.hipHostMalloc(C.byref(p), nbytes, flags); dt = time.perf_counter() - t
    assert rc == 0, rc
    return p, dt
sizes = [27 << 20, 27 << 20, 5 << 20, 5 << 20, 2 << 20, 2 << 20, 2 << 20, 2 << 20]
for rep in range(2):
    p, dt = alloc(27 << 20); print("one 27 MB array: %.2f ms" % (dt * 1e3)); hip.hipHostFree(p)
    t = time.perf_counter(); ps = [alloc(s)[0] for s in sizes]; print("a slot's 8 arrays (72 MB), sequential: %.2f ms" % ((time.perf_counter() - t) * 1e3))
    for p in ps: hip.hipHostFree(p)
    res = [None] * 8
    def w(i): res[i] = alloc(sizes[i])
    t = time.perf_counter(); th = [threading.Thread(target=w, args=(i,)) for i in range(8)]; [x.start() for x in th]; [x.join() for x in th]
    print("the same in 8 threads: %.2f ms (per call %s)" % ((time.perf_counter() - t) * 1e3, ["%.1f" % (r[1] * 1e3) for r in res]))
    for r in res: hip.hipHostFree(r[0])
    p, dt = alloc(sum(sizes)); print("one 72 MB arena: %.2f ms" % (dt * 1e3)); hip.hipHostFree(p)
    p, dt = alloc(6 * sum(sizes)); print("one 432 MB arena (6 slots): %.2f ms" % (dt * 1e3)); hip.hipHostFree(p)
