"""Poor man's profiler for a box without perf: run a command, sample /proc/<pid>/task/*/stack of its threads every few ms for the first
seconds and print the most frequent kernel call chains.  python sample_kernel_stacks.py seconds cmd..."""
import collections, os, subprocess, sys, time
secs = float(sys.argv[1])
p = subprocess.Popen(sys.argv[2:])
cnt = collections.Counter(); top = collections.Counter()
t0 = time.time(); n = 0
time.sleep(float(os.environ.get("SKIP_S", "0.6")))
while time.time() - t0 < secs and p.poll() is None:
    try:
        for tid in os.listdir("/proc/%d/task" % p.pid):
            try:
                st = open("/proc/%d/task/%s/stack" % (p.pid, tid)).read().split("\n")
            except OSError:
                continue
            fr = [x.split("] ")[-1].split("+")[0] for x in st if x.strip()]
            if fr:
                cnt[" <- ".join(fr[:6])] += 1; top[fr[0]] += 1; n += 1
    except OSError:
        break
    time.sleep(0.002)
p.wait()
print("samples with a kernel stack:", n)
for k, v in top.most_common(12):
    print("%6d  %s" % (v, k))
print("--- chains")
for k, v in cnt.most_common(12):
    print("%6d  %s" % (v, k))
