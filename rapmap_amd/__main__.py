"""`python -m rapmap_amd quasiindex|quasimap ...` -- the reference's command line on the MI355X path.

Flag names, defaults and validation follow `rapmap quasiindex` (src/RapMapSAIndexer.cpp:821-927) and
`rapmap quasimap` (src/RapMapSAMapper.cpp:984-1189, validateOpts :911-954).  Options that the device path does
not implement (--recoverOrphans) are accepted by the parser and rejected, never silently ignored.  -c/--chaining is
accepted like the reference accepts it: on its own it changes nothing there either (MappingConfig::doChaining is set from
--selAln alone, src/RapMapSAMapper.cpp:182,410).
"""
import argparse
import os
import sys
import time

import numpy as np


def _quasiindex(argv):
    ap = argparse.ArgumentParser(prog="rapmap_amd quasiindex", description="RapMap Indexer (MI355X build)")
    ap.add_argument("-t", "--transcripts", required=True, help="The transcript file to be indexed")
    ap.add_argument("-i", "--index", required=True, help="The location where the index should be written")
    ap.add_argument("-k", "--klen", type=int, default=31, help="The length of k-mer to index (odd, <= 31)")
    ap.add_argument("-p", "--perfectHash", action="store_true", help="Use a perfect hash instead of dense hash (BooPHF)")
    ap.add_argument("-n", "--noClip", action="store_true", help="Don't clip poly-A tails from the ends of target sequences")
    ap.add_argument("--keepDuplicates", action="store_true", help="Retain and index exact sequence-level duplicates")
    ap.add_argument("-s", "--headerSep", default=None, help="Instead of a space or tab, break the header at the first occurrence of "
                    "(one of the characters of) this string, and name the transcript as the token before the first separator")
    ap.add_argument("-x", "--numThreads", type=int, default=4, help="Threads for the k-mer interval scan / perfect hash")
    a = ap.parse_args(argv)
    if a.klen % 2 == 0 or a.klen > 31 or a.klen < 1:
        sys.exit("K-mer length should be odd and <= 31 (RapMapSAIndexer.cpp:870-877)")
    import rapmap_amd as ra
    t = time.time()
    ra.build_index(a.transcripts, a.index, k=a.klen, no_clip_poly_a=a.noClip, keep_duplicates=a.keepDuplicates,
                   threads=a.numThreads, perfect_hash=a.perfectHash, header_sep=a.headerSep)
    print("[rapmap_amd] index written to %s (%.1fs)" % (a.index, time.time() - t), file=sys.stderr)


def _quasimap(argv):
    ap = argparse.ArgumentParser(prog="rapmap_amd quasimap", description="RapMap Mapper (MI355X build)")
    ap.add_argument("-i", "--index", required=True, help="The location of the quasiindex")
    ap.add_argument("-1", "--leftMates", default="", help="The location of the left paired-end reads")
    ap.add_argument("-2", "--rightMates", default="", help="The location of the right paired-end reads")
    ap.add_argument("-r", "--unmatedReads", default="", help="The location of single-end reads")
    ap.add_argument("-t", "--numThreads", type=int, default=16, help="host threads for read parsing and SAM formatting (the GPU does the mapping)")
    ap.add_argument("-m", "--maxNumHits", type=int, default=200, help="Reads mapping to more than this many loci are discarded")
    ap.add_argument("-o", "--output", default="", help="The output file (default: stdout)")
    ap.add_argument("-z", "--quasiCoverage", type=float, default=None)
    ap.add_argument("-n", "--noOutput", action="store_true", help="Don't write out any alignments (for speed testing purposes)")
    ap.add_argument("--noSensitive", action="store_true")
    ap.add_argument("--noStrictCheck", action="store_true")
    ap.add_argument("-f", "--fuzzyIntersection", action="store_true")
    ap.add_argument("-c", "--chaining", action="store_true")
    ap.add_argument("-x", "--compressed", action="store_true", help="Compress the output SAM file using zlib")
    ap.add_argument("-q", "--quiet", action="store_true")
    ap.add_argument("-u", "--writeUnmapped", action="store_true")
    ap.add_argument("--recoverOrphans", action="store_true")
    ap.add_argument("--noDovetail", action="store_true")
    ap.add_argument("--noOrphans", action="store_true")
    ap.add_argument("-s", "--selAln", action="store_true", help="Perform selective alignment to validate mapping hits")
    ap.add_argument("--go", type=int, default=4, help="[only with selAln]: gap open penalty")
    ap.add_argument("--ge", type=int, default=2, help="[only with selAln]: gap extend penalty")
    ap.add_argument("--mm", type=int, default=-4, help="[only with selAln]: mismatch penalty")
    ap.add_argument("--ma", type=int, default=2, help="[only with selAln]: match score")
    ap.add_argument("--dpBandwidth", type=int, default=15)
    ap.add_argument("--minScoreFrac", type=float, default=0.65)
    ap.add_argument("--consensusSlack", type=float, default=0.2)
    ap.add_argument("--hardFilter", action="store_true")
    ap.add_argument("--mimicBT2", action="store_true")
    ap.add_argument("--mimicStrictBT2", action="store_true")
    ap.add_argument("--maxMMPExtension", type=int, default=7)
    ap.add_argument("--device", type=int, default=0, help="GPU to use")
    ap.add_argument("--devices", default="", help="comma-separated GPUs to use, or 'all': the read batches are dealt round-robin to the devices "
                    "(each holds its own index replica), results come back in input order")
    ap.add_argument("--chunk", type=int, default=1 << 18, help="read pairs per GPU batch")
    a = ap.parse_args(argv)

    paired = bool(a.leftMates and a.rightMates)
    single = bool(a.unmatedReads)
    # validateOpts (src/RapMapSAMapper.cpp:911-954, :1179-1189)
    if paired == single:
        sys.exit("You must provide either paired-end (-1 and -2) or single-end (-r) reads, and not both")
    zset = a.quasiCoverage is not None                     # TCLAP isSet(): the flag was given, whatever its value
    if not zset:
        a.quasiCoverage = 0.0
    if not (0.0 <= a.quasiCoverage <= 1.0):
        sys.exit("quasiCoverage must be in [0,1]")
    if a.recoverOrphans:
        sys.exit("--recoverOrphans is not implemented on the MI355X path")
    if zset and a.noSensitive:                             # src/RapMapSAMapper.cpp:1178-1181
        print("The --quasiCoverage option is set to %g, but the --noSensitive flag was also set. The former forbids the later. "
              "Enabling sensitive mode." % a.quasiCoverage, file=sys.stderr)
        a.noSensitive = False
    if a.chaining and not (a.selAln or a.mimicBT2 or a.mimicStrictBT2):
        print("--chaining without --selAln does not change the mapping (as in the reference: doChaining follows --selAln)", file=sys.stderr)

    import rapmap_amd as ra
    opts = ra.default_opts(sensitive=0 if a.noSensitive else 1, strict_check=0 if a.noStrictCheck else 1,
                           max_num_hits=a.maxNumHits, no_orphans=int(a.noOrphans), no_dovetail=int(a.noDovetail),
                           fuzzy=int(a.fuzzyIntersection), sel_aln=int(a.selAln), quasi_cov=a.quasiCoverage)
    # --selAln and what it implies (src/RapMapSAMapper.cpp:1124-1175)
    if a.mimicBT2 and a.mimicStrictBT2:
        sys.exit("Cannot set --mimicBT2 and --mimicStrictBT2 simultaneously.  Please choose one")
    if a.mimicBT2 or a.mimicStrictBT2:
        opts.sel_aln = 1
    if opts.sel_aln:
        opts.gap_open, opts.gap_extend, opts.mismatch_penalty, opts.match_score = a.go, a.ge, a.mm, a.ma
        opts.dp_bandwidth, opts.min_score_fraction, opts.consensus_slack = a.dpBandwidth, a.minScoreFrac, a.consensusSlack
        opts.hard_filter, opts.max_mmp_extension = int(a.hardFilter), a.maxMMPExtension
        if a.mimicBT2 or a.mimicStrictBT2:
            opts.aln_policy = 2 if a.mimicStrictBT2 else 1
            opts.no_orphans = 1; opts.no_dovetail = 1; opts.consensus_slack = 0.35; opts.max_num_hits = 1000
        if a.mimicStrictBT2:
            opts.min_score_fraction = 0.8; opts.match_score = 1; opts.mismatch_penalty = 0; opts.gap_open = 25; opts.gap_extend = 25
    if a.devices:
        if a.devices == "all":
            import ctypes as _C
            nd = _C.c_int(0)
            hip = _C.CDLL("libamdhip64.so")
            if hip.hipGetDeviceCount(_C.byref(nd)) != 0 or nd.value <= 0:
                sys.exit("no HIP device visible")
            devices = list(range(nd.value))
        else:
            try:
                devices = [int(x) for x in a.devices.split(",") if x != ""]
            except ValueError:
                sys.exit("--devices takes a comma-separated list of device numbers, or 'all'")
            if not devices:
                sys.exit("--devices: no device given")
    else:
        devices = [a.device]
    # pinned memory for the stream's slots (every device of the run has its own), pinned in the background while the index is
    # opened and uploaded
    ra.reserve_stream_memory((1280 << 20) * len(devices))
    qi = ra.QuasiIndex(a.index)
    log = (lambda *x: None) if a.quiet else (lambda *x: print(*x, file=sys.stderr, flush=True))
    out = None
    direct_fd = None
    if not a.noOutput:
        out = open(a.output, "wb") if a.output else sys.stdout.buffer
        out.flush()
        try:
            direct_fd = out.fileno()
        except (OSError, ValueError, AttributeError):
            direct_fd = None                                 # a stdout that is not a descriptor: text goes through Python
    # header and records go from the library straight to the descriptor; -x: as gzip members compressed by the formatter's
    # workers (the reference wraps its output stream in a zlib compressor, src/RapMapSAMapper.cpp:832-833)
    writer = None
    if direct_fd is not None:
        writer = ra.SamWriter(qi, direct_fd, max_num_hits=opts.max_num_hits, threads=max(1, a.numThreads), gzip=bool(a.compressed))
        writer.header()
    elif out is not None:
        if a.compressed:
            import gzip
            out = gzip.GzipFile(fileobj=out, mode="wb")
        out.write(ra.sam_header_text(qi))
    tot = {"numReads": 0, "totHits": 0, "peHits": 0, "seHits": 0, "tooManyHits": 0}
    t0 = time.time()
    gpu_ms = 0.0
    nthr = max(1, a.numThreads)
    # ingest, mapping and result download are the library's pipelined stream (qm_stream_*): a reader thread parses the next
    # batch into pinned memory while two device contexts map the previous ones; SAM text (qm_sam_*) is formatted here from
    # the batch's pinned arrays.  -t sets the reader's / the formatter's worker count.
    if paired:
        files1, files2 = a.leftMates.split(","), a.rightMates.split(",")
        if len(files1) != len(files2):
            sys.exit("the number of left and right read files differs")
        pairs = list(zip(files1, files2))
    else:
        pairs = [(f, None) for f in a.unmatedReads.split(",")]
    # one context per device lives for the whole run: the index replica stays resident between the files of a multi-file run
    # (a stream's own contexts share it)
    keep = [ra.QuasiMapper(qi, d) for d in sorted(set(devices))]
    for f1, f2 in pairs:
        os.environ.setdefault("QM_INGEST_PIN", "1")   # a whole-machine job with one ingest engine: its workers on the NUMA node that holds the files' pages
        st = ra.MappedStream(qi, f1, f2, opts=opts, device=devices, batch_units=a.chunk, threads=nthr, names=out is not None)
        for b in st:
            gpu_ms += b.gpu_ms
            for kk in tot:
                tot[kk] += b.counters[kk]
            if out is not None:
                if writer is not None:
                    writer.put(b, b.hit_offsets, b.hits)       # formats now; the writer's thread writes while the next batch comes in
                else:
                    out.write(ra.sam_records_text(qi, b, b.hit_offsets, b.hits, max_num_hits=opts.max_num_hits, threads=nthr))
            if paired:
                log("saw %d reads : pe / read = %.4f : se / read = %.4f" % (
                    tot["numReads"], tot["peHits"] / max(1, tot["numReads"]), tot["seHits"] / max(1, tot["numReads"])))
        log("stream: " + ", ".join("%s %.3f" % kv for kv in st.stats().items()))
        st.close()
    for k_ in keep:
        k_.close()
    if writer is not None:
        writer.close()
    if out is not None and out is not sys.stdout.buffer:
        out.close()
    log("Done mapping reads.")
    log("In total saw %d reads." % tot["numReads"])
    log("Final # hits per read = %g" % (tot["totHits"] / max(1, tot["numReads"])))       # RapMapSAMapper.cpp:892-893
    log("Elapsed time: %.3fs (GPU mapping %.3fs)" % (time.time() - t0, gpu_ms / 1e3))


def main():
    if len(sys.argv) < 2 or sys.argv[1] in ("-h", "--help"):
        print("usage: python -m rapmap_amd {quasiindex,quasimap} [options]\n"
              "  quasiindex  build a suffix array-based (SA) index\n"
              "  quasimap    map reads using the SA-based index (on an MI355X)")
        return
    if sys.argv[1] in ("-v", "--version"):
        import rapmap_amd as ra
        print("rapmap_amd", ra.api.lib().qm_version().decode())
        return
    cmd = sys.argv[1]
    if cmd == "quasiindex":
        _quasiindex(sys.argv[2:])
    elif cmd == "quasimap":
        _quasimap(sys.argv[2:])
    else:
        sys.exit("the command %s is not yet implemented" % cmd)       # src/RapMap.cpp:86-94


if __name__ == "__main__":
    main()
