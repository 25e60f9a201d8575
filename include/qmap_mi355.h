/* qmap_mi355.h -- C ABI of libqmap_mi355.so, the MI355X-native drop-in for the
 * `rapmap quasimap` hot path (RapMap v0.6.0).
 *
 * RapMap has no FFI/plugin registry: its callers compile three C++ template
 * entry points into themselves.  This header is the
 * batched, language-neutral face of the same three steps; each entry point
 * cites the reference interface it replaces (paths relative to the reference
 * tree).  Plain pointers and sizes only; no C++/torch types.  All functions
 * return 0 on success or a negative qm_status; none of them calls exit().
 *
 * One call of qm_map_pairs() == for every read pair i:
 *     SACollector::operator()(left)  ; SACollector::operator()(right)       include/SACollector.hpp:108-362
 *     hit_manager::hitsToMappingsSimple(..., PAIRED_END_LEFT / _RIGHT, ...)  src/HitManager.cpp:691-882
 *     utils::mergeLeftRightHits(...)                                         include/RapMapUtils.hpp:1185-1264
 *     + the per-pair bookkeeping of processReadsPairSA                       src/RapMapSAMapper.cpp:461-551,684-701
 * executed on the GPU: one 64-lane wavefront per read for the first two steps, one thread per pair for the merge.
 */
#ifndef QMAP_MI355_H
#define QMAP_MI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum qm_status {
  QM_OK = 0,
  QM_E_ARG = -1,         /* bad argument */
  QM_E_IO = -2,          /* index file missing / malformed */
  QM_E_NOGPU = -3,       /* no HIP device, or HIP call failed */
  QM_E_UNSUPPORTED = -4, /* option / index variant not implemented on the device path */
  QM_E_TOOLONG = -5,     /* a read is longer than QM_MAX_LONG_READ_LEN */
  QM_E_NOMEM = -6,
  QM_E_STATE = -7,       /* call order (e.g. fetch before map) */
  QM_E_FORMAT = -8       /* malformed FASTA/FASTQ input */
} qm_status;

/* Reads of up to QM_MAX_READ_LEN characters are mapped by the main kernels (64-character slot classes 2 / 3 / 4 / 8).  Longer
 * reads of a batch -- up to QM_MAX_LONG_READ_LEN -- are set aside by the main launch and mapped by a second, small launch of
 * 32-slot kernels (the reference takes any std::string, include/SACollector.hpp:108); with -s the same happens in the
 * collector pass and the alignment kernel runs in its long-image editions (with --dpBandwidth beyond 97 or negative: on blocks
 * in device memory, a ring that holds every column of a 2048-base alignment -- slow, and rare).  A read beyond QM_MAX_LONG_READ_LEN is
 * skipped, not mapped (round 5): it comes back without hits and on the skip list (qm_fetch_skipped), the batch goes on; the stage
 * entries that take caller-supplied lengths (qm_hits_to_mappings, ...) still answer QM_E_TOOLONG. */
#define QM_MAX_READ_LEN 512
#define QM_MAX_LONG_READ_LEN 2048

/* Mirrors MappingOpts (src/RapMapSAMapper.cpp:114-152) for the fields that
 * reach the hot path.  Defaults (qm_opts_default) == `rapmap quasimap` defaults
 * (src/RapMapSAMapper.cpp:992-1023,1113-1114). */
typedef struct qm_opts {
  int32_t sensitive;     /* 1 unless --noSensitive (disableNIP, SACollector.hpp:40)        */
  int32_t strict_check;  /* 1 unless --noStrictCheck (SACollector.hpp:61)                  */
  int32_t max_num_hits;  /* -m, 200                                                       */
  int32_t no_orphans;    /* --noOrphans                                                   */
  int32_t no_dovetail;   /* --noDovetail                                                  */
  int32_t fuzzy;         /* -f  mergeLeftRightHitsFuzzy, include/RapMapUtils.hpp:864-1183   */
  int32_t max_interval;  /* SACollector::setMaxInterval, 1000 (SACollector.hpp:54,77)     */
  int32_t sel_aln;       /* -s  selective alignment (src/RapMapSAMapper.cpp:554-667)        */
  double quasi_cov;      /* -z  (SACollector::setCoverageRequirement)                     */
  /* sub-options of -s (src/RapMapSAMapper.cpp:1011-1023,1135-1175); read only when sel_aln != 0 */
  int32_t hard_filter;        /* --hardFilter                                             */
  int32_t match_score;        /* --ma, 2                                                  */
  int32_t mismatch_penalty;   /* --mm, -4                                                 */
  int32_t gap_open;           /* --go, 4                                                  */
  int32_t gap_extend;         /* --ge, 2                                                  */
  int32_t dp_bandwidth;       /* --dpBandwidth, 15                                        */
  int32_t max_mmp_extension;  /* --maxMMPExtension, 7                                     */
  int32_t aln_policy;         /* 0 default, 1 --mimicBT2, 2 --mimicStrictBT2              */
  double min_score_fraction;  /* --minScoreFrac, 0.65                                     */
  double consensus_slack;     /* --consensusSlack, 0.2; a negative value -f gives MappingConfig::consensusFraction = f directly */
} qm_opts;

/* POD image of rapmap::utils::QuasiAlignment (include/RapMapUtils.hpp:399-502),
 * restricted to the fields that are defined on this path.  For orphan / single-end
 * hits the reference leaves matePos and mateLen uninitialised; here they are 0. */
typedef struct qm_hit {
  uint32_t tid;
  int32_t pos;
  int32_t mate_pos;
  uint32_t frag_len;
  uint32_t read_len;
  uint32_t mate_len;
  uint8_t fwd;
  uint8_t mate_is_fwd;
  uint8_t is_paired;
  uint8_t mate_status; /* MateStatus: 0 SINGLE_END 1 PE_LEFT 2 PE_RIGHT 3 PE_PAIRED (RapMapUtils.hpp:356-362) */
  int32_t aln_score;   /* alnScore_, 0 without -s */
} qm_hit;

/* rapmap::utils::HitCounters (include/RapMapUtils.hpp:208-216) + mapped units */
typedef struct qm_counters {
  uint64_t pe_hits, se_hits, tot_hits, num_reads, too_many_hits, mapped;
} qm_counters;

/* rapmap::utils::SAIntervalHit<OffsetT> (include/RapMapUtils.hpp:516-525) + which list it sits in.  begin / end are the device's
 * 32-bit offsets: for a BigSA index (OffsetT = int64_t in the reference) read them as uint32_t */
typedef struct qm_sa_interval_hit {
  int32_t begin, end;
  uint32_t len, query_pos;
  uint8_t query_rc;
  uint8_t list; /* 0 left-fwd, 1 left-rc, 2 right-fwd, 3 right-rc */
  uint16_t pad;
} qm_sa_interval_hit;

typedef struct qm_index_info {
  int32_t k;
  int32_t big_sa;
  int32_t perfect_hash;
  int32_t pad;
  int64_t text_len;
  int64_t n_txps;
  int64_t n_keys;
} qm_index_info;

typedef struct qm_index qm_index; /* host image of the on-disk quasi-index */
typedef struct qm_ctx qm_ctx;     /* one device context: index replica in HBM + work buffers */

const char* qm_last_error(void);
const char* qm_version(void);
int qm_opts_default(qm_opts* o);

/* RapMapSAIndex<int32_t, RegHashT>::load (src/RapMapSAIndex.cpp:97-176): reads
 * header.json, sa.bin, txpInfo.bin, rsd.bin and hash.bin -- or, for a perfect-hash (-p) index
 * (RapMapSAIndex<int32_t, PerfectHashT>, include/FrugalBooMap.hpp), hash_info.bph + hash_info.val --
 * of a "q5" index directory (files are mmap'd, not deserialised). */
int qm_index_open(const char* dir, qm_index** out);
int qm_index_close(qm_index* ix);
int qm_index_info_get(const qm_index* ix, qm_index_info* info);
const char* qm_index_txp_name(const qm_index* ix, int64_t tid); /* rmi.txpNames[tid] */
int64_t qm_index_txp_len(const qm_index* ix, int64_t tid);      /* rmi.txpLens[tid]  */
/* Read-only views of rmi.seq (the '$'-separated text) and rmi.txpOffsets (include/RapMapSAIndex.hpp:70-82);
 * valid until qm_index_close.  The offsets are UNSIGNED 32-bit values (read them as uint32_t for a BigSA index, whose text may
 * pass 2^31 characters; the int64 vectors of such an index are narrowed at open and its text must stay below 2^32 - 2). */
int qm_index_arrays(const qm_index* ix, const uint8_t** text, int64_t* text_len, const int32_t** txp_offsets,
                    int64_t* n_txps);
/* The other arrays of an open index as the library holds them, for inspection and cross-checks (profiles/r04/big_index_crosscheck.py
 * holds them against the oracle's independent numpy reader and the reference's own container): which = QM_RAW_SA -- the suffix
 * array as uint32_t[count] (a BigSA index's int64 entries narrowed at open), QM_RAW_HASH -- the dense hash's records, count x 16
 * bytes {uint64 k-mer word, uint32 lb, uint32 ub} in file order (NULL / 0 for a -p index), QM_RAW_COMPLETE_LENS -- uint32_t[count]. */
enum { QM_RAW_SA = 0, QM_RAW_HASH = 1, QM_RAW_COMPLETE_LENS = 2 };
int qm_index_raw(const qm_index* ix, int which, const void** data, int64_t* count);

/* Replicates the index into the HBM of `device_id` as flat SoA arrays and
 * allocates the per-context work buffers.  One ctx per GPU / per host thread (a context is not thread-safe); the
 * contexts of one index on one device share a single replica, which is freed with the last of them.  Destroy the
 * contexts before closing the index. */
int qm_ctx_create(const qm_index* ix, int device_id, qm_ctx** out);
/* flags: QM_CTX_PH_COMPACT -- a perfect-hash (-p) index keeps the reference's frugal structure on the device (re-blocked
 * BooPHF levels walked per lookup, include/BooPHF.hpp:971-1009, + one 16-byte record per k-mer: 1.3 GB for 80 M k-mers).
 * By default a -p index is expanded at load time into the same one-sector bucket table a dense index gets (8.6 GB), after
 * every record has been looked up through the BooPHF walk itself: identical answers, a single HBM round trip per lookup. */
/* QM_CTX_NO_PAIR_KERNEL (round 6) -- paired reads of up to 128 characters take qm_lean_kernel alone instead of the pair kernel / a part
 * of each kind (tests, A/B timing; the answers are the same).  QM_CTX_WIDE_READS -- build the wide extension table (reads of 129 .. 256
 * characters: 64 bytes per suffix-array entry) with the replica instead of inside the first call that has such reads. */
enum { QM_CTX_PH_COMPACT = 1, QM_CTX_NO_PAIR_KERNEL = 2, QM_CTX_WIDE_READS = 4 };
int qm_ctx_create_ex(const qm_index* ix, int device_id, uint32_t flags, qm_ctx** out);
int qm_ctx_destroy(qm_ctx* ctx);
int64_t qm_ctx_device_bytes(const qm_ctx* ctx);

/* Map n read pairs held in HOST memory.  seqX = concatenated read bytes (ASCII,
 * any case, N allowed), offX[n+1] = byte offsets.  Results stay in the context;
 * *n_hits receives the total number of hits.  counters may be NULL.
 * The buffers may be pageable: the characters are uploaded in chunks on a copy
 * stream and every chunk is mapped as soon as it has arrived, so transfer and
 * mapping overlap inside one call; nothing is in flight once the call returns. */
int qm_map_pairs(qm_ctx* ctx, const qm_opts* opts, int64_t n, const char* seq1, const int64_t* off1,
                 const char* seq2, const int64_t* off2, int64_t* n_hits, qm_counters* counters);
/* Same for single-end reads (processReadsSingleSA, src/RapMapSAMapper.cpp:232-250). */
int qm_map_reads(qm_ctx* ctx, const qm_opts* opts, int64_t n, const char* seq, const int64_t* off,
                 int64_t* n_hits, qm_counters* counters);
/* Same, with the inputs already resident in this device's memory (d_* are device
 * pointers; seq2/off2 NULL for single-end).  `max_read_len` bounds every read. */
int qm_map_device(qm_ctx* ctx, const qm_opts* opts, int64_t n, const void* d_seq1, const void* d_off1,
                  const void* d_seq2, const void* d_off2, int32_t max_read_len, int64_t* n_hits,
                  qm_counters* counters);
/* ---- 2-bit packed reads (replaces the per-record std::strings of src/FastxParser.cpp:229-328 on the way
 * to the device).  A host-buffer call moves 100 bytes per 100-bp read over PCIe; packed it moves 26.  The format keeps EVERY
 * character (the collector treats lower case, N, IUPAC codes and U differently, include/SACollector.hpp:176-192,498-536):
 *   read i's characters [off[i], off[i+1]) sit four to a byte -- character j of a group in bits 2j, 2j+1; A 0, C 1, G 2, T 3,
 *   upper case only -- from byte qm_packed_offset(off, i) = (off[i] >> 2) + i on (reads never share a byte: any number of
 *   threads may pack different reads of one batch side by side);
 *   every other character is an EXCEPTION {position in the concatenated sequence, character} and has code 0 in the packed bytes.
 * The device unpacks into the ASCII image the kernels read (one thread per group of four, then one per exception) and maps as
 * qm_map_pairs / qm_map_reads do: same results, bit for bit.  qm_pack_reads is the host-side packer (the ingest engine packs
 * in its copy tasks with the same routine); QM_E_ARG when the exceptions outgrow exc_cap (callers then send the plain characters). */
typedef struct qm_pack_exc { uint32_t pos; uint32_t ch; } qm_pack_exc;
int64_t qm_packed_offset(const int64_t* off, int64_t i);   /* (off[i] >> 2) + i: where read i's packed bytes start */
int64_t qm_packed_bytes(const int64_t* off, int64_t n);    /* (off[n] >> 2) + n + 8: bytes a packed batch of n reads takes */
int qm_pack_reads(const char* seq, const int64_t* off, int64_t n, uint8_t* packed, qm_pack_exc* exc, int64_t exc_cap, int64_t* n_exc);
int qm_map_pairs_packed(qm_ctx* ctx, const qm_opts* opts, int64_t n, const uint8_t* packed1, const int64_t* off1, const qm_pack_exc* exc1,
                        int64_t n_exc1, const uint8_t* packed2, const int64_t* off2, const qm_pack_exc* exc2, int64_t n_exc2,
                        int64_t* n_hits, qm_counters* counters);
int qm_map_reads_packed(qm_ctx* ctx, const qm_opts* opts, int64_t n, const uint8_t* packed, const int64_t* off, const qm_pack_exc* exc,
                        int64_t n_exc, int64_t* n_hits, qm_counters* counters);
/* Copy the results of the last map call to host memory:
 * hit_offsets[n+1] (exclusive prefix sum), hits[n_hits].  Large results come
 * down through pinned staging buffers and are placed by several host threads. */
int qm_fetch_hits(qm_ctx* ctx, int64_t* hit_offsets, qm_hit* hits);
/* The same into PAGE-LOCKED destination memory (hipHostMalloc / hipHostRegister): straight DMA, no staging, no host copy. */
int qm_fetch_hits_pinned(qm_ctx* ctx, int64_t* hit_offsets, qm_hit* hits);
/* Device pointers of the same arrays (valid until the next map call on ctx). */
int qm_result_device(qm_ctx* ctx, const void** d_hit_offsets, const void** d_hits);

/* ---- the reference's three entry points as calls of their own ---------------------------------------------------------
 * RapMap's callers (and Salmon) drive the path per read through three C++ entry points; include/qmap_rapmap_compat.hpp
 * gives them back with the reference's signatures and fills them from the batched calls below.  Each call works on a
 * whole batch; results stay in the context until the next call on it.
 *
 * (1) SACollector::operator()(read, saSearcher, hcInfo)        include/SACollector.hpp:108-362
 *     qm_collect_reads: n reads in, per read the SA-interval hits (HitCollectorInfo::fwdSAInts then rcSAInts,
 *     include/HitManager.hpp:59-72) and the call's return value (foundHit).  opts: sensitive, strict_check, quasi_cov,
 *     max_interval, and with sel_aln the chain-scoring collector (enableChainScoring / setMaxMMPExtension).
 *     qm_fetch_intervals: int_offsets[n+1], ints[int_offsets[n]] (ints = NULL: offsets only); the `list` field tells the
 *     strand (and, after qm_map_pairs / qm_map_pairs_stages on a context with qm_ctx_set_debug, the mate: a pair's four lists
 *     follow each other).  No cap on the number of intervals.
 * (2) hit_manager::hitsToMappingsSimple(rmi, mc, mateStatus, hcinfo, hits)   include/HitManager.hpp:130-135, src/HitManager.cpp:691-882
 *     qm_hits_to_mappings: per read its length and its SA-interval hits (forward-strand records first) in, the read's hit
 *     list out.  opts: fuzzy keeps both orientations of a transcript (what the -f merge consumes); sel_aln = MappingConfig
 *     {doChaining, considerMultiPos} with consensus_slack.  qm_fetch_read_lists: list_offsets[n+1], words[].
 *     List words without sel_aln: one per hit, ascending, tid << 33 | isRC << 32 | (uint32) hitPos.
 *     With sel_aln: per hit  tid | primaryRC << 32 | chainStatus << 33 | nP << 36,  then  (uint32) pos | nO << 32,
 *     then nP positions of the surviving orientation (QuasiAlignment::allPositions) and nO of the other
 *     (oppositeStrandPositions), each in the low 32 bits of a word.
 * (3) utils::mergeLeftRightHits / mergeLeftRightHitsFuzzy          include/RapMapUtils.hpp:1185-1264, :864-1183
 *     qm_merge_lists: per pair the two mates' lists (same word format), their lengths and -- for the fuzzy merge
 *     (opts.fuzzy or opts.sel_aln) -- leftMatches / rightMatches; jointHits come back through qm_fetch_hits, the
 *     tooManyHits out-parameter through qm_fetch_too_many (bit 0; bit 1: the mates shared a transcript), the HitCounters the merge bumps (peHits, seHits, tooManyHits)
 *     in `counters`.  Nothing of the caller's bookkeeping that follows the merge in processReadsPairSA is applied
 *     (src/RapMapSAMapper.cpp:534-551,684-701).  With sel_aln the hits' aln_score carries the chain statuses
 *     (left | right << 4; rapmap::utils::ChainStatus), not a score: the merge does not align.
 * qm_map_pairs_stages = (1)-(3) for a batch of pairs in one fused pass, every stage's output kept: intervals
 * (qm_fetch_intervals, four lists per pair), foundHit (qm_fetch_found, [2n]: left, right, left, ...), per-read lists
 * (qm_fetch_read_lists, [2n+1] offsets), merge results (qm_fetch_hits, qm_fetch_too_many). */
int qm_ctx_set_debug(qm_ctx* ctx, int keep_intervals);   /* fused calls keep the SA-interval hits for qm_fetch_intervals */
int qm_collect_reads(qm_ctx* ctx, const qm_opts* opts, int64_t n, const char* seq, const int64_t* off, int64_t* n_intervals);
int qm_fetch_intervals(qm_ctx* ctx, int64_t* int_offsets, qm_sa_interval_hit* ints, int64_t cap);
int qm_fetch_found(qm_ctx* ctx, uint8_t* found);
int qm_hits_to_mappings(qm_ctx* ctx, const qm_opts* opts, int64_t n, const int32_t* read_len, const int64_t* int_offsets,
                        const qm_sa_interval_hit* ints, int64_t* n_words);
int qm_fetch_read_lists(qm_ctx* ctx, int64_t* list_offsets, uint64_t* words, int64_t cap);
int qm_merge_lists(qm_ctx* ctx, const qm_opts* opts, int64_t n, const int64_t* loff_left, const uint64_t* words_left,
                   const int64_t* loff_right, const uint64_t* words_right, const uint8_t* found_left, const uint8_t* found_right,
                   const int32_t* len_left, const int32_t* len_right, int64_t* n_hits, qm_counters* counters);
int qm_fetch_too_many(qm_ctx* ctx, uint8_t* too_many);
int qm_map_pairs_stages(qm_ctx* ctx, const qm_opts* opts, int64_t n, const char* seq1, const int64_t* off1, const char* seq2,
                        const int64_t* off2, int64_t* n_hits, qm_counters* counters);
/* Round 6: the same pass for callers that do not look inside the collector's SA-interval records -- the reference's own caller hands its
 * HitCollectorInfo straight on to hitsToMappingsSimple (src/RapMapSAMapper.cpp:466-486) -- and / or keep their batches 2-bit packed:
 * QM_STAGES_NO_INTERVALS leaves the interval arrays of the stage view empty (every read's count 0; foundHit, lists, hits and tooMany as
 * before), which lets the pass run on the pair / lean stage-A kernels and brings half the bytes down; _packed takes the reads as
 * qm_map_pairs_packed does (26 bytes up per 100-bp read instead of 100).  stage_flags 0 = qm_map_pairs_stages. */
enum { QM_STAGES_NO_INTERVALS = 1 };
int qm_map_pairs_stages_ex(qm_ctx* ctx, const qm_opts* opts, int64_t n, const char* seq1, const int64_t* off1, const char* seq2,
                           const int64_t* off2, uint32_t stage_flags, int64_t* n_hits, qm_counters* counters);
int qm_map_pairs_stages_packed(qm_ctx* ctx, const qm_opts* opts, int64_t n, const uint8_t* packed1, const int64_t* off1, const qm_pack_exc* exc1,
                               int64_t n_exc1, const uint8_t* packed2, const int64_t* off2, const qm_pack_exc* exc2, int64_t n_exc2,
                               uint32_t stage_flags, int64_t* n_hits, qm_counters* counters);
/* Everything qm_map_pairs_stages kept, brought to the host in ONE go (round 4; what a caller of the reference's per-read call
 * surface needs for a parser chunk of ~10 000 pairs -- src/RapMapSAMapper.cpp:853 -- where five separate fetches, each a
 * synchronous copy of a bump-allocated device buffer, cost far more than the mapping itself).  The per-read interval records
 * and list words are compacted into CSR order ON THE DEVICE (scan + gather), then the eight arrays come down with asynchronous
 * copies on the context's stream and one synchronisation, laid out in an arena the CALLER owns -- page-locked memory
 * (qm_pinned_alloc) makes the copies true DMA; pageable memory works, slower.  qm_stage_bytes says how large the arena must be
 * for the last qm_map_pairs_stages call.  The view's pointers point into the arena.
 *   iv_off[n_reads + 1] / iv[]       per READ (2u = left mate of pair u, 2u + 1 = right): SA-interval hits, forward strand first
 *   found[n_reads]                   SACollector::operator()'s return value
 *   list_off[n_reads + 1] / words[]  per read: hitsToMappingsSimple's list ("List words" above)
 *   hit_off[n_units + 1] / hits[]    per pair: the merge's jointHits;  too_many[n_units]: bit 0 tooManyHits, bit 1 shared transcript */
typedef struct qm_stage_view {
  int64_t n_units, n_reads;
  const int64_t* iv_off; const qm_sa_interval_hit* iv; const uint8_t* found;
  const int64_t* list_off; const uint64_t* words;
  const int64_t* hit_off; const qm_hit* hits; const uint8_t* too_many;
} qm_stage_view;
int qm_stage_bytes(qm_ctx* ctx, int64_t* bytes);
int qm_fetch_stages(qm_ctx* ctx, void* arena, int64_t arena_bytes, qm_stage_view* view);
/* Page-locked host memory for callers that have no HIP runtime of their own to ask (input buffers of the qm_map_* calls,
 * arenas of qm_fetch_stages): hipHostMalloc / hipHostFree. */
void* qm_pinned_alloc(int64_t bytes);
void qm_pinned_free(void* p);

/* Timing of the dominant kernel of the last map call, measured with HIP events on
 * the context's stream (milliseconds); n_launches kernels were timed. */
int qm_last_kernel_ms(const qm_ctx* ctx, double* map_kernel_ms, double* total_ms);

/* Diagnostics of the last map call on ctx (tests, tuning): which = QM_STAT_RELAUNCHES -- how often stage A was run again
 * because the per-read hit lists outgrew their buffer (the buffer is grown and the batch redone; results are unaffected),
 * QM_STAT_LIST_WORDS -- capacity of that buffer in 8-byte words, QM_STAT_SLOW_READS -- reads that took the per-read
 * overflow path of -s (more suffixes than the wave's scratch holds). */
enum { QM_STAT_RELAUNCHES = 0, QM_STAT_LIST_WORDS = 1, QM_STAT_SLOW_READS = 2, QM_STAT_LEAN_READS = 3, QM_STAT_LEAN_DEFERRED = 4, QM_STAT_SKIPPED_READS = 5,
       QM_STAT_SEL_QUESTIONS = 6,    /* -s: alignments the reference would look at beyond PERFECT chains (one per hit and mate) ... */
       QM_STAT_KSW2_ALIGNMENTS = 7,  /* ... the ksw2 alignments the device ran for them (the others: alignment-cache hits, ungapped chains, answers known without ksw2) */
       QM_STAT_STRIP_ALIGNMENTS = 8, /* ... and the alignments answered by the exact strip DP instead (gapless path within q + 7 e of the best possible) */
       QM_STAT_PAIR_KERNEL_PAIRS = 9, /* pairs the pair kernel (both mates in one wavefront, merged there) was launched over; -1: not used.  An unsplit call only
                                         (a call mapped in parts reports QM_STAT_LEAN_READS / _DEFERRED summed over its parts and -1 here) */
       QM_STAT_PAIRS_MERGED = 10,    /* ... of which it merged itself (the others had a mate left to the general kernel, or the call kept lists) */
       /* why the reads of QM_STAT_LEAN_DEFERRED were left to the general kernel (they add up to it): */
       QM_STAT_DEFER_DIRTY = 11,     /* a character that is not A C G T (an N ...), or more characters than the kernel's lanes hold */
       QM_STAT_DEFER_HOMOPOLYMER = 12, /* a window of k equal bases (isHomoPolymer, include/Kmer.hpp:484-487) */
       QM_STAT_DEFER_WIDE = 13,      /* an SA interval wider than the kernel's lanes, more suffixes / intervals than its stash, a match beyond its extension table */
       QM_STAT_DEFER_BOTH_STRANDS = 14, /* k-mers of the other orientation seen on the way: the reference maps the other strand as well (SACollector.hpp:258,271) */
       QM_STAT_N_PASS_READS = 15 };  /* reads with N's that the N-aware second pass of stage A mapped (it runs when QM_NPASS_MIN reads or more -- 2 048 -- were left
                                        for a character outside A C G T; such k-mers are stepped over, SACollector.hpp:172-181,497-512).  They are not part of
                                        QM_STAT_LEAN_DEFERRED or QM_STAT_DEFER_DIRTY, which count what the general kernel took */
int qm_ctx_stat(const qm_ctx* ctx, int which, int64_t* value);

/* Reads of the last map call on ctx that were SKIPPED, not mapped (round 5; before, one such read failed the whole batch): a read
 * longer than QM_MAX_LONG_READ_LEN characters (code 1: the reference takes any std::string, include/SACollector.hpp:108 -- the
 * device kernels' longest slot class does not), a read whose SA-interval lists outgrow the per-wave scratch (code 2: only with
 * max_interval above its default of 1000, include/SACollector.hpp:54,77).  Such a read has an empty result -- no intervals, no
 * hits, foundHit false; its mate is mapped as usual -- and everything else in the batch is mapped as if it were not there.
 * *total = how many there were; the first min(total, cap, 4096) are listed: reads[i] = index of the read in the call (paired
 * calls: 2 * pair + mate), codes[i] as above.  Either array may be null. */
int qm_fetch_skipped(const qm_ctx* ctx, int64_t* reads, int32_t* codes, int64_t cap, int64_t* total);

/* `rapmap quasiindex [-p]` (src/RapMapSAIndexer.cpp:449-819): FASTA -> q5 index directory readable by
 * qm_index_open and by the reference (sa.bin, txpInfo.bin, rsd.bin and -- with perfect_hash --
 * hash_info.bph / hash_info.val come out byte-identical to the reference's).  A text of more than
 * 2^31 - 2 characters gets the reference's int64 form ("BigSA": true: 8-byte transcript starts, suffix
 * array entries and interval bounds, :682-683,711-722,743-765); the environment variable QM_FORCE_BIGSA=1
 * writes that form for any text (tests).  Host only. */
int qm_build_index(const char* fasta_path, const char* out_dir, int32_t k, int32_t no_clip_poly_a,
                   int32_t keep_duplicates, int32_t n_threads, int32_t perfect_hash);
/* ... with `-s / --headerSep` (src/RapMapSAIndexer.cpp:833-835,868,588): the transcript's name is its header up to the first
 * of these characters (NULL: space or tab, the default) */
int qm_build_index_ex(const char* fasta_path, const char* out_dir, int32_t k, int32_t no_clip_poly_a,
                      int32_t keep_duplicates, int32_t n_threads, int32_t perfect_hash, const char* header_sep);

/* XXH64 as the index builder uses it: KmerKeyHasher (include/RapMapUtils.hpp:236-238: XXH64 of the key's 8 bytes, seed 0) places
 * the records of hash.bin so that the reference's spp::sparse_hash_map finds them after unserialize(); also the key of the
 * duplicate-transcript filter.  (Diagnostics: tests hold it against the reference's src/xxhash.c.) */
uint64_t qm_xxh64(const void* data, uint64_t len, uint64_t seed);

/* ---- host-side callers of the path -------------------------------------------
 * Read ingest: replaces fastx_parser::FastxParser<ReadPair|ReadSeq> (include/FastxParser.hpp:62-66,
 * src/FastxParser.cpp:229-328: one kseq producer thread, per-record std::strings).  FASTA/FASTQ, plain or
 * gzip'd; path2 == NULL for single-end.  qm_reader_next hands out up to max_units records as packed batches
 * in exactly the form qm_map_pairs / qm_map_reads take; the pointers stay valid until the next call.
 * n_units == 0 means end of input.  Qualities are dropped, as the reference's parser does. */
typedef struct qm_reader qm_reader;
int qm_reader_open(const char* path1, const char* path2, int32_t n_threads, qm_reader** out);
int qm_reader_next(qm_reader* r, int64_t max_units, int64_t* n_units, const char** seq1, const int64_t** off1,
                   const char** names1, const int64_t** name_off1, const char** seq2, const int64_t** off2,
                   const char** names2, const int64_t** name_off2);
void qm_reader_close(qm_reader* r);
const char* qm_io_last_error(void);

/* FASTA/FASTQ files -> mapped batches, pipelined (the ingest side of the path; replaces the single kseq
 * producer + per-record std::strings of src/FastxParser.cpp:229-328, and the worker threads of spawnProcessReadsThreads,
 * src/RapMapSAMapper.cpp:752-799).  `reader_threads` workers parse the files chunk-parallel (plain files are cut at byte
 * offsets with record resync; .gz files are inflated by one thread per file, pipelined with the parsers) and pack batches
 * straight into pinned host slots, several batches in flight, while four device contexts per device (QM_STREAM_CTX_PER_DEVICE) (sharing that device's
 * index replica) upload, map and download the previous ones; with several devices consecutive batches go to different
 * devices (the static sharding of SURVEY.md section 8e inside one process).  The caller drains the batches IN INPUT ORDER,
 * whichever device mapped them.  Everything a batch points to -- reads, names, hit offsets, hits -- is pinned memory owned by
 * the stream and stays valid until the next qm_stream_next call.  n_units == 0: end of input.  path2 == NULL: single-end.
 * Counters come per batch; a run's HitCounters are their sum. */
typedef struct qm_stream qm_stream;
typedef struct qm_stream_batch {
  int64_t n_units;
  const char* seq1; const int64_t* off1; const char* names1; const int64_t* name_off1;
  const char* seq2; const int64_t* off2; const char* names2; const int64_t* name_off2;
  const int64_t* hit_offsets; const qm_hit* hits; int64_t n_hits;
  qm_counters counters;
  double gpu_ms;
  int32_t device;  /* the device that mapped this batch */
  int32_t pad;
} qm_stream_batch;
int qm_stream_open(const qm_index* ix, int device_id, uint32_t ctx_flags, const qm_opts* opts, const char* path1, const char* path2,
                   int64_t batch_units, int32_t reader_threads, qm_stream** out);
/* ... on the devices devices[0..n_devices).  stream_flags: QM_STREAM_NO_NAMES -- read names are not kept (names* / name_off*
 * of the batches are NULL): for callers that only want hits. */
#define QM_STREAM_NO_NAMES 1u
int qm_stream_open_ex(const qm_index* ix, const int32_t* devices, int32_t n_devices, uint32_t ctx_flags, const qm_opts* opts,
                      const char* path1, const char* path2, int64_t batch_units, int32_t reader_threads, uint32_t stream_flags,
                      qm_stream** out);
/* Pinning host memory is slow (about 5.5 GB/s on this platform, whatever the number of threads): a stream's slots -- a few hundred
 * MB -- cost as much as mapping millions of pairs.  qm_stream_reserve(bytes) pins a process-wide pool in the background and
 * returns at once; streams carve their slots out of it and hand them back when they close, so only the first reservation of a
 * process pays.  Call it as early as possible (the CLI does, before it uploads the index); 512 MB serve one single-device
 * stream of 2^18-pair batches of 2 x 100 bp reads.  Optional: without it a stream pins its own slots while it starts. */
int qm_stream_reserve(int64_t bytes);
int qm_stream_next(qm_stream* s, qm_stream_batch* batch);
void qm_stream_close(qm_stream* s);
/* seconds spent so far: [0] the ingest engine, open to its last batch packed (wall), [1] upload + kernels (summed over the
 * contexts), [2] download (summed), [3] the caller waiting in qm_stream_next, [4] qm_stream_open, [5] growing the pinned result
 * buffers; qm_stream_stats_ex(n <= 12) adds [6] open to the first batch packed, [7] parse tasks (CPU seconds over all workers),
 * [8] copy tasks, [9] inflate threads, [10] bytes parsed, [11] open to the last batch mapped and downloaded (wall) */
int qm_stream_stats(qm_stream* s, double* out6);
int qm_stream_stats_ex(qm_stream* s, double* out, int32_t n);
const char* qm_stream_last_error(void);

/* SAM text of a mapped batch, byte for byte what `rapmap quasimap -o` writes: header = writeSAMHeader
 * (include/RapMapUtils.hpp:97-115); records = writeAlignmentsToStream / writeUnalignedPairToStream
 * (src/RapMapUtils.cpp:137-196,198-311,313-588) with getSamFlags / adjustOverhang
 * (include/RapMapUtils.hpp:687-810).  seq2 == NULL: single-end records.  The text is malloc'd; release it
 * with qm_buf_free. */
int qm_sam_header(const qm_index* ix, char** out, int64_t* out_len);
int qm_sam_records(const qm_index* ix, int64_t n, const char* names1, const int64_t* name_off1, const char* seq1,
                   const int64_t* off1, const char* names2, const int64_t* name_off2, const char* seq2,
                   const int64_t* off2, const int64_t* hit_offsets, const qm_hit* hits, int32_t max_num_hits,
                   int32_t n_threads, char** out, int64_t* out_len);
/* the same text written straight to an open file descriptor (no copy through the caller); formatted by n_threads
 * workers, written in order by the calling thread */
int qm_sam_write(const qm_index* ix, int64_t n, const char* names1, const int64_t* name_off1, const char* seq1,
                 const int64_t* off1, const char* names2, const int64_t* name_off2, const char* seq2,
                 const int64_t* off2, const int64_t* hit_offsets, const qm_hit* hits, int32_t max_num_hits,
                 int32_t n_threads, int fd, int64_t* bytes_written);
/* A writer for a whole run: qm_sam_writer_put formats a batch with the writer's worker threads and returns; a thread of
 * the writer's own puts the text on the descriptor in batch order while the caller fetches and formats the next batch
 * (two batches of text are buffered; put blocks while both wait).  The batch's arrays are not referenced after put
 * returns.  A write error is reported by the next put and by close; close drains, reports the bytes written and
 * releases the writer (the descriptor stays the caller's). */
typedef struct qm_sam_writer qm_sam_writer;
int qm_sam_writer_open(const qm_index* ix, int fd, int32_t max_num_hits, int32_t n_threads, qm_sam_writer** out);
/* flags: QM_SAM_GZIP = `-x / --compressed` (src/RapMapSAMapper.cpp:832-833 wraps the output buffer in a zstr::ostream, i.e. a
 * zlib deflate stream): every formatter part becomes a gzip member of its own, compressed by the worker that formatted it,
 * written in order -- the concatenation is one valid .gz stream.  Bits 8-11: deflate level 1..9 (0: level 1).
 * qm_sam_writer_header puts the SAM header through the same queue (compressed like everything else). */
#define QM_SAM_GZIP 1u
int qm_sam_writer_open_ex(const qm_index* ix, int fd, int32_t max_num_hits, int32_t n_threads, uint32_t flags, qm_sam_writer** out);
int qm_sam_writer_header(qm_sam_writer* w);
int qm_sam_writer_put(qm_sam_writer* w, int64_t n, const char* names1, const int64_t* name_off1, const char* seq1,
                      const int64_t* off1, const char* names2, const int64_t* name_off2, const char* seq2,
                      const int64_t* off2, const int64_t* hit_offsets, const qm_hit* hits);
int qm_sam_writer_close(qm_sam_writer* w, int64_t* bytes_written);
void qm_buf_free(char* p);

#ifdef __cplusplus
}
#endif
#endif /* QMAP_MI355_H */
