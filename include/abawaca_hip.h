/* abawaca_hip.h — C ABI of libabawaca_hip.so: the MI355X (gfx950) implementation of abawaca's
 * data-parallel hot path.  Plain pointers and sizes only; nothing throws across this boundary.
 *
 * Citations "ref:" are into the reference tree (CK7/abawaca, paths relative to its root) and name
 * the interface or code each entry point stands in for.  INTEGRATION.md shows the C++ that a
 * maintainer adds on the reference side to bind them.
 *
 * Conventions
 *   - every function returns ABW_OK (0) or a negative ABW_ERR_* code; abw_last_error() gives text.
 *   - "host" pointers are ordinary memory; "dev" pointers are HIP device memory on the context's GPU.
 *   - a context is used by one host thread at a time (ref: ClusterSeparator::separate() is called
 *     serially from main, src/abawaca.cpp:109-112); different contexts are independent.
 *   - work is enqueued on the context's stream; functions that return results in host memory
 *     synchronise that stream before returning, *_dev functions do not.
 */
#ifndef ABAWACA_HIP_H
#define ABAWACA_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ABW_OK               0
#define ABW_ERR_ARG         -1   /* bad argument (null pointer, size, ordering) */
#define ABW_ERR_HIP         -2   /* a HIP runtime call failed */
#define ABW_ERR_CAPACITY    -3   /* caller-provided output capacity too small */
#define ABW_ERR_LOWER_N     -4   /* lower-case 'n' in a sequence (ref: src/String.cpp:45-49 throws) */
#define ABW_ERR_STATE       -5   /* call order (e.g. abw_sep_node before abw_sep_load) */
#define ABW_ERR_NOMEM       -6
#define ABW_ERR_FORMAT      -7   /* text input the parser does not take (abw_lrn_parse) */

#define ABW_SIG_NDIMS  180       /* canonical 1..4-mers incl. "A" (ref: src/abawaca-build.cpp:75-100) */
#define ABW_SIG_NCOLS  179       /* columns written to abawaca.lrn: all but "A" (ref: :585,591,602) */
#define ABW_SIG_NMERS  340

typedef struct abw_ctx abw_ctx;

/* ---- context ------------------------------------------------------------------------------ */
/* device: HIP device ordinal.  Creates a stream owned by the context. */
/* number of HIP devices this process sees (for a caller that spreads contexts over them without including HIP itself) */
int  abw_device_count(int32_t* n);
int  abw_ctx_create(int device, abw_ctx** out);
void abw_ctx_destroy(abw_ctx* ctx);
/* Run on a caller-owned hipStream_t instead (NULL restores the context's own stream). */
int  abw_ctx_set_stream(abw_ctx* ctx, void* hip_stream);
int  abw_ctx_synchronize(abw_ctx* ctx);
const char* abw_last_error(const abw_ctx* ctx);
/* Library/ABI version, bumped on any signature change. */
int  abw_abi_version(void);

/* Measurement aid (no counterpart in the reference): when on, the library brackets each of its
 * kernel launches with HIP events recorded on the context's stream.  abw_prof_read synchronises
 * the stream and returns, for the kernel family `name` ("sig_count", "sig_segment", "sep_accum",
 * "sep_eval", "km_step", "km_update"), the number of launches and their summed duration since the
 * last abw_prof_reset.  Used by bench.py for the roofline figure.  An event pair costs the stream about 9 us (it is a
 * dependency of its own between two launches): on = 1 brackets every launch; on = n > 1 brackets every n-th launch of a
 * family, a sample -- abw_prof_read's count is then the number of launches that were timed. */
int  abw_prof_enable(abw_ctx* ctx, int on);
int  abw_prof_reset(abw_ctx* ctx);
int  abw_prof_read(abw_ctx* ctx, const char* name, uint64_t* launches, double* total_ms);

/* ---- stage 1: signature build (ref: src/abawaca-build.cpp:103-228, 595-607) ----------------- */

/* Names of the 180 dimensions in the reference's order ("A","C","AA",...), NUL-separated into buf
 * (needs >= 900 bytes).  ref: dimension_order, src/abawaca-build.cpp:69,82.  Host only. */
int abw_sig_dim_names(char* buf, uint64_t buf_len);

/* Segmentation of n_scaf scaffolds held as one concatenated ASCII blob (scaffold i =
 * ascii[scaf_off[i] .. scaf_off[i+1])).  Replaces the loop of Scaf::Scaf (ref: :198-228) together
 * with DNAString's upper-casing rule (ref: src/String.cpp:37-51; bytes are NOT modified, the
 * counting kernel upper-cases on the fly).  Writes, in scaffold order, for every segment its
 * scaffold index and 1-based inclusive [start,end] within the scaffold (ref: segment_coords, :216).
 * Host memory in, host memory out; multi-threaded on the host.  *n_seg receives the count even when
 * it exceeds seg_cap (then ABW_ERR_CAPACITY). */
int abw_sig_segment_host(const uint8_t* ascii, const uint64_t* scaf_off, uint64_t n_scaf, uint32_t window,
                         uint64_t seg_cap, uint32_t* seg_scaf, uint64_t* seg_start, uint64_t* seg_end,
                         uint64_t* n_seg);

/* Same on the GPU: d_ascii/d_scaf_off and all outputs are device pointers; *n_seg is host memory
 * (the call synchronises).  total_len = scaf_off[n_scaf]. */
int abw_sig_segment_dev(abw_ctx* ctx, const uint8_t* d_ascii, uint64_t total_len, const uint64_t* d_scaf_off,
                        uint64_t n_scaf, uint32_t window, uint64_t seg_cap, uint32_t* d_seg_scaf,
                        uint64_t* d_seg_start, uint64_t* d_seg_end, uint64_t* n_seg);

/* The same segmentation for callers that CANNOT ALLOCATE (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  No counterpart in the reference.  It calls no hipMalloc / hipFree / hipHostMalloc or any other
 * allocating routine and does not synchronise: the working arrays that abw_sig_segment_dev keeps in grow-only buffers
 * (block counts and their prefix sums, per-scaffold segment counts and offsets, scan temporaries) live in a file-scope
 * device array of the library, sized for at most ABW_SIG_STATIC_MAX_LEN bytes of sequence and ABW_SIG_STATIC_MAX_SCAF
 * scaffolds per call (about 11 MB, held by every GPU the library runs on until the process ends); larger inputs are
 * rejected with ABW_ERR_CAPACITY.  Inputs and outputs are the caller's device memory as in abw_sig_segment_dev, except
 * that nothing comes back to the host: d_result (device, 2 x uint64) receives [0] the number of segments of the input
 * (also when it exceeds seg_cap: then only the first seg_cap segments were written) and [1] non-zero iff the sequence
 * holds a lower-case 'n' (ABW_ERR_LOWER_N of the other entry points).  Asynchronous on the context's stream.
 * Every call on one GPU -- from any context -- works in that GPU's one array: calls that overlap on a GPU race. */
#define ABW_SIG_STATIC_MAX_LEN   (64ull << 20)
#define ABW_SIG_STATIC_MAX_SCAF  (256ull << 10)
int abw_sig_segment_static(abw_ctx* ctx, const uint8_t* d_ascii, uint64_t total_len, const uint64_t* d_scaf_off,
                           uint64_t n_scaf, uint32_t window, uint64_t seg_cap, uint32_t* d_seg_scaf,
                           uint64_t* d_seg_start, uint64_t* d_seg_end, uint64_t* d_result);

/* Counting + finalisation of n_seg segments: the body of Scaf_segment::Scaf_segment (ref: :103-174)
 * for every segment at once.  Segment s covers blob bytes
 *   [d_scaf_off[d_seg_scaf[s]] + d_seg_start[s] - 1,  d_scaf_off[d_seg_scaf[s]] + d_seg_end[s]).
 * Outputs (each may be NULL):
 *   d_dims   [n_seg][180] f64 — Scaf_segment::dimensions, bit-exact (ref: :171)
 *   d_milli  [n_seg][179] i32 — int(1000.0*x) of every column but "A" (ref: :603)
 *   d_counts [n_seg][340] u32 — raw k-mer window counts, k=1..4 blocks, base-4 A,C,G,T=0..3
 * Asynchronous on the context's stream. */
int abw_sig_count_dev(abw_ctx* ctx, const uint8_t* d_ascii, uint64_t total_len, const uint64_t* d_scaf_off,
                      const uint32_t* d_seg_scaf, const uint64_t* d_seg_start, const uint64_t* d_seg_end,
                      uint64_t n_seg, double* d_dims, int32_t* d_milli, uint32_t* d_counts);

/* Known-answer aid (no counterpart in the reference): the finalisation divides counts by totals with a hand-scheduled
 * fp64 sequence (reciprocal hoisted per total) that must equal the IEEE division of ref :171 bit for bit.  For the
 * divisors t = t_lo + i * t_step, i < count (all in [1, 2^31)) and every dividend n in [0, t] -- all of them when
 * t <= n_max, else the n_max + 1 values floor(j * t / n_max) -- *mismatches receives the number of (n, t) whose quotient
 * differs from the compiler's own fp64 division on the device, and checksums[i] (host, may be NULL) the sum mod 2^64 of
 * the quotients' bit patterns, which a host can compare with its own division.  tests/test_gpu_signature.py runs the
 * whole domain a segment below 65 536 bases can produce (t in [1, 65535], n in [0, t]). */
int abw_sig_selftest_division(abw_ctx* ctx, uint32_t t_lo, uint32_t t_step, uint32_t count, uint32_t n_max,
                              uint64_t* mismatches, uint64_t* checksums);

/* Whole stage with host buffers: segmentation on the GPU, H2D, counting, D2H.  This is what the
 * reference's FASTA loop (ref: src/abawaca-build.cpp:483-490) would call once per batch of
 * scaffolds instead of constructing Scaf objects.  dims/milli may be NULL. */
int abw_sig_build(abw_ctx* ctx, const uint8_t* ascii, const uint64_t* scaf_off, uint64_t n_scaf, uint32_t window,
                  uint64_t seg_cap, uint32_t* seg_scaf, uint64_t* seg_start, uint64_t* seg_end,
                  double* dims, int32_t* milli, uint64_t* n_seg);

/* ---- stage 2: cluster separator (ref: src/ClusterSeparator.cpp:25-135,
 *               src/ClusterSeparatorBySensitivitySpecificity.cpp:91-161, src/ClusterQuality.cpp:96-136)
 *
 * abw_sep_load puts the whole column store on the GPU once (ref: ClusterData loaded at
 * src/abawaca.cpp:86); abw_sep_node then runs ClusterSeparator::separate() for one node of the
 * recursion (ref: src/abawaca.cpp:109-112) given only the node's scaffold list.
 */
typedef struct {
    int32_t found;          /* some threshold of this dimension passed size check and gate */
    int32_t key;            /* its value (the caller's integer key; thousandths for abawaca.lrn) */
    double  sensitivity;    /* ref: ClusterStats::get_sensitivity, ...Specificity.h:75 */
    double  specificity;    /* ref: ...Specificity.h:74 */
} abw_dimbest;

typedef struct {
    int32_t split;          /* return value of separate() (ref: ClusterSeparator.cpp:57-135) */
    int32_t dimension;      /* get_separating_dimension(), 1-based; -1 when none/reset */
    int32_t key;            /* get_separating_value() as integer key; -1 when none/reset */
    int32_t legal;          /* is_best_separation_legal() of the global winner before the size check */
    double  sensitivity;    /* get_best_sensitivity() */
    double  specificity;    /* get_best_specificity() */
    int64_t ndps1, ndps2;   /* final cluster1/cluster2 sizes after dps followed their scaffolds */
} abw_split;

/* keys: host [D][N] int32 column-major (dimension-major), an order-preserving integer image of the
 * .lrn values — thousandths for abawaca-build output (ref: src/abawaca-build.cpp:603 writes
 * 3 decimals; src/ClusterData.cpp:159 reads them back).  dp2scaf: host [N], 0-based scaffold of
 * each dp; dps of one scaffold must be contiguous and scaffolds ascending, which is how
 * ScafDpData numbers them (ref: src/ScafDpData.cpp:91-99).  Every scaffold needs >= 2 dps
 * (ref: :92-93 drops the others).  scaf_len: [S] sequence lengths (ref: ClusterQuality.cpp:111-115).
 * scg_off [S+1] / scg_ids: CSR scaffold -> SCG id in [0, n_scg) (ref: SCGdb::scaf2scgs,
 * src/SCGdb.h:25); both may be NULL when n_scg == 0. */
int abw_sep_load(abw_ctx* ctx, const int32_t* keys, int32_t D, int64_t N, const int32_t* dp2scaf,
                 const int64_t* scaf_len, int32_t S, const int32_t* scg_off, const int32_t* scg_ids,
                 int32_t n_scg);
/* The same with the keys already on the device (d_keys: [D][N] int32 on ctx's device, e.g. what a loader left there; it is
 * copied, the caller keeps its buffer); every other argument on the host as above.  Both calls build the table on the
 * device: the columns are classified, the wide ones (keys outside [0, 1024): the coverage columns of ref
 * src/abawaca-build.cpp:603) sorted and ranked by kernels.  The same rows of the reference as abw_sep_load replace
 * (src/ClusterData.cpp:117-163, src/Dimension.cpp). */
int abw_sep_load_dev(abw_ctx* ctx, const int32_t* d_keys, int32_t D, int64_t N, const int32_t* dp2scaf,
                     const int64_t* scaf_len, int32_t S, const int32_t* scg_off, const int32_t* scg_ids,
                     int32_t n_scg);

/* One node.  scaf_ids: host, the node's scaffolds (any order, no duplicates); NULL = all S
 * scaffolds (the root, ref: init_cluster, src/abawaca.cpp:242-251).  min_ndps is
 * cluster_ndps_threshold (100; ref: ClusterSeparator.h:72).
 * Outputs (host; per_dim/scaf_side/dp_raw_side may be NULL):
 *   per_dim     [D]      per-dimension winner (ref: best_dimension_separation, ...Specificity.cpp:115)
 *   scaf_side   [n_scaf] 1/2 = scaffold assigned to cluster1/cluster2 (ref: ClusterSeparator.cpp:94-101),
 *                        in the order of scaf_ids (ascending scaffold id when scaf_ids is NULL); 0 if !split
 *   dp_raw_side [n dps]  1/2 = raw_dps_cluster1/2 (ref: :87-90) for the node's dps, scaffold by scaffold
 *                        in the order of scaf_ids, dps ascending within a scaffold; 0 if no legal separation */
int abw_sep_node(abw_ctx* ctx, const int32_t* scaf_ids, int32_t n_scaf, int32_t min_ndps,
                 abw_split* out, abw_dimbest* per_dim, uint8_t* scaf_side, uint8_t* dp_raw_side);

/* The second half of separate() for a node whose winner is known (ref: src/ClusterSeparator.cpp:35-53 reconstruct_best_clusters,
 * :87-132 raw sets, scaffold assignment, datapoints following their scaffolds, size check -- given best_separation): for a
 * caller that has run the whole recursion in one call (abw_sep_tree below) and needs a node's raw sets and scaffold sides to
 * rebuild the reference's Cluster objects.  dimension (1-based) / key: the node's legal winner as abw_sep_tree reported it.
 * out: split, ndps1 / ndps2 and the winner handed in (sensitivity / specificity are not recomputed: 0); scaf_side / dp_raw_side
 * as abw_sep_node.  After abw_sep_node on the same node with the same winner the outputs are identical. */
int abw_sep_node_assign(abw_ctx* ctx, const int32_t* scaf_ids, int32_t n_scaf, int32_t dimension, int32_t key, int32_t min_ndps,
                        abw_split* out, uint8_t* scaf_side, uint8_t* dp_raw_side);

/* ---- the whole recursion in one call: the work-list loop of ref src/abawaca.cpp:95-197 around separate(), on the table of
 *      abw_sep_load.  The reference takes the node with the smallest id from its std::map and gives the children of a split
 *      the next two ids (ref :98-100,155-194), i.e. it visits the nodes breadth first: abw_sep_tree evaluates every LEVEL of
 *      that order with one set of kernel launches (per node abw_sep_node costs about 14 launches and a synchronisation, and
 *      most nodes of a deep recursion are small).  Node ids, winners (dimension, key, sensitivity / specificity doubles),
 *      children and their sizes are those of calling abw_sep_node node by node in id order.
 *   nodes_out [node_cap] : the visited nodes in id order (id 1 = all scaffolds); *n_nodes = their number, also when it
 *                          exceeds node_cap (then ABW_ERR_CAPACITY; the first node_cap are written)
 *   max_nodes            : stop after that many nodes (<= 0: no limit), as a driver that cuts the work list would
 *   scaf2cluster [S]     : host, may be NULL: the id of the terminal node every scaffold ended in (0: its node was not visited)
 * dimension is 1-based, -1 when the node found no threshold or was reset by the final size check (ref
 * src/ClusterSeparator.cpp:125-132); the raw datapoint sets of a split are {value <= key} and its complement in the
 * winning dimension, cluster1 being the smaller (ref :49-53), should a caller need them. */
typedef struct {
    int32_t id;             /* 1.. in the reference's work-list order */
    int32_t n_scaf;         /* scaffolds of the node */
    int64_t ndps;           /* datapoints of the node */
    int32_t split;          /* separate() returned true */
    int32_t dimension;      /* as abw_split */
    int32_t key;
    int32_t legal;
    double  sensitivity, specificity;
    int32_t child1, child2; /* ids of the children (0 when !split) */
    int64_t ndps1, ndps2;   /* their sizes */
} abw_tree_node;
int abw_sep_tree(abw_ctx* ctx, int32_t min_ndps, int64_t max_nodes, abw_tree_node* nodes_out, int64_t node_cap,
                 int64_t* n_nodes, int32_t* scaf2cluster);

/* ---- the same recursion level by level, for a table whose DIMENSIONS are spread over several contexts / processes / GPUs.
 *      The sweep of a dimension needs nothing from the other dimensions (ref: src/ClusterSeparator.cpp:70-78 runs them as
 *      independent threads and compares their results afterwards, :11-16), so every context loads ALL scaffolds but only its
 *      own columns (abw_sep_load with D = its number of columns) and the contexts meet twice per LEVEL with a few bytes:
 *        eval    : this context's best threshold of every node of the level (best[k]; *n_level = 0: the recursion is over;
 *                  ABW_ERR_CAPACITY with *n_level set when cap is too small -- call again)
 *        -- the caller fills best[k].global_dimension (1-based dimension in the whole table), gathers the records of all
 *           contexts and picks every node's winner as the reference does: larger sensitivity x specificity, then the smaller
 *           dimension (abw_sep_tree_merge) --
 *        assign  : winners[k] = the node's winner; .dimension >= 0 only in the record handed back to the context that owns
 *                  it, -1 elsewhere.  The owner assigns the node's scaffolds: *d_side = device array [*side_count = S] of
 *                  bytes, 1 / 2 for the scaffolds this context decided, 0 elsewhere.  Asynchronous.
 *        -- the caller combines the side arrays of all contexts (MAX or SUM: every scaffold is decided by at most one) --
 *        advance : children sizes, the size check (ref: src/ClusterSeparator.cpp:125-132), next level, partition.
 *        result  : the nodes and the bins, as abw_sep_tree returns them (dimension = global_dimension).
 *      Every context walks the same tree.  abw_sep_tree is this loop on one context.  Between begin and result no other entry
 *      point may run on the context (the run lives in its scratch buffers; one that does is noticed: the next level call returns
 *      ABW_ERR_STATE and the recursion has to be begun again).  Nothing of this exists in the reference's
 *      single process; the partition follows its thread-per-dimension loop. */
typedef struct abw_tree_winner {
    int32_t found;              /* 0: no threshold passed the gates */
    int32_t dimension;          /* 0-based column of THIS context's table; -1: not this context's */
    int32_t key;
    int32_t legal;
    double  sensitivity, specificity;
    int32_t global_dimension;   /* 1-based dimension of the whole table (eval: dimension + 1; the caller translates) */
    int32_t reserved;
} abw_tree_winner;
int abw_sep_tree_begin(abw_ctx* ctx, int32_t min_ndps, int64_t max_nodes);
int abw_sep_tree_eval(abw_ctx* ctx, abw_tree_winner* best, int64_t cap, int64_t* n_level);
int abw_sep_tree_assign(abw_ctx* ctx, const abw_tree_winner* winners, int64_t n, uint8_t** d_side, int64_t* side_count);
int abw_sep_tree_advance(abw_ctx* ctx);
int abw_sep_tree_result(abw_ctx* ctx, abw_tree_node* nodes_out, int64_t node_cap, int64_t* n_nodes, int32_t* scaf2cluster);
/* host only: all = [n_ctx][n] records of eval (global_dimension filled in); out[k] = the winner, owner[k] = its context (-1: none) */
int abw_sep_tree_merge(const abw_tree_winner* all, int32_t n_ctx, int64_t n, abw_tree_winner* out, int32_t* owner);


/* ---- the separator with the SCAFFOLDS sharded over several contexts / processes / GPUs (SURVEY §8(e):
 *      "scaffold-sharded" full recursive binning; the reference itself is one process with a thread per
 *      dimension, ref: src/ClusterSeparator.cpp:72-78).  A shard holds whole scaffolds (a node of the
 *      recursion always does, ref: src/ClusterSeparator.cpp:104-122).  Every flip value t_s is local to the
 *      shard that owns scaffold s and every quantity the sweep looks at is a count, so the histograms of the
 *      shards simply add; the SCG gate needs min / max of the per-SCG flip values.  The library runs each
 *      shard's local part, the CALLER combines three device buffers in between (abawaca_amd/multi.py: RCCL /
 *      gloo all-reduce).  All counts are integers: the result is bit-identical to abw_sep_node on the
 *      unsharded table for any number of shards.
 *
 *   load   : as abw_sep_load for the shard's own scaffolds (local ids 0..S-1), plus the bin layout every shard
 *            must share: dict_off [D+1] / dict_vals — for dimension d an EMPTY range means "narrow": all keys
 *            of all shards lie in [0,1024) (every signature column: thousandths in [0,1000]); otherwise the
 *            strictly ascending distinct keys of column d over ALL shards (coverage columns).
 *   accum  : the node's scaffolds on this shard (local ids; n_scaf < 0 = all of them; n_scaf = 0 = none, the
 *            shard still takes part in the exchange) -> ex: d_hist (hist_count int64 words: integer SUM over
 *            shards), d_scg_lo (MIN) and d_scg_hi (MAX), scg_count int32 each.  Asynchronous.
 *   eval   : on the combined buffers: every shard finds the same per-dimension winners and global winner
 *            (out->dimension/key/sensitivity/specificity/legal as abw_sep_node), then assigns ITS scaffolds
 *            (scaf_side, dp_raw_side as abw_sep_node, local order).  out->ndps1/ndps2 = dps of this shard's
 *            scaffolds sent to cluster1/cluster2; out->split stays 0.
 *   decide : host only.  The caller sums ndps1/ndps2 over the shards; this applies the final size check
 *            (ref: src/ClusterSeparator.cpp:125-132) and sets out->split / resets out as separate() would.
 *            When it resets (legal winner, a side too small) the caller discards scaf_side.
 * Needs the node's dps over all shards to stay below 2^31. */
typedef struct abw_sep_exchange {
    int64_t* d_hist;    uint64_t hist_count;
    int32_t* d_scg_lo;  int32_t* d_scg_hi;  uint64_t scg_count;
} abw_sep_exchange;
int abw_sep_shard_load(abw_ctx* ctx, const int32_t* keys, int32_t D, int64_t N, const int32_t* dp2scaf,
                       const int64_t* scaf_len, int32_t S, const int32_t* scg_off, const int32_t* scg_ids,
                       int32_t n_scg, const int32_t* dict_vals, const int64_t* dict_off);
int abw_sep_shard_accum(abw_ctx* ctx, const int32_t* scaf_ids, int32_t n_scaf, abw_sep_exchange* ex);
int abw_sep_shard_eval(abw_ctx* ctx, int32_t min_ndps, abw_split* out, abw_dimbest* per_dim,
                       uint8_t* scaf_side, uint8_t* dp_raw_side);
int abw_sep_shard_decide(abw_split* out, int64_t ndps1_total, int64_t ndps2_total, int32_t min_ndps);
/* The exchange between accum and eval for a caller that holds ALL shards in one process (the reference's driver is one
 * process: one context per GPU of the node, or several contexts on one GPU): the three buffers of every context become
 * their SUM / MIN / MAX over the contexts, by peer copies and an add kernel (direct reduce-scatter + all-gather; every
 * owner on its own stream).  What ncclAllReduce does between processes; the reference has no
 * counterpart (it is single-threaded per dimension, ref: src/ClusterSeparator.cpp:70-78).  Every context must have an
 * accumulated node (abw_sep_shard_accum) under the same bin layout.  Returns when all contexts hold the result. */
int abw_sep_shard_combine(abw_ctx* const* ctxs, int32_t n_ctx);

/* ---- ... and with the SCAFFOLDS sharded for the signature columns (SURVEY section 8(e) "separator sweep (B)", BASELINE configs[4]:
 *      "scaffold-sharded at 2/4/8 GPUs"; the reference is one process, ref: src/ClusterSeparator.cpp:70-78).  The coverage columns
 *      (few, about as many distinct keys as datapoints) are spread over the ranks by dimension and walked with the level calls
 *      above over all scaffolds; the signature columns (many, keys below 1024) are cut by SCAFFOLD: rank r loads a contiguous
 *      range of the scaffolds with those columns only (abw_sep_load on a second context) and follows the walk level by level:
 *        abw_sep_tree_level (walking context) : the level's nodes, and the device arrays with every scaffold's node / side
 *        accum  : the shard's scaffolds of the nodes that are the SMALLER child of their parent (the root; a node without
 *                 sibling) -> histogram slices; SCG lists of all nodes from the shard's carriers.  ex: the slices packed side by
 *                 side (int64 words; integer SUM over the shards) and the lists (MIN of scg_lo, MAX of scg_hi)
 *        eval   : after the exchange: bigger child = parent - smaller child on every shard, every threshold of every signature
 *                 column -> best[k] (identical on all shards; key = the column's value, global_dimension = dimension + 1 in
 *                 THIS table: the caller translates it)
 *        assign : winners[k].dimension >= 0 where this table's column won node k: the shard's scaffolds there get their sides
 *                 in d_side (the walking context's array; the ranks then combine it by MAX as among themselves)
 *      Per level the ranks exchange 20 B x signature bins x accumulated nodes (about 0.1 MB a node for abawaca-build's tables),
 *      the SCG lists, one record per node and one byte per scaffold.  Integers only: the tree is abw_sep_tree's bit for bit. */
/* The shards add their histogram slices bin by bin, so every shard's table needs the same number of bins in every column:
 * before abw_sep_load of a shard, bins[d] = the largest key of column d over ALL shards + 1 (one MAX over the ranks). */
int abw_sep_narrow_bins(abw_ctx* ctx, const int32_t* bins, int32_t D);
typedef struct abw_tree_level_node { int32_t id; int32_t n_scaf; int64_t ndps; int32_t parent; int32_t reserved; } abw_tree_level_node;
typedef struct abw_tree_level { int64_t n_nodes; int64_t n_scaf; const int32_t* d_scaf_slot; uint8_t* d_side; } abw_tree_level;
int abw_sep_tree_level(abw_ctx* ctx, abw_tree_level* info, abw_tree_level_node* nodes_out, int64_t cap);
int abw_sep_stree_begin(abw_ctx* ctx, int32_t scaf_first);
int abw_sep_stree_accum(abw_ctx* ctx, const abw_tree_level_node* nodes, int64_t n_nodes, const int32_t* d_scaf_slot, abw_sep_exchange* ex);
int abw_sep_stree_eval(abw_ctx* ctx, int32_t min_ndps, abw_tree_winner* best, int64_t cap);
int abw_sep_stree_assign(abw_ctx* ctx, const abw_tree_winner* winners, int64_t n, const int32_t* d_scaf_slot, uint8_t* d_side);

/* Device pointer to the resident column store [D][N] int32 (for abw_km_features_from_keys). */
const int32_t* abw_sep_dev_keys(const abw_ctx* ctx);

/* ---- next row N1: per-segment read coverage (ref: src/abawaca-build.cpp:231-244 Scaf::add_mapped_read,
 *      :177-185 Scaf_segment::add_mapped_read; the SAM filter of :546 and the parser stay with the caller) ----
 * Segments as produced by abw_sig_segment_* (grouped by scaffold, ascending start; 1-based inclusive
 * coordinates).  Read i maps to scaffold read_scaf[i] (an id >= n_scaf is ignored, like a reference name
 * the reference does not know, ref :550-552) at the 0-based position read_pos[i] (= SAM POS - 1, ref
 * src/ReadMapping.cpp:41) with read_len[i] bases, in SAM order.  cov[g] += the read's overlap with
 * segment g divided by the segment's length, added in read order, exactly as the reference's fp64
 * accumulator; call once per chunk of a SAM file (in order) with the same cov to continue the sums, and
 * with a zeroed cov for the next sample.  At most 2^32-3 segments, scaffolds and reads (and 2^32-2 contributions) per call;
 * a segment is shorter than 2^32 bases. */
int abw_cov_accumulate(abw_ctx* ctx, const uint32_t* seg_scaf, const uint64_t* seg_start, const uint64_t* seg_end,
                       uint64_t n_seg, uint64_t n_scaf, const uint32_t* read_scaf, const uint64_t* read_pos,
                       const uint32_t* read_len, uint64_t n_reads, double* cov);
/* Same with every pointer on the device; asynchronous except for one or two internal read-backs of a few words. */
int abw_cov_accumulate_dev(abw_ctx* ctx, const uint32_t* d_seg_scaf, const uint64_t* d_seg_start, const uint64_t* d_seg_end,
                           uint64_t n_seg, uint64_t n_scaf, const uint32_t* d_read_scaf, const uint64_t* d_read_pos,
                           const uint32_t* d_read_len, uint64_t n_reads, double* d_cov);

/* ---- next row N1, the reader side: SAM text -> the arrays abw_cov_accumulate takes ------------------------------------
 * What the reference does per line of a SAM file before Scaf::add_mapped_read sees it: SAMReader::next_mapping (ref
 * src/ReadMappingReader.cpp:80-116: '@' lines skipped, line ends chomped), ReadMapping's constructor (ref
 * src/ReadMapping.cpp:23-72: FLAG, RNAME, POS - 1, the read length = SEQ's), determine_snps (ref :78-185: SNPs counted from
 * the MD:Z and CIGAR strings), the filter of ref src/abawaca-build.cpp:546 (unmapped, more than max_snps SNPs -- the
 * reference's constant is 15 --, secondary) and the name lookup of ref :549-551.
 * text: the file, or a part of it cut at line ends.  names / name_off[n_scaf + 1]: the scaffold names, concatenated (host).
 * Outputs, each of capacity cap or NULL: read_* = the records that pass the filter, in file order -- scaffold index (n_scaf
 * for a name that is none of the scaffolds; abw_cov_accumulate ignores those), 0-based position (POS 0 wraps as the
 * reference's size_t does), length -- ready for abw_cov_accumulate[_dev]; rec_* = the same and FLAG and the SNP count of
 * EVERY record.  counts: lines, records (lines that do not start with '@') and records kept.
 * A line the reference would not survive (it throws, or indexes a field that is not there), or one of the few things this
 * parser is stricter about (FLAG and POS must be 1..9 digits; "MD:Z:" inside a field; numbers of more than 9 digits),
 * fails the call with ABW_ERR_FORMAT (ABW_ERR_LOWER_N for a lower-case n in SEQ) and abw_last_error names the line: fall
 * back to the reference's reader.  More records than cap: ABW_ERR_CAPACITY with counts filled.  The read-pair links of the
 * `-c` sample (ref src/abawaca-build.cpp:531-545) are not made here. */
typedef struct abw_sam_counts { uint64_t lines, records, kept; } abw_sam_counts;
int abw_sam_parse(abw_ctx* ctx, const uint8_t* text, uint64_t len, const uint8_t* names, const uint64_t* name_off, uint64_t n_scaf,
                  uint32_t max_snps, uint64_t cap, uint32_t* read_scaf, uint64_t* read_pos, uint32_t* read_len, uint32_t* rec_scaf,
                  uint64_t* rec_pos, uint32_t* rec_len, uint32_t* rec_flag, uint32_t* rec_nsnps, abw_sam_counts* counts);
/* The name table once per assembly (a SAM file read in pieces, several SAM files; ref src/abawaca-build.cpp:549-551 looks every
 * RNAME up in ONE map built at :483-490): after it, abw_sam_parse[_dev] with names == NULL, name_off == NULL and the same
 * n_scaf uses the table resident on the device instead of hashing and uploading the names again (ABW_ERR_STATE when none
 * is loaded or n_scaf differs).  Held until the next load (n_scaf = 0: dropped) or abw_ctx_destroy. */
int abw_sam_names_load(abw_ctx* ctx, const uint8_t* names, const uint64_t* name_off, uint64_t n_scaf);
/* The same with the text and every output array on the device (names on the host); one or two internal read-backs. */
int abw_sam_parse_dev(abw_ctx* ctx, const uint8_t* d_text, uint64_t len, const uint8_t* names, const uint64_t* name_off,
                      uint64_t n_scaf, uint32_t max_snps, uint64_t cap, uint32_t* d_read_scaf, uint64_t* d_read_pos,
                      uint32_t* d_read_len, uint32_t* d_rec_scaf, uint64_t* d_rec_pos, uint32_t* d_rec_len, uint32_t* d_rec_flag,
                      uint32_t* d_rec_nsnps, abw_sam_counts* counts);

/* ---- next row N4: the .lrn feature table, text -> integer keys (loader) and integer keys -> text (writer)
 *      (ref: ClusterData::ClusterData(esom_lrn_file, scaf_db), src/ClusterData.cpp:27-168; the file is written by
 *      src/abawaca-build.cpp:578-607 with every value printed "%.3lf").
 * abw_lrn_header (host only): the four header lines (ref :36-111) -> number of datapoints, number of dimensions
 * (without the key column) and, when names_buf is given, the dimension names NUL-separated.
 * abw_lrn_parse: the data rows (ref :117-163) on the GPU.  keys [ndims][ndps] int32 (dimension-major, rows in file
 * order) = thousandths of every value, i.e. exactly atof(field) * 1000 for 3-decimal text; dp_names [ndps] = the key
 * column (ref :146 atoi; mapping names to ScafDpData ids, ref :146-150, stays with the caller).  Lines are chomped,
 * empty lines and lines starting with '%' are skipped (ref :118-124); a wrong field count (ref :135-140) or row
 * count (ref :127-130,164-167) is ABW_ERR_FORMAT where the reference exits.  Fields must be [-]digits[.digits] with
 * at most 3 significant decimals -- what abawaca-build writes; anything else (exponents, inf, nan) is ABW_ERR_FORMAT
 * with the line number, and the caller falls back to the reference's loader.  Host memory in, host memory out. */
int abw_lrn_header(const uint8_t* text, uint64_t len, int64_t* ndps, int32_t* ndims, char* names_buf, uint64_t names_len);
int abw_lrn_parse(abw_ctx* ctx, const uint8_t* text, uint64_t len, int64_t ndps, int32_t ndims, int32_t* dp_names, int32_t* keys);
/* Writer side (ref: src/abawaca-build.cpp:595-607): the data rows of abawaca.lrn.  milli: host [n_rows][n_cols] int32,
 * row-major, = int(1000.0*x) of every printed column (abw_sig_count_dev's d_milli for the signature columns, the
 * truncated coverage sums behind them); row r is printed as "<first_key + r>\t%.3lf\t...\n" with every value
 * int(1000.0*x)/1000.0 -- byte for byte what fprintf writes (ref :600-605).  The four header lines (ref :579-593) are
 * the caller's.  *text_len receives the size even when it exceeds text_cap (then ABW_ERR_CAPACITY).  n_cols <= 960. */
int abw_lrn_emit(abw_ctx* ctx, const int32_t* milli, int64_t n_rows, int32_t n_cols, uint64_t first_key,
                 uint8_t* text, uint64_t text_cap, uint64_t* text_len);
/* Cluster writer (ref: ClusterEsomWriter::write_clustering_info_file, src/ClusterWriter.cpp:92-98): the data rows of one
 * cluster's .lrn, "<dp>" then "\t%lf" of the datapoint's value in every dimension, "\n".  "%lf" is reproduced exactly
 * (six decimals of the binary value, round to nearest, ties to even).  values: [n_cols][ld] doubles, dimension-major, the
 * value of datapoint dp (1-based internal id) in column c at values[c*ld + dp-1]; dps: the cluster's datapoints in the
 * order of the rows (the reference walks an unordered_set: the order is the caller's); the four header lines (ref :81-90)
 * stay with the caller.  *text_len = bytes the rows take, also when they do not fit text_cap (ABW_ERR_CAPACITY; text may
 * then be NULL: the call only sizes).  One launch prints and sizes, so a caller that passes a buffer of a safe size --
 * n_rows * (23 n_cols + 24) at most -- needs one call; after ABW_ERR_CAPACITY, ABW_ERR_ARG (a dp outside [1, ld]) or
 * ABW_ERR_FORMAT (a value that is not finite or not below 9.2e12 in magnitude) the first text_cap bytes of text are
 * unspecified, nothing behind them is touched.  _dev: values, dps and text are device pointers (the table stays resident
 * between clusters); d_row_off (device, n_rows + 1 entries, or NULL) receives the byte offset of every row, so that one
 * call can print the concatenated datapoint lists of many clusters and the caller cuts the text at the clusters' first
 * rows.  The other form takes host pointers and copies.  n_cols <= 710. */
int abw_lrn_emit_rows_dev(abw_ctx* ctx, const double* d_values, int64_t ld, int32_t n_cols, const int64_t* d_dps, int64_t n_rows,
                          uint8_t* d_text, uint64_t text_cap, uint64_t* text_len, uint64_t* d_row_off);
int abw_lrn_emit_rows(abw_ctx* ctx, const double* values, int64_t ld, int32_t n_cols, const int64_t* dps, int64_t n_rows,
                      uint8_t* text, uint64_t text_cap, uint64_t* text_len);
/* Keeps a copy of the table on the device: after abw_lrn_table_load (host pointer) or abw_lrn_table_load_dev (device
 * pointer; same [n_cols][ld] layout), abw_lrn_emit_rows and abw_lrn_emit_rows_dev with values == NULL (and the same ld,
 * n_cols) print from the resident copy -- what a driver that writes one .lrn per cluster of the recursion (ref:
 * src/abawaca.cpp:161-182) wants.  The copy is kept TRANSPOSED, a datapoint's values side by side: a row is then one
 * coalesced read, where the [n_cols][ld] form costs a 128-byte line per value (1.6 times the rows per second, DESIGN.md
 * 3.7).  values == NULL here drops the copy. */
int abw_lrn_table_load(abw_ctx* ctx, const double* values, int64_t ld, int32_t n_cols);
int abw_lrn_table_load_dev(abw_ctx* ctx, const double* d_values, int64_t ld, int32_t n_cols);

/* ---- extension: k-means restarts (NO counterpart in the reference, see DESIGN.md "K-means";
 *      RNG = TinyMT32, ref: src/misc.cpp:24-36, src/TinyMT-src-1.0.1/tinymt/tinymt32.{c,h}) -------- */

/* The chunk length abw_km_run_dev uses when it is passed chunk = 0: the multiple of 32 (at most
 * 2048) that cuts N points into an even number of rounds over the device's compute units for R
 * restarts.  Results depend on the chunk (it bounds the fp32 partial sums), so a caller that wants
 * bit-reproducible output across devices passes an explicit chunk instead.  (No reference counterpart.) */
int32_t abw_km_auto_chunk(const abw_ctx* ctx, int64_t N, int32_t R);
/* The same rule for a device of num_cu compute units, without a context (no GPU call): what an adapter that wants a run
 * reproducible bit for bit across devices passes as `chunk` -- integration/ClusterSeparatorKMeans uses num_cu = 256 (MI355X)
 * whatever the device is.  abw_km_auto_chunk(ctx, N, R) == abw_km_chunk_for(the context's compute units, N, R). */
int32_t abw_km_chunk_for(int32_t num_cu, int64_t N, int32_t R);

/* The arithmetic of the Lloyd passes that feed the centroid update (abw_km_run_dev, abw_km_shard_step, abw_km_node):
 *   ABW_KM_ARITH_F32    (default) exact fp32 fmaf chains on the fp32 matrix cores: bit-identical to the CPU restatement;
 *   ABW_KM_ARITH_BF16X3 points and discriminant split exactly into three bf16 planes each, the six plane products of weight
 *                       >= 2^-16 on the bf16 matrix cores with fp32 accumulation (about fp32 accuracy, hardware summation
 *                       order): twice the step rate, held to the documented Lloyd tolerance (per pass <= 1e-5 of the
 *                       points in the other cluster, final inertia to 1e-6 relative) instead of bit parity.  Shapes without
 *                       such a kernel (fewer than 33 restarts, d + 1 < 129, 64 restarts with d > 191, (d - 1) N >= 2^30) run the exact one.
 * The final assign pass, the fp64 reductions, the centroid update and the inertia are the exact path's either way.
 * Stays set on the context until changed.  (No reference counterpart.) */
#define ABW_KM_ARITH_F32    0
#define ABW_KM_ARITH_BF16X3 1
int abw_km_set_arith(abw_ctx* ctx, int32_t arith);

/* d_x: device [d][N] fp32 dimension-major points, d <= 223.  k = 2.  R restarts are run
 * together; restart r seeds TinyMT32 with base_seed + r.  chunk: points per fp32 partial sum
 * (multiple of 32; 0 = abw_km_auto_chunk).  Outputs are device pointers (any may be NULL):
 *   d_assign    [R][N] u8   cluster of every point after the final assign pass
 *   d_centroids [R][2][d] fp32
 *   d_inertia   [R] f64     sum of squared distances to the assigned centroid
 * Asynchronous on the context's stream. */
int abw_km_run_dev(abw_ctx* ctx, const float* d_x, int64_t N, int32_t d, int32_t R, uint32_t base_seed,
                   int32_t iters, int32_t chunk, uint8_t* d_assign, float* d_centroids, double* d_inertia);

/* ---- k-means with the POINTS sharded over several contexts / processes / GPUs ("scaffolds
 *      are sharded with all-reduce of the per-iteration centroid partial sums").  No reference counterpart.
 * The library runs each shard's local part; between the phases the CALLER combines three small device
 * buffers across the shards with whatever transport it has (abawaca_amd/multi.py: RCCL / gloo all-gather,
 * added in shard order so that every shard holds identical bits).  abw_km_run_dev is these phases with
 * one shard and nothing to combine.
 *
 *   begin   : shard owns points [first_point, first_point + N_local) of N_total, d_x = its [d][N_local] rows.
 *             Restart r draws its two seed points from [0, N_total) exactly as the unsharded run does.
 *   combine : d_init    -- init_count 32-bit words: fp32 bit patterns of the initial centroids, all-zero
 *                          where another shard owns the seed point -> INTEGER sum over shards;
 *             d_moments -- moments_count fp64 (per-dimension totals, local point count, sums of squares)
 *                          -> fp64 sum in shard order.
 *   ready   : builds the discriminants from the combined centroids.
 *   per pass (iters update passes, then one final pass):
 *     step   : local assign + cluster-1 sums -> d_sums (sums_count fp64); d_assign [R][N_local] u8 or NULL
 *     combine: d_sums -> fp64 sum in shard order
 *     update : last = 0 moves the centroids, last = 1 forms the inertia terms instead
 *   finish  : d_centroids [R][2][d] fp32 and d_inertia [R] f64 (either may be NULL), identical on every shard.
 * Asynchronous on the context's stream; the caller orders its combine steps after the library's work
 * (abw_ctx_synchronize, or by running them on the same stream). */
typedef struct abw_km_exchange {
    void*    d_init;     uint64_t init_count;
    double*  d_moments;  uint64_t moments_count;
    double*  d_sums;     uint64_t sums_count;
} abw_km_exchange;
/* The fp64 "sum in shard order" of the combine steps above as one kernel, for a caller that has the shards' buffers side
 * by side on its device (d_parts: [n_parts][n], e.g. the output of one RCCL all-gather): d_out[i] = ((part 0 + part 1) +
 * part 2) + ...  No counterpart in the reference (it is one process). */
int abw_km_sum_ordered(abw_ctx* ctx, const double* d_parts, int32_t n_parts, int64_t n, double* d_out);
int abw_km_shard_begin(abw_ctx* ctx, const float* d_x, int64_t N_local, int64_t first_point, int64_t N_total, int32_t d,
                       int32_t R, uint32_t base_seed, int32_t chunk, abw_km_exchange* ex);
int abw_km_shard_ready(abw_ctx* ctx);
int abw_km_shard_step(abw_ctx* ctx, uint8_t* d_assign);
int abw_km_shard_update(abw_ctx* ctx, int32_t last);
int abw_km_shard_finish(abw_ctx* ctx, float* d_centroids, double* d_inertia);

/* ---- the k-means separator on a RESIDENT feature table (SURVEY §8(f) row N2 applied to the extension; no
 *      reference counterpart for the k-means itself).  abw_km_load: x host [D][N] fp32 (the caller's scaling,
 *      e.g. z-scores over the root), dp2scaf as abw_sep_load.  abw_km_node = one node of the recursion given its
 *      scaffolds (strictly ascending ids; NULL = all): R restarts x iters Lloyd iterations over the node's dps in
 *      ascending dp order, the restart of smallest inertia (ties: the earlier), then what every separator of the
 *      reference does with a raw split (ref: src/ClusterSeparator.cpp:49-53 the smaller raw side is cluster1,
 *      tie: the library's side 0; :94-101 a scaffold goes to cluster1 iff at least half of its dps are there;
 *      :103-122 dps follow their scaffold; :125-132 both children need min_ndps dps).  Outputs as abw_sep_node:
 *      scaf_side [n_scaf] 1/2, dp_raw_side [node dps] 1/2, zero when there is no split candidate.
 *      chunk: points per fp32 partial sum as in abw_km_run_dev -- it is part of the result's bits; 0 lets the library size
 *      it for this device (abw_km_auto_chunk over the node's dps) and out->chunk reports what was used, so a caller that
 *      wants the same bits on another device passes that value there.
 *      out->split = 1 says the sizes allow a split; ACCEPTING it (agreement level, the SCG gate
 *      ClusterQuality::is_split_better, ref: src/ClusterQuality.cpp:96-136) stays with the caller. */
typedef struct {
    int32_t split;          /* raw side and both children reach min_ndps */
    int32_t best_restart;   /* -1 when the node is too small to try (n < 2*min_ndps or n < 32) */
    double  inertia;        /* of the best restart */
    double  agreement;      /* share of the node's dps that lie on their scaffold's side */
    int64_t ndps;           /* dps of the node */
    int64_t raw1;           /* dps on the smaller raw side */
    int64_t ndps1, ndps2;   /* children sizes after dps followed their scaffolds */
    int32_t chunk;          /* points per fp32 partial sum the run used (= the argument, or the library's choice for 0) */
    int32_t reserved;
} abw_km_split;
int abw_km_load(abw_ctx* ctx, const float* x, int32_t D, int64_t N, const int32_t* dp2scaf, int32_t S);
int abw_km_node(abw_ctx* ctx, const int32_t* scaf_ids, int32_t n_scaf, int32_t R, uint32_t base_seed, int32_t iters,
                int32_t chunk, int32_t min_ndps, abw_km_split* out, uint8_t* scaf_side, uint8_t* dp_raw_side);

/* Host-buffer convenience: x is host [d][N]; outputs host. */
int abw_km_run(abw_ctx* ctx, const float* x, int64_t N, int32_t d, int32_t R, uint32_t base_seed,
               int32_t iters, int32_t chunk, uint8_t* assign, float* centroids, double* inertia);

/* d_x[j][p] = (float)keys[j][p] * scale[j]  (scale host [D], NULL = all ones) — turns the resident
 * integer column store into k-means features without a second copy of the table on the host. */
int abw_km_features_from_keys(abw_ctx* ctx, const int32_t* d_keys, int32_t D, int64_t N, const float* scale,
                              float* d_x);

/* First n outputs of the TinyMT32 stream for `seed` computed ON THE DEVICE (known-answer check
 * against ref: tinymt/check32.out.txt).  out: host [n]. */
int abw_km_tinymt_selftest(abw_ctx* ctx, uint32_t seed, int32_t n, uint32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* ABAWACA_HIP_H */
