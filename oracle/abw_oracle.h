/* TEST INFRASTRUCTURE — NOT part of the product path.
 *
 * CPU restatement ("oracle") of the abawaca binning hot path, written from the reference's
 * behaviour, each function citing the reference file:line it follows (paths relative to
 * /root/reference).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library — as the checker, never as the thing measured or shipped.
 *
 * Pinning: the restatement is checked in tests/test_oracle_golden.py against golden vectors
 * produced by running the compiled reference in the build container
 * (tests/golden/make_golden.py -> tests/golden/*.npz):
 *   signature  — pinned (abawaca.lrn of reference abawaca-build, 214 segments incl. edge cases)
 *   separator  — pinned (per-node dimension/value/specificity/sensitivity and final bins of the
 *                reference's ClusterSeparatorBySensitivitySpecificity on 6 synthetic assemblies)
 *   tinymt32   — pinned (upstream check32.out.txt + reference tinymt32.c for 6 seeds)
 *   k-means    — PARITY UNPINNED: the reference contains no k-means; the
 *                functions below define the algorithm the HIP path must reproduce.
 */
#ifndef ABW_ORACLE_H
#define ABW_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ABW_OR_NMERS 340   /* 4 + 16 + 64 + 256 */
#define ABW_OR_NDIMS 180   /* 2 + 10 + 32 + 136 canonical 1..4-mers, "A" first */

/* mer (k=1..4 blocks, base-4 digits A,C,G,T = 0..3, most significant first) -> canonical
 * dimension index, numbered in order of first appearance.  reference src/abawaca-build.cpp:75-100 */
void abw_or_mer_table(int32_t mer2dim[ABW_OR_NMERS]);

/* Upper-cases like Bio::DNAString::validate_string (reference src/String.cpp:37-51).
 * Returns 0, or -1 if a lower-case 'n' is met (the reference throws Illegal_DNAString). */
int abw_or_upcase(uint8_t* seq, uint64_t len);

/* Segmentation of one (already upper-cased) scaffold.  reference src/abawaca-build.cpp:198-228.
 * start/end are 1-based inclusive like segment_coords (:216).  Returns the number of segments
 * (writes at most cap of them). */
uint64_t abw_or_segments(const uint8_t* seq, uint64_t len, uint32_t window,
                         uint64_t* start, uint64_t* end, uint64_t cap);

/* 1..4-mer signature of one segment: dims[180] as Scaf_segment::dimensions after the constructor,
 * counts[340] the raw window counts (our numbering), totals[4] the per-k window totals.
 * reference src/abawaca-build.cpp:103-174.  counts/totals may be NULL. */
void abw_or_signature(const uint8_t* seg, uint64_t len, double dims[ABW_OR_NDIMS],
                      uint32_t* counts, uint32_t* totals);

/* int(1000.0*x) — the truncation of reference src/abawaca-build.cpp:603 (printed as %.3lf). */
int32_t abw_or_milli(double x);

/* ------------------------------------------------------------------------------------------
 * Separator (reference src/ClusterSeparator.cpp:25-135,
 *            src/ClusterSeparatorBySensitivitySpecificity.cpp:6-161, .h:25-35,
 *            src/ClusterQuality.cpp:96-136, src/SCGdb.cpp:21-38)
 *
 * keys      : [D][N] int32, column-major over ALL dps (0-based dp ids); any order-preserving
 *             integer image of the .lrn values (thousandths for abawaca-build output).
 * dp2scaf   : [N] 0-based scaffold of each dp;  scaf_ndps [S]: ScafDpData::ndps(scaf);
 * scaf_len  : [S] sequence length (ClusterQuality.cpp:111-115);
 * scg_off/scg_ids : CSR scaffold -> SCG ids (SCGdb::scaf2scgs), n_scg distinct names.
 * node_dps  : [n] the dps of this node.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t found;        /* 1 if some threshold passed size check + gate in this dimension */
    int32_t key;          /* separating key (value) */
    double  sensitivity;
    double  specificity;
} abw_or_dimbest;

typedef struct {
    int32_t split;        /* separate() returned true */
    int32_t dimension;    /* 1-based like the reference; -1 if none / reset */
    int32_t key;          /* separating key; -1 if none */
    int32_t legal;        /* is_legal() of the global winner */
    double  sensitivity;
    double  specificity;
    int64_t ndps1, ndps2; /* final cluster sizes (cluster1 = the smaller raw side) */
} abw_or_split;

/* Runs one node.  per_dim (may be NULL): [D] per-dimension winners.  dp_raw_side (may be NULL):
 * [n] 1/2 = raw_dps_cluster1/2 membership (0 when no legal separation).  scaf_side (may be NULL):
 * [S] 1/2 final cluster of every scaffold present in the node, 0 elsewhere / when not split.
 * min_ndps is cluster_ndps_threshold (100, ClusterSeparator.h:72).  nthreads <= 0: all cores. */
int abw_or_separate(const int32_t* keys, int64_t N, int32_t D,
                    const int32_t* dp2scaf, const int32_t* scaf_ndps, const int64_t* scaf_len, int32_t S,
                    const int32_t* scg_off, const int32_t* scg_ids, int32_t n_scg,
                    const int32_t* node_dps, int64_t n, int32_t min_ndps, int nthreads,
                    abw_or_split* out, abw_or_dimbest* per_dim,
                    uint8_t* dp_raw_side, uint8_t* scaf_side);

/* ------------------------------------------------------------------------------------------
 * TinyMT32 (reference src/TinyMT-src-1.0.1/tinymt/tinymt32.c:62-76, tinymt32.h:70-107,168-171)
 * with the parameter triple of reference src/misc.cpp:29-31.
 * ------------------------------------------------------------------------------------------ */
typedef struct { uint32_t s[4]; } abw_or_tinymt;
void     abw_or_tinymt_init(abw_or_tinymt* t, uint32_t seed);
uint32_t abw_or_tinymt_next(abw_or_tinymt* t);

/* ------------------------------------------------------------------------------------------
 * Row N1, the reader side: one SAM record line as Bio::ReadMapping::ReadMapping(const char*) takes it apart
 * (reference src/ReadMapping.cpp:23-72) and determine_snps counts its SNPs (:78-185).
 * line: the record without its end of line (SAMReader::next_mapping chomps "\n" and "\r", src/ReadMappingReader.cpp:113,
 * src/common.cpp:38-49).  Returns 0, or the code of what the reference would do to the line instead of returning a
 * mapping (it throws, or indexes a field that is not there), or of what this restatement is stricter about:
 *   1  fewer than 11 tab-separated fields (:38-54 index fs[0] .. fs[10])
 *   2  FLAG or POS is not 1..9 digits (atoi would stop at the first other character: refused here)
 *   3  a lower-case n in SEQ (src/String.cpp:45-49 throws)
 *   4  the MD:Z string (throws :86-88, :102-103, :120-121; at() :115; a field behind the 11th that holds "MD:Z:" anywhere
 *      but at its start, or a number of more than 9 digits: refused here)
 *   5  the CIGAR string of a record with an MD:Z tag (throws :134-137, :147-149, :176-177)
 *   6  SNP offsets and CIGAR lengths do not add up to the read length (:182-183)
 * nsnps = num_snps(): one per mismatched and per deleted base of MD:Z, one per I operation of the CIGAR string.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t flag;              /* atoi(fs[1]) */
    uint64_t pos0;              /* atoi(fs[3]) - 1 as size_t: POS 0 wraps (:41) */
    uint32_t seq_len;           /* SEQ.size() (:53) */
    uint32_t nsnps;
    uint64_t rname_off, rname_len;   /* fs[2] inside the line */
} abw_or_sam_rec;
int abw_or_sam_line(const uint8_t* line, uint64_t len, abw_or_sam_rec* out);
/* A whole SAM text: lines that start with '@' are skipped (src/ReadMappingReader.cpp:105-107), every other line is a record,
 * in file order; scaf = index of RNAME among the n_scaf names (names: blob, name_off[n_scaf + 1]) or n_scaf when it is none
 * of them (src/abawaca-build.cpp:549-551).  Returns the number of records (at most cap are written), or -(1-based line
 * number) with *err_code set when a line is refused. */
int64_t abw_or_sam_parse(const uint8_t* text, uint64_t len, const uint8_t* names, const uint64_t* name_off, uint64_t n_scaf,
                         uint32_t* scaf, uint64_t* pos0, uint32_t* seq_len, uint32_t* flag, uint32_t* nsnps, uint64_t cap,
                         int32_t* err_code);

/* ------------------------------------------------------------------------------------------
 * Row N1: per-segment read coverage (reference src/abawaca-build.cpp:231-244, :177-185).
 * cov[g] += overlap/length for every read, in read order (fp64, the reference's accumulator).
 * Segments grouped by scaffold, ascending; read_pos 0-based (SAM POS - 1).
 * ------------------------------------------------------------------------------------------ */
int abw_or_coverage(const uint32_t* seg_scaf, const uint64_t* seg_start, const uint64_t* seg_end, uint64_t n_seg,
                    uint64_t n_scaf, const uint32_t* read_scaf, const uint64_t* read_pos, const uint32_t* read_len,
                    uint64_t n_reads, double* cov);

/* ------------------------------------------------------------------------------------------
 * K-means restarts (EXTENSION, parity unpinned — see DESIGN.md "K-means").
 * x: [d][N] fp32 dimension-major points; k = 2; R restarts.  Restart r seeds TinyMT32 with
 * base_seed + r and takes as initial centroids the points  u0 % N  and  u1 % N  (u1 redrawn, at
 * most 64 times, while it names the same point).
 * Assign (k = 2, one linear discriminant per restart): D_j = c1_j - c0_j (fp32) for j < d and
 *   D_d = 0.5f*|c0|^2 - 0.5f*|c1|^2 (the norms fp32 fmaf chains from 0) against x_d := 1;
 *   g = chain0 + chain1 (fp32), two fmaf chains from 0 over j = 0..d, j going to chain (j>>1)&1;
 *   a point goes to cluster 1 iff g > 0 (tie -> 0): the nearer centroid in exact arithmetic.
 * Sums: points are cut into chunks of `chunk` consecutive points; per chunk and dimension the
 *   members of cluster 1 are summed in fp32 in point order (and counted the same way); chunk
 *   partials are added in fp64 in chunk order inside groups of 32 consecutive chunks, then the
 *   group sums in group order.  Cluster 0 is totals minus cluster 1 (fp64), the totals T_j built as
 *   in km_moments() in abw_oracle.c, T_d = N.  centroid = (float)(sum / count); an empty cluster
 *   keeps its centroid.
 * `iters` assign+update rounds are followed by one assign-only pass whose sums give
 *   inertia[r] = (X2 + term_0) + term_1,  term_c = count_c*|c_c|^2 - 2*(c_c . sum_c)  in fp64
 *   (j ascending), X2 = sum |x|^2 as built by km_moments() in abw_oracle.c.
 * Outputs: assign [R][N] uint8, centroids [R][2][d] fp32, inertia [R] fp64.
 * ------------------------------------------------------------------------------------------ */
int abw_or_kmeans(const float* x, int64_t N, int32_t d, int32_t R, uint32_t base_seed, int32_t iters,
                  int32_t chunk, int nthreads, uint8_t* assign, float* centroids, double* inertia);

/* The same run with the points cut into `nshards` contiguous shards [bounds[s], bounds[s+1]) (bounds[0] = 0,
 * bounds[nshards] = N), as the device runs it over several GPUs: chunks restart at every shard, a shard's
 * sums are its groups added in order, the shards' sums and moments are added in shard order.
 * nshards = 1 is abw_or_kmeans. */
int abw_or_kmeans_sharded(const float* x, int64_t N, int32_t d, int32_t R, uint32_t base_seed, int32_t iters,
                          int32_t chunk, int nthreads, int32_t nshards, const int64_t* bounds,
                          uint8_t* assign, float* centroids, double* inertia);

#ifdef __cplusplus
}
#endif
#endif

/* ------------------------------------------------------------------------------------------
 * .lrn loader, the reference's way: ClusterData::ClusterData(esom_lrn_file, scaf_db),
 * reference src/ClusterData.cpp:27-168.  Four header lines ("% ndps", "% ncols", weights, names), then per
 * line: chomp trailing white space (:119-120), skip empty and '%' lines (:123), ncols tab-separated fields
 * (:135-140), field 0 = atoi -> dp name (:146), the others atof (:159,162).  The name -> internal id mapping
 * of ScafDpData (:146-150) is not part of this function.
 * values: [ndims][row_cap] doubles in file row order; dp_names [row_cap].  Returns the number of data rows,
 * -1 on a malformed header, -2 on a wrong field count, -3 when there are more rows than row_cap.
 * ------------------------------------------------------------------------------------------ */
int64_t abw_or_lrn_parse(const uint8_t* text, uint64_t len, int64_t row_cap, int32_t* ndps_declared, int32_t* ndims,
                         int32_t* dp_names, double* values);

/* ------------------------------------------------------------------------------------------
 * One cluster's ESOM table, the rows of ClusterEsomWriter::write_clustering_info_file, reference
 * src/ClusterWriter.cpp:92-98: per datapoint "%lu" of its id, then "\t%lf" of its value in every dimension, "\n".
 * The reference walks an unordered_set, so the row order is the caller's (dps, 1-based ids).
 * values: [n_cols][ld] doubles, the value of datapoint dp in column c at values[c*ld + dp-1].
 * Returns the number of bytes the rows take; writes them when they fit cap.
 * ------------------------------------------------------------------------------------------ */
uint64_t abw_or_lrn_emit_rows(const double* values, int64_t ld, int32_t n_cols, const int64_t* dps, int64_t n_rows,
                              uint8_t* text, uint64_t cap);
