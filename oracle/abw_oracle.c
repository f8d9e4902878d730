/* TEST INFRASTRUCTURE — NOT part of the product path.  See abw_oracle.h for scope and pinning.
 *
 * Plain-C restatement of the reference's algorithms, deliberately written the way the REFERENCE
 * computes (rolling decimal keys; sort + sweep with TP/FP bookkeeping) and not the way the HIP
 * kernels do (2-bit codes + suffix marginals; histograms + scans), so that agreement between the
 * two is evidence and not tautology.
 */
#include "abw_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------------------------------- */
/* mer table — reference src/abawaca-build.cpp:75-100                                        */
/* ---------------------------------------------------------------------------------------- */
static int mer_block_offset(int k) { return k == 1 ? 0 : k == 2 ? 4 : k == 3 ? 20 : 84; }

void abw_or_mer_table(int32_t mer2dim[ABW_OR_NMERS])
{
    /* ids are handed out in enumeration order (k = 1..4, lexicographic A<C<G<T) to the first of
     * {mer, reverse complement} that appears (:79-86); the other one re-uses the id. */
    int32_t next_id = 0;
    for (int i = 0; i < ABW_OR_NMERS; i++) mer2dim[i] = -1;
    for (int k = 1; k <= 4; k++) {
        int off = mer_block_offset(k), n = 1 << (2 * k);
        for (int m = 0; m < n; m++) {
            int rc = 0;
            for (int i = 0; i < k; i++) {             /* digit i (from least significant) */
                int dgt = (m >> (2 * i)) & 3;
                rc = (rc << 2) | (3 - dgt);           /* complement, reversed */
            }
            if (mer2dim[off + rc] >= 0) mer2dim[off + m] = mer2dim[off + rc];
            else mer2dim[off + m] = next_id++;
        }
    }
}

/* ---------------------------------------------------------------------------------------- */
/* DNAString upper-casing — reference src/String.cpp:37-51                                   */
/* ---------------------------------------------------------------------------------------- */
int abw_or_upcase(uint8_t* seq, uint64_t len)
{
    for (uint64_t i = 0; i < len; i++) {
        uint8_t c = seq[i];
        if (c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N') continue;
        if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 'a' + 'A');   /* toupper, C locale */
        seq[i] = c;
        if (c == 'N') return -1;                                   /* :47-49 throws */
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------- */
/* segmentation — reference src/abawaca-build.cpp:198-228                                    */
/* ---------------------------------------------------------------------------------------- */
uint64_t abw_or_segments(const uint8_t* seq, uint64_t len, uint32_t window,
                         uint64_t* start, uint64_t* end, uint64_t cap)
{
    uint64_t Ns = 0;
    for (uint64_t i = 0; i < len; i++) Ns += (seq[i] == 'N');      /* Bio::Ns, String.cpp:134-146 */
    uint64_t nsegments = (len - Ns) / window;                       /* :202 */
    if (nsegments == 0) nsegments = 1;                              /* :204-205 */
    uint64_t nbps = (len - Ns) / nsegments;                         /* :206 */
    uint64_t nbps_segment = 0, sit = 0, n = 0;
    for (uint64_t it = 0; it < len; it++) {                         /* :209-225 */
        if (seq[it] != 'N') nbps_segment++;
        if (nbps_segment == nbps) {
            if (n < cap) { start[n] = sit + 1; end[n] = it + 1; }   /* :216, 1-based inclusive */
            n++;
            sit = it + 1;
            nbps_segment = 0;
        }
    }
    return n;                                                       /* tail dropped, :227 */
}

/* ---------------------------------------------------------------------------------------- */
/* signature — reference src/abawaca-build.cpp:103-174                                       */
/* ---------------------------------------------------------------------------------------- */
static int dec_digit(uint8_t c)
{
    switch (c) { case 'A': return 2; case 'C': return 3; case 'G': return 4; case 'T': return 5; default: return 0; }
}

void abw_or_signature(const uint8_t* seg, uint64_t len, double dims[ABW_OR_NDIMS],
                      uint32_t* counts, uint32_t* totals)
{
    static int32_t mer2dim[ABW_OR_NMERS];
    static int have_table = 0;
    if (!have_table) { abw_or_mer_table(mer2dim); have_table = 1; }

    for (int i = 0; i < ABW_OR_NDIMS; i++) dims[i] = 0.0;
    if (counts) memset(counts, 0, ABW_OR_NMERS * sizeof(uint32_t));
    if (totals) memset(totals, 0, 4 * sizeof(uint32_t));

    int factor = 10, N_limit = 1;                                   /* :112-113 */
    for (int k = 1; k <= 4; k++, N_limit *= 10, factor *= 10) {     /* :114 */
        if (len < (uint64_t)k) continue;                            /* :116-117 */
        uint32_t cnt[6000];                                         /* indexed by decimal key <= 5555 */
        memset(cnt, 0, sizeof(cnt));
        int key = 0, total = 0;
        uint64_t pos = 0;
        while (pos < (uint64_t)k) {                                 /* first mer, :124-133 */
            key *= 10;
            int dgt = dec_digit(seg[pos++]);
            if (dgt) key += dgt; else key = 0;
        }
        if (key > N_limit) { cnt[key]++; total++; }                 /* :134-137 */
        while (pos < len) {                                         /* :139-155 */
            key *= 10;
            key = key % factor;
            int dgt = dec_digit(seg[pos++]);
            if (dgt) key += dgt; else key = 0;
            if (key > N_limit) { cnt[key]++; total++; }
        }
        if (totals) totals[k - 1] = (uint32_t)total;
        for (int key2 = 0; key2 < 6000; key2++) {                   /* :157-172 (map order is irrelevant:   */
            if (!cnt[key2]) continue;                               /*  each dim gets at most two addends)  */
            int m = 0, ok = 1;
            for (int p = factor / 10; p > 0; p /= 10) {             /* decode the k digits */
                int dg = (key2 / p) % 10;
                if (dg < 2 || dg > 5) { ok = 0; break; }
                m = (m << 2) | (dg - 2);
            }
            if (!ok) continue;                                      /* cannot happen for key > N_limit */
            if (counts) counts[mer_block_offset(k) + m] = cnt[key2];
            dims[mer2dim[mer_block_offset(k) + m]] += (total != 0) ? ((double)cnt[key2] / total) : 0;   /* :171 */
        }
    }
}

int32_t abw_or_milli(double x) { return (int32_t)(1000.0 * x); }   /* :603 */

/* ---------------------------------------------------------------------------------------- */
/* separator                                                                                 */
/* ---------------------------------------------------------------------------------------- */
typedef struct {
    int64_t tp, fp, tot;          /* true_positives, false_positives, total_dps_for_assigned_scafs */
} or_stats;

typedef struct { int32_t key; int32_t local; } or_kv;

static int cmp_kv(const void* a, const void* b)
{
    int32_t x = ((const or_kv*)a)->key, y = ((const or_kv*)b)->key;
    return (x > y) - (x < y);
}

/* ClusteringResult::is_better — reference src/ClusterSeparatorBySensitivitySpecificity.h:25-30
 * with the base-class tie-break of src/ClusterSeparator.cpp:11-16.  dim: size_t, (size_t)-1 = none. */
static int cr_is_better(double sens, double spec, uint64_t dim, double value,
                        double osens, double ospec, uint64_t odim, double ovalue)
{
    double a = sens * spec, b = osens * ospec;
    if (a != b) return a >= b;
    if (dim != odim) return dim < odim;
    return value < ovalue;
}

/* ClusterQuality::is_split_better — reference src/ClusterQuality.cpp:96-136 (symmetric in its
 * two arguments).  side[ls] in {1,2}: cluster the node-local scaffold ls is assigned to. */
static int gate_is_split_better(const uint8_t* side, const int32_t* node_scafs, int32_t m,
                                const int64_t* scaf_len, const int32_t* scg_off, const int32_t* scg_ids,
                                int32_t n_scg, int32_t* scratch /* 2*n_scg */)
{
    int32_t* c1 = scratch;
    int32_t* c2 = scratch + n_scg;
    memset(scratch, 0, sizeof(int32_t) * 2 * (size_t)n_scg);
    int64_t size1 = 0, size2 = 0;
    for (int32_t ls = 0; ls < m; ls++) {
        int32_t s = node_scafs[ls];
        if (side[ls] == 1) size1 += scaf_len[s]; else size2 += scaf_len[s];
        if (!scg_off) continue;
        for (int32_t e = scg_off[s]; e < scg_off[s + 1]; e++) {
            if (side[ls] == 1) c1[scg_ids[e]]++; else c2[scg_ids[e]]++;
        }
    }
    int64_t g1 = 0, g2 = 0, both = 0;
    for (int32_t g = 0; g < n_scg; g++) {
        g1 += c1[g] > 0; g2 += c2[g] > 0; both += (c1[g] > 0 && c2[g] > 0);
    }
    if (g1 == 0 || g2 == 0) return (size1 >= 500000) && (size2 >= 500000);   /* :118-120 */
    if ((double)both / (double)(g1 < g2 ? g1 : g2) < 0.2) return 0;            /* :130-131 */
    return 1;
}

int abw_or_separate(const int32_t* keys, int64_t N, int32_t D,
                    const int32_t* dp2scaf, const int32_t* scaf_ndps, const int64_t* scaf_len, int32_t S,
                    const int32_t* scg_off, const int32_t* scg_ids, int32_t n_scg,
                    const int32_t* node_dps, int64_t n, int32_t min_ndps, int nthreads,
                    abw_or_split* out, abw_or_dimbest* per_dim,
                    uint8_t* dp_raw_side, uint8_t* scaf_side)
{
    if (n <= 0 || D <= 0) return -1;
    /* node-local scaffold numbering */
    int32_t* g2l = (int32_t*)malloc(sizeof(int32_t) * (size_t)S);
    int32_t* node_scafs = (int32_t*)malloc(sizeof(int32_t) * (size_t)S);
    int32_t* lscaf = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    if (!g2l || !node_scafs || !lscaf) return -2;
    for (int32_t s = 0; s < S; s++) g2l[s] = -1;
    int32_t m = 0;
    for (int64_t i = 0; i < n; i++) {
        int32_t s = dp2scaf[node_dps[i]];
        if (g2l[s] < 0) { g2l[s] = m; node_scafs[m++] = s; }
        lscaf[i] = g2l[s];
    }

    abw_or_dimbest* dimbest = per_dim ? per_dim : (abw_or_dimbest*)malloc(sizeof(abw_or_dimbest) * (size_t)D);
    int err = 0;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif

    /* one "thread" per dimension like reference src/ClusterSeparator.cpp:72-78 */
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int32_t d = 0; d < D; d++) {
        /* separate_dimension — reference src/ClusterSeparatorBySensitivitySpecificity.cpp:91-143 */
        or_kv* dps = (or_kv*)malloc(sizeof(or_kv) * (size_t)n);
        int32_t* cnt1 = (int32_t*)calloc((size_t)m, sizeof(int32_t));
        int32_t* cnt2 = (int32_t*)calloc((size_t)m, sizeof(int32_t));
        uint8_t* side = (uint8_t*)malloc((size_t)m);            /* which cluster the scaffold is assigned to */
        int32_t* scratch = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(n_scg > 0 ? n_scg : 1));
        if (!dps || !cnt1 || !cnt2 || !side || !scratch) { err = 1; free(dps); free(cnt1); free(cnt2); free(side); free(scratch); continue; }
        or_stats s1 = {0, 0, 0}, s2 = {0, 0, 0};
        const int32_t* col = keys + (int64_t)d * N;
        /* :105-112 — everything starts in cluster2, every scaffold assigned to cluster2 */
        for (int64_t i = 0; i < n; i++) {
            dps[i].key = col[node_dps[i]];
            dps[i].local = (int32_t)i;
            int32_t ls = lscaf[i];
            if (scaf_ndps[node_scafs[ls]] < 2) continue;        /* add_dp :10-11 (never true after ScafDpData.cpp:92-93) */
            cnt2[ls]++;
            s2.fp++;                                            /* add_dp before assign: false positive */
        }
        for (int32_t ls = 0; ls < m; ls++) {
            side[ls] = 2;
            if (scaf_ndps[node_scafs[ls]] < 2) continue;
            s2.fp -= cnt2[ls]; s2.tp += cnt2[ls]; s2.tot += scaf_ndps[node_scafs[ls]];   /* assign_scaf :42-58 */
        }
        qsort(dps, (size_t)n, sizeof(or_kv), cmp_kv);           /* :114 */

        int found = 0;
        double bsens = 0, bspec = 0, bvalue = -1;
        uint64_t bdim = (uint64_t)-1;                            /* ClusteringResult() : dimension(-1), value(-1) */
        int32_t bkey = -1;
        for (int64_t i = 0; i < n;) {                            /* :117-134 */
            int32_t key = dps[i].key;
            while (i < n && dps[i].key == key) {
                int32_t ls = lscaf[dps[i].local];
                int32_t tot_s = scaf_ndps[node_scafs[ls]];
                if (tot_s >= 2) {
                    cnt1[ls]++; if (side[ls] == 1) s1.tp++; else s1.fp++;      /* add_dp :6-21 */
                    cnt2[ls]--; if (side[ls] == 2) s2.tp--; else s2.fp--;      /* remove_dp :24-39 */
                    if (side[ls] == 2 && 2 * (int64_t)cnt1[ls] > tot_s) {      /* :125 */
                        s1.fp -= cnt1[ls]; s1.tp += cnt1[ls]; s1.tot += tot_s; /* assign_scaf */
                        s2.fp += cnt2[ls]; s2.tp -= cnt2[ls]; s2.tot -= tot_s; /* unassign_scaf :61-76 */
                        side[ls] = 1;
                    }
                }
                i++;
            }
            /* separate_specific :146-161 */
            int64_t n1 = s1.tp + s1.fp, n2 = s2.tp + s2.fp;
            if (n1 < min_ndps || n2 < min_ndps) continue;
            const or_stats* small = (n1 < n2) ? &s1 : &s2;
            double sens = (double)small->tp / (double)small->tot;              /* .h:75 */
            double spec = (double)small->tp / (double)(small->tp + small->fp); /* .h:74 */
            double value = (double)key;                          /* order-isomorphic to the .lrn double */
            if (cr_is_better(sens, spec, (uint64_t)(d + 1), value, bsens, bspec, bdim, bvalue) &&
                gate_is_split_better(side, node_scafs, m, scaf_len, scg_off, scg_ids, n_scg, scratch)) {
                bsens = sens; bspec = spec; bdim = (uint64_t)(d + 1); bvalue = value; bkey = key; found = 1;
            }
        }
        dimbest[d].found = found;
        dimbest[d].key = bkey;
        dimbest[d].sensitivity = bsens;
        dimbest[d].specificity = bspec;
        free(dps); free(cnt1); free(cnt2); free(side); free(scratch);
    }
    if (err) { free(g2l); free(node_scafs); free(lscaf); if (!per_dim) free(dimbest); return -2; }

    /* global winner :137-140 — order independent (is_better is a strict total order on distinct
     * (product, dimension, value) triples) */
    double gsens = 0, gspec = 0, gvalue = -1;
    uint64_t gdim = (uint64_t)-1;
    int32_t gkey = -1;
    for (int32_t d = 0; d < D; d++) {
        double sens = dimbest[d].found ? dimbest[d].sensitivity : 0.0;
        double spec = dimbest[d].found ? dimbest[d].specificity : 0.0;
        uint64_t dim = dimbest[d].found ? (uint64_t)(d + 1) : (uint64_t)-1;
        double value = dimbest[d].found ? (double)dimbest[d].key : -1.0;
        if (cr_is_better(sens, spec, dim, value, gsens, gspec, gdim, gvalue)) {
            gsens = sens; gspec = spec; gdim = dim; gvalue = value; gkey = dimbest[d].key;
        }
    }

    memset(out, 0, sizeof(*out));
    out->dimension = (gdim == (uint64_t)-1) ? -1 : (int32_t)gdim;
    out->key = (gdim == (uint64_t)-1) ? -1 : gkey;
    out->sensitivity = gsens;
    out->specificity = gspec;
    /* is_legal .h:31-35 */
    out->legal = (gsens >= 0.8) && (gspec >= 0.8) && (gsens + gspec >= 1.6) && (gsens * gspec >= 0.8);
    if (dp_raw_side) memset(dp_raw_side, 0, (size_t)n);
    if (scaf_side) memset(scaf_side, 0, (size_t)S);

    if (out->legal && gdim != (uint64_t)-1) {
        /* reconstruct_best_clusters — reference src/ClusterSeparator.cpp:25-54 */
        const int32_t* col = keys + (int64_t)(gdim - 1) * N;
        int64_t le = 0, gt = 0;
        for (int64_t i = 0; i < n; i++) {
            int useful = scaf_ndps[dp2scaf[node_dps[i]]] > 1;
            if (col[node_dps[i]] <= gkey) le += useful; else gt += useful;
        }
        int swap = gt < le;                                      /* :49-53: cluster1 = the smaller one */
        int32_t* c1cnt = (int32_t*)calloc((size_t)m, sizeof(int32_t));
        int32_t* c2cnt = (int32_t*)calloc((size_t)m, sizeof(int32_t));
        uint8_t* raw = (uint8_t*)malloc((size_t)n);
        for (int64_t i = 0; i < n; i++) {
            int in_le = col[node_dps[i]] <= gkey;
            int c = (in_le != swap) ? 1 : 2;
            raw[i] = (uint8_t)c;
            if (c == 1) c1cnt[lscaf[i]]++; else c2cnt[lscaf[i]]++;
        }
        if (dp_raw_side) memcpy(dp_raw_side, raw, (size_t)n);    /* :87-90 */
        /* scaffold assignment :94-101, dp moves :104-122 */
        int64_t nd1 = 0, nd2 = 0;
        uint8_t* fin = (uint8_t*)malloc((size_t)m);
        for (int32_t ls = 0; ls < m; ls++) {
            int32_t tot_s = scaf_ndps[node_scafs[ls]];
            int a1 = (c1cnt[ls] > 0) && (2 * (int64_t)c1cnt[ls] >= tot_s);
            int a2 = (c2cnt[ls] > 0) && (2 * (int64_t)c2cnt[ls] > tot_s);
            /* dps of a scaffold assigned to neither stay where they are (cannot happen for whole scaffolds) */
            if (a1) { fin[ls] = 1; nd1 += c1cnt[ls] + c2cnt[ls]; }
            else if (a2) { fin[ls] = 2; nd2 += c1cnt[ls] + c2cnt[ls]; }
            else { fin[ls] = 3; nd1 += c2cnt[ls]; nd2 += c1cnt[ls]; }   /* both halves swap sides (:104-122) */
        }
        if (nd1 >= min_ndps && nd2 >= min_ndps) {                /* :125-132 */
            out->split = 1;
            out->ndps1 = nd1; out->ndps2 = nd2;
            if (scaf_side) for (int32_t ls = 0; ls < m; ls++) scaf_side[node_scafs[ls]] = fin[ls];
        } else {
            out->dimension = -1; out->key = -1; out->sensitivity = 0; out->specificity = 0;   /* best_separation->reset() :130 */
            if (dp_raw_side) { /* raw sets are left filled by the reference (:87-90 ran); keep them */ }
        }
        free(c1cnt); free(c2cnt); free(raw); free(fin);
    }
    free(g2l); free(node_scafs); free(lscaf);
    if (!per_dim) free(dimbest);
    return 0;
}

/* ---------------------------------------------------------------------------------------- */
/* TinyMT32 — reference tinymt32.c:62-76, tinymt32.h:70-107                                  */
/* ---------------------------------------------------------------------------------------- */
#define OR_MAT1 0x8f7011eeU   /* reference src/misc.cpp:29 */
#define OR_MAT2 0xfc78ff1fU   /* :30 */
#define OR_TMAT 0x3793fdffU   /* :31 */

static void tinymt_next_state(abw_or_tinymt* t)
{
    uint32_t y = t->s[3];
    uint32_t x = (t->s[0] & 0x7fffffffU) ^ t->s[1] ^ t->s[2];
    x ^= (x << 1);
    y ^= (y >> 1) ^ x;
    t->s[0] = t->s[1];
    t->s[1] = t->s[2];
    t->s[2] = x ^ (y << 10);
    t->s[3] = y;
    uint32_t mask = (uint32_t)(-(int32_t)(y & 1));
    t->s[1] ^= mask & OR_MAT1;
    t->s[2] ^= mask & OR_MAT2;
}

void abw_or_tinymt_init(abw_or_tinymt* t, uint32_t seed)
{
    t->s[0] = seed; t->s[1] = OR_MAT1; t->s[2] = OR_MAT2; t->s[3] = OR_TMAT;
    for (uint32_t i = 1; i < 8; i++)
        t->s[i & 3] ^= i + 1812433253U * (t->s[(i - 1) & 3] ^ (t->s[(i - 1) & 3] >> 30));
    if ((t->s[0] & 0x7fffffffU) == 0 && t->s[1] == 0 && t->s[2] == 0 && t->s[3] == 0) {
        t->s[0] = 'T'; t->s[1] = 'I'; t->s[2] = 'N'; t->s[3] = 'Y';
    }
    for (int i = 0; i < 8; i++) tinymt_next_state(t);
}

uint32_t abw_or_tinymt_next(abw_or_tinymt* t)
{
    tinymt_next_state(t);
    uint32_t t0 = t->s[3];
    uint32_t t1 = t->s[0] + (t->s[2] >> 8);
    t0 ^= t1;
    t0 ^= (uint32_t)(-(int32_t)(t1 & 1)) & OR_TMAT;
    return t0;
}

/* ---------------------------------------------------------------------------------------- */
/* Row N1: per-segment read coverage                                                         */
/* ---------------------------------------------------------------------------------------- */
/* ref: Scaf::add_mapped_read src/abawaca-build.cpp:231-244 (walk the scaffold's segments in order: skip
 * while s > end, stop when e < start) and Scaf_segment::add_mapped_read :177-185 (clip, divide, add).
 * s = read_pos (0-based, ref src/ReadMapping.cpp:41) and e = s + len - 1 are compared with the 1-based
 * segment coordinates as they are -- the reference's own off-by-one. */
int abw_or_coverage(const uint32_t* seg_scaf, const uint64_t* seg_start, const uint64_t* seg_end, uint64_t n_seg,
                    uint64_t n_scaf, const uint32_t* read_scaf, const uint64_t* read_pos, const uint32_t* read_len,
                    uint64_t n_reads, double* cov)
{
    uint64_t* first = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n_scaf + 2));
    if (!first) return -2;
    uint64_t g = 0;
    for (uint64_t sc = 0; sc <= n_scaf; sc++) {
        while (g < n_seg && seg_scaf[g] < sc) g++;
        first[sc] = g;
    }
    first[n_scaf + 1] = n_seg;
    for (uint64_t r = 0; r < n_reads; r++) {
        const uint64_t sc = read_scaf[r];
        if (sc >= n_scaf) continue;                                     /* ref :550-552: unknown reference name */
        /* read_len 0: e wraps like the reference's size_t (ref :233) */
        const uint64_t s = read_pos[r], e = read_pos[r] + read_len[r] - 1;
        for (uint64_t q = first[sc]; q < first[sc + 1]; q++) {
            if (s > seg_end[q]) continue;                               /* ref :235-236 */
            if (e < seg_start[q]) break;                                /* ref :237-238 */
            const uint64_t a = (s < seg_start[q]) ? seg_start[q] : s;   /* ref :181 */
            const uint64_t b = (e > seg_end[q]) ? seg_end[q] : e;       /* ref :182 */
            cov[q] += (double)(b - a + 1) / (double)(seg_end[q] - seg_start[q] + 1);   /* ref :183 */
        }
    }
    free(first);
    return 0;
}

/* ---------------------------------------------------------------------------------------- */
/* Row N1, the reader side: SAM records                                                      */
/* ---------------------------------------------------------------------------------------- */
static int plain_number(const uint8_t* p, uint64_t n, uint64_t* v)      /* 1..9 digits */
{
    if (n == 0 || n > 9) return 0;
    uint64_t x = 0;
    for (uint64_t i = 0; i < n; i++) { if (p[i] < '0' || p[i] > '9') return 0; x = x * 10 + (uint64_t)(p[i] - '0'); }
    *v = x;
    return 1;
}

int abw_or_sam_line(const uint8_t* line, uint64_t len, abw_or_sam_rec* out)
{
    /* split('\t', ...), src/common.cpp:95-108: n tabs make n + 1 fields (an empty line makes none) */
    uint64_t fo[12], fl[12];          /* the first 11 fields; [11]: everything behind the 11th tab */
    int nf = 0;
    uint64_t p1 = 0;
    if (len > 0) {
        for (uint64_t i = 0; i <= len && nf < 11; i++) {
            if (i == len || line[i] == '\t') { fo[nf] = p1; fl[nf] = i - p1; nf++; p1 = i + 1; }
        }
    }
    if (nf < 11) return 1;
    uint64_t v;
    if (!plain_number(line + fo[1], fl[1], &v)) return 2;              /* FLAG = atoi(fs[1]), ref :39 */
    out->flag = (uint32_t)v;
    if (!plain_number(line + fo[3], fl[3], &v)) return 2;              /* POS = atoi(fs[3]) - 1, ref :41 */
    out->pos0 = v - 1;
    out->rname_off = fo[2]; out->rname_len = fl[2];                    /* ref :40 */
    out->seq_len = (uint32_t)fl[9];                                    /* ref :53 */
    for (uint64_t i = 0; i < fl[9]; i++) if (line[fo[9] + i] == 'n') return 3;      /* src/String.cpp:45-49 */
    const uint64_t qual_len = fl[10];
    out->nsnps = 0;
    /* the MD:Z field, ref :57-69: the first field behind the 11th that holds "MD:Z:" */
    const uint8_t* md = 0;
    uint64_t md_len = 0;
    for (uint64_t a = p1; a <= len && !md; ) {
        uint64_t b = a;
        while (b < len && line[b] != '\t') b++;
        for (uint64_t i = a; i + 5 <= b; i++) {
            if (line[i] == 'M' && line[i + 1] == 'D' && line[i + 2] == ':' && line[i + 3] == 'Z' && line[i + 4] == ':') {
                if (i != a) return 4;                                  /* inside a field: refused here */
                md = line + a; md_len = b - a;
                break;
            }
        }
        a = b + 1;
    }
    if (!md) return 0;                                                 /* ref :80-81 */
    for (uint64_t i = 0; i < md_len; i++) if (md[i] == ' ') return 4;  /* ref :60-66 keeps the blank, :102-103 throws on it */
    /* determine_snps, ref :85-128 */
    uint64_t q = 5, read_pos = 0, nd = 0;
    while (q < md_len && md[q] >= '0' && md[q] <= '9') { read_pos = read_pos * 10 + (uint64_t)(md[q++] - '0'); if (++nd > 9) return 4; }
    read_pos--;                                                        /* ref :98 (wraps when the string starts with 0) */
    uint32_t snps = 0;
    while (q < md_len) {
        if (md[q] != '^' && (md[q] < 'A' || md[q] > 'Z')) return 4;   /* ref :102-103 */
        if (md[q] == '^') {
            q++;
            while (q < md_len && (md[q] < '0' || md[q] > '9')) { snps++; q++; }     /* ref :107-111: one per deleted base */
        } else {
            read_pos++;
            if (read_pos >= out->seq_len || read_pos >= qual_len) return 4;          /* ref :115: get() / at() throw */
            snps++;
            q++;
        }
        if (q >= md_len || md[q] < '0' || md[q] > '9') return 4;       /* ref :120-121 */
        uint64_t o = 0; nd = 0;
        while (q < md_len && md[q] >= '0' && md[q] <= '9') { o = o * 10 + (uint64_t)(md[q++] - '0'); if (++nd > 9) return 4; }
        read_pos += o;
    }
    /* the CIGAR string, ref :132-179 */
    const uint8_t* cg = line + fo[5];
    const uint64_t cl = fl[5];
    if (cl == 0) return 5;
    if (cg[cl - 1] != 'I' && cg[cl - 1] != 'D' && cg[cl - 1] != 'M') return 5;       /* ref :134-137 */
    uint64_t read_pos2 = 0;
    for (uint64_t i = 0; i < cl; ) {
        uint64_t o = 0; nd = 0;
        while (i < cl && cg[i] >= '0' && cg[i] <= '9') { o = o * 10 + (uint64_t)(cg[i++] - '0'); if (++nd > 9) return 5; }
        if (o == 0) return 5;                                          /* ref :147-149 */
        const uint8_t op = cg[i++];                                    /* the last character is a letter: i < cl here */
        if (op == 'M') read_pos2 += o;
        else if (op == 'D') { }
        else if (op == 'I') { read_pos += o; snps++; read_pos2 += o; } /* ref :161-173: one SNP per insertion */
        else return 5;                                                 /* ref :176-177 */
    }
    if (read_pos + 1 != (uint64_t)out->seq_len || read_pos2 != (uint64_t)out->seq_len) return 6;   /* ref :182-183 */
    out->nsnps = snps;
    return 0;
}

int64_t abw_or_sam_parse(const uint8_t* text, uint64_t len, const uint8_t* names, const uint64_t* name_off, uint64_t n_scaf,
                         uint32_t* scaf, uint64_t* pos0, uint32_t* seq_len, uint32_t* flag, uint32_t* nsnps, uint64_t cap,
                         int32_t* err_code)
{
    int64_t n = 0, line_no = 0;
    for (uint64_t a = 0; a < len; ) {
        uint64_t b = a;
        while (b < len && text[b] != '\n') b++;
        uint64_t e = b;
        line_no++;
        if (text[a] != '@') {                                          /* src/ReadMappingReader.cpp:105-107 */
            while (e > a && (text[e - 1] == '\r' || text[e - 1] == '\n')) e--;   /* chomp, src/common.cpp:38-49 */
            abw_or_sam_rec r;
            const int rc = abw_or_sam_line(text + a, e - a, &r);
            if (rc) { if (err_code) *err_code = rc; return -line_no; }
            if ((uint64_t)n < cap) {
                uint64_t id = n_scaf;                                  /* scafs.find(ref_name()), src/abawaca-build.cpp:549 */
                for (uint64_t s = 0; s < n_scaf; s++) {
                    if (name_off[s + 1] - name_off[s] == r.rname_len && memcmp(names + name_off[s], text + a + r.rname_off, r.rname_len) == 0) { id = s; break; }
                }
                scaf[n] = (uint32_t)id; pos0[n] = r.pos0; seq_len[n] = r.seq_len; flag[n] = r.flag; nsnps[n] = r.nsnps;
            }
            n++;
        }
        a = b + 1;
    }
    return n;
}

/* ---------------------------------------------------------------------------------------- */
/* K-means restarts (extension; semantics fixed in DESIGN.md "K-means")                      */
/* ---------------------------------------------------------------------------------------- */
#define KM_GROUP 32   /* chunk partials are first added in groups of 32 consecutive chunks, then the groups */

/* Per dimension: T[j] = sum_p x[j][p] and sum_p x[j][p]^2, each as 256 strided fp64 partials,
 * xor-butterfly inside each group of 64, then (w0 + w1) + (w2 + w3).  T[d] = N.  Returns X2 = the
 * squares summed over dimensions in index order. */
static double km_butterfly(const double* part)
{
    double w[4];
    for (int g = 0; g < 4; g++) {
        double v[64], nv[64];
        for (int l = 0; l < 64; l++) v[l] = part[g * 64 + l];
        for (int o = 32; o > 0; o >>= 1) {
            for (int l = 0; l < 64; l++) nv[l] = v[l] + v[l ^ o];
            memcpy(v, nv, sizeof(v));
        }
        w[g] = v[0];
    }
    return (w[0] + w[1]) + (w[2] + w[3]);
}

/* moments of the points [b, e) of x (rows of length N); results ADDED to T[0..d] and returned (X2) */
static double km_moments(const float* x, int64_t N, int64_t b, int64_t e, int32_t d, double* T)
{
    double total = 0.0;
    for (int32_t j = 0; j < d; j++) {
        const float* row = x + (int64_t)j * N + b;
        double p1[256], p2[256];
        for (int t = 0; t < 256; t++) {
            double a1 = 0.0, a2 = 0.0;
            for (int64_t p = t; p < e - b; p += 256) { double v = (double)row[p]; a1 += v; a2 += v * v; }
            p1[t] = a1; p2[t] = a2;
        }
        T[j] += km_butterfly(p1);
        total += km_butterfly(p2);
    }
    T[d] += (double)(e - b);
    return total;
}

int abw_or_kmeans(const float* x, int64_t N, int32_t d, int32_t R, uint32_t base_seed, int32_t iters,
                  int32_t chunk, int nthreads, uint8_t* assign, float* centroids, double* inertia)
{
    const int64_t bounds[2] = { 0, N };
    return abw_or_kmeans_sharded(x, N, d, R, base_seed, iters, chunk, nthreads, 1, bounds, assign, centroids, inertia);
}

int abw_or_kmeans_sharded(const float* x, int64_t N, int32_t d, int32_t R, uint32_t base_seed, int32_t iters,
                          int32_t chunk, int nthreads, int32_t nshards, const int64_t* bounds,
                          uint8_t* assign, float* centroids, double* inertia)
{
    if (N < 2 || d < 1 || R < 1 || chunk < 1 || nshards < 1 || !bounds || bounds[0] != 0 || bounds[nshards] != N) return -1;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    double* T = (double*)calloc((size_t)(d + 1), sizeof(double));
    double* x2d = (double*)calloc((size_t)d, sizeof(double));
    if (!T || !x2d) { free(T); free(x2d); return -2; }
    /* moments: per shard, then added in shard order -- per dimension for the squares too, the
     * dimensions being added last (the device keeps one sum of squares per dimension) */
    for (int32_t sh = 0; sh < nshards; sh++) {
        double* Ts = (double*)calloc((size_t)(d + 1), sizeof(double));
        if (!Ts) { free(T); free(x2d); return -2; }
        for (int32_t j = 0; j < d; j++) {
            double one[2] = { 0.0, 0.0 };
            /* one dimension at a time through the same routine */
            const double sq = km_moments(x + (int64_t)j * N, N, bounds[sh], bounds[sh + 1], 1, one);
            Ts[j] = one[0];
            x2d[j] += sq;
        }
        for (int32_t j = 0; j < d; j++) T[j] += Ts[j];
        T[d] += (double)(bounds[sh + 1] - bounds[sh]);
        free(Ts);
    }
    double x2 = 0.0;
    for (int32_t j = 0; j < d; j++) x2 += x2d[j];
    free(x2d);
    int err = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int32_t r = 0; r < R; r++) {
        float* c = centroids + (size_t)r * 2 * d;                  /* [2][d] */
        uint8_t* a = assign + (size_t)r * N;
        float* part = (float*)malloc(sizeof(float) * (size_t)(d + 1));       /* cluster 1 only */
        double* S = (double*)malloc(sizeof(double) * 2 * (size_t)(d + 1));   /* [2][d+1], S[0] derived */
        double* Sg = (double*)malloc(sizeof(double) * (size_t)(d + 1));
        double* Ss = (double*)malloc(sizeof(double) * (size_t)(d + 1));      /* one shard's sums */
        float* D = (float*)malloc(sizeof(float) * (size_t)(d + 1));
        if (!part || !S || !Sg || !Ss || !D) { err = 1; free(part); free(S); free(Sg); free(Ss); free(D); continue; }
        abw_or_tinymt t;
        abw_or_tinymt_init(&t, base_seed + (uint32_t)r);
        int64_t i0 = (int64_t)(abw_or_tinymt_next(&t) % (uint64_t)N);
        int64_t i1 = i0;
        for (int tries = 0; tries < 64 && i1 == i0; tries++) i1 = (int64_t)(abw_or_tinymt_next(&t) % (uint64_t)N);
        for (int32_t j = 0; j < d; j++) { c[j] = x[(int64_t)j * N + i0]; c[d + j] = x[(int64_t)j * N + i1]; }
        for (int32_t it = 0; it <= iters; it++) {
            /* k = 2: one linear discriminant per restart.  D = c1 - c0 (fp32), and at position d
             * (against an implicit x_d = 1) e = 0.5|c0|^2 - 0.5|c1|^2, the norms fp32 fmaf chains. */
            float n0 = 0.0f, n1 = 0.0f;
            for (int32_t j = 0; j < d; j++) { n0 = fmaf(c[j], c[j], n0); n1 = fmaf(c[d + j], c[d + j], n1); D[j] = c[d + j] - c[j]; }
            D[d] = 0.5f * n0 - 0.5f * n1;
            double* S1 = S + (d + 1);
            for (int32_t q = 0; q <= d; q++) S1[q] = 0.0;
            for (int32_t sh = 0; sh < nshards; sh++) {
                const int64_t sb = bounds[sh], se = bounds[sh + 1];
                const int64_t nchunks = (se - sb + chunk - 1) / chunk;
                for (int32_t q = 0; q <= d; q++) { Ss[q] = 0.0; Sg[q] = 0.0; }
                for (int64_t ch = 0; ch < nchunks; ch++) {
                    const int64_t p0 = sb + ch * chunk, p1 = p0 + chunk < se ? p0 + chunk : se;
                    for (int32_t q = 0; q <= d; q++) part[q] = 0.0f;
                    for (int64_t p = p0; p < p1; p++) {
                        /* Two fmaf chains: dimension pairs (j>>1) alternate between them, position d
                         * included; g is their fp32 sum.  (The GPU issues the pairs as alternating MFMA
                         * k-steps into two accumulators so that no MFMA waits for its predecessor.) */
                        float g[2] = { 0.0f, 0.0f };
                        for (int32_t j = 0; j < d; j++) g[(j >> 1) & 1] = fmaf(x[(int64_t)j * N + p], D[j], g[(j >> 1) & 1]);
                        g[(d >> 1) & 1] = fmaf(1.0f, D[d], g[(d >> 1) & 1]);
                        const int as = (g[0] + g[1]) > 0.0f;             /* tie -> cluster 0 */
                        a[p] = (uint8_t)as;
                        if (as) {
                            for (int32_t j = 0; j < d; j++) part[j] = fmaf(1.0f, x[(int64_t)j * N + p], part[j]);
                            part[d] = fmaf(1.0f, 1.0f, part[d]);        /* the count column */
                        }
                    }
                    for (int32_t q = 0; q <= d; q++) Sg[q] += (double)part[q];
                    if ((ch % KM_GROUP) == KM_GROUP - 1 || ch == nchunks - 1)      /* group of 32 chunks complete */
                        for (int32_t q = 0; q <= d; q++) { Ss[q] += Sg[q]; Sg[q] = 0.0; }
                }
                for (int32_t q = 0; q <= d; q++) S1[q] += Ss[q];                     /* shards in order */
            }
            for (int32_t q = 0; q <= d; q++) S[q] = T[q] - S1[q];     /* cluster 0 = totals - cluster 1 */
            if (it < iters) {
                for (int cl = 0; cl < 2; cl++) {
                    const double cnt = S[(size_t)cl * (d + 1) + d];
                    if (cnt > 0.0)
                        for (int32_t j = 0; j < d; j++) c[(size_t)cl * d + j] = (float)(S[(size_t)cl * (d + 1) + j] / cnt);
                }
            } else {
                double term[2];
                for (int cl = 0; cl < 2; cl++) {
                    double dot = 0.0, n2 = 0.0;
                    for (int32_t j = 0; j < d; j++) {
                        dot += (double)c[(size_t)cl * d + j] * S[(size_t)cl * (d + 1) + j];
                        n2 += (double)c[(size_t)cl * d + j] * (double)c[(size_t)cl * d + j];
                    }
                    term[cl] = S[(size_t)cl * (d + 1) + d] * n2 - 2.0 * dot;
                }
                inertia[r] = (x2 + term[0]) + term[1];
            }
        }
        free(part); free(S); free(Sg); free(Ss); free(D);
    }
    free(T);
    return err ? -2 : 0;
}

/* ------------------------------------------------------------------------------------------
 * .lrn loader, reference src/ClusterData.cpp:27-168 (see abw_oracle.h)
 * ------------------------------------------------------------------------------------------ */
#include <ctype.h>
#include <stdio.h>
int64_t abw_or_lrn_parse(const uint8_t* text, uint64_t len, int64_t row_cap, int32_t* ndps_declared, int32_t* ndims,
                         int32_t* dp_names, double* values)
{
    uint64_t pos = 0;
    int lineno = 0;
    unsigned ndps = 0, ncols = 0;
    int64_t entry = 0;
    char* line = (char*)malloc(8192);
    if (!line) return -1;
    while (pos < len) {
        /* fgets(line, 8192, fp): up to 8191 bytes, through the newline (ref :36,40,117) */
        uint64_t n = 0;
        while (pos < len && n < 8191) { line[n++] = (char)text[pos++]; if (line[n - 1] == '\n') break; }
        line[n] = 0;
        lineno++;
        if (lineno == 1) { char c; if (sscanf(line, "%c %u", &c, &ndps) != 2 || c != '%') { free(line); return -1; } continue; }   /* ref :48 */
        if (lineno == 2) { char c; if (sscanf(line, "%c %u", &c, &ncols) != 2 || c != '%' || ncols < 2) { free(line); return -1; } continue; }   /* ref :61 */
        if (lineno == 3) { unsigned t = 1; for (char* p = line; *p; p++) t += *p == '\t'; if (t != ncols) { free(line); return -1; } continue; }  /* ref :74-82 */
        while (strlen(line) > 0 && isspace((unsigned char)line[strlen(line) - 1])) line[strlen(line) - 1] = 0;   /* ref :92-93, :119-120 */
        if (lineno == 4) { unsigned t = 1; for (char* p = line; *p; p++) t += *p == '\t'; if (t != ncols) { free(line); return -1; } continue; }  /* ref :97-111 */
        if (line[0] == '%' || strlen(line) == 0) continue;                                  /* ref :123 */
        unsigned t = 1;
        for (char* p = line; *p; p++) t += *p == '\t';
        if (t != ncols) { free(line); return -2; }                                          /* ref :135-140 */
        if (entry >= row_cap) { free(line); return -3; }
        char* p1 = line;
        char* p2 = strchr(p1, '\t');
        *p2 = 0;
        dp_names[entry] = atoi(p1);                                                          /* ref :146 */
        p1 = p2 + 1;
        for (unsigned d = 0; d + 1 < ncols; d++) {
            p2 = strchr(p1, '\t');
            if (p2) *p2 = 0;
            values[(int64_t)d * row_cap + entry] = atof(p1);                                 /* ref :159,162 */
            p1 = p2 ? p2 + 1 : p1 + strlen(p1);
        }
        entry++;
    }
    free(line);
    *ndps_declared = (int32_t)ndps; *ndims = (int32_t)ncols - 1;
    return entry;
}

/* reference src/ClusterWriter.cpp:92-98 (the C library's own "%lf": correctly rounded, ties to even) */
uint64_t abw_or_lrn_emit_rows(const double* values, int64_t ld, int32_t n_cols, const int64_t* dps, int64_t n_rows,
                              uint8_t* text, uint64_t cap)
{
    uint64_t n = 0;
    char f[400];
    for (int64_t r = 0; r < n_rows; r++) {
        int l = snprintf(f, sizeof f, "%lu", (unsigned long)dps[r]);
        if (n + (uint64_t)l <= cap) memcpy(text + n, f, (size_t)l);
        n += (uint64_t)l;
        for (int32_t c = 0; c < n_cols; c++) {
            l = snprintf(f, sizeof f, "\t%lf", values[(int64_t)c * ld + dps[r] - 1]);
            if (n + (uint64_t)l <= cap) memcpy(text + n, f, (size_t)l);
            n += (uint64_t)l;
        }
        if (n + 1 <= cap) text[n] = '\n';
        n += 1;
    }
    return n;
}
