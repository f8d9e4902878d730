// Definitions shared by the separator's translation units (abw_sep.hip: one node at a time; abw_tree.hip: the whole
// recursion level by level): histogram layout, candidate records and their order, the flip-value selection.
#pragma once
#include "abw_internal.h"
#include <limits.h>
#include <vector>

constexpr int NARROW_T = 1024;      // dimensions with keys in [0, NARROW_T) use the key itself as bin
constexpr int SCAN_CH = 2048;       // bins per scan chunk
constexpr int TILE_A = 1024;        // scaffolds per workgroup of the narrow accumulation (256 threads x 4)
constexpr int GRP_BINS = 1024;      // LDS bins of one group of narrow dimensions (a narrow dimension has at most NARROW_T)
constexpr int MAX_SCG = 4096;

// A group of up to 8 narrow dimensions whose values travel together: the narrow columns are kept a second time as
// [group][dp][8] x 16 bit (value | (value > t_s) << 15, t_s = the flip value of the dp's scaffold in that dimension) with the
// flip values beside them as [group][scaffold][8] x 16 bit (t_s in the low 10 bits; above them the number of the scaffold's
// dps with a value > t_s, 63 standing for "63 or more: count them").  A node is a set of whole scaffolds, so t_s -- the
// (n_s / 2 + 1)-th smallest of the scaffold's own values -- is the same in every node: both tables are built once per
// loaded table.  One 16-byte load brings a dp's values in 8 dimensions, and the dps of a scaffold are contiguous: the
// scattered scaffolds of a node deep in the recursion cost 64 useful bytes per cache line instead of 16 (in the
// [dimension][dp] layout every (scaffold, dimension) is a cache line of its own).
// The dimensions of a group are consecutive in SepHost::nsort (ascending number of bins); their bins lie side by side in
// LDS (loff: offsets by the values the table really holds, `total` <= GRP_BINS in all).
// goff: offset of the group in a slice laid out in nsort order; toff[q]: offset of dimension q behind goff there (bins T per
// dimension, which may exceed the table's own span when tables share a layout: abw_sep_narrow_bins)
struct NGroup { int32_t cnt, total; int32_t loff[8]; int32_t dim[8]; int32_t toff[8]; int64_t goff; };

struct DimBestDev { int32_t found; int32_t bin; double sens; double spec; double prod; long long size1; long long n; };

struct ChunkDesc { int32_t dim; int32_t start; int32_t len; int32_t first_chunk_of_dim; };

struct SepDev {
    const int32_t* bins; int64_t N;
    const int32_t* scaf_first; const int32_t* scaf_ndps; const int64_t* scaf_len;
    const int32_t* scg_off; const int32_t* scg_ids; int32_t n_scg;
    const int32_t* T; const int64_t* Toff;
    unsigned long long* h_st;          // per bin: size (low 32 bits) and tp1 (high 32 bits), updated by one 64-bit atomic
    uint32_t* h_tot; unsigned long long* h_bp;
    int32_t* scg_lo; int32_t* scg_hi;
    int32_t* range;                    // [D][2] bins the current node touched in each dimension (lo, hi inclusive; hi < lo: none)
};


// k-th smallest (0-based) of the n values at col[0..n), n > 8: binary search on the answer.
static __device__ inline int32_t select_kth_mem(const int32_t* __restrict__ col, int n, int k, int T)
{
    int lo = 0;
    int bit = 1;
    while ((bit << 1) < T && bit < (1 << 29)) bit <<= 1;
    for (; bit > 0; bit >>= 1) {
        const int cand = lo | bit;
        if (cand >= T) continue;
        int cnt = 0;
        for (int i = 0; i < n; i++) cnt += col[i] < cand;
        if (cnt <= k) lo = cand;
    }
    return lo;
}


struct Sums { unsigned long long size, tp1, tot, bp; };

static __device__ __forceinline__ Sums sums_add(Sums a, Sums b) { return {a.size + b.size, a.tp1 + b.tp1, a.tot + b.tot, a.bp + b.bp}; }


struct Cand { double prod; int32_t bin; int32_t found; double sens; double spec; long long size1; };

static __device__ __forceinline__ bool cand_better(const Cand& a, const Cand& b)   // is a better than b
{
    if (!a.found) return false;
    if (!b.found) return true;
    if (a.prod != b.prod) return a.prod > b.prod;
    return a.bin < b.bin;                        // equal products keep the smaller threshold
}

static __device__ __forceinline__ Cand cand_shfl_xor(const Cand& c, int o)
{
    Cand r;
    r.prod = __shfl_xor(c.prod, o, 64); r.bin = __shfl_xor(c.bin, o, 64); r.found = __shfl_xor(c.found, o, 64);
    r.sens = __shfl_xor(c.sens, o, 64); r.spec = __shfl_xor(c.spec, o, 64); r.size1 = __shfl_xor(c.size1, o, 64);
    return r;
}


struct GlobalRes { int32_t found; int32_t dim; int32_t bin; int32_t legal; double sens; double spec; long long size1; long long n; };


struct SepHost {
    std::vector<int32_t> narrow_dims, wide_dims;
    std::vector<ChunkDesc> chunks;
    std::vector<int32_t> dim_first_chunk, dim_nchunks;
    int32_t *d_narrow = nullptr, *d_wide = nullptr, *d_dim_first_chunk = nullptr, *d_dim_nchunks = nullptr;
    ChunkDesc* d_chunks = nullptr;
    Sums *d_chunk_sums = nullptr, *d_dim_total = nullptr;
    Cand* d_chunk_best = nullptr;
    // wide dimensions: dps sorted by bin (static) and the per-node flip-value records
    int32_t *d_sscaf = nullptr, *d_sbin = nullptr, *d_sdp = nullptr, *d_wts = nullptr, *d_wbelow = nullptr;
    int32_t* d_range = nullptr;        // [D][2] bins touched by the current node
    // narrow dimensions in groups (see NGroup), built by abw_sep_load
    std::vector<int32_t> nsort, nslot; // narrow dimensions by ascending number of bins; group * 8 + lane of each
    std::vector<int64_t> toffn;        // [n_narrow + 1] offsets of the dimensions, in nsort order, in a slice of narrow histograms only
    std::vector<NGroup> groups;
    NGroup* d_groups = nullptr;
    int32_t* d_nsort = nullptr;
    int64_t* d_toffn = nullptr;
    int32_t* d_dim_slot = nullptr;     // [D] narrow: group * 8 + lane; wide: -(index among the wide dimensions) - 1
    void* d_ts8 = nullptr;             // [groups][S][8] x 16 bit flip values
    void* d_nk16 = nullptr;            // [groups][N][8] x 16 bit re-coded narrow columns
    void* d_srec = nullptr;            // [S] x 16 bytes: first dp, dps, length (lo, hi)
    // abw_sep_tree's static tables of the wide dimensions, built at its first call on this table (abw_tree.hip)
    uint32_t* d_wword = nullptr;       // [n_wide][N] per sorted position: scaffold | (v >= t_s) << 31 | (owner) << 30
    uint32_t* d_wext = nullptr;        // [n_wide][S][4] per scaffold: dps below t_s, dps, min(length, 500000), bin of t_s
    uint32_t* d_sinfo = nullptr;       // [S] dps | min(length, 500000) << 7: the inline form of the position words (abw_tree.hip), else NULL
    int32_t* d_posof = nullptr;        // [n_wide][N] the position of every dp in the static order (the inverse of d_sdp)
    uint32_t* d_wpack = nullptr;       // [n_wide][S] the first three of d_wext in one word (scaffolds of <= 127 dps), else all ones
    int32_t* d_carriers = nullptr;     // scaffolds that carry at least one SCG
    int32_t n_carriers = 0;
    int32_t *d_g2c_off = nullptr, *d_g2c = nullptr;   // the (scaffold, SCG) pairs by SCG: [n_scg + 1] offsets, scaffolds
    uint8_t* d_member = nullptr;       // [S] a byte per scaffold: in the node in hand (abw_sep_node on a subset)
    void* d_nw = nullptr;              // abw_sep_node's scan of the wide dimensions: member bytes, tile sums, tile bests (abw_tree.hip)
    size_t nw_cap = 0;
    bool packed = false;               // every scaffold has at most PACK_MAX_NDPS dps: accum_group_tile<true>
    bool node_scan = false;            // the node accumulated last evaluates its wide dimensions by that scan
    void* stree_run = nullptr;         // this table's part in a scaffold-sharded recursion (abw_sep_stree_*; abw_tree.hip)
    void* tree_run = nullptr;          // the recursion in progress (abw_sep_tree_begin .. abw_sep_tree_result; abw_tree.hip)
    bool sharded = false;              // table loaded with a shared bin layout (abw_sep_shard_load)
};


struct abw_sep_extra { SepHost h; };
SepHost& abw_sep_host(abw_ctx* ctx);     // abw_sep.hip
void abw_tree_run_free(void* run);       // abw_tree.hip
void abw_stree_run_free(void* run);
int abw_tree_static_ensure(abw_ctx* ctx);
bool abw_node_wide_usable(abw_ctx* ctx);
int abw_node_wide_accum(abw_ctx* ctx, const uint8_t* d_member);
int abw_node_wide_eval(abw_ctx* ctx, const uint8_t* d_member, bool exact_per_dim, int32_t min_ndps, const int32_t* scg_lo, const int32_t* scg_hi, struct DimBestDev* d_dimbest);

// sorts the 8 ints x[0..8) ascending in place (Batcher's 19-comparator network; unused slots padded with INT_MAX)
#define ABW_CX8(x, a, b) { const int lo_ = min(x[a], x[b]), hi_ = max(x[a], x[b]); x[a] = lo_; x[b] = hi_; }
#define ABW_SORT8(x)                                                              \
    do {                                                                          \
        ABW_CX8(x, 0, 1) ABW_CX8(x, 2, 3) ABW_CX8(x, 4, 5) ABW_CX8(x, 6, 7)        \
        ABW_CX8(x, 0, 2) ABW_CX8(x, 1, 3) ABW_CX8(x, 4, 6) ABW_CX8(x, 5, 7)        \
        ABW_CX8(x, 1, 2) ABW_CX8(x, 5, 6)                                          \
        ABW_CX8(x, 0, 4) ABW_CX8(x, 1, 5) ABW_CX8(x, 2, 6) ABW_CX8(x, 3, 7)        \
        ABW_CX8(x, 2, 4) ABW_CX8(x, 3, 5)                                          \
        ABW_CX8(x, 1, 2) ABW_CX8(x, 3, 4) ABW_CX8(x, 5, 6)                         \
    } while (0)

// ---- narrow dimensions: histograms of one tile of scaffolds in one group of dimensions -------------------------------
// One thread per scaffold of the tile [begin, end) (positions in `list`, or the scaffold ids themselves when it is NULL),
// 256 threads striding over it.  Per dp and dimension size[v] += 1 and, above the flip value, tp1[v] += 1; per scaffold
// and dimension tp1[t_s] += its dps at or below t_s (they count from threshold t_s on), tot[t_s] += n and bp[t_s] += len
// -- into workgroup-private LDS histograms, handed bin by bin to flush(b, size | tp1 << 32, tot, bp) at the end (b: bin
// in the group's LDS layout; only non-empty bins).  scg_lo / scg_hi (NULL: none): [D][n_scg] min / max flip value over
// the scaffolds that carry each SCG.
// PACKED (tables whose scaffolds all have at most PACK_MAX_NDPS dps, so that a tile of TILE_A scaffolds holds fewer than 65 536
// dps): a bin's size and tp1 counts share one 32-bit LDS word (16 bits each), so do what a scaffold puts on bin t_s (its dps |
// its dps at or below t_s), and bp sums min(len, 500 000) in 32 bits -- one LDS atomic per dp and dimension instead of up to
// two, two per scaffold and dimension instead of three of which one of 64 bits (experiments/accum_range.hip: 121 -> 92 us at
// the root of the bench table).  The clamped lengths change no verdict: bp only ever meets ">= 500 000" (ref
// ClusterQuality.cpp:118-121), and a sum of clamped lengths reaches 500 000 exactly when the sum of the lengths does.
constexpr int PACK_MAX_NDPS = 63;
template <bool PACKED, typename Flush>
static __device__ __forceinline__ void accum_group_tile(const NGroup& G, const uint4* __restrict__ codes, const uint4* __restrict__ tsv,
                                                        const int4* __restrict__ srec, const int32_t* __restrict__ list, int begin, int end,
                                                        const SepDev& P, int32_t* __restrict__ scg_lo, int32_t* __restrict__ scg_hi, Flush flush)
{
    __shared__ uint32_t s_size[GRP_BINS];                  // LDS atomics: 32-bit ones are the fast ones
    __shared__ uint32_t s_tp1[PACKED ? 1 : GRP_BINS];
    __shared__ uint32_t s_tot[GRP_BINS];
    __shared__ unsigned long long s_bp[PACKED ? 1 : GRP_BINS];
    __shared__ uint32_t s_bp32[PACKED ? GRP_BINS : 1];
    // k-mer frequencies crowd into a handful of bins, and LDS atomics of one instruction on one word are served one after
    // the other: the histograms are kept in as many copies as fit (a power of two, at most 16), BIN-major -- bin b of copy
    // c at b * copies + c, the copy of a lane = lane mod copies -- so that lanes of different copies never share a bank.
    int lg = 0;
    while (lg < 4 && (G.total << (lg + 1)) <= GRP_BINS) lg++;
    const int copies = 1 << lg, mine = threadIdx.x & (copies - 1);
    for (int i = threadIdx.x; i < (G.total << lg); i += 256) {
        s_size[i] = 0; s_tot[i] = 0;
        if (PACKED) s_bp32[i] = 0; else { s_tp1[i] = 0; s_bp[i] = 0; }
    }
    __syncthreads();
    // one dp's value in one dimension: size += 1 and, above the flip value, tp1 += 1
    auto add_dp = [&](int at, uint32_t code) {
        if (PACKED) atomicAdd(&s_size[at], 1u + ((code & 0x8000u) << 1));
        else { atomicAdd(&s_size[at], 1u); if (code & 0x8000u) atomicAdd(&s_tp1[at], 1u); }
    };
    // what a scaffold puts on bin t_s of one dimension: tp1 += its dps at or below t_s, tot += n, bp += len
    auto add_scaf = [&](int at, uint32_t n, uint32_t le, unsigned long long len) {
        if (PACKED) { atomicAdd(&s_tot[at], n | (le << 16)); atomicAdd(&s_bp32[at], (uint32_t)(len < 500000ull ? len : 500000ull)); }
        else { atomicAdd(&s_tp1[at], le); atomicAdd(&s_tot[at], n); atomicAdd(&s_bp[at], len); }
    };
    if (!list) {
        // The tile is a RANGE of scaffold ids: its dps are one contiguous run, and with the flags and the counts above the flip
        // value precomputed nothing a dp needs depends on its scaffold.  Two loops without a dependent load: a thread per dp
        // (16-byte loads side by side), then a thread per scaffold for what lands on bin t_s.
        const int4 r0 = srec[begin], r1 = srec[end - 1];
        for (int dp = r0.x + (int)threadIdx.x; dp < r1.x + r1.y; dp += 256) {
            typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
            const u32x4s c = __builtin_nontemporal_load(reinterpret_cast<const u32x4s*>(codes + dp));    // read once per node: streamed
            const uint32_t cw[4] = {c[0], c[1], c[2], c[3]};
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (q < G.cnt) {
                    const uint32_t code = (cw[q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
                    add_dp(((G.loff[q] + (int)(code & 0x7FFFu)) << lg) + mine, code);
                }
            }
        }
        for (int s = begin + (int)threadIdx.x; s < end; s += 256) {
            const int4 rec = srec[s];
            const uint4 t8 = tsv[s];
            const int f = rec.x, n = rec.y;
            const uint32_t tw[4] = {t8.x, t8.y, t8.z, t8.w};
            const unsigned long long len = (unsigned long long)(uint32_t)rec.z | ((unsigned long long)(uint32_t)rec.w << 32);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (q < G.cnt) {
                    const uint32_t e = (tw[q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
                    const int ts = (int)(e & 0x3FFu);
                    uint32_t gt = e >> 10;
                    if (gt == 63u) {                                     // a long scaffold: count
                        gt = 0;
                        for (int i = 0; i < n; i++) {
                            const uint4 c = codes[f + i];
                            const uint32_t w = (q >> 1) == 0 ? c.x : (q >> 1) == 1 ? c.y : (q >> 1) == 2 ? c.z : c.w;
                            gt += (w >> (16 * (q & 1) + 15)) & 1u;
                        }
                    }
                    add_scaf(((G.loff[q] + ts) << lg) + mine, (uint32_t)n, (uint32_t)n - gt, len);
                    if (scg_lo && P.n_scg > 0) {
                        for (int k = P.scg_off[s]; k < P.scg_off[s + 1]; k++) {
                            const int g = P.scg_ids[k];
                            atomicMin(&scg_lo[(int64_t)G.dim[q] * P.n_scg + g], ts);
                            atomicMax(&scg_hi[(int64_t)G.dim[q] * P.n_scg + g], ts);
                        }
                    }
                }
            }
        }
    } else
    for (int ls = begin + threadIdx.x; ls < end; ls += 256) {
        const int s = list[ls];
        const int4 rec = srec[s];
        const uint4 t8 = tsv[s];
        const int f = rec.x, n = rec.y;
        uint32_t le[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i0 = 0; i0 < n; i0 += 4) {
            uint4 c[4];
#pragma unroll
            for (int i = 0; i < 4; i++) c[i] = (i0 + i < n) ? codes[f + i0 + i] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (i0 + i < n) {
                    const uint32_t cw[4] = {c[i].x, c[i].y, c[i].z, c[i].w};
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        if (q < G.cnt) {
                            const uint32_t code = (cw[q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
                            add_dp(((G.loff[q] + (int)(code & 0x7FFFu)) << lg) + mine, code);
                            if (!(code & 0x8000u)) le[q]++;
                        }
                    }
                }
            }
        }
        const uint32_t tw[4] = {t8.x, t8.y, t8.z, t8.w};
        const unsigned long long len = (unsigned long long)(uint32_t)rec.z | ((unsigned long long)(uint32_t)rec.w << 32);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (q < G.cnt) add_scaf(((G.loff[q] + (int)((tw[q >> 1] >> (16 * (q & 1))) & 0x3FFu)) << lg) + mine, (uint32_t)n, le[q], len);
        }
        if (scg_lo && P.n_scg > 0) {
            for (int e = P.scg_off[s]; e < P.scg_off[s + 1]; e++) {
                const int g = P.scg_ids[e];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    if (q < G.cnt) {
                        const int ts = (int)((tw[q >> 1] >> (16 * (q & 1))) & 0x3FFu);
                        atomicMin(&scg_lo[(int64_t)G.dim[q] * P.n_scg + g], ts);
                        atomicMax(&scg_hi[(int64_t)G.dim[q] * P.n_scg + g], ts);
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < G.total; b += 256) {
        unsigned long long a = 0, e = 0;
        uint32_t c3 = 0;
        for (int c = 0; c < copies; c++) {
            const int i = (b << lg) + ((c + b) & (copies - 1));          // rotated: the lanes of a wave read different banks
            if (PACKED) {
                const uint32_t w0 = s_size[i], w1 = s_tot[i];
                a += (unsigned long long)(w0 & 0xFFFFu) | ((unsigned long long)((w0 >> 16) + (w1 >> 16)) << 32);
                c3 += w1 & 0xFFFFu;
                e += s_bp32[i];
            } else {
                a += (unsigned long long)s_size[i] | ((unsigned long long)s_tp1[i] << 32);
                c3 += s_tot[i];
                e += s_bp[i];
            }
        }
        if (a | c3 | e) flush(b, a, c3, e);
    }
}
