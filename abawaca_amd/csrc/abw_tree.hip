// Stage 2, the whole recursion in one call: abw_sep_tree = the work-list loop of reference src/abawaca.cpp:95-197 around
// ClusterSeparator::separate() (src/ClusterSeparator.cpp:25-135, ...BySensitivitySpecificity.cpp:91-161,
// ClusterQuality.cpp:96-136), run LEVEL BY LEVEL on the resident table of abw_sep_load.
//
// The reference takes the node with the smallest id from a std::map and gives the children of a split the next two ids
// (src/abawaca.cpp:98-100,155-194): nodes are therefore visited in breadth-first order, and the nodes of one level can be
// evaluated together -- one set of launches per LEVEL instead of per node (abw_sep_node costs ~14 launches and one
// synchronisation per node, and most nodes of a deep recursion are small).  Same integer formulation as abw_sep.hip, plus
// three observations:
//   1. A node always holds WHOLE scaffolds (ref src/ClusterSeparator.cpp:104-122), so the flip value t_s of a scaffold in a
//      dimension -- the (n_s/2 + 1)-th smallest of its own values -- is the same in every node of the recursion.  It is
//      computed once per table (tree_static_*): per node nothing is sorted or selected any more.
//   2. With static flip values a node's histograms are plain sums over its scaffolds, hence
//      hist(bigger child) = hist(parent) - hist(smaller child): only the SMALLER child of every split is accumulated from
//      the table, its sibling comes from a subtraction of two histogram slices.  Every dp is accumulated O(log) times
//      over the whole recursion instead of once per level of a (typically very unbalanced) tree.
//   3. Wide dimensions (coverage) need NO histogram.  Each keeps its dps sorted by (node, bin, dp): initially the table's
//      static (bin, dp) order, after every level a stable partition of each node's segment into its two children.  A
//      node's thresholds in ascending order are then the run ends of its segment and size1 / TP1 / tot1 / bp1 are prefix
//      sums along it: two passes over 8 B per position (tile sums; rescan + evaluate).  What a position contributes is
//      static, too, and travels in the position's word: scaffold | [v >= t_s] << 31 | [owner] << 30, the owner being the
//      first dp of its scaffold on bin t_s, which also carries what the scaffold puts on that bin (one 16-byte gather).
//   Narrow dimensions (keys < 1024) keep their histograms: LDS-private per tile of scaffolds, flushed into the slice of
//   the tile's node; one workgroup per (node, dimension) scans the <= 1024 bins and evaluates every threshold.
//   The scaffolds of a node deep in the tree lie scattered over the table, and in the [dimension][dp] layout each
//   (scaffold, dimension) would cost a whole cache line for 16 bytes.  The narrow columns are therefore re-coded once per
//   table in groups of up to 8 dimensions: [group][dp][8] x 16 bit = value | (value > t_s) << 15 -- one 16-byte load per
//   dp and group, a scaffold's dps contiguous -- with the flip values as [group][scaffold][8] x 16 bit beside them.
//   Then per node the winner over the dimensions, the scaffold sides, the children's sizes: one small read-back per level.
#include "abw_sep_common.h"
#include <string.h>
#include <algorithm>

namespace {

// Tiles are cut at ABSOLUTE multiples of 2048 in the arrays they index (a node's first and last tile are partial), and the
// per-dimension stride of the position arrays is a multiple of 8: thread t of a tile owns the 8 consecutive items at
// (tile & ~2047) + 8 t, 32-byte aligned -- two 16-byte loads per array, a wave reads 2 KB in one piece.
constexpr int TILE_S = 2048;     // scaffolds per tile (256 threads x 8)
constexpr int TILE_P = 2048;     // sorted positions per tile (256 threads x 8 consecutive)
constexpr uint32_t PACK_ESCAPE = 0xFFFFFFFFu;

struct NodeDev {
    int32_t scaf_begin, scaf_end;       // its scaffolds in `order`
    int32_t dp_begin, dp_end;           // its SEGMENT in every wide dimension's sorted arrays: its own positions and dead ones
    int32_t n;                          // its datapoints (the live positions of the segment)
    int32_t stile_begin, stile_end;     // its scaffold tiles
    int32_t ptile_begin, ptile_end;     // its position tiles
};
struct Tile { int32_t node, begin, end; };

struct NodeRes {                        // per node of a level, device -> host
    GlobalRes win;
    unsigned long long n1, n2;          // dps of the scaffolds sent to cluster1 / cluster2
    uint32_t m1, m2;                    // scaffolds sent to cluster1 / cluster2
};
struct NodePart {                       // per node, host -> device for the partition
    int32_t split, id;
    int32_t scaf_base1, scaf_base2;     // where the children's scaffolds start in the next level's order
    int32_t dp_base1, dp_base2;         // ... and where their positions are copied to in the sorted arrays (-1: that child KEEPS the
                                        // node's segment; the positions of its sibling are marked dead there)
    int32_t slot1, slot2;               // the children's slots in the next level
    int32_t bits, reserved;             // bits: the positions that leave are marked in the bit map (tree_mark_kernel), not looked up
};

struct WSums { unsigned long long tp1, tot, bp; };
struct WSums4 { unsigned long long size, tp1, tot, bp; };

struct TreeDev {
    SepDev P;
    const int32_t* narrow; int32_t n_narrow; const int64_t* toffn; int64_t TBn;       // narrow dims, their offsets inside a node's slice
    const int32_t* wide; int32_t n_wide;
    const int32_t* dp2scaf;
    const int32_t* dim_slot;            // [D]: narrow: group * 8 + lane in the group; wide: -(index among the wide ones) - 1
    const NGroup* groups; int32_t n_groups;
    const uint4* nk16;                  // [n_groups][N]: 8 x 16 bit per dp: value | (value > t_s) << 15
    const uint4* ts8;                   // [n_groups][S]: 8 x 16 bit per scaffold: t_s
    const int4* srec;                   // [S]: first dp, dps, length (lo, hi)
    const uint4* wext;                  // [n_wide][S]: x = dps below t_s, y = dps, z = min(length, 500000), w = bin of t_s
    const uint32_t* wpack;              // [n_wide][S]: x | y << 6 | z << 13 where y <= 127 (4 MB per million scaffolds: L2), else PACK_ESCAPE
    const uint32_t* sinfo;              // the INLINE form of the position words (at most 2^22 scaffolds of at most 63 dps each): [S] y | z << 7,
                                        // and x of the dimension sits in the word itself -- no gather from a table per dimension.  NULL: not in use
    const uint8_t* stot;                // [S] y alone (1 MB per million scaffolds: stays in the L2), for the screen, which does not need z
    uint32_t sc_mask;                   // the scaffold of a position word: W_SC, or W_SC22 in the inline form
    int32_t D, S;
};

// W_END (static, abw_sep_node's scan over the table's own order only): the last position of its bin in the table
// W_DEAD (the recursion's own copies only): the position's dp has left the node whose segment this is (it went to the
// smaller child, which was copied out; the bigger child keeps the segment)
constexpr uint32_t W_GE = 0x80000000u, W_OWN = 0x40000000u, W_END = 0x20000000u, W_DEAD = 0x10000000u, W_SC = 0x0FFFFFFFu;
// the inline form: bits 22..27 of a W_OWN word hold the scaffold's dps below its flip value in this dimension (<= 31)
constexpr uint32_t W_SC22 = 0x003FFFFFu;
constexpr int W_X_SHIFT = 22;

__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long* sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// exclusive prefix of v over the 256 threads of the block (sh: 5 words)
__device__ __forceinline__ unsigned long long block_excl_scan_u64(unsigned long long v, unsigned long long* sh, unsigned long long* total)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned long long x = wave_prefix_u64(v);
    __syncthreads();
    if (lane == 63) sh[wid] = x;
    __syncthreads();
    unsigned long long off = 0;
    for (int k = 0; k < wid; k++) off += sh[k];
    if (total) *total = sh[0] + sh[1] + sh[2] + sh[3];
    return off + x - v;
}

// the 8 consecutive items of this thread in an aligned tile [begin, end); i0 is a multiple of 8 and p + i0 32-byte aligned
template <typename T>
__device__ __forceinline__ void load8(const T* __restrict__ p, int i0, int begin, int end, T (&v)[8], T fill)
{
    static_assert(sizeof(T) == 4, "32-bit items");
    if (i0 >= begin && i0 + 8 <= end) {
        const uint4 a = *reinterpret_cast<const uint4*>(p + i0), b = *reinterpret_cast<const uint4*>(p + i0 + 4);
        v[0] = (T)a.x; v[1] = (T)a.y; v[2] = (T)a.z; v[3] = (T)a.w; v[4] = (T)b.x; v[5] = (T)b.y; v[6] = (T)b.z; v[7] = (T)b.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = (i0 + i >= begin && i0 + i < end) ? p[i0 + i] : fill;
    }
}

// the same for arrays that are streamed once per kernel beside tables that are gathered from (the per-scaffold records of a
// wide dimension): loaded with the non-temporal hint, so that the stream does not push the tables out of the L2
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ void load8_stream(const T* __restrict__ p, int i0, int begin, int end, T (&v)[8], T fill)
{
    static_assert(sizeof(T) == 4, "32-bit items");
    if (i0 >= begin && i0 + 8 <= end) {
        const u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + i0)), b = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + i0 + 4));
        v[0] = (T)a.x; v[1] = (T)a.y; v[2] = (T)a.z; v[3] = (T)a.w; v[4] = (T)b.x; v[5] = (T)b.y; v[6] = (T)b.z; v[7] = (T)b.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = (i0 + i >= begin && i0 + i < end) ? p[i0 + i] : fill;
    }
}

// ---- level set-up: the SCG lo / hi lists of every (node, dimension) start empty ------------------------------------
__global__ void tree_scg_reset_kernel(int32_t* __restrict__ lo, int32_t* __restrict__ hi, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { lo[i] = INT_MAX; hi[i] = -1; }
}

// ---- the static tables (once per loaded table) ------------------------------------------------------------------------
// wide dimensions: flip value (as a bin), dps below it, one byte per dp (bit 0: v >= t_s; bit 1: the first dp of the
// scaffold on bin t_s).  grid (ceil(S / 256), wide dimensions)
__global__ __launch_bounds__(256) void tree_static_wide_kernel(SepDev P, const int32_t* __restrict__ wide, int32_t S, uint8_t* __restrict__ flags,
                                                              uint4* __restrict__ wext, uint32_t* __restrict__ wpack, uint32_t* __restrict__ sinfo)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    const int w = blockIdx.y;
    const int d = wide[w];
    const int32_t* __restrict__ col = P.bins + (int64_t)d * P.N;
    uint8_t* __restrict__ fl = flags + (int64_t)w * P.N;
    const int f = P.scaf_first[s], n = P.scaf_ndps[s], k = n >> 1;
    int ts, nb = 0;
    bool owned = false;
    if (n <= 8) {
        int v[8], x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { v[i] = (i < n) ? col[f + i] : INT_MAX; x[i] = v[i]; }
        ABW_SORT8(x);
        ts = (k == 1) ? x[1] : (k == 2) ? x[2] : (k == 3) ? x[3] : x[4];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (i < n) {
                nb += v[i] < ts;
                const bool own = !owned && v[i] == ts;
                owned |= own;
                fl[f + i] = (uint8_t)((v[i] >= ts ? 1 : 0) | (own ? 2 : 0));
            }
        }
    } else {
        ts = select_kth_mem(col + f, n, k, P.T[d]);
        for (int i = 0; i < n; i++) {
            const int v = col[f + i];
            nb += v < ts;
            const bool own = !owned && v == ts;
            owned |= own;
            fl[f + i] = (uint8_t)((v >= ts ? 1 : 0) | (own ? 2 : 0));
        }
    }
    const long long len = P.scaf_len[s];
    const uint32_t lenc = (uint32_t)(len < 500000 ? (len < 0 ? 0 : len) : 500000);
    wext[(int64_t)w * S + s] = make_uint4((uint32_t)nb, (uint32_t)n, lenc, (uint32_t)ts);
    wpack[(int64_t)w * S + s] = (n <= 127) ? ((uint32_t)nb | (uint32_t)n << 6 | lenc << 13) : PACK_ESCAPE;      // nb <= n / 2 <= 63
    if (sinfo && w == 0) { sinfo[s] = (uint32_t)n | lenc << 7; reinterpret_cast<uint8_t*>(sinfo + S)[s] = (uint8_t)n; }      // [S] words, then [S] bytes
}

// the word of every sorted position: scaffold | flags.  grid (blocks, wide dimensions)
__global__ void tree_static_words_kernel(const int32_t* __restrict__ sdp, const int32_t* __restrict__ sbin, const uint8_t* __restrict__ flags,
                                         const int32_t* __restrict__ dp2scaf, int64_t N, int64_t Np, const uint4* __restrict__ wext, int32_t S, int inl,
                                         uint32_t* __restrict__ wword)
{
    const int64_t base = (int64_t)blockIdx.y * N, obase = (int64_t)blockIdx.y * Np;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const int dp = sdp[base + i];
        const uint8_t f = flags[base + dp];
        const bool end = (i + 1 == N) || sbin[base + i + 1] != sbin[base + i];
        const int sc = dp2scaf[dp];
        uint32_t word = (uint32_t)sc | ((f & 1) ? W_GE : 0u) | ((f & 2) ? W_OWN : 0u) | (end ? W_END : 0u);
        if (inl && (f & 2)) word |= wext[(int64_t)blockIdx.y * S + sc].x << W_X_SHIFT;
        wword[obase + i] = word;
    }
}

// ---- narrow dimensions ----------------------------------------------------------------------------------------------
// Histograms of one tile of scaffolds in one GROUP of narrow dimensions (accum_group_tile, abw_sep_common.h), flushed with
// one atomic per non-empty bin into the slice of the tile's node.  grid (tiles to accumulate, groups)
template <bool PACKED>
__global__ __launch_bounds__(256) void tree_accum_kernel(TreeDev Tr, const Tile* __restrict__ tiles, const int32_t* __restrict__ order,
                                                        unsigned long long* __restrict__ h_st, unsigned long long* __restrict__ h_bp,
                                                        uint32_t* __restrict__ h_tot)
{
    const Tile t = tiles[blockIdx.x];
    const int j = blockIdx.y;
    const NGroup G = Tr.groups[j];
    const int64_t at = (int64_t)t.node * Tr.TBn + G.goff;
    accum_group_tile<PACKED>(G, Tr.nk16 + (int64_t)j * Tr.P.N, Tr.ts8 + (int64_t)j * Tr.S, Tr.srec, order, t.begin, t.end, Tr.P, nullptr, nullptr,
                     [&](int b, unsigned long long a, uint32_t c3, unsigned long long e) {
                         int lo = 0, to = 0;                            // the bin's dimension: from the LDS layout (spans) to the slice's (bins)
#pragma unroll
                         for (int i = 1; i < 8; i++) if (i < G.cnt && b >= G.loff[i]) { lo = G.loff[i]; to = G.toff[i]; }
                         const int64_t x = at + to + (b - lo);
                         if (a) atomicAdd(&h_st[x], a);
                         if (c3) atomicAdd(&h_tot[x], c3);
                         if (e) atomicAdd(&h_bp[x], e);
                     });
}

// slices of the nodes that are accumulated start at zero.  grid (nodes to zero, ceil(TBn / 1024))
__global__ __launch_bounds__(256) void tree_zero_kernel(const int32_t* __restrict__ slots, int64_t TBn, unsigned long long* __restrict__ h_st,
                                                       unsigned long long* __restrict__ h_bp, uint32_t* __restrict__ h_tot)
{
    const int64_t at = (int64_t)slots[blockIdx.x] * TBn;
    for (int64_t b = (int64_t)blockIdx.y * 1024 + threadIdx.x; b < std::min<int64_t>(TBn, (int64_t)(blockIdx.y + 1) * 1024); b += 256) {
        h_st[at + b] = 0; h_bp[at + b] = 0; h_tot[at + b] = 0;
    }
}

// hist(bigger child) = hist(parent) - hist(smaller child), slice by slice (the packed size | tp1 words subtract without a
// borrow between the halves: the child's counts never exceed the parent's).  grid (pairs, ceil(TBn / 1024))
struct SubJob { int32_t big, parent, small; };
__global__ __launch_bounds__(256) void tree_sub_kernel(const SubJob* __restrict__ jobs, int64_t TBn, unsigned long long* __restrict__ c_st,
                                                      unsigned long long* __restrict__ c_bp, uint32_t* __restrict__ c_tot,
                                                      const unsigned long long* __restrict__ p_st, const unsigned long long* __restrict__ p_bp,
                                                      const uint32_t* __restrict__ p_tot)
{
    const SubJob j = jobs[blockIdx.x];
    const int64_t big = (int64_t)j.big * TBn, par = (int64_t)j.parent * TBn, sm = (int64_t)j.small * TBn;
    for (int64_t b = (int64_t)blockIdx.y * 1024 + threadIdx.x; b < std::min<int64_t>(TBn, (int64_t)(blockIdx.y + 1) * 1024); b += 256) {
        c_st[big + b] = p_st[par + b] - c_st[sm + b];
        c_bp[big + b] = p_bp[par + b] - c_bp[sm + b];
        c_tot[big + b] = p_tot[par + b] - c_tot[sm + b];
    }
}

// SCG lo / hi of every (node, dimension): min / max of the static flip values over the node's scaffolds that carry the
// SCG.  One thread per (carrier scaffold, dimension); scaf_slot = the scaffold's node in this level, -1 when it has none.  K: nodes of the level.
__global__ __launch_bounds__(256) void tree_scg_kernel(TreeDev Tr, const int32_t* __restrict__ carriers, int32_t n_carriers, const int32_t* __restrict__ scaf_slot,
                                                      int32_t K, int32_t* __restrict__ scg_lo, int32_t* __restrict__ scg_hi)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n_carriers * Tr.D) return;
    const int s = carriers[idx / Tr.D], d = (int)(idx % Tr.D);
    const int k = scaf_slot[s];
    if (k < 0 || k >= K) return;                              // no node, or a node the cut work list does not reach
    const int slot = Tr.dim_slot[d];
    const int ts = slot >= 0 ? (int)(reinterpret_cast<const uint16_t*>(Tr.ts8)[((int64_t)(slot >> 3) * Tr.S + s) * 8 + (slot & 7)] & 0x3FFu)
                             : (int)Tr.wext[(int64_t)(-slot - 1) * Tr.S + s].w;
    const int64_t sg = ((int64_t)k * Tr.D + d) * Tr.P.n_scg;
    for (int e = Tr.P.scg_off[s]; e < Tr.P.scg_off[s + 1]; e++) {
        const int g = Tr.P.scg_ids[e];
        atomicMin(&scg_lo[sg + g], ts);
        atomicMax(&scg_hi[sg + g], ts);
    }
}

// the same lists with the minima / maxima taken in LDS: one workgroup per dimension holds the lists of all K nodes of the
// level (K x n_scg <= SCG_LEVEL_LDS entries), walks the carriers once and writes every list whole -- no reset pass, no
// global atomics.  grid (dimensions), 1024 threads: the walk is a chain of three dependent loads per carrier, so the more carriers
// in flight the better (256 threads: 47 us per level at 1M scaffolds)
constexpr int SCG_LEVEL_LDS = 6144;
__global__ __launch_bounds__(1024) void tree_scg_lds_kernel(TreeDev Tr, const int32_t* __restrict__ carriers, int32_t n_carriers, const int32_t* __restrict__ scaf_slot,
                                                          int32_t K, int32_t* __restrict__ scg_lo, int32_t* __restrict__ scg_hi)
{
    __shared__ int32_t s_lo[SCG_LEVEL_LDS], s_hi[SCG_LEVEL_LDS];
    const int d = blockIdx.x, n_scg = Tr.P.n_scg, total = K * n_scg;
    for (int i = threadIdx.x; i < total; i += (int)blockDim.x) { s_lo[i] = INT_MAX; s_hi[i] = -1; }
    __syncthreads();
    const int slot = Tr.dim_slot[d];
    for (int i = threadIdx.x; i < n_carriers; i += (int)blockDim.x) {
        const int s = carriers[i];
        const int k = scaf_slot[s];
        if (k < 0 || k >= K) continue;
        const int ts = slot >= 0 ? (int)(reinterpret_cast<const uint16_t*>(Tr.ts8)[((int64_t)(slot >> 3) * Tr.S + s) * 8 + (slot & 7)] & 0x3FFu)
                                 : (int)Tr.wext[(int64_t)(-slot - 1) * Tr.S + s].w;
        for (int e = Tr.P.scg_off[s]; e < Tr.P.scg_off[s + 1]; e++) {
            const int g = Tr.P.scg_ids[e];
            atomicMin(&s_lo[k * n_scg + g], ts);
            atomicMax(&s_hi[k * n_scg + g], ts);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < total; i += (int)blockDim.x) {
        const int64_t at = ((int64_t)(i / n_scg) * Tr.D + d) * n_scg + (i % n_scg);
        scg_lo[at] = s_lo[i]; scg_hi[at] = s_hi[i];
    }
}

constexpr int SCG_LDS = 256;
struct ScgView { const int32_t* lo; const int32_t* hi; int n; bool compact; };     // compact: lo / hi are LDS arrays holding the present SCGs only
typedef const __attribute__((address_space(3))) int32_t* LdsI32;

// One threshold of a (node, dimension): size check, small side, sensitivity x specificity in fp64 exactly as the
// reference's doubles, the SCG / size gate (ref ...Specificity.cpp:146-161, ClusterQuality.cpp:96-136).
__device__ __forceinline__ void eval_threshold(long long size1, long long tp1, long long tot1, unsigned long long bp1, long long n,
                                               unsigned long long bp_total, int32_t min_ndps, int bin, const ScgView& scg, Cand& best,
                                               double bar = 0.0)
{
    const long long size2 = n - size1, tot2 = n - tot1;
    const long long tp2 = n - (size1 + tot1 - tp1);
    if (size1 < min_ndps || size2 < min_ndps) return;              // ref :151
    const bool small1 = size1 < size2;                             // ref :155 (tie -> cluster2)
    const long long TP = small1 ? tp1 : tp2, TOT = small1 ? tot1 : tot2, SZ = small1 ? size1 : size2;
    {
        // Most thresholds lose by far against what is already known (this thread's best; `bar`: a product some other dimension
        // of the node has reached, which the node's winner must reach as well).  TP^2 / (TOT SZ) below that by more than
        // the rounding of the three products and of the reference's two divisions and product (< 1e-15 relative each) cannot
        // win and cannot tie: skipped without the two fp64 divisions.  Anything closer goes through the exact path.
        const double need = best.found ? (best.prod > bar ? best.prod : bar) : bar;
        const double tp = (double)TP;
        if (tp * tp < need * ((double)TOT * (double)SZ) * (1.0 - 1e-9)) return;
    }
    const double sens = (double)(int)TP / (double)(int)TOT;        // ref .h:75 (int fields)
    const double spec = (double)(int)TP / (double)(int)SZ;         // ref .h:74
    const double prod = sens * spec;
    if (!(prod == prod)) return;                                   // NaN is never better (ref .h:28)
    const Cand cnd = {prod, bin, 1, sens, spec, size1};
    if (!cand_better(cnd, best)) return;
    int cl = 0, chh = 0, ng = scg.n;
    if (scg.compact) {
        // the staged lists live in LDS: say so, or every read is a flat load that waits on both memory paths
        const LdsI32 lo3 = (LdsI32)scg.lo, hi3 = (LdsI32)scg.hi;
#pragma unroll 4
        for (int g = 0; g < scg.n; g++) { cl += lo3[g] <= bin; chh += hi3[g] <= bin; }
    } else {
        ng = 0;
        for (int g = 0; g < scg.n; g++) { const int l = scg.lo[g]; if (l != INT_MAX) { ng++; cl += l <= bin; chh += scg.hi[g] <= bin; } }
    }
    const int G1 = cl, G2 = ng - chh, both = cl - chh;
    bool pass;
    if (G1 == 0 || G2 == 0) pass = (bp1 >= 500000ull) && (bp_total - bp1 >= 500000ull);
    else pass = !((double)both / (double)(G1 < G2 ? G1 : G2) < 0.2);
    if (pass) best = cnd;
}

// The screen in front of eval_threshold for the wide dimensions' walks (one candidate per run end: tens of millions a level,
// of which a handful can win): the same size check and small side in 32-bit integers (all counts are below 2^31), and
// TP^2 against need x TOT x SZ in fp32 with a margin of 1e-5 -- far above what the five fp32 roundings and the rounding of
// `need` itself can add up to (< 1e-6) -- so that it only drops candidates the exact test below would drop as well.
__device__ __forceinline__ float screen_need(const Cand& best, double bar)
{
    const double need = best.found ? (best.prod > bar ? best.prod : bar) : bar;
    return (float)need * (1.0f - 1e-5f);
}
__device__ __forceinline__ bool screen32(uint32_t size1, uint32_t tp1, uint32_t tot1, uint32_t n, uint32_t min_ndps, float need)
{
    const uint32_t size2 = n - size1;
    if (size1 < min_ndps || size2 < min_ndps) return false;
    const bool small1 = size1 < size2;
    const uint32_t TP = small1 ? tp1 : n - (size1 + tot1 - tp1), TOT = small1 ? tot1 : n - tot1, SZ = small1 ? size1 : size2;
    const float tp = (float)TP;
    return !(tp * tp < need * ((float)TOT * (float)SZ));
}

// the same with the quantity itself returned (fp32, for ORDERING candidates only): TP^2 / (TOT x SZ) of the scored side, -1 where
// the sizes fail
__device__ __forceinline__ float screen_est32(uint32_t size1, uint32_t tp1, uint32_t tot1, uint32_t n, uint32_t min_ndps)
{
    const uint32_t size2 = n - size1;
    if (size1 < min_ndps || size2 < min_ndps) return -1.0f;
    const bool small1 = size1 < size2;
    const uint32_t TP = small1 ? tp1 : n - (size1 + tot1 - tp1), TOT = small1 ? tot1 : n - tot1, SZ = small1 ? size1 : size2;
    const float tp = (float)TP, den = (float)TOT * (float)SZ;
    return den > 0.0f ? tp * tp / den : 3.0e38f;
}

// SCGs present in the (node, dimension): lo != INT_MAX.  Up to SCG_LDS SCGs are compacted into LDS (the gate of a candidate
// then walks the present ones only); a table with more is read in place.  Call by all threads.
__device__ __forceinline__ ScgView stage_scgs(const int32_t* __restrict__ lo, const int32_t* __restrict__ hi, int n_scg, int32_t* s_lo, int32_t* s_hi, int* s_ng)
{
    if (n_scg > SCG_LDS) return ScgView{lo, hi, n_scg, false};
    if (threadIdx.x == 0) *s_ng = 0;
    __syncthreads();
    for (int g = threadIdx.x; g < n_scg; g += blockDim.x) {
        const int l = lo[g];
        if (l != INT_MAX) { const int slot = atomicAdd(s_ng, 1); s_lo[slot] = l; s_hi[slot] = hi[g]; }
    }
    __syncthreads();
    return ScgView{s_lo, s_hi, *s_ng, true};
}

__device__ __forceinline__ Cand block_best(Cand best, Cand* wbest)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const Cand other = cand_shfl_xor(best, o); if (cand_better(other, best)) best = other; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = best;
    __syncthreads();
    Cand b = wbest[0];
    for (int k = 1; k < 4; k++) if (cand_better(wbest[k], b)) b = wbest[k];
    return b;
}

// grid (nodes, narrow dimensions): scan the node's slice (T <= 1024 bins: 4 per thread), evaluate every threshold present,
// keep the best.  The slice stays: the node's children are derived from it.
__global__ __launch_bounds__(256) void tree_eval_narrow_kernel(TreeDev Tr, const NodeDev* __restrict__ nodes, int32_t min_ndps,
                                                              const unsigned long long* __restrict__ h_st, const unsigned long long* __restrict__ h_bp,
                                                              const uint32_t* __restrict__ h_tot, const int32_t* __restrict__ scg_lo,
                                                              const int32_t* __restrict__ scg_hi, Cand* __restrict__ cand)
{
    __shared__ unsigned long long sh[3][5];
    __shared__ int32_t s_lo[SCG_LDS], s_hi[SCG_LDS];
    __shared__ int s_ng;
    __shared__ Cand wbest[4];
    const int k = blockIdx.x, dn = blockIdx.y;
    const int d = Tr.narrow[dn];
    const int T = Tr.P.T[d];
    const NodeDev nd = nodes[k];
    const long long n = nd.n;
    const int64_t at = (int64_t)k * Tr.TBn + Tr.toffn[dn];
    const int64_t sg = ((int64_t)k * Tr.D + d) * Tr.P.n_scg;
    const ScgView scg = stage_scgs(scg_lo + sg, scg_hi + sg, Tr.P.n_scg, s_lo, s_hi, &s_ng);
    unsigned long long st[4], bp[4];
    uint32_t tot[4];
    unsigned long long a_size = 0, a_tp1 = 0, a_tot = 0, a_bp = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int t = 4 * threadIdx.x + i;
        st[i] = 0; bp[i] = 0; tot[i] = 0;
        if (t < T) {
            st[i] = h_st[at + t]; bp[i] = h_bp[at + t]; tot[i] = h_tot[at + t];
        }
        a_size += (uint32_t)st[i]; a_tp1 += st[i] >> 32; a_tot += tot[i]; a_bp += bp[i];
    }
    // size and tp1 fit 32 bits each (n < 2^31): one packed scan for the pair, one for tot | -, one for bp
    unsigned long long tot_st = 0, tot_bp = 0, tot_tt = 0;
    unsigned long long o_st = block_excl_scan_u64(a_size | (a_tp1 << 32), sh[0], &tot_st);
    unsigned long long o_tt = block_excl_scan_u64(a_tot, sh[1], &tot_tt);
    unsigned long long o_bp = block_excl_scan_u64(a_bp, sh[2], &tot_bp);
    Cand best = {0.0, 0, 0, 0.0, 0.0, 0};
    long long size1 = (long long)(uint32_t)o_st, tp1 = (long long)(o_st >> 32), tot1 = (long long)o_tt;
    unsigned long long bp1 = o_bp;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t here = (uint32_t)st[i];
        size1 += here; tp1 += (long long)(st[i] >> 32); tot1 += tot[i]; bp1 += bp[i];
        if (here) eval_threshold(size1, tp1, tot1, bp1, n, tot_bp, min_ndps, 4 * threadIdx.x + i, scg, best);
    }
    const Cand b = block_best(best, wbest);
    if (threadIdx.x == 0) cand[(int64_t)k * Tr.D + d] = b;
}

// the best product the narrow dimensions of every node have reached: the bar for the wide ones (see eval_threshold).
// grid (nodes), one wave
__global__ __launch_bounds__(64) void tree_bar_kernel(TreeDev Tr, const Cand* __restrict__ cand, double* __restrict__ bar)
{
    double b = 0.0;
    for (int i = threadIdx.x; i < Tr.n_narrow; i += 64) {
        const Cand c = cand[(int64_t)blockIdx.x * Tr.D + Tr.narrow[i]];
        if (c.found && c.prod > b) b = c.prod;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double y = __shfl_xor(b, o, 64); if (y > b) b = y; }
    if (threadIdx.x == 0) bar[blockIdx.x] = b;
}

// ---- wide dimensions ------------------------------------------------------------------------------------------------
// What the dps at a thread's 8 sorted positions add to the prefix sums of their node (all static: see tree_static_wide_kernel).
// First the packed records of the scaffolds that OWN one of the positions (x | y << 6 | z << 13; 0 where the position adds
// no record: not W_OWN, dead, or its bit of `use` clear; PACK_ESCAPE: a scaffold too large for the packed form) -- as eight
// independent loads with no branch around them: a load under a branch is waited for before the next is issued, and eight
// round trips in a row were most of these kernels' time.  A position without a record reads entry 0.
template <bool TOT_ONLY = false>
__device__ __forceinline__ void load_packs(const TreeDev& Tr, int w, const uint32_t (&wd)[8], uint32_t use, uint32_t (&pk)[8])
{
    uint32_t own = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) own |= (uint32_t)((wd[i] & (W_OWN | W_DEAD)) == W_OWN) << i;
    own &= use;
    if (TOT_ONLY && Tr.sinfo) {                                 // z left out (0)
        uint8_t y[8];
#pragma unroll
        for (int i = 0; i < 8; i++) y[i] = Tr.stot[((own >> i) & 1u) ? (wd[i] & W_SC22) : 0u];
#pragma unroll
        for (int i = 0; i < 8; i++) pk[i] = ((own >> i) & 1u) ? (((wd[i] >> W_X_SHIFT) & 63u) | (uint32_t)y[i] << 6) : 0u;
    } else if (Tr.sinfo) {
        uint32_t si[8];
#pragma unroll
        for (int i = 0; i < 8; i++) si[i] = Tr.sinfo[((own >> i) & 1u) ? (wd[i] & W_SC22) : 0u];
#pragma unroll
        for (int i = 0; i < 8; i++) pk[i] = ((own >> i) & 1u) ? (((wd[i] >> W_X_SHIFT) & 63u) | (si[i] & 127u) << 6 | (si[i] >> 7) << 13) : 0u;
    } else {
        const uint32_t* __restrict__ tab = Tr.wpack + (int64_t)w * Tr.S;
        uint32_t v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = tab[((own >> i) & 1u) ? (wd[i] & W_SC) : 0u];
#pragma unroll
        for (int i = 0; i < 8; i++) pk[i] = ((own >> i) & 1u) ? v[i] : 0u;
    }
}
// ... then the terms of one position from its word and its record (live: the position counts at all)
__device__ __forceinline__ WSums pack_wsums(const TreeDev& Tr, int w, uint32_t word, uint32_t pk, bool live)
{
    WSums r = {live ? (unsigned long long)(word >> 31) : 0ull, 0, 0};
    if (pk == PACK_ESCAPE) { const uint4 e = Tr.wext[(int64_t)w * Tr.S + (word & W_SC)]; r.tp1 += e.x; r.tot = e.y; r.bp = e.z; }      // rare
    else { r.tp1 += pk & 63u; r.tot = (pk >> 6) & 127u; r.bp = pk >> 13; }
    return r;
}
__device__ __forceinline__ void pack_terms(uint32_t word, uint32_t pk, uint32_t& tp, uint32_t& tot)      // pk != PACK_ESCAPE
{
    tp = ((word & W_DEAD) ? 0u : (word >> 31)) + (pk & 63u);
    tot = (pk >> 6) & 127u;
}
// one byte per scaffold (a node's members, the sides of a split) for the scaffolds of 8 position words, the same way
__device__ __forceinline__ void load_scaf_bytes(const uint8_t* __restrict__ tab, const uint32_t (&wd)[8], uint32_t use, uint32_t sc_mask, uint8_t (&out)[8])
{
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = tab[((use >> i) & 1u) ? (wd[i] & sc_mask) : 0u];
}

__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t* sh)          // sh: 4 words
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t x = wave_prefix_u32(v);
    __syncthreads();
    if (lane == 63) sh[wid] = x;
    __syncthreads();
    uint32_t off = 0;
    for (int k = 0; k < wid; k++) off += sh[k];
    return off + x - v;
}

// pass 1: sums of every position tile (live positions only; a tile's count and tot share a word: both < 2^32).
// grid (position tiles, wide dimensions)
// grid (tiles to sum, wide dimensions); which: the tiles' indices (NULL: all of them); the sums go to tile_sums[w * ostride + tile]
__global__ __launch_bounds__(256) void tree_wide_sums_kernel(TreeDev Tr, const Tile* __restrict__ tiles, const int32_t* __restrict__ which, const uint32_t* __restrict__ wword,
                                                            int64_t wstride, WSums4* __restrict__ tile_sums, int64_t ostride)
{
    __shared__ unsigned long long sh[8];
    const int tile = which ? which[blockIdx.x] : (int)blockIdx.x;
    const Tile t = tiles[tile];
    const int w = blockIdx.y;
    const uint32_t* __restrict__ words = wword + (int64_t)w * wstride;
    WSums4 a = {0, 0, 0, 0};
    const int i0 = (t.begin & ~(TILE_P - 1)) + 8 * threadIdx.x;
    uint32_t wd[8];
    load8_stream(words, i0, t.begin, t.end, wd, W_DEAD);             // a dead word adds nothing
    uint32_t pk[8];
    load_packs(Tr, w, wd, 0xFFu, pk);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const bool live = !(wd[i] & W_DEAD);
        const WSums r = pack_wsums(Tr, w, wd[i], pk[i], live);
        a.size += live ? 1u : 0u; a.tp1 += r.tp1; a.tot += r.tot; a.bp += r.bp;
    }
    const unsigned long long s0 = block_sum_u64(a.size | (a.tot << 32), sh);           // inside a tile every sum fits 32 bits (see the eval kernel)
    const unsigned long long s1 = block_sum_u64(a.tp1 | (a.bp << 32), sh + 4);
    if (threadIdx.x == 0) tile_sums[(int64_t)w * ostride + tile] = WSums4{s0 & 0xFFFFFFFFull, s1 & 0xFFFFFFFFull, s0 >> 32, s1 >> 32};
}

// pass 1a for the tiles of a node that KEPT its parent's segment: last level's sums less what left in last level's partition
// (tree_part_scatter_kernel leaves that per tile) -- the giant of a caterpillar is not summed again.  grid (blocks), src[tile] < 0: a fresh tile
__global__ void tree_wide_inherit_kernel(const int32_t* __restrict__ src, int32_t n_tiles, int32_t n_wide, const WSums4* __restrict__ prev,
                                         const WSums4* __restrict__ left, WSums4* __restrict__ cur, int64_t stride)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)n_tiles * n_wide; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i / n_tiles), t = (int)(i - (int64_t)w * n_tiles);
        const int from = src[t];
        if (from < 0) continue;
        const WSums4 a = prev[(int64_t)w * stride + from], b = left[(int64_t)w * stride + from];
        cur[(int64_t)w * stride + t] = WSums4{a.size - b.size, a.tp1 - b.tp1, a.tot - b.tot, a.bp - b.bp};
    }
}

// pass 1b: exclusive scan of the tile sums inside every node; node totals.  grid (nodes, wide dimensions), one wave
__global__ __launch_bounds__(64) void tree_wide_scan_kernel(const NodeDev* __restrict__ nodes, int32_t n_tiles, const WSums4* __restrict__ raw, int64_t rstride,
                                                           WSums4* __restrict__ tile_sums, WSums4* __restrict__ node_total, int32_t K)
{
    const int k = blockIdx.x, w = blockIdx.y, lane = threadIdx.x;
    const NodeDev nd = nodes[k];
    const WSums4* __restrict__ in = raw + (int64_t)w * rstride;
    WSums4* ts = tile_sums + (int64_t)w * n_tiles;
    WSums4 run = {0, 0, 0, 0};
    const WSums4 none = {0, 0, 0, 0};
    WSums4 nxt = (nd.ptile_begin + lane < nd.ptile_end) ? in[nd.ptile_begin + lane] : none;      // the next 64 tiles are on their way during the scan
    for (int i0 = nd.ptile_begin; i0 < nd.ptile_end; i0 += 64) {
        const int i = i0 + lane;
        const WSums4 v = nxt;
        nxt = (i + 64 < nd.ptile_end) ? in[i + 64] : none;
        const WSums4 x = {wave_prefix_u64(v.size), wave_prefix_u64(v.tp1), wave_prefix_u64(v.tot), wave_prefix_u64(v.bp)};
        if (i < nd.ptile_end) ts[i] = WSums4{run.size + x.size - v.size, run.tp1 + x.tp1 - v.tp1, run.tot + x.tot - v.tot, run.bp + x.bp - v.bp};
        run.size += wave_last_u64(x.size); run.tp1 += wave_last_u64(x.tp1); run.tot += wave_last_u64(x.tot); run.bp += wave_last_u64(x.bp);
    }
    if (lane == 0) node_total[(int64_t)w * K + k] = run;
}

// Pass 2a', the screen of the screen: can ANY run end of a tile reach the node's bar?  Along a segment the four sums only grow,
// so inside a tile size1, TP1 and tot1 lie between the tile's carry and carry + the tile's own sums, and
// TP^2 / (TOT x SZ) of either side of the split is at most (largest TP)^2 / (smallest TOT x smallest SZ).  A tile whose bound
// stays below the bar (with the margin of screen32) is done without reading one of its positions: next to the best
// threshold of a dimension the bound is tight, everywhere else it is far below -- a few per cent of the tiles remain.
// The same for the size check (ref ...Specificity.cpp:151): no run end of the tile leaves both sides their minimum.
// c: the tile's carry, o: its own sums.  -> the tile must be looked at
__device__ __forceinline__ bool tile_may_reach(const WSums4& c, const WSums4& o, long long n, int32_t min_ndps, double bar)
{
    const long long size_lo = (long long)c.size, size_hi = (long long)(c.size + o.size);
    if (size_hi < min_ndps || n - size_lo < min_ndps) return false;            // size1 <= size_hi, size2 <= n - size_lo
    if (!(bar > 0.0)) return true;
    const double need = bar * (1.0 - 1e-5);
    const double tp_hi = (double)(c.tp1 + o.tp1), tot_lo = (double)c.tot, tot_hi = (double)(c.tot + o.tot), dn = (double)n;
    // Only the SMALLER side of a threshold is scored (ref :155; a tie scores side 2): side 1 where 2 size1 < n.
    // the side "<= threshold": TP1 <= tp_hi, tot1 >= tot_lo, size1 >= size_lo
    if (2 * size_lo < n) {
        const double den1 = tot_lo * (double)size_lo;
        if (!(den1 > 0.0) || tp_hi * tp_hi >= need * den1) return true;
    }
    // the other side: TP2 = n - size1 - tot1 + TP1, tot2 = n - tot1, size2 = n - size1
    if (2 * size_hi >= n) {
        const double tp2_hi = dn - (double)size_lo - tot_lo + tp_hi, den2 = (dn - tot_hi) * (dn - (double)size_hi);
        if (!(den2 > 0.0) || tp2_hi * tp2_hi >= need * den2) return true;
    }
    return false;
}

// grid: blocks over (position tiles x wide dimensions).  Lists the tiles to look at; every tile's best starts as "none".
__global__ __launch_bounds__(256) void tree_wide_tilescreen_kernel(const Tile* __restrict__ tiles, int32_t n_tiles, int32_t n_wide, const NodeDev* __restrict__ nodes,
                                                                  int32_t K, int32_t min_ndps, const WSums4* __restrict__ tile_carry, const WSums4* __restrict__ node_total,
                                                                  const double* __restrict__ bar, Cand* __restrict__ tile_best, int32_t* __restrict__ list,
                                                                  int32_t* __restrict__ list_n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    if (i < (int64_t)n_tiles * n_wide) {
        const int w = (int)(i / n_tiles), t = (int)(i - (int64_t)w * n_tiles);
        const Tile tl = tiles[t];
        const NodeDev nd = nodes[tl.node];
        const WSums4 c = tile_carry[i];
        const WSums4 e = (t + 1 < nd.ptile_end) ? tile_carry[i + 1] : node_total[(int64_t)w * K + tl.node];      // the carry behind the tile
        keep = tile_may_reach(c, WSums4{e.size - c.size, e.tp1 - c.tp1, e.tot - c.tot, e.bp - c.bp}, nd.n, min_ndps, bar[tl.node]);
        tile_best[i] = Cand{0.0, 0, 0, 0.0, 0.0, 0};
    }
    const unsigned long long m = __ballot(keep);
    if (m) {
        const int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == 0) base = atomicAdd(list_n, __popcll(m));
        base = __shfl(base, 0, 64);
        if (keep) list[base + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)i;
    }
}

// pass 2a, the screen: rescan every tile with its carry and ask at every run end (the last position of a bin inside the
// node's segment) whether its threshold could reach the node's bar at all (screen32).  The tiles where one does -- a few, once
// the narrow dimensions have set a bar -- are listed for pass 2b; the others are done.  Kept lean (no candidate, no SCG lists,
// no fp64: 8 waves per SIMD), because a block lives about as long as a handful of memory round trips.
// grid: any; the workgroups take the tiles listed by tree_wide_tilescreen_kernel in turn
__global__ __launch_bounds__(256) void tree_wide_screen_kernel(TreeDev Tr, const Tile* __restrict__ tiles, int32_t n_tiles, const NodeDev* __restrict__ nodes,
                                                              int32_t min_ndps, const int32_t* __restrict__ wbin, const uint32_t* __restrict__ wword,
                                                              int64_t wstride, const WSums4* __restrict__ tile_carry, const double* __restrict__ bar,
                                                              const int32_t* __restrict__ todo, const int32_t* __restrict__ todo_n,
                                                              int32_t* __restrict__ list, int32_t* __restrict__ list_n)
{
    __shared__ unsigned long long sh[5];
    __shared__ uint32_t sh32[4];
    const int n_todo = *todo_n;
  for (int li = blockIdx.x; li < n_todo; li += gridDim.x) {           // (a workgroup's turns are separated by the barrier of the vote below)
    const int entry = todo[li];
    const int w = entry / n_tiles, tile = entry - w * n_tiles;
    const Tile t = tiles[tile];
    const int32_t* __restrict__ bins = wbin + (int64_t)w * wstride;
    const uint32_t* __restrict__ words = wword + (int64_t)w * wstride;
    const int i0 = (t.begin & ~(TILE_P - 1)) + 8 * threadIdx.x;
    uint32_t wd[8], pk[8];
    int b[9];
    load8_stream(words, i0, t.begin, t.end, wd, W_DEAD);
    {
        int b8[8];
        load8_stream(bins, i0, t.begin, t.end, b8, -1);
#pragma unroll
        for (int i = 0; i < 8; i++) b[i] = b8[i];
    }
    const int next_bin = bins[min((int64_t)i0 + 8, wstride - 1)];   // (not behind the node's record: one round trip less)
    const NodeDev nd = nodes[t.node];
    const WSums4 carry = tile_carry[(int64_t)w * n_tiles + tile];
    const double node_bar = bar[t.node];
    load_packs<true>(Tr, w, wd, 0xFFu, pk);
    b[8] = (i0 + 8 < nd.dp_end) ? next_bin : -1;                  // -1 past the segment's end (where a partial tile ends, too): the last run ends there
    // all the walk needs of a position, in one register: live | run end << 1 | tp << 2 | tot << 9
    uint32_t info[8];
    uint32_t a_size = 0, a_tp = 0, a_tot = 0;
    bool hit = false;                                         // a tile with a scaffold too large for the packed form goes to pass 2b as it is
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t tp, tot;
        hit |= pk[i] == PACK_ESCAPE;
        pack_terms(wd[i], pk[i], tp, tot);
        const uint32_t live = (wd[i] & W_DEAD) ? 0u : 1u;
        const uint32_t run_end = (i0 + i >= t.begin && i0 + i < t.end && b[i + 1] != b[i]) ? 2u : 0u;      // threshold = this bin
        info[i] = live | run_end | tp << 2 | tot << 9;
        a_size += live; a_tp += tp; a_tot += tot;
    }
    // inside a tile every sum fits 32 bits (tp1, tot: a scaffold owns one position per dimension, so they stay below N + 2048)
    const unsigned long long o0 = block_excl_scan_u64((unsigned long long)a_size | ((unsigned long long)a_tot << 32), sh, nullptr);
    const uint32_t o1 = block_excl_scan_u32(a_tp, sh32);
    uint32_t size1 = (uint32_t)(carry.size + (o0 & 0xFFFFFFFFull)), tp1 = (uint32_t)carry.tp1 + o1, tot1 = (uint32_t)(carry.tot + (o0 >> 32));
    const uint32_t n32 = (uint32_t)nd.n, mn32 = (uint32_t)max(min_ndps, 0);
    const Cand none = {0.0, 0, 0, 0.0, 0.0, 0};
    const float need = screen_need(none, node_bar);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        size1 += info[i] & 1u; tp1 += (info[i] >> 2) & 127u; tot1 += (info[i] >> 9) & 127u;      // (nothing outside the tile: dead words there)
        hit |= (info[i] & 2u) && screen32(size1, tp1, tot1, n32, mn32, need);
        __builtin_amdgcn_sched_barrier(0);                      // one position after the other: eight at once cost 20 registers
    }
    const int any = __syncthreads_or(hit);
    if (threadIdx.x == 0 && any) list[atomicAdd(list_n, 1)] = entry;
  }
}

// pass 2b: the listed tiles again, with everything: evaluate the threshold at every run end (the last position of a bin inside
// the node's segment), keep the tile's best.  A run end whose bin holds only dead positions repeats its predecessor's sums
// at a larger threshold: never better (see node_wide_eval_kernel).  grid (position tiles, wide dimensions)
__global__ __launch_bounds__(256) void tree_wide_eval_kernel(TreeDev Tr, const Tile* __restrict__ tiles, int32_t n_tiles, const NodeDev* __restrict__ nodes,
                                                            int32_t K, int32_t min_ndps, const int32_t* __restrict__ wbin, const uint32_t* __restrict__ wword,
                                                            int64_t wstride, const WSums4* __restrict__ tile_carry, const WSums4* __restrict__ node_total,
                                                            const int32_t* __restrict__ scg_lo, const int32_t* __restrict__ scg_hi,
                                                            const double* __restrict__ bar, Cand* __restrict__ tile_best,
                                                            const int32_t* __restrict__ list, const int32_t* __restrict__ list_n)
{
    __shared__ unsigned long long sh[2][5];
    __shared__ int32_t s_lo[SCG_LDS], s_hi[SCG_LDS];
    __shared__ int s_ng;
    __shared__ Cand wbest[4];
    const int n_listed = *list_n;
  for (int li = blockIdx.x; li < n_listed; li += gridDim.x) {         // (a block's turns are separated by the barriers of block_best)
    const int entry = list[li];
    const int w = entry / n_tiles, tile = entry - w * n_tiles;
    const Tile t = tiles[tile];
    const int d = Tr.wide[w];
    const NodeDev nd = nodes[t.node];
    const long long n = nd.n;
    const int32_t* __restrict__ bins = wbin + (int64_t)w * wstride;
    const uint32_t* __restrict__ words = wword + (int64_t)w * wstride;
    const int i0 = (t.begin & ~(TILE_P - 1)) + 8 * threadIdx.x;
    int b[9];
    uint32_t wd[8];
    uint32_t r_tp[8], r_tot[8], r_bp[8];                 // one position's terms fit 32 bits each (dps of a scaffold; a clamped length)
    WSums4 a = {0, 0, 0, 0};
    // the tile's own loads first: a block lives about as long as a handful of memory round trips, and the SCG staging below
    // (a round trip, LDS atomics, two barriers) would otherwise stand in front of them
    load8_stream(words, i0, t.begin, t.end, wd, W_DEAD);
    {
        int b8[8];
        load8_stream(bins, i0, t.begin, t.end, b8, -1);
#pragma unroll
        for (int i = 0; i < 8; i++) b[i] = b8[i];
        b[8] = (i0 + 8 < nd.dp_end) ? bins[i0 + 8] : -1;            // -1 past the segment's end (where a partial tile ends, too): the last run ends there
    }
    const WSums4 carry = tile_carry[(int64_t)w * n_tiles + tile];
    const unsigned long long bp_total = node_total[(int64_t)w * K + t.node].bp;
    const double node_bar = bar[t.node];
    {
        uint32_t pk[8];
        load_packs(Tr, w, wd, 0xFFu, pk);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const bool live = !(wd[i] & W_DEAD);
            const WSums r = pack_wsums(Tr, w, wd[i], pk[i], live);
            r_tp[i] = (uint32_t)r.tp1; r_tot[i] = (uint32_t)r.tot; r_bp[i] = (uint32_t)r.bp;
            a.size += live ? 1u : 0u; a.tp1 += r.tp1; a.tot += r.tot; a.bp += r.bp;
        }
    }
    // inside a tile every sum fits 32 bits (tp1, tot: a scaffold owns one position per dimension, so they stay below
    // N + 2048; bp: 2048 lengths clamped to 500 000)
    const unsigned long long o0 = block_excl_scan_u64(a.size | (a.tot << 32), sh[0], nullptr);
    const unsigned long long o1 = block_excl_scan_u64(a.tp1 | (a.bp << 32), sh[1], nullptr);
    uint32_t size1 = (uint32_t)(carry.size + (o0 & 0xFFFFFFFFull)), tp1 = (uint32_t)(carry.tp1 + (o1 & 0xFFFFFFFFull)), tot1 = (uint32_t)(carry.tot + (o0 >> 32));
    unsigned long long bp1 = carry.bp + (o1 >> 32);
    Cand best = {0.0, 0, 0, 0.0, 0.0, 0};
    const uint32_t n32 = (uint32_t)n, mn32 = (uint32_t)max(min_ndps, 0);
    float need = screen_need(best, node_bar);
    const int64_t sg = ((int64_t)t.node * Tr.D + d) * Tr.P.n_scg;
    const ScgView scg = stage_scgs(scg_lo + sg, scg_hi + sg, Tr.P.n_scg, s_lo, s_hi, &s_ng);
    // The thread's run ends (threshold = the bin): the one that looks best first (fp32 estimate), then the others against what it
    // set -- near a dimension's best threshold the products rise from run end to run end, and in walking order every one of
    // a thread's eight would go through the two IEEE divisions and the gate.  Any order gives the same best (cand_better).
    uint32_t p_sz[8], p_tp[8], p_tot[8];
    unsigned long long p_bp[8];
    uint32_t todo = 0;
    float e_best = -2.0f;
    int i_best = -1;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        size1 += (wd[i] & W_DEAD) ? 0u : 1u; tp1 += r_tp[i]; tot1 += r_tot[i]; bp1 += r_bp[i];       // (nothing outside the tile: dead words there)
        p_sz[i] = size1; p_tp[i] = tp1; p_tot[i] = tot1; p_bp[i] = bp1;
        if (i0 + i >= t.begin && i0 + i < t.end && b[i + 1] != b[i] && screen32(size1, tp1, tot1, n32, mn32, need)) {
            todo |= 1u << i;
            const float e = screen_est32(size1, tp1, tot1, n32, mn32);
            if (e > e_best) { e_best = e; i_best = i; }
        }
    }
    if (i_best >= 0) {
        uint32_t sz = 0, tp = 0, tt = 0; unsigned long long bp = 0; int bin = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) if (i == i_best) { sz = p_sz[i]; tp = p_tp[i]; tt = p_tot[i]; bp = p_bp[i]; bin = b[i]; }
        eval_threshold((long long)sz, (long long)tp, (long long)tt, bp, n, bp_total, min_ndps, bin, scg, best, node_bar);
        need = screen_need(best, node_bar);
        todo &= ~(1u << i_best);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (((todo >> i) & 1u) && screen32(p_sz[i], p_tp[i], p_tot[i], n32, mn32, need)) {
            eval_threshold((long long)p_sz[i], (long long)p_tp[i], (long long)p_tot[i], p_bp[i], n, bp_total, min_ndps, b[i], scg, best, node_bar);
            need = screen_need(best, node_bar);
        }
    }
    const Cand bb = block_best(best, wbest);
    if (threadIdx.x == 0) tile_best[(int64_t)w * n_tiles + tile] = bb;
  }
}

// best tile of every (node, wide dimension).  grid (nodes, wide dimensions), one wave
__global__ __launch_bounds__(64) void tree_wide_best_kernel(TreeDev Tr, const NodeDev* __restrict__ nodes, int32_t n_tiles, const Cand* __restrict__ tile_best,
                                                           Cand* __restrict__ cand)
{
    const int k = blockIdx.x, w = blockIdx.y, lane = threadIdx.x;
    const NodeDev nd = nodes[k];
    Cand c = {0.0, 0, 0, 0.0, 0.0, 0};
    for (int t = nd.ptile_begin + lane; t < nd.ptile_end; t += 64) { const Cand o = tile_best[(int64_t)w * n_tiles + t]; if (cand_better(o, c)) c = o; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const Cand other = cand_shfl_xor(c, o); if (cand_better(other, c)) c = other; }
    if (lane == 0) cand[(int64_t)k * Tr.D + Tr.wide[w]] = c;
}

// ---- abw_sep_node's wide dimensions: the same prefix sums along the TABLE's own static (bin, dp) order -------------------
// One node, no partition: the positions of a wide dimension in the table's order, a node below the root picked out by one
// byte per scaffold (member; NULL: the root).  Thresholds are evaluated at the static run ends (W_END) whether or not the
// node holds a dp of that bin: a bin the node does not hold leaves all four sums, and the SCG counts (the lists hold flip
// values, which are values of the node), as its predecessor left them -- the same product at a larger threshold, which
// never replaces the smaller one (ref ...Specificity.h:25-30); before the node's first value size1 = 0 fails the size
// check, or 0 / 0 = NaN is never better.  The bin is read only for a candidate that gets as far as the gate.
template <bool MASKED>
__global__ __launch_bounds__(256) void node_wide_sums_kernel(TreeDev Tr, int32_t N, int32_t n_tiles, const uint32_t* __restrict__ wword, int64_t wstride,
                                                            const uint8_t* __restrict__ member, WSums4* __restrict__ tile_sums)
{
    __shared__ unsigned long long sh[8];
    const int w = blockIdx.y;
    const uint32_t* __restrict__ words = wword + (int64_t)w * wstride;
    const int begin = blockIdx.x * TILE_P, end = min(N, begin + TILE_P);
    const int i0 = begin + 8 * threadIdx.x;
    uint32_t wd[8];
    load8_stream(words, i0, begin, end, wd, 0u);
    WSums4 a = {0, 0, 0, 0};
    uint32_t in = 0;                                          // the positions of the tile that belong to the node
#pragma unroll
    for (int i = 0; i < 8; i++) in |= (uint32_t)(i0 + i < end) << i;
    if (MASKED) {
        uint8_t mb[8];
        load_scaf_bytes(member, wd, in, Tr.sc_mask, mb);
#pragma unroll
        for (int i = 0; i < 8; i++) if (!mb[i]) in &= ~(1u << i);
    }
    uint32_t pk[8];
    load_packs(Tr, w, wd, in, pk);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const bool live = (in >> i) & 1u;
        const WSums r = pack_wsums(Tr, w, wd[i], pk[i], live);
        a.size += live ? 1u : 0u; a.tp1 += r.tp1; a.tot += r.tot; a.bp += r.bp;
    }
    const unsigned long long s0 = block_sum_u64(a.size | (a.tot << 32), sh);          // a tile holds 2048 positions: every sum of a tile < 2^32
    const unsigned long long s1 = block_sum_u64(a.tp1 | (a.bp << 32), sh + 4);
    if (threadIdx.x == 0) tile_sums[(int64_t)w * n_tiles + blockIdx.x] = WSums4{s0 & 0xFFFFFFFFull, s1 & 0xFFFFFFFFull, s0 >> 32, s1 >> 32};
}

// exclusive scan of the tile sums of every wide dimension; totals.  grid (wide dimensions), one wave
__global__ __launch_bounds__(64) void node_wide_scan_kernel(int32_t n_tiles, WSums4* __restrict__ tile_sums, WSums4* __restrict__ total)
{
    const int w = blockIdx.x, lane = threadIdx.x;
    WSums4* ts = tile_sums + (int64_t)w * n_tiles;
    WSums4 run = {0, 0, 0, 0};
    const WSums4 none = {0, 0, 0, 0};
    WSums4 nxt = (lane < n_tiles) ? ts[lane] : none;          // the next 64 tiles are on their way during the scan (they lie behind what this turn writes)
    for (int i0 = 0; i0 < n_tiles; i0 += 64) {
        const int i = i0 + lane;
        const WSums4 v = nxt;
        nxt = (i + 64 < n_tiles) ? ts[i + 64] : none;
        const WSums4 x = {wave_prefix_u64(v.size), wave_prefix_u64(v.tp1), wave_prefix_u64(v.tot), wave_prefix_u64(v.bp)};
        if (i < n_tiles) ts[i] = WSums4{run.size + x.size - v.size, run.tp1 + x.tp1 - v.tp1, run.tot + x.tot - v.tot, run.bp + x.bp - v.bp};
        run.size += wave_last_u64(x.size); run.tp1 += wave_last_u64(x.tp1); run.tot += wave_last_u64(x.tot); run.bp += wave_last_u64(x.bp);
    }
    if (lane == 0) total[w] = run;
}

// eval_threshold with the bin fetched by get_bin() only when the candidate gets past the product test
template <typename BinFn>
__device__ __forceinline__ void eval_threshold_lazy(long long size1, long long tp1, long long tot1, unsigned long long bp1, long long n,
                                                    unsigned long long bp_total, int32_t min_ndps, BinFn get_bin, const ScgView& scg, Cand& best, double bar)
{
    const long long size2 = n - size1, tot2 = n - tot1;
    const long long tp2 = n - (size1 + tot1 - tp1);
    if (size1 < min_ndps || size2 < min_ndps) return;              // ref :151
    const bool small1 = size1 < size2;                             // ref :155 (tie -> cluster2)
    const long long TP = small1 ? tp1 : tp2, TOT = small1 ? tot1 : tot2, SZ = small1 ? size1 : size2;
    {
        const double need = best.found ? (best.prod > bar ? best.prod : bar) : bar;     // see eval_threshold
        const double tp = (double)TP;
        if (tp * tp < need * ((double)TOT * (double)SZ) * (1.0 - 1e-9)) return;
    }
    const double sens = (double)(int)TP / (double)(int)TOT;        // ref .h:75 (int fields)
    const double spec = (double)(int)TP / (double)(int)SZ;         // ref .h:74
    const double prod = sens * spec;
    if (!(prod == prod)) return;                                   // NaN is never better (ref .h:28)
    if (best.found && prod < best.prod) return;
    const int bin = get_bin();
    if (best.found && prod == best.prod && bin >= best.bin) return;   // equal products keep the smaller threshold (the walk may come in any order)
    int cl = 0, chh = 0, ng = scg.n;
    if (scg.compact) {
        // the staged lists live in LDS: say so, or every read is a flat load that waits on both memory paths
        const LdsI32 lo3 = (LdsI32)scg.lo, hi3 = (LdsI32)scg.hi;
#pragma unroll 4
        for (int g = 0; g < scg.n; g++) { cl += lo3[g] <= bin; chh += hi3[g] <= bin; }
    } else {
        ng = 0;
        for (int g = 0; g < scg.n; g++) { const int l = scg.lo[g]; if (l != INT_MAX) { ng++; cl += l <= bin; chh += scg.hi[g] <= bin; } }
    }
    const int G1 = cl, G2 = ng - chh, both = cl - chh;
    bool pass;
    if (G1 == 0 || G2 == 0) pass = (bp1 >= 500000ull) && (bp_total - bp1 >= 500000ull);
    else pass = !((double)both / (double)(G1 < G2 ? G1 : G2) < 0.2);
    if (pass) best = Cand{prod, bin, 1, sens, spec, size1};
}

// pass 2a' (as tree_wide_tilescreen_kernel): the tiles whose sums could reach the bar at all.  grid: blocks over (tiles x wide dimensions)
__global__ __launch_bounds__(256) void node_wide_tilescreen_kernel(int32_t n_tiles, int32_t n_wide, int32_t min_ndps, const WSums4* __restrict__ tile_carry,
                                                                  const WSums4* __restrict__ total, const double* __restrict__ bar, Cand* __restrict__ tile_best,
                                                                  int32_t* __restrict__ list, int32_t* __restrict__ list_n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    if (i < (int64_t)n_tiles * n_wide) {
        const int w = (int)(i / n_tiles), t = (int)(i - (int64_t)w * n_tiles);
        const WSums4 c = tile_carry[i], tot = total[w];
        const WSums4 e = (t + 1 < n_tiles) ? tile_carry[i + 1] : tot;
        keep = tile_may_reach(c, WSums4{e.size - c.size, e.tp1 - c.tp1, e.tot - c.tot, e.bp - c.bp}, (long long)tot.size, min_ndps, bar[0]);
        tile_best[i] = Cand{0.0, 0, 0, 0.0, 0.0, 0};
    }
    const unsigned long long m = __ballot(keep);
    if (m) {
        const int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == 0) base = atomicAdd(list_n, __popcll(m));
        base = __shfl(base, 0, 64);
        if (keep) list[base + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)i;
    }
}

// pass 2a, the screen (as tree_wide_screen_kernel): which of those tiles hold a static run end whose threshold could reach
// the node's bar.  grid: any; the workgroups take the listed tiles in turn
template <bool MASKED>
__global__ __launch_bounds__(256) void node_wide_screen_kernel(TreeDev Tr, int32_t N, int32_t n_tiles, int32_t min_ndps, const uint32_t* __restrict__ wword,
                                                              int64_t wstride, const uint8_t* __restrict__ member, const WSums4* __restrict__ tile_carry,
                                                              const WSums4* __restrict__ total, const double* __restrict__ bar,
                                                              const int32_t* __restrict__ todo, const int32_t* __restrict__ todo_n,
                                                              int32_t* __restrict__ list, int32_t* __restrict__ list_n)
{
    __shared__ unsigned long long sh[5];
    __shared__ uint32_t sh32[4];
    const int n_todo = *todo_n;
  for (int li = blockIdx.x; li < n_todo; li += gridDim.x) {
    const int entry = todo[li];
    const int w = entry / n_tiles, tile = entry - w * n_tiles;
    const uint32_t* __restrict__ words = wword + (int64_t)w * wstride;
    const int begin = tile * TILE_P, end = min(N, begin + TILE_P);
    const int i0 = begin + 8 * threadIdx.x;
    uint32_t wd[8], pk[8];
    load8_stream(words, i0, begin, end, wd, 0u);
    const WSums4 carry = tile_carry[(int64_t)w * n_tiles + tile];
    const uint32_t n32 = (uint32_t)total[w].size;
    const double node_bar = bar[0];
    uint32_t in = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) in |= (uint32_t)(i0 + i < end) << i;
    const uint32_t inside = in;
    if (MASKED) {
        uint8_t mb[8];
        load_scaf_bytes(member, wd, in, Tr.sc_mask, mb);
#pragma unroll
        for (int i = 0; i < 8; i++) if (!mb[i]) in &= ~(1u << i);
    }
    load_packs<true>(Tr, w, wd, in, pk);
    uint32_t info[8];                                         // live | run end << 1 | tp << 2 | tot << 9
    uint32_t a_size = 0, a_tp = 0, a_tot = 0;
    bool hit = false;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t live = (in >> i) & 1u;
        hit |= pk[i] == PACK_ESCAPE;
        const uint32_t tp = (live ? (wd[i] >> 31) : 0u) + (pk[i] & 63u), tot = (pk[i] >> 6) & 127u;
        const uint32_t run_end = (((inside >> i) & 1u) && (wd[i] & W_END)) ? 2u : 0u;
        info[i] = live | run_end | tp << 2 | tot << 9;
        a_size += live; a_tp += tp; a_tot += tot;
    }
    const unsigned long long o0 = block_excl_scan_u64((unsigned long long)a_size | ((unsigned long long)a_tot << 32), sh, nullptr);
    const uint32_t o1 = block_excl_scan_u32(a_tp, sh32);
    uint32_t size1 = (uint32_t)(carry.size + (o0 & 0xFFFFFFFFull)), tp1 = (uint32_t)carry.tp1 + o1, tot1 = (uint32_t)(carry.tot + (o0 >> 32));
    const uint32_t mn32 = (uint32_t)max(min_ndps, 0);
    const Cand none = {0.0, 0, 0, 0.0, 0.0, 0};
    const float need = screen_need(none, node_bar);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        size1 += info[i] & 1u; tp1 += (info[i] >> 2) & 127u; tot1 += (info[i] >> 9) & 127u;
        hit |= (info[i] & 2u) && screen32(size1, tp1, tot1, n32, mn32, need);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int any = __syncthreads_or(hit);
    if (threadIdx.x == 0 && any) list[atomicAdd(list_n, 1)] = entry;
  }
}

// pass 2b: the listed tiles (list == NULL: every tile) again with everything; evaluate at the static run ends.  grid: any
template <bool MASKED>
__global__ __launch_bounds__(256) void node_wide_eval_kernel(TreeDev Tr, int32_t N, int32_t n_tiles, int32_t min_ndps, const int32_t* __restrict__ sbin,
                                                            const uint32_t* __restrict__ wword, int64_t wstride, const uint8_t* __restrict__ member,
                                                            const WSums4* __restrict__ tile_carry, const WSums4* __restrict__ total,
                                                            const int32_t* __restrict__ scg_lo, const int32_t* __restrict__ scg_hi,
                                                            const double* __restrict__ bar, Cand* __restrict__ tile_best,
                                                            const int32_t* __restrict__ list, const int32_t* __restrict__ list_n)
{
    __shared__ unsigned long long sh[2][5];
    __shared__ int32_t s_lo[SCG_LDS], s_hi[SCG_LDS];
    __shared__ int s_ng;
    __shared__ Cand wbest[4];
    const int n_listed = list ? *list_n : n_tiles * Tr.n_wide;
  for (int li = blockIdx.x; li < n_listed; li += gridDim.x) {         // (a block's turns are separated by the barriers of block_best)
    const int entry = list ? list[li] : li;
    const int w = entry / n_tiles, tile = entry - w * n_tiles;
    const int d = Tr.wide[w];
    const WSums4 tot = total[w];
    const long long n = (long long)tot.size;
    const int32_t* __restrict__ bins = sbin + (int64_t)w * N;
    const uint32_t* __restrict__ words = wword + (int64_t)w * wstride;
    const double node_bar = bar[0];
    const int begin = tile * TILE_P, end = min(N, begin + TILE_P);
    const int i0 = begin + 8 * threadIdx.x;
    uint32_t wd[8];
    uint32_t r_tp[8], r_tot[8], r_bp[8];
    uint32_t in = 0;
    WSums4 a = {0, 0, 0, 0};
    load8_stream(words, i0, begin, end, wd, 0u);
#pragma unroll
    for (int i = 0; i < 8; i++) in |= (uint32_t)(i0 + i < end) << i;
    if (MASKED) {
        uint8_t mb[8];
        load_scaf_bytes(member, wd, in, Tr.sc_mask, mb);
#pragma unroll
        for (int i = 0; i < 8; i++) if (!mb[i]) in &= ~(1u << i);
    }
    {
        uint32_t pk[8];
        load_packs(Tr, w, wd, in, pk);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const bool live = (in >> i) & 1u;
            const WSums r = pack_wsums(Tr, w, wd[i], pk[i], live);
            r_tp[i] = (uint32_t)r.tp1; r_tot[i] = (uint32_t)r.tot; r_bp[i] = (uint32_t)r.bp;
            a.size += live ? 1u : 0u; a.tp1 += r.tp1; a.tot += r.tot; a.bp += r.bp;
        }
    }
    const WSums4 carry = tile_carry[(int64_t)w * n_tiles + tile];
    const unsigned long long o0 = block_excl_scan_u64(a.size | (a.tot << 32), sh[0], nullptr);
    const unsigned long long o1 = block_excl_scan_u64(a.tp1 | (a.bp << 32), sh[1], nullptr);
    uint32_t size1 = (uint32_t)(carry.size + (o0 & 0xFFFFFFFFull)), tp1 = (uint32_t)(carry.tp1 + (o1 & 0xFFFFFFFFull)), tot1 = (uint32_t)(carry.tot + (o0 >> 32));
    unsigned long long bp1 = carry.bp + (o1 >> 32);
    Cand best = {0.0, 0, 0, 0.0, 0.0, 0};
    const uint32_t n32 = (uint32_t)n, mn32 = (uint32_t)max(min_ndps, 0);
    float need = screen_need(best, node_bar);
    const int64_t sg = (int64_t)d * Tr.P.n_scg;
    const ScgView scg = stage_scgs(scg_lo + sg, scg_hi + sg, Tr.P.n_scg, s_lo, s_hi, &s_ng);
    // (the most promising run end first, as in tree_wide_eval_kernel)
    uint32_t p_sz[8], p_tp[8], p_tot[8];
    unsigned long long p_bp[8];
    uint32_t todo = 0;
    float e_best = -2.0f;
    int i_best = -1;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (in & (1u << i)) { size1++; tp1 += r_tp[i]; tot1 += r_tot[i]; bp1 += r_bp[i]; }
        p_sz[i] = size1; p_tp[i] = tp1; p_tot[i] = tot1; p_bp[i] = bp1;
        if (i0 + i < end && (wd[i] & W_END) && screen32(size1, tp1, tot1, n32, mn32, need)) {
            todo |= 1u << i;
            const float e = screen_est32(size1, tp1, tot1, n32, mn32);
            if (e > e_best) { e_best = e; i_best = i; }
        }
    }
    if (i_best >= 0) {
        uint32_t sz = 0, tp = 0, tt = 0; unsigned long long bp = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) if (i == i_best) { sz = p_sz[i]; tp = p_tp[i]; tt = p_tot[i]; bp = p_bp[i]; }
        eval_threshold_lazy((long long)sz, (long long)tp, (long long)tt, bp, n, tot.bp, min_ndps, [&]() { return bins[i0 + i_best]; }, scg, best, node_bar);
        need = screen_need(best, node_bar);
        todo &= ~(1u << i_best);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (((todo >> i) & 1u) && screen32(p_sz[i], p_tp[i], p_tot[i], n32, mn32, need)) {
            eval_threshold_lazy((long long)p_sz[i], (long long)p_tp[i], (long long)p_tot[i], p_bp[i], n, tot.bp, min_ndps, [&]() { return bins[i0 + i]; }, scg, best, node_bar);
            need = screen_need(best, node_bar);
        }
    }
    const Cand bb = block_best(best, wbest);
    if (threadIdx.x == 0) tile_best[(int64_t)w * n_tiles + tile] = bb;
  }
}

// best tile of every wide dimension -> the dimension's record of abw_sep_node.  grid (wide dimensions), one wave
__global__ __launch_bounds__(64) void node_wide_best_kernel(TreeDev Tr, int32_t n_tiles, const Cand* __restrict__ tile_best, const WSums4* __restrict__ total,
                                                           DimBestDev* __restrict__ out)
{
    const int w = blockIdx.x, lane = threadIdx.x;
    Cand c = {0.0, 0, 0, 0.0, 0.0, 0};
    for (int t = lane; t < n_tiles; t += 64) { const Cand o = tile_best[(int64_t)w * n_tiles + t]; if (cand_better(o, c)) c = o; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const Cand other = cand_shfl_xor(c, o); if (cand_better(other, c)) c = other; }
    if (lane == 0) {
        DimBestDev& r = out[Tr.wide[w]];
        r.found = c.found; r.bin = c.bin; r.sens = c.sens; r.spec = c.spec; r.prod = c.prod; r.size1 = c.size1; r.n = (long long)total[w].size;
    }
}

// the best product of the node's narrow dimensions: the bar of eval_threshold for the wide ones.  one wave
__global__ __launch_bounds__(64) void node_bar_kernel(TreeDev Tr, const DimBestDev* __restrict__ db, double* __restrict__ bar)
{
    double b = 0.0;
    for (int i = threadIdx.x; i < Tr.n_narrow; i += 64) { const DimBestDev c = db[Tr.narrow[i]]; if (c.found && c.prod > b) b = c.prod; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double y = __shfl_xor(b, o, 64); if (y > b) b = y; }
    if (threadIdx.x == 0) bar[0] = b;
}

// ---- the winner of every node ----------------------------------------------------------------------------------------
// ClusteringResult::is_better over the dimensions in dimension order (ref ...Specificity.h:25-30 + ClusterSeparator.cpp:11-16):
// the largest product, on equal products the smallest dimension; then is_legal (ref .h:31-35).  grid (nodes), one wave.
__global__ __launch_bounds__(64) void tree_winner_kernel(TreeDev Tr, const NodeDev* __restrict__ nodes, const Cand* __restrict__ cand, NodeRes* __restrict__ res)
{
    const int k = blockIdx.x, lane = threadIdx.x;
    const NodeDev nd = nodes[k];
    Cand bc = {0.0, 0, 0, 0.0, 0.0, 0};
    int bd = -1;
    // every lane takes dimensions lane, lane + 64, ... in ascending order: a later dimension must be strictly better
    for (int d = lane; d < Tr.D; d += 64) {
        const Cand c = cand[(int64_t)k * Tr.D + d];
        if (c.found && (bd < 0 || c.prod > bc.prod)) { bc = c; bd = d; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const Cand oc = cand_shfl_xor(bc, o);
        const int od = __shfl_xor(bd, o, 64);
        if (od >= 0 && (bd < 0 || oc.prod > bc.prod || (oc.prod == bc.prod && od < bd))) { bc = oc; bd = od; }
    }
    if (lane != 0) return;
    NodeRes r;
    r.win.found = bd >= 0; r.win.dim = bd; r.win.bin = bd >= 0 ? bc.bin : -1;
    r.win.sens = bd >= 0 ? bc.sens : 0.0; r.win.spec = bd >= 0 ? bc.spec : 0.0;
    r.win.legal = (bd >= 0) && (bc.sens >= 0.8) && (bc.spec >= 0.8) && (bc.sens + bc.spec >= 1.6) && (bc.sens * bc.spec >= 0.8);
    r.win.size1 = bd >= 0 ? bc.size1 : 0; r.win.n = nd.n;
    r.n1 = r.n2 = 0; r.m1 = r.m2 = 0;
    res[k] = r;
}

// ---- scaffold sides (ref ClusterSeparator.cpp:35-53,94-122): cluster1 = the raw side with fewer dps (tie: the "<=" side), a
// scaffold goes to cluster1 iff at least half of its dps are there.  Only the nodes whose winner is THIS context's
// (res.win.legal set by the host: a legal winner in one of its dimensions) are assigned here.  grid (scaffold tiles)
// COUNT (one context decides every node: abw_sep_tree): the children's sizes in the same pass (tree_count_kernel's sums).
template <bool COUNT>
__global__ __launch_bounds__(256) void tree_assign_kernel(TreeDev Tr, const Tile* __restrict__ tiles, const int32_t* __restrict__ order,
                                                         NodeRes* __restrict__ res, uint8_t* __restrict__ scaf_side)
{
    __shared__ unsigned long long b[4];
    const Tile t = tiles[blockIdx.x];
    const GlobalRes win = res[t.node].win;
    if (!win.legal) return;                                   // uniform for the tile: nothing to reconstruct (ref :27-28), or not mine
    if (COUNT) { if (threadIdx.x < 4) b[threadIdx.x] = 0; __syncthreads(); }
    const bool swap = win.n - win.size1 < win.size1;
    const int32_t* __restrict__ col = Tr.P.bins + (int64_t)win.dim * Tr.P.N;
    unsigned long long n1 = 0, n2 = 0, m1 = 0, m2 = 0;
    for (int ls = t.begin + threadIdx.x; ls < t.end; ls += 256) {
        const int s = order[ls];
        const int f = Tr.P.scaf_first[s], tot = Tr.P.scaf_ndps[s];
        int c = 0;
        for (int i = 0; i < tot; i++) c += col[f + i] <= win.bin;
        const int c1 = swap ? tot - c : c;
        const bool one = c1 > 0 && 2 * (long long)c1 >= tot;
        scaf_side[s] = one ? 1 : 2;
        if (COUNT) { if (one) { n1 += (unsigned long long)tot; m1++; } else { n2 += (unsigned long long)tot; m2++; } }
    }
    if (COUNT) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { n1 += __shfl_xor(n1, o, 64); n2 += __shfl_xor(n2, o, 64); m1 += __shfl_xor(m1, o, 64); m2 += __shfl_xor(m2, o, 64); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&b[0], n1); atomicAdd(&b[1], n2); atomicAdd(&b[2], m1); atomicAdd(&b[3], m2); }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (b[0]) atomicAdd(&res[t.node].n1, b[0]);
            if (b[1]) atomicAdd(&res[t.node].n2, b[1]);
            if (b[2]) atomicAdd(&res[t.node].m1, (uint32_t)b[2]);
            if (b[3]) atomicAdd(&res[t.node].m2, (uint32_t)b[3]);
        }
    }
}

// the children's sizes from the (combined) sides: dps and scaffolds sent to cluster1 / cluster2.  grid (scaffold tiles)
__global__ __launch_bounds__(256) void tree_count_kernel(TreeDev Tr, const Tile* __restrict__ tiles, const int32_t* __restrict__ order,
                                                        const uint8_t* __restrict__ scaf_side, NodeRes* __restrict__ res)
{
    __shared__ unsigned long long b[4];
    const Tile t = tiles[blockIdx.x];
    if (threadIdx.x < 4) b[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long n1 = 0, n2 = 0, m1 = 0, m2 = 0;
    for (int ls = t.begin + threadIdx.x; ls < t.end; ls += 256) {
        const int s = order[ls];
        const uint8_t sd = scaf_side[s];
        const unsigned long long tot = (unsigned long long)Tr.P.scaf_ndps[s];
        if (sd == 1) { n1 += tot; m1++; } else if (sd == 2) { n2 += tot; m2++; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { n1 += __shfl_xor(n1, o, 64); n2 += __shfl_xor(n2, o, 64); m1 += __shfl_xor(m1, o, 64); m2 += __shfl_xor(m2, o, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&b[0], n1); atomicAdd(&b[1], n2); atomicAdd(&b[2], m1); atomicAdd(&b[3], m2); }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (b[0]) atomicAdd(&res[t.node].n1, b[0]);
        if (b[1]) atomicAdd(&res[t.node].n2, b[1]);
        if (b[2]) atomicAdd(&res[t.node].m1, (uint32_t)b[2]);
        if (b[3]) atomicAdd(&res[t.node].m2, (uint32_t)b[3]);
    }
}

// ---- partition: the next level's order of scaffolds and of every wide dimension's positions ---------------------------
// The scaffold order is rewritten whole every level (a few bytes per scaffold).  The position arrays of the wide dimensions are
// not: a child whose base is -1 KEEPS its parent's segment -- the bigger child of a split, as long as at least half of the
// segment is still its own -- and only its sibling's positions are copied out (to fresh space behind everything in use) and
// marked dead where they were.  The trees of a metagenome are caterpillars (a giant node sheds a genome per level): moving
// the giant at every level was three quarters of the recursion's time.
// Per tile: the number of live items going to cluster 1 and of live items (low / high half of tile_cnt), and their masks
// (two bytes per thread of the tile = 8 items: the scatter does not look the sides up again).  items: scaffold ids (order) or
// position words (scaffold in the low 28 bits, W_DEAD); side: per scaffold.  grid (tiles, arrays); istride: a multiple of 8
// Where a child keeps the segment the items are not looked at: the positions that leave were marked in a bit map by their
// scaffolds (tree_mark_kernel; bits: one per position of a dimension's array), and the masks are those bits -- as "going to
// cluster 1" when cluster 1 leaves, as "live, not cluster 1" when cluster 2 does, so that the scatter's ranks come out the same.
__global__ __launch_bounds__(256) void tree_part_count_kernel(const Tile* __restrict__ tiles, int32_t n_tiles, const NodePart* __restrict__ part,
                                                             const uint32_t* __restrict__ items, int64_t istride,
                                                             const uint8_t* __restrict__ side, uint32_t sc_mask, uint8_t* __restrict__ bits,
                                                             unsigned long long* __restrict__ tile_cnt, uint16_t* __restrict__ mask)
{
    __shared__ unsigned long long sh[4];
    const Tile t = tiles[blockIdx.x];
    const NodePart np = part[t.node];
    if (!np.split) return;                                    // uniform: a terminal node's items go nowhere
    const int i0 = (t.begin & ~2047) + 8 * threadIdx.x;
    uint32_t m1 = 0, ma = 0;
    if (bits && np.bits) {
        uint8_t* __restrict__ mine = bits + (((int64_t)blockIdx.y * istride + i0) >> 3);      // read by this thread alone: left clear for the next level
        uint32_t m = *mine;
        if (m) *mine = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) if (i0 + i < t.begin || i0 + i >= t.end) m &= ~(1u << i);
        ma = m; m1 = np.dp_base1 >= 0 ? m : 0u;
    } else {
        const uint32_t* __restrict__ it = items + (int64_t)blockIdx.y * istride;
        uint32_t v[8];
        load8(it, i0, t.begin, t.end, v, W_DEAD);
#pragma unroll
        for (int i = 0; i < 8; i++) ma |= (uint32_t)(i0 + i >= t.begin && i0 + i < t.end && !(v[i] & W_DEAD)) << i;
        uint8_t sd[8];
        load_scaf_bytes(side, v, ma, sc_mask, sd);             // eight loads in flight, not one after the other
#pragma unroll
        for (int i = 0; i < 8; i++) m1 |= (uint32_t)(sd[i] == 1) << i;
        m1 &= ma;
    }
    mask[((int64_t)blockIdx.y * n_tiles + blockIdx.x) * 256 + threadIdx.x] = (uint16_t)(m1 | (ma << 8));      // by tile: two nodes may share 8 aligned items
    const unsigned long long s = block_sum_u64((unsigned long long)__popc(m1) | ((unsigned long long)__popc(ma) << 32), sh);
    if (threadIdx.x == 0) tile_cnt[(int64_t)blockIdx.y * n_tiles + blockIdx.x] = s;
}

// exclusive scan of the tile counts inside every split node.  grid (nodes, arrays), one wave.  scaf = 1: scaffold tiles
__global__ __launch_bounds__(64) void tree_part_scan_kernel(const NodeDev* __restrict__ nodes, const NodePart* __restrict__ part, int32_t n_tiles,
                                                           unsigned long long* __restrict__ tile_cnt, int scaf)
{
    const int k = blockIdx.x, lane = threadIdx.x;
    if (!part[k].split) return;
    const NodeDev nd = nodes[k];
    const int tb = scaf ? nd.stile_begin : nd.ptile_begin, te = scaf ? nd.stile_end : nd.ptile_end;
    unsigned long long* tc = tile_cnt + (int64_t)blockIdx.y * n_tiles;
    unsigned long long run = 0;
    unsigned long long nxt = (tb + lane < te) ? tc[tb + lane] : 0;
    for (int i0 = tb; i0 < te; i0 += 64) {
        const int i = i0 + lane;
        const unsigned long long v = nxt;
        nxt = (i + 64 < te) ? tc[i + 64] : 0;
        const unsigned long long x = wave_prefix_u64(v);       // both halves stay below 2^31: no carry between them
        if (i < te) tc[i] = run + x - v;
        run += wave_last_u64(x);
    }
}

// stable scatter of a split node's live items into its children's ranges (a child with base -1 stays where it is, and the
// items that leave are marked dead in place: `marks`, the array itself).  For scaffolds (scaf = 1) every scaffold also learns
// the slot of its node in the next level (-1: none) and, in a terminal node, its final cluster id.  grid (tiles, arrays)
__global__ __launch_bounds__(256) void tree_part_scatter_kernel(const Tile* __restrict__ tiles, int32_t n_tiles, const NodeDev* __restrict__ nodes,
                                                               const NodePart* __restrict__ part, const uint32_t* __restrict__ items,
                                                               const int32_t* __restrict__ items2, int64_t istride,
                                                               const uint16_t* __restrict__ mask, const unsigned long long* __restrict__ tile_cnt,
                                                               uint32_t* __restrict__ out, int32_t* __restrict__ out2,
                                                               uint32_t* __restrict__ marks, int scaf, int32_t* __restrict__ scaf2cluster, int32_t* __restrict__ scaf_slot,
                                                               TreeDev Tr, WSums4* __restrict__ left, int64_t lstride)
{
    __shared__ unsigned long long sh[5];
    __shared__ unsigned long long shs[8];
    const Tile t = tiles[blockIdx.x];
    const NodePart np = part[t.node];
    const uint32_t* __restrict__ it = items + (int64_t)blockIdx.y * istride;
    const int i0 = (t.begin & ~2047) + 8 * threadIdx.x;
    uint32_t v[8];
    if (!np.split) {
        if (scaf) {
            load8(it, i0, t.begin, t.end, v, 0u);
#pragma unroll
            for (int i = 0; i < 8; i++) if (i0 + i >= t.begin && i0 + i < t.end) { scaf_slot[v[i]] = -1; if (np.id > 0) scaf2cluster[v[i]] = np.id; }
        }
        return;
    }
    const int base1 = scaf ? np.scaf_base1 : np.dp_base1, base2 = scaf ? np.scaf_base2 : np.dp_base2;
    const uint32_t mk = mask[((int64_t)blockIdx.y * n_tiles + blockIdx.x) * 256 + threadIdx.x];
    const uint32_t m1 = mk & 0xFFu, ma = mk >> 8;
    // the items are read only by the threads that have one to copy: where a child keeps the segment, almost none do
    const uint32_t moving = (base1 >= 0 ? m1 : 0u) | (base2 >= 0 ? (ma & ~m1) : 0u);
    // a tile of a kept segment that nothing leaves (most tiles of a caterpillar's giant) is done here
    const bool one_keeps = !scaf && (base1 < 0) != (base2 < 0);
    const int any_moving = one_keeps ? __syncthreads_or(moving != 0) : 1;
    if (!any_moving) {
        if (left && threadIdx.x == 0) left[(int64_t)blockIdx.y * lstride + blockIdx.x] = WSums4{0, 0, 0, 0};
        return;
    }
    int32_t v2[8];
    if (moving || scaf) {
        load8(it, i0, t.begin, t.end, v, W_DEAD);
        if (items2) load8(items2 + (int64_t)blockIdx.y * istride, i0, t.begin, t.end, v2, 0);
    }
    const unsigned long long before = block_excl_scan_u64((unsigned long long)__popc(m1) | ((unsigned long long)__popc(ma) << 32), sh, nullptr) +
                                      tile_cnt[(int64_t)blockIdx.y * n_tiles + blockIdx.x];                 // inside the node
    long long ones_before = (long long)(before & 0xFFFFFFFFull), alive_before = (long long)(before >> 32);
    uint32_t* o1 = out + (int64_t)blockIdx.y * istride;
    int32_t* o2 = out2 ? out2 + (int64_t)blockIdx.y * istride : nullptr;
    uint32_t* mkp = marks ? marks + (int64_t)blockIdx.y * istride : nullptr;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if ((ma >> i) & 1u) {
            const uint32_t one = (m1 >> i) & 1u;
            const int base = one ? base1 : base2;
            if (base >= 0) {
                const long long dest = base + (one ? ones_before : alive_before - ones_before);
                o1[dest] = v[i];
                if (o2) o2[dest] = v2[i];
                if (mkp && (one ? base2 : base1) < 0) mkp[i0 + i] = v[i] | W_DEAD;      // its sibling keeps the segment
            }
            if (scaf) scaf_slot[v[i]] = one ? np.slot1 : np.slot2;
            ones_before += one; alive_before++;
        }
    }
    // One child keeps the segment: the sums of what leaves this tile, for tree_wide_inherit_kernel at the next level (the
    // tile's new sums are its old ones less these).  Uniform for the workgroup.
    if (left && one_keeps) {
        const int w = blockIdx.y;
        WSums4 a = {0, 0, 0, 0};
        {
            if (!moving) {
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = W_DEAD;             // (this thread loaded nothing)
            }
            uint32_t pk[8];
            load_packs(Tr, w, v, moving, pk);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if ((moving >> i) & 1u) {
                    const WSums r = pack_wsums(Tr, w, v[i], pk[i], true);
                    a.size += 1u; a.tp1 += r.tp1; a.tot += r.tot; a.bp += r.bp;
                }
            }
            const unsigned long long s0 = block_sum_u64(a.size | (a.tot << 32), shs), s1 = block_sum_u64(a.tp1 | (a.bp << 32), shs + 4);
            a = WSums4{s0 & 0xFFFFFFFFull, s1 & 0xFFFFFFFFull, s0 >> 32, s1 >> 32};
        }
        if (threadIdx.x == 0) left[(int64_t)w * lstride + blockIdx.x] = a;
    }
}

// The node that still lives in the ROOT's segment (the table's static order, where every dp's position is a static table:
// pos_of) -- the giant of a caterpillar -- finds the positions that leave it from the scaffolds that leave, not by looking at
// all of its positions: the leaving scaffolds (a range of the next level's scaffold order, which the scaffolds' own partition
// has just written) mark their dps' positions in every wide dimension's bit map (tree_mark_kernel).
// grid (ceil(listed scaffolds / 256), wide dimensions)
__global__ __launch_bounds__(256) void tree_mark_kernel(TreeDev Tr, const int32_t* __restrict__ list, int32_t n_list, const int32_t* __restrict__ pos_of, int64_t pstride,
                                                       uint32_t* __restrict__ bits, int64_t bstride)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_list) return;
    const int s = list[i];
    const int32_t* __restrict__ pof = pos_of + (int64_t)blockIdx.y * pstride;
    uint32_t* __restrict__ bw = bits + (int64_t)blockIdx.y * bstride;
    const int f = Tr.P.scaf_first[s], n = Tr.P.scaf_ndps[s];
    for (int k = 0; k < n; k++) { const int p = pof[f + k]; atomicOr(&bw[p >> 5], 1u << (p & 31)); }
}

// pos_of at the root: the inverse of the table's static order.  grid (blocks, wide dimensions)
__global__ void tree_posof_kernel(const int32_t* __restrict__ sdp, int64_t N, int32_t* __restrict__ pos_of)
{
    const int64_t base = (int64_t)blockIdx.y * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) pos_of[base + sdp[base + i]] = (int32_t)i;
}

__global__ void tree_iota_kernel(uint32_t* p, int32_t n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}

void make_tiles(std::vector<Tile>& tiles, int node, int begin, int end, int step, int32_t& tb, int32_t& te)
{
    tb = (int32_t)tiles.size();
    for (int a = begin; a < end; a = (a / step + 1) * step) tiles.push_back(Tile{node, a, std::min(end, (a / step + 1) * step)});   // cut at absolute multiples
    te = (int32_t)tiles.size();
}

// a block of host data for one upload: appends v, returns its byte offset (64-byte aligned)
template <typename T>
size_t pack(std::vector<uint8_t>& blob, const std::vector<T>& v)
{
    const size_t at = (blob.size() + 63) & ~(size_t)63;
    blob.resize(at + v.size() * sizeof(T));
    if (!v.empty()) memcpy(blob.data() + at, v.data(), v.size() * sizeof(T));
    return at;
}

// The static tables of the loaded table (once; kept with the table until abw_sep_load / abw_ctx_destroy frees them).
int tree_static_build(abw_ctx* ctx, SepHost& H, const SepDev& P, const int32_t* d_dp2scaf)
{
    abw_sep_state& s = ctx->sep;
    hipStream_t st = ctx->stream;
    const int32_t S = s.S, n_wide = (int32_t)H.wide_dims.size();
    const int64_t N = s.N;
    // a build that failed half-way may have left some of them behind
    if (H.d_wword) { (void)hipFree(H.d_wword); H.d_wword = nullptr; }
    if (H.d_wext) { (void)hipFree(H.d_wext); H.d_wext = nullptr; }
    if (H.d_wpack) { (void)hipFree(H.d_wpack); H.d_wpack = nullptr; }
    if (H.d_posof) { (void)hipFree(H.d_posof); H.d_posof = nullptr; }
    if (H.d_sinfo) { (void)hipFree(H.d_sinfo); H.d_sinfo = nullptr; }
    if (n_wide) {
        const bool inl = H.packed && S <= (int32_t)W_SC22 + 1 && !getenv("ABW_SEP_NO_INLINE");
        if (inl) ABW_HIP(ctx, hipMalloc((void**)&H.d_sinfo, (size_t)S * 5 + 64));
        const int64_t Np = (N + 7) & ~(int64_t)7;
        ABW_HIP(ctx, hipMalloc((void**)&H.d_wword, (size_t)n_wide * Np * 4));
        ABW_HIP(ctx, hipMalloc((void**)&H.d_wext, (size_t)n_wide * S * 16));
        ABW_HIP(ctx, hipMalloc((void**)&H.d_wpack, (size_t)n_wide * S * 4));
        ABW_HIP(ctx, hipMalloc((void**)&H.d_posof, (size_t)n_wide * N * 4));
        int rc;
        if ((rc = abw_reserve(ctx, ctx->scratch[3], (size_t)n_wide * N))) return rc;         // per-dp flags, temporary
        hipLaunchKernelGGL(tree_static_wide_kernel, dim3((unsigned)((S + 255) / 256), (unsigned)n_wide), dim3(256), 0, st, P, H.d_wide, S,
                           (uint8_t*)ctx->scratch[3].p, (uint4*)H.d_wext, H.d_wpack, H.d_sinfo);
        ABW_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(tree_static_words_kernel, dim3((unsigned)(ctx->num_cu * 8), (unsigned)n_wide), dim3(256), 0, st, H.d_sdp, H.d_sbin,
                           (const uint8_t*)ctx->scratch[3].p, d_dp2scaf, N, Np, (const uint4*)H.d_wext, S, inl ? 1 : 0, H.d_wword);
        ABW_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(tree_posof_kernel, dim3((unsigned)(ctx->num_cu * 8), (unsigned)n_wide), dim3(256), 0, st, (const int32_t*)H.d_sdp, N, H.d_posof);
        ABW_LAUNCH_CHECK(ctx);
    }
    ABW_HIP(ctx, hipStreamSynchronize(st));
    return ABW_OK;
}

// ---- the recursion as a sequence of levels that a caller can interleave with an exchange ------------------------------
// node of a level: id, size, and where its histogram comes from
struct HostNode { int32_t id; int32_t m; int64_t n; int32_t parent; int64_t seg_begin, seg_len; bool root_seg; int32_t inherit; };   // inherit: the slot (previous level) whose segment it keeps, or -1   // parent: its slot in the previous level
                                                                          // (-1: the root); seg: its segment of the wide position arrays

struct TreeRun {
    abw_ctx* ctx = nullptr;
    int32_t min_ndps = 0;
    int64_t max_nodes = 0, visited = 0;
    int32_t next_id = 2;
    int cur = 0;                         // scaffold order and histogram pool of the level in flight
    int wcur = 0;                        // the buffer the wide dimensions' position arrays live in (changes only when they are compacted)
    int64_t Wp = 0, arena_end = 0;       // stride of those arrays (2 Np: room to copy children out behind what is in use), and what is in use
    int phase = 0;                       // 0: a level waits for eval, 1: for assign, 2: for advance, 3: the recursion is over
    uint64_t scratch_epoch = 0;          // ctx->scratch_epoch when the last phase returned (TreeGuard)
    std::vector<HostNode> level;
    std::vector<abw_tree_node> nodes;
    std::vector<abw_tree_winner> final_winners;
    TreeDev Tr;
    // sizes and strides
    int32_t D = 0, S = 0, n_narrow = 0, n_wide = 0, n_groups = 0;
    int64_t N = 0, Np = 0, Sp = 0, TBn = 1;
    // whole-call device arrays
    uint32_t* order[2] = {nullptr, nullptr};
    int32_t* wbin[2] = {nullptr, nullptr};
    uint32_t* wword[2] = {nullptr, nullptr};
    uint32_t* bits = nullptr;               // [n_wide][Wp / 32] the positions that leave the root's segment this level
    uint8_t* d_scaf_side = nullptr;
    int32_t* d_scaf_slot = nullptr;
    // the level in flight
    int32_t K = 0, n_st = 0, n_pt = 0;
    size_t o_res = 0, o_part = 0;
    const NodeDev* d_nodes = nullptr;
    const Tile *d_stiles = nullptr, *d_ptiles = nullptr;
    NodeRes* d_res = nullptr;
    NodePart* d_part = nullptr;
    unsigned long long *d_pcnt = nullptr, *d_scnt = nullptr;
    uint16_t *d_pmask = nullptr, *d_smask = nullptr;

    int32_t key_of(int32_t d, int32_t bin) const
    {
        const abw_sep_state& s = ctx->sep;
        return s.h_binval_off[d] < 0 ? bin : s.h_binval[(size_t)s.h_binval_off[d] + bin];
    }
    // solo (abw_sep_tree: one context, nobody between the phases): the winners stay on the device from the evaluation through
    // the sides to the children's sizes -- one read-back and one wait per level instead of three -- and the partition runs on
    // while the host lays out the next level
    bool solo = false;
    // the tiles' own sums (before the scan), kept from level to level: a node that keeps its parent's segment inherits them
    // (tree_wide_inherit_kernel) -- buffers of the run's own, the context's scratch slots are re-reserved by every level
    WSums4* raw[2] = {nullptr, nullptr};    // [n_wide][raw_cap], this level's / the previous level's
    WSums4* left = nullptr;                 // [n_wide][raw_cap] what left every tile in the last partition
    int64_t raw_cap = 0;
    int rcur = 0;
    bool prev_raw = false;                  // raw[rcur ^ 1] and left hold the previous level
    std::vector<int32_t> prev_tb, tile_begin;   // first position tile of every slot of the previous level / of this one (and the count behind the last)
    hipEvent_t ev_part = nullptr;           // the level's NodePart records have left the pinned block
    bool part_pending = false;
    ~TreeRun()
    {
        if (ev_part) (void)hipEventDestroy(ev_part);
        for (void* q : {(void*)raw[0], (void*)raw[1], (void*)left}) if (q) (void)hipFree(q);
    }
    int begin(int32_t min_ndps_, int64_t max_nodes_);
    int eval(abw_tree_winner* best, int64_t cap, int64_t* n_level);
    int assign(const abw_tree_winner* winners, int64_t n);
    int advance();
};

int TreeRun::begin(int32_t min_ndps_, int64_t max_nodes_)
{
    abw_sep_state& s = ctx->sep;
    SepHost& H = abw_sep_host(ctx);
    hipStream_t st = ctx->stream;
    min_ndps = min_ndps_;
    max_nodes = max_nodes_ <= 0 ? INT64_MAX : max_nodes_;
    int rc;
    D = s.D; S = s.S; n_narrow = (int32_t)H.narrow_dims.size(); n_wide = (int32_t)H.wide_dims.size();
    N = s.N; Np = (N + 7) & ~(int64_t)7; Sp = ((int64_t)S + 7) & ~(int64_t)7;      // strides of the position / scaffold arrays
    Wp = (2 * Np + 31) & ~(int64_t)31;                          // (a multiple of 32: a dimension's bit map is whole words)
    if (Wp >= ((int64_t)1 << 31) - TILE_P) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_tree: more than 2^30 datapoints");
    if (S > (int32_t)W_SC) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_tree: more than 2^28 scaffolds");
    n_groups = (int32_t)H.groups.size();
    TBn = std::max<int64_t>(1, n_narrow ? H.toffn[n_narrow] : 0);      // a node's slice: the narrow dimensions in H.nsort order

    // buffers of the run (grow-only scratch of the context: no other entry point may run on the context until the run is over)
    abw_devbuf &b_order = ctx->scratch[0], &b_wbin = ctx->scratch[1], &b_wword = ctx->scratch[2], &b_side = ctx->scratch[4],
               &b_s2c = ctx->scratch[9];
    if ((rc = abw_reserve(ctx, b_order, (size_t)2 * Sp * 4))) return rc;
    if ((rc = abw_reserve(ctx, b_side, (size_t)S + 64 + (size_t)S * 4))) return rc;
    if ((rc = abw_reserve(ctx, b_s2c, (size_t)S * 4))) return rc;
    if (n_wide) {
        if ((rc = abw_reserve(ctx, b_wbin, (size_t)2 * n_wide * Wp * 4))) return rc;
        if ((rc = abw_reserve(ctx, b_wword, (size_t)2 * n_wide * Wp * 4))) return rc;
    }

    Tr.P.bins = s.d_bins ? s.d_bins : s.d_keys; Tr.P.N = N;
    Tr.P.scaf_first = s.d_scaf_first; Tr.P.scaf_ndps = s.d_scaf_ndps; Tr.P.scaf_len = s.d_scaf_len;
    Tr.P.scg_off = s.d_scg_off; Tr.P.scg_ids = s.d_scg_ids; Tr.P.n_scg = s.n_scg;
    Tr.P.T = s.d_T; Tr.P.Toff = s.d_Toff;
    Tr.P.h_st = nullptr; Tr.P.h_tot = nullptr; Tr.P.h_bp = nullptr; Tr.P.scg_lo = nullptr; Tr.P.scg_hi = nullptr; Tr.P.range = nullptr;
    if ((rc = abw_tree_static_ensure(ctx))) return rc;
    if (n_wide) {                                              // (after the static build, which uses this slot itself)
        abw_devbuf& b_bits = ctx->scratch[3];
        if ((rc = abw_reserve(ctx, b_bits, (size_t)n_wide * (Wp / 8) + 64))) return rc;
        bits = (uint32_t*)b_bits.p;
        ABW_HIP(ctx, hipMemsetAsync(bits, 0, (size_t)n_wide * (Wp / 8), st));      // once: tree_part_count_kernel clears what it reads
    }
    Tr.narrow = H.d_nsort; Tr.n_narrow = n_narrow; Tr.toffn = H.d_toffn; Tr.TBn = TBn;
    Tr.groups = H.d_groups; Tr.n_groups = n_groups; Tr.nk16 = (const uint4*)H.d_nk16; Tr.ts8 = (const uint4*)H.d_ts8; Tr.srec = (const int4*)H.d_srec;
    Tr.wide = H.d_wide; Tr.n_wide = n_wide; Tr.dp2scaf = nullptr; Tr.dim_slot = H.d_dim_slot;
    Tr.wext = (const uint4*)H.d_wext; Tr.wpack = H.d_wpack; Tr.D = D; Tr.S = S;
    Tr.sinfo = H.d_sinfo; Tr.stot = H.d_sinfo ? (const uint8_t*)(H.d_sinfo + S) : nullptr; Tr.sc_mask = H.d_sinfo ? W_SC22 : W_SC;

    order[0] = (uint32_t*)b_order.p; order[1] = (uint32_t*)b_order.p + Sp;
    wbin[0] = (int32_t*)b_wbin.p; wbin[1] = n_wide ? (int32_t*)b_wbin.p + (int64_t)n_wide * Wp : nullptr;
    wword[0] = (uint32_t*)b_wword.p; wword[1] = n_wide ? (uint32_t*)b_wword.p + (int64_t)n_wide * Wp : nullptr;
    d_scaf_side = (uint8_t*)b_side.p;
    d_scaf_slot = (int32_t*)((uint8_t*)b_side.p + (((size_t)S + 63) & ~(size_t)63));
    hipLaunchKernelGGL(tree_iota_kernel, dim3(ctx->num_cu * 4), dim3(256), 0, st, order[0], S);
    ABW_LAUNCH_CHECK(ctx);
    ABW_HIP(ctx, hipMemsetAsync(d_scaf_slot, 0, (size_t)S * 4, st));          // every scaffold starts in node slot 0 (the root)
    ABW_HIP(ctx, hipMemsetAsync(b_s2c.p, 0, (size_t)S * 4, st));
    if (n_wide) {
        ABW_HIP(ctx, hipMemcpy2DAsync(wbin[0], (size_t)Wp * 4, H.d_sbin, (size_t)N * 4, (size_t)N * 4, (size_t)n_wide, hipMemcpyDeviceToDevice, st));
        ABW_HIP(ctx, hipMemcpy2DAsync(wword[0], (size_t)Wp * 4, H.d_wword, (size_t)Np * 4, (size_t)N * 4, (size_t)n_wide, hipMemcpyDeviceToDevice, st));

    }
    level.assign(1, HostNode{1, S, N, -1, 0, N, true, -1});        // root_seg: still in the static order, where pos_of holds
    nodes.clear();
    next_id = 2; visited = 0; cur = 0; wcur = 0; arena_end = N; phase = 0;
    return ABW_OK;
}

// One level up to this context's best threshold of every node.  *n_level = 0: the recursion is over.
int TreeRun::eval(abw_tree_winner* best, int64_t cap, int64_t* n_level)
{
    abw_sep_state& s = ctx->sep;
    SepHost& H = abw_sep_host(ctx);
    hipStream_t st = ctx->stream;
    int rc;
    if (phase == 3 || level.empty() || visited >= max_nodes) { phase = 3; *n_level = 0; return ABW_OK; }
    if (phase != 0) return abw_fail(ctx, ABW_ERR_STATE, "abw_sep_tree_eval: the level in flight waits for %s", phase == 1 ? "abw_sep_tree_assign" : "abw_sep_tree_advance");
    if ((int64_t)level.size() > max_nodes - visited) level.resize((size_t)(max_nodes - visited));   // the work list is cut in id order
    K = (int32_t)level.size();
    *n_level = K;
    if (cap < K || !best) return abw_fail(ctx, ABW_ERR_CAPACITY, "abw_sep_tree_eval: %d nodes in the level, capacity %lld", K, (long long)cap);
    if (part_pending) { ABW_HIP(ctx, hipEventSynchronize(ev_part)); part_pending = false; }      // the pinned block is written below
    abw_devbuf &b_scg = ctx->scratch[5], &b_lvl = ctx->scratch[6], &b_tile = ctx->scratch[7];
    abw_devbuf* b_hist[2] = {&ctx->scratch[10], &ctx->scratch[11]};
    // ---- node table, tiles, and the plan for the histograms of this level
    std::vector<NodeDev> nd((size_t)K);
    std::vector<Tile> stiles, ptiles, atiles;
    std::vector<int32_t> zero_slots;
    std::vector<SubJob> subs;
    int32_t sb = 0;
    for (int32_t k = 0; k < K; k++) {
        nd[k].scaf_begin = sb; nd[k].scaf_end = sb + level[k].m;
        nd[k].dp_begin = (int32_t)level[k].seg_begin; nd[k].dp_end = (int32_t)(level[k].seg_begin + level[k].seg_len); nd[k].n = (int32_t)level[k].n;
        make_tiles(stiles, k, nd[k].scaf_begin, nd[k].scaf_end, TILE_S, nd[k].stile_begin, nd[k].stile_end);
        make_tiles(ptiles, k, nd[k].dp_begin, nd[k].dp_end, TILE_P, nd[k].ptile_begin, nd[k].ptile_end);
        sb += level[k].m;
    }
    // children come in pairs (2j, 2j + 1) of one parent: the smaller is accumulated, the bigger subtracted; a node without
    // its sibling in this (cut) level, and the root, are accumulated
    // (a level that accumulates few scaffolds -- the smaller children of a caterpillar -- cuts them into smaller tiles: the
    // kernel is a chain of dependent loads per scaffold, and a few hundred workgroups do not fill the device)
    std::vector<int32_t> acc_nodes;
    int64_t acc_scaf = 0;
    for (int32_t k = 0; k < K; k++) {
        const int32_t sib = (level[k].parent >= 0) ? (k ^ 1) : -1;
        const bool has_sib = sib >= 0 && sib < K && level[sib].parent == level[k].parent;
        const bool bigger = has_sib && (level[k].n > level[sib].n || (level[k].n == level[sib].n && k > sib));
        if (bigger) subs.push_back(SubJob{k, level[k].parent, sib});
        else { zero_slots.push_back(k); acc_nodes.push_back(k); acc_scaf += level[k].m; }
    }
    const int tile_a = acc_scaf * std::max(1, n_groups) <= (int64_t)TILE_A * ctx->num_cu * 8 ? TILE_A / 4 : TILE_A;
    for (int32_t k : acc_nodes) {
        int32_t tb, te;
        make_tiles(atiles, k, nd[k].scaf_begin, nd[k].scaf_end, tile_a, tb, te);
    }
    n_st = (int32_t)stiles.size(); n_pt = (int32_t)ptiles.size();
    const int32_t n_at = (int32_t)atiles.size();
    // position tiles: inherited from the previous level (same segment, same cuts) or fresh
    if (n_wide && n_pt > raw_cap) {                              // (grow-only; a level that grows them inherits nothing)
        ABW_HIP(ctx, hipStreamSynchronize(st));
        for (WSums4** q : {&raw[0], &raw[1], &left}) { if (*q) (void)hipFree(*q); *q = nullptr; }
        raw_cap = (int64_t)n_pt + n_pt / 2 + 64;
        for (WSums4** q : {&raw[0], &raw[1], &left}) ABW_HIP(ctx, hipMalloc((void**)q, (size_t)n_wide * raw_cap * sizeof(WSums4)));
        prev_raw = false;
    }
    std::vector<int32_t> src_tile((size_t)n_pt, -1), fresh;
    tile_begin.assign((size_t)K + 1, n_pt);
    for (int32_t k = 0; k < K && n_wide; k++) {
        const int32_t from = level[k].inherit;
        const int32_t cnt = nd[k].ptile_end - nd[k].ptile_begin;
        tile_begin[(size_t)k] = nd[k].ptile_begin;
        const bool ok = prev_raw && from >= 0 && from + 1 < (int32_t)prev_tb.size() && prev_tb[(size_t)from + 1] - prev_tb[(size_t)from] == cnt;
        for (int32_t j = 0; j < cnt; j++) {
            if (ok) src_tile[(size_t)nd[k].ptile_begin + j] = prev_tb[(size_t)from] + j;
            else fresh.push_back(nd[k].ptile_begin + j);
        }
    }
    const int32_t n_fresh = (int32_t)fresh.size();
    std::vector<uint8_t> blob;
    const size_t o_nodes = pack(blob, nd), o_st = pack(blob, stiles), o_pt = pack(blob, ptiles), o_at = pack(blob, atiles),
                 o_zero = pack(blob, zero_slots), o_sub = pack(blob, subs), o_src = pack(blob, src_tile), o_fresh = pack(blob, fresh);
    const size_t lvl_bytes = (blob.size() + 63) & ~(size_t)63;
    o_res = lvl_bytes; o_part = o_res + (((size_t)K * sizeof(NodeRes) + 63) & ~(size_t)63);
    const size_t all_bytes = o_part + (size_t)K * sizeof(NodePart) + 64;
    if ((rc = abw_reserve(ctx, b_lvl, all_bytes))) return rc;
    if ((rc = abw_reserve_pinned(ctx, all_bytes))) return rc;
    uint8_t* hp = (uint8_t*)ctx->pinned;
    memcpy(hp, blob.data(), blob.size());
    ABW_HIP(ctx, hipMemcpyAsync(b_lvl.p, hp, lvl_bytes, hipMemcpyHostToDevice, st));
    uint8_t* dl = (uint8_t*)b_lvl.p;
    d_nodes = (const NodeDev*)(dl + o_nodes);
    d_stiles = (const Tile*)(dl + o_st);
    d_ptiles = (const Tile*)(dl + o_pt);
    const Tile* d_atiles = (const Tile*)(dl + o_at);
    const int32_t* d_zero = (const int32_t*)(dl + o_zero);
    const SubJob* d_subs = (const SubJob*)(dl + o_sub);
    const int32_t* d_src_tile = (const int32_t*)(dl + o_src);
    const int32_t* d_fresh = (const int32_t*)(dl + o_fresh);
    d_res = (NodeRes*)(dl + o_res);
    d_part = (NodePart*)(dl + o_part);

    // per-level buffers: SCG lo / hi, candidates, tile sums / bests / counts
    const int64_t scg_n = (int64_t)K * D * s.n_scg;
    if ((rc = abw_reserve(ctx, b_scg, (size_t)std::max<int64_t>(1, scg_n) * 8))) return rc;
    int32_t* d_scg_lo = (int32_t*)b_scg.p; int32_t* d_scg_hi = d_scg_lo + scg_n;
    const size_t t_cand = ((size_t)K * D * sizeof(Cand) + 63) & ~(size_t)63,
                 t_sums = ((size_t)std::max(1, n_wide) * n_pt * sizeof(WSums4) + 63) & ~(size_t)63,
                 t_best = ((size_t)std::max(1, n_wide) * n_pt * sizeof(Cand) + 63) & ~(size_t)63,
                 t_tot = ((size_t)std::max(1, n_wide) * K * sizeof(WSums4) + 63) & ~(size_t)63,
                 t_cnt = ((size_t)(std::max(1, n_wide) * (size_t)n_pt + n_st) * 8 + 63) & ~(size_t)63,
                 t_mask = (((size_t)std::max(1, n_wide) * n_pt + n_st) * 512 + 63) & ~(size_t)63;
    if ((int64_t)std::max(1, n_wide) * n_pt >= ((int64_t)1 << 31)) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_tree: too many position tiles");
    if ((rc = abw_reserve(ctx, b_tile, t_cand + t_sums + t_best + t_tot + t_cnt + t_mask + (size_t)K * 8 + 64 + 2 * (size_t)std::max(1, n_wide) * n_pt * 4))) return rc;
    Cand* d_cand = (Cand*)b_tile.p;
    WSums4* d_tsums = (WSums4*)((uint8_t*)b_tile.p + t_cand);
    Cand* d_tbest = (Cand*)((uint8_t*)b_tile.p + t_cand + t_sums);
    WSums4* d_ntot = (WSums4*)((uint8_t*)b_tile.p + t_cand + t_sums + t_best);
    d_pcnt = (unsigned long long*)((uint8_t*)b_tile.p + t_cand + t_sums + t_best + t_tot);
    d_scnt = d_pcnt + (size_t)std::max(1, n_wide) * n_pt;
    d_pmask = (uint16_t*)((uint8_t*)b_tile.p + t_cand + t_sums + t_best + t_tot + t_cnt);
    d_smask = d_pmask + (size_t)std::max(1, n_wide) * n_pt * 256;
    double* d_bar = (double*)((uint8_t*)d_pmask + t_mask);
    int32_t* d_wlist_n = (int32_t*)(d_bar + K);                                             // tiles listed by the screen for the full evaluation
    int32_t* d_wlist = d_wlist_n + 16;                                                      // (d_wlist_n[1]: the tiles listed for the screen itself)
    int32_t* d_wtodo = d_wlist + (size_t)std::max(1, n_wide) * n_pt;
    if (scg_n > 0 && (int64_t)K * s.n_scg <= SCG_LEVEL_LDS) {
        hipLaunchKernelGGL(tree_scg_lds_kernel, dim3((unsigned)D), dim3(1024), 0, st, Tr, H.d_carriers, H.n_carriers, d_scaf_slot, K, d_scg_lo, d_scg_hi);
        ABW_LAUNCH_CHECK(ctx);
    } else if (scg_n > 0) {
        hipLaunchKernelGGL(tree_scg_reset_kernel, dim3((unsigned)std::min<int64_t>((scg_n + 255) / 256, ctx->num_cu * 8)), dim3(256), 0, st, d_scg_lo, d_scg_hi, scg_n);
        ABW_LAUNCH_CHECK(ctx);
        if (H.n_carriers > 0) {
            const int64_t work = (int64_t)H.n_carriers * D;
            hipLaunchKernelGGL(tree_scg_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, Tr, H.d_carriers, H.n_carriers, d_scaf_slot, K, d_scg_lo, d_scg_hi);
            ABW_LAUNCH_CHECK(ctx);
        }
    }

    // ---- narrow dimensions: this level's slices (pool cur) from the table or from the parents' slices (pool cur ^ 1)
    if (n_narrow) {
        const size_t slice = (size_t)TBn * 20;
        if ((size_t)K * slice > ((size_t)48 << 30)) return abw_fail(ctx, ABW_ERR_NOMEM, "abw_sep_tree: %d nodes in one level need %zu bytes of histograms", K, (size_t)K * slice);
        if ((rc = abw_reserve(ctx, *b_hist[cur], (size_t)K * slice + 64))) return rc;
        const int64_t capc = (int64_t)((b_hist[cur]->cap - 64) / slice), capp = b_hist[cur ^ 1]->cap > 64 ? (int64_t)((b_hist[cur ^ 1]->cap - 64) / slice) : 0;
        unsigned long long* c_st = (unsigned long long*)b_hist[cur]->p;
        unsigned long long* c_bp = c_st + capc * TBn;
        uint32_t* c_tot = (uint32_t*)(c_bp + capc * TBn);
        const unsigned long long* p_st = (const unsigned long long*)b_hist[cur ^ 1]->p;
        const unsigned long long* p_bp = p_st ? p_st + capp * TBn : nullptr;
        const uint32_t* p_tot = p_st ? (const uint32_t*)(p_bp + capp * TBn) : nullptr;
        const unsigned gy = (unsigned)((TBn + 1023) / 1024);
        ABW_PROF(ctx, "sep_accum");
        if (!zero_slots.empty()) {
            hipLaunchKernelGGL(tree_zero_kernel, dim3((unsigned)zero_slots.size(), gy), dim3(256), 0, st, d_zero, TBn, c_st, c_bp, c_tot);
            ABW_LAUNCH_CHECK(ctx);
        }
        if (n_at) {
            // level 0 holds all scaffolds in id order: no list (accum_group_tile then takes its tiles as ranges of ids)
            const bool identity = (K == 1 && level[0].parent < 0);
            if (H.packed) hipLaunchKernelGGL(tree_accum_kernel<true>, dim3((unsigned)n_at, (unsigned)n_groups), dim3(256), 0, st, Tr, d_atiles,
                                             identity ? (const int32_t*)nullptr : (const int32_t*)order[cur], c_st, c_bp, c_tot);
            else hipLaunchKernelGGL(tree_accum_kernel<false>, dim3((unsigned)n_at, (unsigned)n_groups), dim3(256), 0, st, Tr, d_atiles,
                                    identity ? (const int32_t*)nullptr : (const int32_t*)order[cur], c_st, c_bp, c_tot);
            ABW_LAUNCH_CHECK(ctx);
        }
        if (!subs.empty()) {
            hipLaunchKernelGGL(tree_sub_kernel, dim3((unsigned)subs.size(), gy), dim3(256), 0, st, d_subs, TBn, c_st, c_bp, c_tot, p_st, p_bp, p_tot);
            ABW_LAUNCH_CHECK(ctx);
        }
        hipLaunchKernelGGL(tree_eval_narrow_kernel, dim3((unsigned)K, (unsigned)n_narrow), dim3(256), 0, st, Tr, d_nodes, min_ndps,
                           (const unsigned long long*)c_st, (const unsigned long long*)c_bp, (const uint32_t*)c_tot, d_scg_lo, d_scg_hi, d_cand);
        ABW_LAUNCH_CHECK(ctx);
    }
    // ---- wide dimensions: prefix sums along every node's sorted segment
    if (n_wide) {
        ABW_PROF(ctx, "sep_wide");
        hipLaunchKernelGGL(tree_bar_kernel, dim3((unsigned)K), dim3(64), 0, st, Tr, d_cand, d_bar);
        ABW_LAUNCH_CHECK(ctx);
        if (n_fresh < n_pt) {
            hipLaunchKernelGGL(tree_wide_inherit_kernel, dim3((unsigned)std::min<int64_t>(((int64_t)n_pt * n_wide + 255) / 256, (int64_t)ctx->num_cu * 4)), dim3(256), 0, st,
                               d_src_tile, n_pt, n_wide, (const WSums4*)raw[rcur ^ 1], (const WSums4*)left, raw[rcur], raw_cap);
            ABW_LAUNCH_CHECK(ctx);
        }
        if (n_fresh) {
            hipLaunchKernelGGL(tree_wide_sums_kernel, dim3((unsigned)n_fresh, (unsigned)n_wide), dim3(256), 0, st, Tr, d_ptiles, n_fresh < n_pt ? d_fresh : (const int32_t*)nullptr,
                               wword[wcur], Wp, raw[rcur], raw_cap);
            ABW_LAUNCH_CHECK(ctx);
        }
        hipLaunchKernelGGL(tree_wide_scan_kernel, dim3((unsigned)K, (unsigned)n_wide), dim3(64), 0, st, d_nodes, n_pt, (const WSums4*)raw[rcur], raw_cap, d_tsums, d_ntot, K);
        ABW_LAUNCH_CHECK(ctx);
        ABW_HIP(ctx, hipMemsetAsync(d_wlist_n, 0, 8, st));
        const int64_t n_items = (int64_t)n_pt * n_wide;
        hipLaunchKernelGGL(tree_wide_tilescreen_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, st, d_ptiles, n_pt, n_wide, d_nodes, K, min_ndps,
                           (const WSums4*)d_tsums, (const WSums4*)d_ntot, (const double*)d_bar, d_tbest, d_wtodo, d_wlist_n + 1);
        ABW_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(tree_wide_screen_kernel, dim3((unsigned)std::min<int64_t>(n_items, (int64_t)ctx->num_cu * 6)), dim3(256), 0, st, Tr, d_ptiles, n_pt, d_nodes, min_ndps,
                           wbin[wcur], wword[wcur], Wp, d_tsums, d_bar, (const int32_t*)d_wtodo, (const int32_t*)(d_wlist_n + 1), d_wlist, d_wlist_n);
        ABW_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(tree_wide_eval_kernel, dim3((unsigned)std::min<int64_t>((int64_t)n_pt * n_wide, (int64_t)ctx->num_cu * 4)), dim3(256), 0, st, Tr, d_ptiles, n_pt, d_nodes, K, min_ndps,
                           wbin[wcur], wword[wcur], Wp, d_tsums, d_ntot, d_scg_lo, d_scg_hi, d_bar, d_tbest, (const int32_t*)d_wlist, (const int32_t*)d_wlist_n);
        ABW_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(tree_wide_best_kernel, dim3((unsigned)K, (unsigned)n_wide), dim3(64), 0, st, Tr, d_nodes, n_pt, d_tbest, d_cand);
        ABW_LAUNCH_CHECK(ctx);
    }
    // ---- this context's winner of every node
    {
        ABW_PROF(ctx, "sep_eval");
        hipLaunchKernelGGL(tree_winner_kernel, dim3((unsigned)K), dim3(64), 0, st, Tr, d_nodes, d_cand, d_res);
        ABW_LAUNCH_CHECK(ctx);
    }
    if (solo) { phase = 1; return ABW_OK; }                    // the winners are read back with the children's sizes (advance)
    NodeRes* h_res = (NodeRes*)(hp + o_res);
    ABW_HIP(ctx, hipMemcpyAsync(h_res, d_res, (size_t)K * sizeof(NodeRes), hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    for (int32_t k = 0; k < K; k++) {
        const GlobalRes& w = h_res[k].win;
        abw_tree_winner& b = best[k];
        memset(&b, 0, sizeof(b));
        b.found = w.found; b.dimension = w.found ? w.dim : -1; b.key = w.found ? key_of(w.dim, w.bin) : -1;
        b.legal = w.legal; b.sensitivity = w.sens; b.specificity = w.spec;
        b.global_dimension = w.found ? w.dim + 1 : -1;
    }
    phase = 1;
    return ABW_OK;
}

// winners[k]: the node's winner over ALL contexts; dimension >= 0 when it is this context's own best (as eval returned it).
// The scaffolds of those nodes get their sides; every other entry of the level's side array stays 0.  Asynchronous.
int TreeRun::assign(const abw_tree_winner* winners, int64_t n)
{
    hipStream_t st = ctx->stream;
    if (phase != 1) return abw_fail(ctx, ABW_ERR_STATE, "abw_sep_tree_assign: no evaluated level (abw_sep_tree_eval)");
    if (solo) {                                                 // every winner is this context's own, and already where the kernel reads it
        ABW_HIP(ctx, hipMemsetAsync(d_scaf_side, 0, (size_t)S, st));
        ABW_PROF(ctx, "sep_eval");
        hipLaunchKernelGGL(tree_assign_kernel<true>, dim3((unsigned)n_st), dim3(256), 0, st, Tr, d_stiles, (const int32_t*)order[cur], d_res, d_scaf_side);
        ABW_LAUNCH_CHECK(ctx);
        phase = 2;
        return ABW_OK;
    }
    if (n != K || !winners) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_tree_assign: %lld winners for a level of %d nodes", (long long)n, K);
    NodeRes* h_res = (NodeRes*)((uint8_t*)ctx->pinned + o_res);
    final_winners.assign(winners, winners + K);
    for (int32_t k = 0; k < K; k++) {
        const abw_tree_winner& w = winners[k];
        GlobalRes& mine = h_res[k].win;
        const bool is_mine = w.found && w.dimension >= 0 && mine.found && w.dimension == mine.dim && w.key == key_of(mine.dim, mine.bin);
        if (w.found && w.dimension >= 0 && !is_mine)
            return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_tree_assign: node %d: the winner named as this context's is not the one abw_sep_tree_eval returned", k);
        if (!(is_mine && w.legal)) mine.legal = 0;              // nothing to assign here (illegal winner, or another context's)
        h_res[k].n1 = h_res[k].n2 = 0; h_res[k].m1 = h_res[k].m2 = 0;
    }
    ABW_HIP(ctx, hipMemcpyAsync(d_res, h_res, (size_t)K * sizeof(NodeRes), hipMemcpyHostToDevice, st));
    ABW_HIP(ctx, hipMemsetAsync(d_scaf_side, 0, (size_t)S, st));
    {
        ABW_PROF(ctx, "sep_eval");
        hipLaunchKernelGGL(tree_assign_kernel<false>, dim3((unsigned)n_st), dim3(256), 0, st, Tr, d_stiles, (const int32_t*)order[cur], d_res, d_scaf_side);
        ABW_LAUNCH_CHECK(ctx);
    }
    phase = 2;
    return ABW_OK;
}

// After the side array holds the sides decided by all contexts: children sizes, the verdicts of ref ClusterSeparator.cpp:125-132,
// the next level, the partition.
int TreeRun::advance()
{
    hipStream_t st = ctx->stream;
    if (phase != 2) return abw_fail(ctx, ABW_ERR_STATE, "abw_sep_tree_advance: no assigned level (abw_sep_tree_assign)");
    uint8_t* hp = (uint8_t*)ctx->pinned;
    NodeRes* h_res = (NodeRes*)(hp + o_res);
    if (!solo) {                                                // (solo: counted with the sides, tree_assign_kernel<true>)
        ABW_PROF(ctx, "sep_eval");
        hipLaunchKernelGGL(tree_count_kernel, dim3((unsigned)n_st), dim3(256), 0, st, Tr, d_stiles, (const int32_t*)order[cur], (const uint8_t*)d_scaf_side, d_res);
        ABW_LAUNCH_CHECK(ctx);
    }
    ABW_HIP(ctx, hipMemcpyAsync(h_res, d_res, (size_t)K * sizeof(NodeRes), hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    if (solo) {
        final_winners.resize((size_t)K);
        for (int32_t k = 0; k < K; k++) {
            const GlobalRes& w = h_res[k].win;
            abw_tree_winner& b = final_winners[k];
            memset(&b, 0, sizeof(b));
            b.found = w.found; b.dimension = w.found ? w.dim : -1; b.key = w.found ? key_of(w.dim, w.bin) : -1;
            b.legal = w.legal; b.sensitivity = w.sens; b.specificity = w.spec;
            b.global_dimension = w.found ? w.dim + 1 : -1;
        }
    }
    std::vector<HostNode> next;
    NodePart* h_part = (NodePart*)(hp + o_part);
    int32_t nsb = 0;
    std::vector<int32_t> split_nodes;
    for (int32_t k = 0; k < K; k++) {
        const NodeRes& r = h_res[k];
        const abw_tree_winner& w = final_winners[k];
        abw_tree_node o;
        memset(&o, 0, sizeof(o));
        o.id = level[k].id; o.ndps = level[k].n; o.n_scaf = level[k].m; o.dimension = -1; o.key = -1;
        bool split = false;
        if (w.found) {
            o.dimension = w.global_dimension; o.key = w.key;
            o.sensitivity = w.sensitivity; o.specificity = w.specificity; o.legal = w.legal;
            if (w.legal) {
                split = (int64_t)r.n1 >= min_ndps && (int64_t)r.n2 >= min_ndps;
                if (!split) { o.dimension = -1; o.key = -1; o.sensitivity = 0; o.specificity = 0; }      // best_separation->reset(), ref :130
            }
        }
        NodePart& np = h_part[k];
        memset(&np, 0, sizeof(np));
        np.split = split; np.id = o.id; np.slot1 = np.slot2 = -1;
        if (split) {
            o.split = 1; o.child1 = next_id; o.child2 = next_id + 1; o.ndps1 = (int64_t)r.n1; o.ndps2 = (int64_t)r.n2;
            next_id += 2;
            np.scaf_base1 = nsb; np.scaf_base2 = nsb + (int32_t)r.m1;
            np.slot1 = (int32_t)next.size(); np.slot2 = np.slot1 + 1;
            nsb += (int32_t)(r.m1 + r.m2);
            next.push_back(HostNode{o.child1, (int32_t)r.m1, (int64_t)r.n1, k, 0, 0, false, -1});
            next.push_back(HostNode{o.child2, (int32_t)r.m2, (int64_t)r.n2, k, 0, 0, false, -1});
            split_nodes.push_back(k);
        }
        nodes.push_back(o);
        visited++;
    }
    // Where the children's positions in the wide dimensions' arrays go.  The bigger child of a split keeps its parent's segment
    // as long as at least half of the segment is its own (the positions of its sibling are copied out behind everything in use
    // and marked dead where they were); otherwise both children are copied out and the segment is abandoned.  When the
    // copies do not fit behind what is in use, this level compacts: every child is copied into the other buffer, from 0.
    bool compact = false;
    int32_t n_leavers = 0, leavers_at = 0;                     // of the node that still lives in the root's segment, if it keeps it: the child that leaves, in the next order
    if (n_wide) {
        auto keeps = [&](int32_t k) -> int {                   // 1 / 2: that child keeps the segment; 0: both are copied out
            const NodeRes& r = h_res[k];
            const int64_t big = (int64_t)std::max(r.n1, r.n2);
            if (2 * big < level[k].seg_len) return 0;          // (keeping the root's segment longer -- down to 1/16 live -- was slower: 88 -> 95 ms at 2M scaffolds)
            return r.n1 > r.n2 ? 1 : 2;
        };
        int64_t need = 0;
        for (int32_t k : split_nodes) { const int kp = keeps(k); need += (kp == 1 ? 0 : (int64_t)h_res[k].n1) + (kp == 2 ? 0 : (int64_t)h_res[k].n2); }
        compact = arena_end + need > Wp;
        int64_t at = compact ? 0 : arena_end;
        for (int32_t k : split_nodes) {
            const NodeRes& r = h_res[k];
            NodePart& np = h_part[k];
            const int kp = compact ? 0 : keeps(k);
            HostNode &c1 = next[(size_t)np.slot1], &c2 = next[(size_t)np.slot2];
            if (kp && level[k].root_seg) { np.bits = 1; n_leavers = (int32_t)(kp == 1 ? r.m2 : r.m1); leavers_at = kp == 1 ? np.scaf_base2 : np.scaf_base1; (kp == 1 ? c1 : c2).root_seg = true; }
            if (kp == 1) { np.dp_base1 = -1; c1.seg_begin = level[k].seg_begin; c1.seg_len = level[k].seg_len; c1.inherit = k; }
            else { np.dp_base1 = (int32_t)at; c1.seg_begin = at; c1.seg_len = (int64_t)r.n1; at += (int64_t)r.n1; }
            if (kp == 2) { np.dp_base2 = -1; c2.seg_begin = level[k].seg_begin; c2.seg_len = level[k].seg_len; c2.inherit = k; }
            else { np.dp_base2 = (int32_t)at; c2.seg_begin = at; c2.seg_len = (int64_t)r.n2; at += (int64_t)r.n2; }
        }
        arena_end = at;
    }
    // ---- partition into the next level's order (terminal nodes' scaffolds get their cluster id)
    ABW_HIP(ctx, hipMemcpyAsync(d_part, h_part, (size_t)K * sizeof(NodePart), hipMemcpyHostToDevice, st));
    if (solo) {
        if (!ev_part) ABW_HIP(ctx, hipEventCreateWithFlags(&ev_part, hipEventDisableTiming));
        ABW_HIP(ctx, hipEventRecord(ev_part, st));
        part_pending = true;
    }
    {
        ABW_PROF(ctx, "sep_partition");
        hipLaunchKernelGGL(tree_part_count_kernel, dim3((unsigned)n_st, 1), dim3(256), 0, st, d_stiles, n_st, d_part, (const uint32_t*)order[cur], (int64_t)0, d_scaf_side,
                           W_SC, (uint8_t*)nullptr, d_scnt, d_smask);
        ABW_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(tree_part_scan_kernel, dim3((unsigned)K, 1), dim3(64), 0, st, d_nodes, d_part, n_st, d_scnt, 1);
        ABW_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(tree_part_scatter_kernel, dim3((unsigned)n_st, 1), dim3(256), 0, st, d_stiles, n_st, d_nodes, d_part, (const uint32_t*)order[cur],
                           (const int32_t*)nullptr, (int64_t)0, d_smask, d_scnt, order[cur ^ 1], (int32_t*)nullptr,
                           (uint32_t*)nullptr, 1, (int32_t*)ctx->scratch[9].p, d_scaf_slot, Tr, (WSums4*)nullptr, (int64_t)0);
        ABW_LAUNCH_CHECK(ctx);
        if (n_wide && !next.empty()) {
            if (n_leavers > 0) {
                hipLaunchKernelGGL(tree_mark_kernel, dim3((unsigned)((n_leavers + 255) / 256), (unsigned)n_wide), dim3(256), 0, st, Tr, (const int32_t*)(order[cur ^ 1] + leavers_at), n_leavers,
                                   (const int32_t*)abw_sep_host(ctx).d_posof, N, bits, Wp / 32);
                ABW_LAUNCH_CHECK(ctx);
            }
            hipLaunchKernelGGL(tree_part_count_kernel, dim3((unsigned)n_pt, (unsigned)n_wide), dim3(256), 0, st, d_ptiles, n_pt, d_part, (const uint32_t*)wword[wcur], Wp, d_scaf_side,
                               Tr.sc_mask, n_leavers > 0 ? (uint8_t*)bits : (uint8_t*)nullptr, d_pcnt, d_pmask);
            ABW_LAUNCH_CHECK(ctx);
            hipLaunchKernelGGL(tree_part_scan_kernel, dim3((unsigned)K, (unsigned)n_wide), dim3(64), 0, st, d_nodes, d_part, n_pt, d_pcnt, 0);
            ABW_LAUNCH_CHECK(ctx);
            const int wto = compact ? (wcur ^ 1) : wcur;
            hipLaunchKernelGGL(tree_part_scatter_kernel, dim3((unsigned)n_pt, (unsigned)n_wide), dim3(256), 0, st, d_ptiles, n_pt, d_nodes, d_part, (const uint32_t*)wword[wcur],
                               (const int32_t*)wbin[wcur], Wp, d_pmask, d_pcnt, wword[wto], wbin[wto],
                               compact ? (uint32_t*)nullptr : wword[wcur], 0, (int32_t*)nullptr, (int32_t*)nullptr, Tr, compact ? (WSums4*)nullptr : left, raw_cap);
            ABW_LAUNCH_CHECK(ctx);
            wcur = wto;
        }
    }
    // the level's pinned block is re-used by the next level's upload: the copy of h_part must have left it (solo: the next
    // eval waits for ev_part instead, and the partition runs on behind the host)
    if (!solo) ABW_HIP(ctx, hipStreamSynchronize(st));
    cur ^= 1;
    // what the next level may inherit: this level's tile sums (raw[rcur]) and what left every tile (left)
    prev_tb = tile_begin;
    prev_raw = n_wide > 0 && !compact;
    rcur ^= 1;
    level.swap(next);
    phase = (level.empty() || visited >= max_nodes) ? 3 : 0;
    return ABW_OK;
}

TreeRun* run_of(abw_ctx* ctx, const char* who, bool create)
{
    if (!ctx) return nullptr;
    if (!ctx->sep.loaded) { abw_fail(ctx, ABW_ERR_STATE, "%s: call abw_sep_load first", who); return nullptr; }
    SepHost& H = abw_sep_host(ctx);
    if (H.sharded) { abw_fail(ctx, ABW_ERR_STATE, "%s: the table is one shard of a scaffold-sharded problem (abw_sep_shard_load); use abw_sep_shard_*", who); return nullptr; }
    if (create) {
        abw_tree_run_free(H.tree_run);
        TreeRun* r = new TreeRun();
        r->ctx = ctx;
        H.tree_run = r;
    }
    if (!H.tree_run) { abw_fail(ctx, ABW_ERR_STATE, "%s: no recursion in progress (abw_sep_tree_begin)", who); return nullptr; }
    return (TreeRun*)H.tree_run;
}

}  // namespace

void abw_tree_run_free(void* run) { delete (TreeRun*)run; }

static SepDev static_params(const abw_sep_state& s)
{
    SepDev P;
    P.bins = s.d_bins ? s.d_bins : s.d_keys; P.N = s.N;
    P.scaf_first = s.d_scaf_first; P.scaf_ndps = s.d_scaf_ndps; P.scaf_len = s.d_scaf_len;
    P.scg_off = s.d_scg_off; P.scg_ids = s.d_scg_ids; P.n_scg = s.n_scg;
    P.T = s.d_T; P.Toff = s.d_Toff;
    P.h_st = nullptr; P.h_tot = nullptr; P.h_bp = nullptr; P.scg_lo = nullptr; P.scg_hi = nullptr; P.range = nullptr;
    return P;
}

// The static tables of the wide dimensions (position words, per-scaffold records, SCG carriers), built at the first call that
// needs them on a loaded table -- abw_sep_tree_begin, or abw_sep_node on a node large enough for the scan.
int abw_tree_static_ensure(abw_ctx* ctx)
{
    abw_sep_state& s = ctx->sep;
    SepHost& H = abw_sep_host(ctx);
    const size_t n_wide = H.wide_dims.size();
    if (!(n_wide && !(H.d_wword && H.d_wpack))) return ABW_OK;
    int rc;
    if ((rc = abw_reserve(ctx, ctx->scratch[8], (size_t)s.N * 4 + 64))) return rc;
    int32_t* d_dp2scaf = (int32_t*)ctx->scratch[8].p;                        // dp -> scaffold: only this build needs it
    std::vector<int32_t> d2s((size_t)s.N);
    for (int32_t sc = 0; sc < s.S; sc++) std::fill(d2s.begin() + s.h_scaf_first[sc], d2s.begin() + s.h_scaf_first[sc] + s.h_scaf_ndps[sc], sc);
    ABW_HIP(ctx, hipMemcpyAsync(d_dp2scaf, d2s.data(), (size_t)s.N * 4, hipMemcpyHostToDevice, ctx->stream));
    ABW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return tree_static_build(ctx, H, static_params(s), d_dp2scaf);
}

// ---- abw_sep_node's wide dimensions (see node_wide_*_kernel): host side --------------------------------------------------
namespace {
struct NodeWide { TreeDev Tr; int32_t n_tiles, n_wide; WSums4 *tile_sums, *total; Cand* tile_best; double* bar; int32_t *list_n, *list, *todo; };

int node_wide_plan(abw_ctx* ctx, NodeWide& W)
{
    abw_sep_state& s = ctx->sep;
    SepHost& H = abw_sep_host(ctx);
    W.n_wide = (int32_t)H.wide_dims.size();
    W.n_tiles = (int32_t)((s.N + TILE_P - 1) / TILE_P);
    const size_t o_sums = 0, o_tot = o_sums + (size_t)W.n_wide * W.n_tiles * sizeof(WSums4),
                 o_best = o_tot + (size_t)W.n_wide * sizeof(WSums4), o_bar = o_best + (size_t)W.n_wide * W.n_tiles * sizeof(Cand), o_list = o_bar + 64,
                 bytes = o_list + 64 + 2 * (size_t)W.n_wide * W.n_tiles * 4;
    if (H.nw_cap < bytes) {
        if (H.d_nw) { ABW_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(H.d_nw); H.d_nw = nullptr; H.nw_cap = 0; }
        ABW_HIP(ctx, hipMalloc(&H.d_nw, bytes));
        H.nw_cap = bytes;
    }
    uint8_t* p = (uint8_t*)H.d_nw;
    W.tile_sums = (WSums4*)(p + o_sums); W.total = (WSums4*)(p + o_tot); W.tile_best = (Cand*)(p + o_best); W.bar = (double*)(p + o_bar);
    W.list_n = (int32_t*)(p + o_list); W.list = W.list_n + 16; W.todo = W.list + (size_t)W.n_wide * W.n_tiles;      // list_n[1]: the length of todo
    memset(&W.Tr, 0, sizeof(W.Tr));
    W.Tr.P = static_params(s);
    W.Tr.narrow = H.d_nsort; W.Tr.n_narrow = (int32_t)H.narrow_dims.size();
    W.Tr.wide = H.d_wide; W.Tr.n_wide = W.n_wide;
    W.Tr.wext = (const uint4*)H.d_wext; W.Tr.wpack = H.d_wpack; W.Tr.D = s.D; W.Tr.S = s.S;
    W.Tr.sinfo = H.d_sinfo; W.Tr.stot = H.d_sinfo ? (const uint8_t*)(H.d_sinfo + s.S) : nullptr; W.Tr.sc_mask = H.d_sinfo ? W_SC22 : W_SC;
    return ABW_OK;
}
}  // namespace

bool abw_node_wide_usable(abw_ctx* ctx)
{
    const abw_sep_state& s = ctx->sep;
    SepHost& H = abw_sep_host(ctx);
    return !H.sharded && !H.wide_dims.empty() && s.S <= (int32_t)W_SC && s.N < ((int64_t)1 << 31) - TILE_P;
}

// tile sums and their scan for the node (d_member: a byte per scaffold, NULL = every scaffold)
int abw_node_wide_accum(abw_ctx* ctx, const uint8_t* d_member)
{
    abw_sep_state& s = ctx->sep;
    SepHost& H = abw_sep_host(ctx);
    hipStream_t st = ctx->stream;
    int rc;
    if ((rc = abw_tree_static_ensure(ctx))) return rc;
    NodeWide W;
    if ((rc = node_wide_plan(ctx, W))) return rc;
    const bool masked = d_member != nullptr;
    const int64_t Np = (s.N + 7) & ~(int64_t)7;
    const dim3 grid((unsigned)W.n_tiles, (unsigned)W.n_wide);
    if (masked) hipLaunchKernelGGL(node_wide_sums_kernel<true>, grid, dim3(256), 0, st, W.Tr, (int32_t)s.N, W.n_tiles, (const uint32_t*)H.d_wword, Np, d_member, W.tile_sums);
    else hipLaunchKernelGGL(node_wide_sums_kernel<false>, grid, dim3(256), 0, st, W.Tr, (int32_t)s.N, W.n_tiles, (const uint32_t*)H.d_wword, Np, (const uint8_t*)nullptr, W.tile_sums);
    ABW_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(node_wide_scan_kernel, dim3((unsigned)W.n_wide), dim3(64), 0, st, W.n_tiles, W.tile_sums, W.total);
    ABW_LAUNCH_CHECK(ctx);
    return ABW_OK;
}

// every threshold of the wide dimensions, once the narrow dimensions' records are in d_dimbest (their best product is the bar)
// exact_per_dim: the caller wants every dimension's own winner (abw_sep_node's per_dim), not only the node's: no bar then
int abw_node_wide_eval(abw_ctx* ctx, const uint8_t* d_member, bool exact_per_dim, int32_t min_ndps, const int32_t* scg_lo, const int32_t* scg_hi, DimBestDev* d_dimbest)
{
    abw_sep_state& s = ctx->sep;
    SepHost& H = abw_sep_host(ctx);
    hipStream_t st = ctx->stream;
    int rc;
    NodeWide W;
    if ((rc = node_wide_plan(ctx, W))) return rc;
    const int64_t Np = (s.N + 7) & ~(int64_t)7;
    if (exact_per_dim) ABW_HIP(ctx, hipMemsetAsync(W.bar, 0, 8, st));
    else {
        hipLaunchKernelGGL(node_bar_kernel, dim3(1), dim3(64), 0, st, W.Tr, (const DimBestDev*)d_dimbest, W.bar);
        ABW_LAUNCH_CHECK(ctx);
    }
    const dim3 grid((unsigned)W.n_tiles, (unsigned)W.n_wide);
    const int64_t all = (int64_t)W.n_tiles * W.n_wide;
    const int32_t* list = nullptr;
    if (!exact_per_dim) {                                       // with a bar most tiles are done after the screen
        ABW_HIP(ctx, hipMemsetAsync(W.list_n, 0, 8, st));
        hipLaunchKernelGGL(node_wide_tilescreen_kernel, dim3((unsigned)((all + 255) / 256)), dim3(256), 0, st, W.n_tiles, W.n_wide, min_ndps, (const WSums4*)W.tile_sums,
                           (const WSums4*)W.total, (const double*)W.bar, W.tile_best, W.todo, W.list_n + 1);
        ABW_LAUNCH_CHECK(ctx);
        const dim3 sgrid((unsigned)std::min<int64_t>(all, (int64_t)ctx->num_cu * 6));
        if (d_member) hipLaunchKernelGGL(node_wide_screen_kernel<true>, sgrid, dim3(256), 0, st, W.Tr, (int32_t)s.N, W.n_tiles, min_ndps, (const uint32_t*)H.d_wword, Np, d_member,
                                         (const WSums4*)W.tile_sums, (const WSums4*)W.total, (const double*)W.bar, (const int32_t*)W.todo, (const int32_t*)(W.list_n + 1), W.list, W.list_n);
        else hipLaunchKernelGGL(node_wide_screen_kernel<false>, sgrid, dim3(256), 0, st, W.Tr, (int32_t)s.N, W.n_tiles, min_ndps, (const uint32_t*)H.d_wword, Np, (const uint8_t*)nullptr,
                                (const WSums4*)W.tile_sums, (const WSums4*)W.total, (const double*)W.bar, (const int32_t*)W.todo, (const int32_t*)(W.list_n + 1), W.list, W.list_n);
        ABW_LAUNCH_CHECK(ctx);
        list = W.list;
    }
    const dim3 egrid((unsigned)std::min<int64_t>(all, exact_per_dim ? all : (int64_t)ctx->num_cu * 4));
    if (d_member) hipLaunchKernelGGL(node_wide_eval_kernel<true>, egrid, dim3(256), 0, st, W.Tr, (int32_t)s.N, W.n_tiles, min_ndps, (const int32_t*)H.d_sbin, (const uint32_t*)H.d_wword, Np,
                                   d_member, (const WSums4*)W.tile_sums, (const WSums4*)W.total, scg_lo, scg_hi, (const double*)W.bar, W.tile_best, list, (const int32_t*)W.list_n);
    else hipLaunchKernelGGL(node_wide_eval_kernel<false>, egrid, dim3(256), 0, st, W.Tr, (int32_t)s.N, W.n_tiles, min_ndps, (const int32_t*)H.d_sbin, (const uint32_t*)H.d_wword, Np,
                            (const uint8_t*)nullptr, (const WSums4*)W.tile_sums, (const WSums4*)W.total, scg_lo, scg_hi, (const double*)W.bar, W.tile_best, list, (const int32_t*)W.list_n);
    ABW_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(node_wide_best_kernel, dim3((unsigned)W.n_wide), dim3(64), 0, st, W.Tr, W.n_tiles, (const Cand*)W.tile_best, (const WSums4*)W.total, d_dimbest);
    ABW_LAUNCH_CHECK(ctx);
    return ABW_OK;
}

namespace {
// The level of a recursion lives in the context's scratch buffers between the calls of the phase API.  Another entry point
// that reserves one of them in between (it may grow = free it, or simply write it) would leave the run with stale device
// pointers: every phase checks that nobody did, and its own reservations do not count.
struct TreeGuard {
    abw_ctx* ctx; TreeRun* r; bool was_quiet;
    TreeGuard(abw_ctx* c, TreeRun* run) : ctx(c), r(run), was_quiet(c->scratch_quiet) { ctx->scratch_quiet = true; }
    ~TreeGuard() { ctx->scratch_quiet = was_quiet; r->scratch_epoch = ctx->scratch_epoch; }
    int stale(const char* who) const
    {
        if (r->scratch_epoch == ctx->scratch_epoch) return ABW_OK;
        r->phase = 3;
        return abw_fail(ctx, ABW_ERR_STATE, "%s: another entry point used this context's work buffers since the recursion's last call; "
                        "start it again (abw_sep_tree_begin), or give the other work a context of its own", who);
    }
};
}  // namespace

extern "C" {

int abw_sep_tree_begin(abw_ctx* ctx, int32_t min_ndps, int64_t max_nodes)
{
    if (!ctx) return ABW_ERR_ARG;
    if (min_ndps < 0) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_tree_begin: bad argument");
    TreeRun* r = run_of(ctx, "abw_sep_tree_begin", true);
    if (!r) return ABW_ERR_STATE;
    abw_device_guard g(ctx->device);
    TreeGuard tg(ctx, r);
    return r->begin(min_ndps, max_nodes);
}

int abw_sep_tree_eval(abw_ctx* ctx, abw_tree_winner* best, int64_t cap, int64_t* n_level)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!n_level) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_tree_eval: bad argument");
    TreeRun* r = run_of(ctx, "abw_sep_tree_eval", false);
    if (!r) return ABW_ERR_STATE;
    abw_device_guard g(ctx->device);
    TreeGuard tg(ctx, r);
    if (int rc = tg.stale("abw_sep_tree_eval")) return rc;
    return r->eval(best, cap, n_level);
}

int abw_sep_tree_assign(abw_ctx* ctx, const abw_tree_winner* winners, int64_t n, uint8_t** d_side, int64_t* side_count)
{
    if (!ctx) return ABW_ERR_ARG;
    TreeRun* r = run_of(ctx, "abw_sep_tree_assign", false);
    if (!r) return ABW_ERR_STATE;
    abw_device_guard g(ctx->device);
    TreeGuard tg(ctx, r);
    if (int rc0 = tg.stale("abw_sep_tree_assign")) return rc0;
    const int rc = r->assign(winners, n);
    if (!rc && d_side) *d_side = r->d_scaf_side;
    if (!rc && side_count) *side_count = r->S;
    return rc;
}

int abw_sep_tree_advance(abw_ctx* ctx)
{
    if (!ctx) return ABW_ERR_ARG;
    TreeRun* r = run_of(ctx, "abw_sep_tree_advance", false);
    if (!r) return ABW_ERR_STATE;
    abw_device_guard g(ctx->device);
    TreeGuard tg(ctx, r);
    if (int rc = tg.stale("abw_sep_tree_advance")) return rc;
    return r->advance();
}

int abw_sep_tree_result(abw_ctx* ctx, abw_tree_node* nodes_out, int64_t node_cap, int64_t* n_nodes, int32_t* scaf2cluster)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!n_nodes || (node_cap > 0 && !nodes_out)) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_tree_result: bad argument");
    TreeRun* r = run_of(ctx, "abw_sep_tree_result", false);
    if (!r) return ABW_ERR_STATE;
    if (r->phase == 1 || r->phase == 2) return abw_fail(ctx, ABW_ERR_STATE, "abw_sep_tree_result: a level is in flight");
    abw_device_guard g(ctx->device);
    TreeGuard tg(ctx, r);
    if (scaf2cluster) { if (int rc = tg.stale("abw_sep_tree_result")) return rc; }
    const int64_t n = (int64_t)r->nodes.size();
    *n_nodes = n;
    for (int64_t i = 0; i < std::min(n, node_cap); i++) nodes_out[i] = r->nodes[(size_t)i];
    if (scaf2cluster) {
        ABW_HIP(ctx, hipMemcpyAsync(scaf2cluster, ctx->scratch[9].p, (size_t)r->S * 4, hipMemcpyDeviceToHost, ctx->stream));
        ABW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (n > node_cap) return abw_fail(ctx, ABW_ERR_CAPACITY, "abw_sep_tree: %lld nodes, capacity %lld", (long long)n, (long long)node_cap);
    return ABW_OK;
}

// The winner of every node over several contexts' bests: the larger product sensitivity x specificity, on equal products the
// smaller dimension of the whole table (ref ...Specificity.h:25-30 + ClusterSeparator.cpp:11-16, dimensions in ascending order).
// all: [n_ctx][n] as abw_sep_tree_eval returned them, global_dimension filled in by the caller.  out[k].dimension is kept
// from the winning record; the caller hands record k back to context c with dimension = -1 unless c == owner[k].
int abw_sep_tree_merge(const abw_tree_winner* all, int32_t n_ctx, int64_t n, abw_tree_winner* out, int32_t* owner)
{
    if (!all || !out || n_ctx <= 0 || n < 0) return ABW_ERR_ARG;
    for (int64_t k = 0; k < n; k++) {
        int32_t best = -1;
        for (int32_t c = 0; c < n_ctx; c++) {
            const abw_tree_winner& w = all[(int64_t)c * n + k];
            if (!w.found) continue;
            if (best < 0) { best = c; continue; }
            const abw_tree_winner& b = all[(int64_t)best * n + k];
            const double pw = w.sensitivity * w.specificity, pb = b.sensitivity * b.specificity;
            if (pw > pb || (pw == pb && w.global_dimension < b.global_dimension)) best = c;
        }
        if (best >= 0) out[k] = all[(int64_t)best * n + k];
        else { memset(&out[k], 0, sizeof(out[k])); out[k].dimension = -1; out[k].key = -1; out[k].global_dimension = -1; }
        if (owner) owner[k] = best;
    }
    return ABW_OK;
}

// The whole recursion on one context: the loop over the level calls above.
int abw_sep_tree(abw_ctx* ctx, int32_t min_ndps, int64_t max_nodes, abw_tree_node* nodes_out, int64_t node_cap,
                 int64_t* n_nodes, int32_t* scaf2cluster)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!n_nodes || (node_cap > 0 && !nodes_out) || min_ndps < 0) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_tree: bad argument");
    *n_nodes = 0;
    int rc;
    if ((rc = abw_sep_tree_begin(ctx, min_ndps, max_nodes))) return rc;
    ((TreeRun*)abw_sep_host(ctx).tree_run)->solo = true;
    std::vector<abw_tree_winner> best(256);
    for (;;) {
        int64_t k = 0;
        rc = abw_sep_tree_eval(ctx, best.data(), (int64_t)best.size(), &k);
        if (rc == ABW_ERR_CAPACITY && k > (int64_t)best.size()) {           // the level is wider than the buffer: it has not started
            best.resize((size_t)k);
            continue;
        }
        if (rc) return rc;
        if (k == 0) break;
        if ((rc = abw_sep_tree_assign(ctx, best.data(), k, nullptr, nullptr))) return rc;
        if ((rc = abw_sep_tree_advance(ctx))) return rc;
    }
    {
        abw_device_guard g(ctx->device);
        ABW_HIP(ctx, hipStreamSynchronize(ctx->stream));          // the last level's partition may still be running
    }
    return abw_sep_tree_result(ctx, nodes_out, node_cap, n_nodes, scaf2cluster);
}

// The level in flight of a recursion (after abw_sep_tree_eval): its nodes and the device arrays that say where every scaffold
// of the table is -- what a context holding OTHER columns of some of the scaffolds (abw_sep_stree_*) needs to follow the walk.
int abw_sep_tree_level(abw_ctx* ctx, abw_tree_level* info, abw_tree_level_node* nodes_out, int64_t cap)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!info) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_tree_level: bad argument");
    TreeRun* r = run_of(ctx, "abw_sep_tree_level", false);
    if (!r) return ABW_ERR_STATE;
    if (r->phase != 1 && r->phase != 2) return abw_fail(ctx, ABW_ERR_STATE, "abw_sep_tree_level: no evaluated level (abw_sep_tree_eval)");
    TreeGuard tg(ctx, r);
    if (int rc = tg.stale("abw_sep_tree_level")) return rc;
    info->n_nodes = r->K; info->n_scaf = r->S; info->d_scaf_slot = r->d_scaf_slot; info->d_side = r->d_scaf_side;
    if (cap < r->K || (r->K > 0 && !nodes_out)) return abw_fail(ctx, ABW_ERR_CAPACITY, "abw_sep_tree_level: %d nodes in the level, capacity %lld", r->K, (long long)cap);
    for (int32_t k = 0; k < r->K; k++) {
        nodes_out[k].id = r->level[k].id; nodes_out[k].n_scaf = r->level[k].m; nodes_out[k].ndps = r->level[k].n;
        nodes_out[k].parent = r->level[k].parent; nodes_out[k].reserved = 0;
    }
    return ABW_OK;
}

}  // extern "C"

// ---- the recursion with the SCAFFOLDS of the narrow columns sharded ("separator sweep (B)"; BASELINE
// configs[4]) ---------------------------------------------------------------------------------------------------------------
// A shard = one context loaded (abw_sep_load) with a contiguous range of the table's scaffolds and the narrow (signature)
// columns only; the coverage columns are spread over other contexts by DIMENSION, which walk the recursion with the level
// calls above over ALL scaffolds.  Per level a shard follows that walk:
//   accum  : its scaffolds are sorted by the node they are in (their slots come from the walking context's device array),
//            the nodes that are the smaller child of their parent (or have no sibling) are accumulated into histogram
//            slices (tree_accum_kernel), the SCG lists of every node filled from its own carriers -- and both are handed
//            out for the exchange: the slices packed side by side (20 B x narrow bins x accumulated nodes: about 0.1 MB a
//            node at the bench's tables), integer SUM over the shards; the lists MIN / MAX.
//   eval   : slices unpacked, the bigger children = parent - smaller child (every shard holds the same parents), every
//            threshold of every narrow column evaluated: the node's best narrow candidate, identical on all shards.
//   assign : the scaffolds of the shard that lie in nodes won by a narrow column get their sides (the walking contexts
//            combine the sides by MAX, as they do among themselves).
// All quantities exchanged are integers: the tree is the one of abw_sep_tree bit for bit.
namespace {

struct STreeRun {
    abw_ctx* ctx = nullptr;
    int32_t scaf_first = 0, K = 0, cur = 0, phase = 0;       // phase 0: waits for accum, 1: eval, 2: assign
    TreeDev Tr;
    int64_t TBn = 1;
    int32_t n_acc = 0;
    std::vector<abw_tree_level_node> level;
    std::vector<int32_t> acc_nodes;                           // slots accumulated this level, ascending
    std::vector<SubJob> subs;
    abw_devbuf pool[2], lvl, list, cand, ex, scg, canon;
    size_t pool_nodes[2] = {0, 0};
    std::vector<NodeRes> h_res;
    ~STreeRun() { abw_release(pool[0]); abw_release(pool[1]); abw_release(lvl); abw_release(list); abw_release(cand); abw_release(ex); abw_release(scg); abw_release(canon); }
};

// grow-only buffers of the run itself (not the context's scratch: other entry points may use that between the calls)
int own_reserve(abw_ctx* ctx, abw_devbuf& b, size_t bytes)
{
    if (bytes <= b.cap) return ABW_OK;
    if (b.p) { ABW_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    const size_t want = bytes + bytes / 4 + 256;
    ABW_HIP(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return ABW_OK;
}

// scaffolds of the shard per node (slot[s]: the node of local scaffold s in this level; acc[k] != 0: node k is accumulated)
__global__ __launch_bounds__(256) void stree_count_kernel(const int32_t* __restrict__ slot, int32_t S, const int32_t* __restrict__ acc, int32_t K, int32_t* __restrict__ counts)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    const int k = slot[s];
    if (k >= 0 && k < K && acc[k]) atomicAdd(&counts[k], 1);
}
__global__ __launch_bounds__(256) void stree_scatter_kernel(const int32_t* __restrict__ slot, int32_t S, const int32_t* __restrict__ acc, int32_t K,
                                                           const int32_t* __restrict__ off, int32_t* __restrict__ cursor, int32_t* __restrict__ list)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    const int k = slot[s];
    if (k >= 0 && k < K && acc[k]) list[off[k] + atomicAdd(&cursor[k], 1)] = s;
}

// slices of the accumulated nodes <-> one block of 64-bit words per node: st[TBn] | bp[TBn] | tot[TBn] (32-bit, two to a word),
// the columns in ASCENDING COLUMN ORDER there (a slice keeps them in the table's own order of groups, which follows the
// table's own spans and differs from shard to shard; the bins per column are shared, abw_sep_narrow_bins).
// grid (accumulated nodes, narrow columns); canon[dn]: offset of column Tr.narrow[dn] in the exchange order
__global__ __launch_bounds__(256) void stree_pack_kernel(TreeDev Tr, const int32_t* __restrict__ acc_nodes, const int64_t* __restrict__ canon, int64_t words,
                                                        unsigned long long* __restrict__ h_st, unsigned long long* __restrict__ h_bp, uint32_t* __restrict__ h_tot,
                                                        unsigned long long* __restrict__ ex, int unpack)
{
    const int dn = blockIdx.y;
    const int64_t TBn = Tr.TBn;
    const int64_t at = (int64_t)acc_nodes[blockIdx.x] * TBn + Tr.toffn[dn], to = canon[dn];
    const int T = (int)(Tr.toffn[dn + 1] - Tr.toffn[dn]);
    unsigned long long* e = ex + (int64_t)blockIdx.x * words;
    uint32_t* e32 = reinterpret_cast<uint32_t*>(e + 2 * TBn);
    for (int b = threadIdx.x; b < T; b += 256) {
        if (unpack) { h_st[at + b] = e[to + b]; h_bp[at + b] = e[TBn + to + b]; h_tot[at + b] = e32[to + b]; }
        else { e[to + b] = h_st[at + b]; e[TBn + to + b] = h_bp[at + b]; e32[to + b] = h_tot[at + b]; }
    }
    if (!unpack && (TBn & 1) && dn == 0 && threadIdx.x == 0) e32[TBn] = 0;
}

// sides of the shard's scaffolds in the nodes a narrow column won (res[k].win.legal set for exactly those).  slot / side: the
// walking context's arrays, offset to this shard's first scaffold
__global__ __launch_bounds__(256) void stree_assign_kernel(TreeDev Tr, const int32_t* __restrict__ slot, int32_t K, const NodeRes* __restrict__ res, uint8_t* __restrict__ side)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= Tr.S) return;
    const int k = slot[s];
    if (k < 0 || k >= K) return;
    const GlobalRes win = res[k].win;
    if (!win.legal) return;
    const bool swap = win.n - win.size1 < win.size1;
    const int32_t* __restrict__ col = Tr.P.bins + (int64_t)win.dim * Tr.P.N;
    const int f = Tr.P.scaf_first[s], tot = Tr.P.scaf_ndps[s];
    int c = 0;
    for (int i = 0; i < tot; i++) c += col[f + i] <= win.bin;
    const int c1 = swap ? tot - c : c;
    side[s] = (c1 > 0 && 2 * (long long)c1 >= tot) ? 1 : 2;
}

STreeRun* stree_of(abw_ctx* ctx, const char* who, bool create)
{
    if (!ctx) return nullptr;
    if (!ctx->sep.loaded) { abw_fail(ctx, ABW_ERR_STATE, "%s: call abw_sep_load first", who); return nullptr; }
    SepHost& H = abw_sep_host(ctx);
    if (H.sharded) { abw_fail(ctx, ABW_ERR_STATE, "%s: the table was loaded with abw_sep_shard_load (node-by-node shards)", who); return nullptr; }
    if (!H.wide_dims.empty()) { abw_fail(ctx, ABW_ERR_STATE, "%s: a scaffold shard holds narrow columns only (keys below %d); give the others to a context that walks them by dimension", who, NARROW_T); return nullptr; }
    if (create) { delete (STreeRun*)H.stree_run; STreeRun* r = new STreeRun(); r->ctx = ctx; H.stree_run = r; }
    if (!H.stree_run) { abw_fail(ctx, ABW_ERR_STATE, "%s: no recursion in progress (abw_sep_stree_begin)", who); return nullptr; }
    return (STreeRun*)H.stree_run;
}

}  // namespace

void abw_stree_run_free(void* run) { delete (STreeRun*)run; }

extern "C" {

int abw_sep_stree_begin(abw_ctx* ctx, int32_t scaf_first)
{
    if (!ctx) return ABW_ERR_ARG;
    if (scaf_first < 0) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_stree_begin: bad argument");
    STreeRun* r = stree_of(ctx, "abw_sep_stree_begin", true);
    if (!r) return ABW_ERR_STATE;
    abw_sep_state& s = ctx->sep;
    SepHost& H = abw_sep_host(ctx);
    r->scaf_first = scaf_first;
    memset(&r->Tr, 0, sizeof(r->Tr));
    TreeDev& Tr = r->Tr;
    Tr.P = static_params(s);
    const int32_t n_narrow = (int32_t)H.narrow_dims.size();
    r->TBn = std::max<int64_t>(1, n_narrow ? H.toffn[n_narrow] : 0);
    Tr.narrow = H.d_nsort; Tr.n_narrow = n_narrow; Tr.toffn = H.d_toffn; Tr.TBn = r->TBn;
    Tr.groups = H.d_groups; Tr.n_groups = (int32_t)H.groups.size(); Tr.nk16 = (const uint4*)H.d_nk16; Tr.ts8 = (const uint4*)H.d_ts8; Tr.srec = (const int4*)H.d_srec;
    Tr.wide = H.d_wide; Tr.n_wide = 0; Tr.dim_slot = H.d_dim_slot; Tr.D = s.D; Tr.S = s.S;
    r->cur = 0; r->phase = 0; r->K = 0; r->pool_nodes[0] = r->pool_nodes[1] = 0;
    if (n_narrow) {
        // where every column of the slice goes in the exchange order: ascending column number, the shared bins per column
        std::vector<int64_t> by_dim((size_t)s.D + 1, 0), canon((size_t)n_narrow);
        for (int32_t d = 0; d < s.D; d++) by_dim[(size_t)d + 1] = by_dim[d] + s.h_T[d];
        for (int32_t dn = 0; dn < n_narrow; dn++) canon[(size_t)dn] = by_dim[H.nsort[(size_t)dn]];
        abw_device_guard g(ctx->device);
        int rc;
        if ((rc = own_reserve(ctx, r->canon, (size_t)n_narrow * 8))) return rc;
        ABW_HIP(ctx, hipMemcpy(r->canon.p, canon.data(), (size_t)n_narrow * 8, hipMemcpyHostToDevice));
    }
    return ABW_OK;
}

int abw_sep_stree_accum(abw_ctx* ctx, const abw_tree_level_node* nodes, int64_t n_nodes, const int32_t* d_scaf_slot, abw_sep_exchange* ex)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!nodes || n_nodes <= 0 || !d_scaf_slot || !ex) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_stree_accum: bad argument");
    STreeRun* r = stree_of(ctx, "abw_sep_stree_accum", false);
    if (!r) return ABW_ERR_STATE;
    if (r->phase != 0) return abw_fail(ctx, ABW_ERR_STATE, "abw_sep_stree_accum: the level in flight waits for %s", r->phase == 1 ? "abw_sep_stree_eval" : "abw_sep_stree_assign");
    abw_device_guard g(ctx->device);
    abw_sep_state& s = ctx->sep;
    SepHost& H = abw_sep_host(ctx);
    hipStream_t st = ctx->stream;
    const int32_t K = (int32_t)n_nodes, S = s.S, D = s.D;
    const int64_t TBn = r->TBn;
    int rc;
    r->K = K;
    r->level.assign(nodes, nodes + K);
    // the plan of TreeRun::eval: children come in pairs (2j, 2j + 1) of one parent, the smaller is accumulated
    std::vector<int32_t> acc((size_t)K, 0);
    r->acc_nodes.clear(); r->subs.clear();
    for (int32_t k = 0; k < K; k++) {
        const int32_t sib = (nodes[k].parent >= 0) ? (k ^ 1) : -1;
        const bool has_sib = sib >= 0 && sib < K && nodes[sib].parent == nodes[k].parent;
        const bool bigger = has_sib && (nodes[k].ndps > nodes[sib].ndps || (nodes[k].ndps == nodes[sib].ndps && k > sib));
        if (bigger) {
            if ((size_t)nodes[k].parent >= r->pool_nodes[r->cur ^ 1]) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_stree_accum: node %d names a parent the level before did not have", k);
            r->subs.push_back(SubJob{k, nodes[k].parent, sib});
        } else { acc[k] = 1; r->acc_nodes.push_back(k); }
    }
    r->n_acc = (int32_t)r->acc_nodes.size();
    const size_t slice = (size_t)TBn * 20;
    if ((size_t)K * slice > ((size_t)48 << 30)) return abw_fail(ctx, ABW_ERR_NOMEM, "abw_sep_stree_accum: %d nodes in one level need %zu bytes of histograms", K, (size_t)K * slice);
    if ((rc = own_reserve(ctx, r->pool[r->cur], (size_t)K * slice + 64))) return rc;
    r->pool_nodes[r->cur] = (size_t)K;
    unsigned long long* c_st = (unsigned long long*)r->pool[r->cur].p;
    unsigned long long* c_bp = c_st + (int64_t)K * TBn;
    uint32_t* c_tot = (uint32_t*)(c_bp + (int64_t)K * TBn);
    // level block: acc flags | counts | offsets | cursors | acc node list | NodeDev | sub jobs (one upload), tiles follow the counts
    std::vector<NodeDev> nd((size_t)K);
    for (int32_t k = 0; k < K; k++) { memset(&nd[k], 0, sizeof(NodeDev)); nd[k].dp_begin = 0; nd[k].dp_end = (int32_t)nodes[k].ndps; nd[k].n = (int32_t)nodes[k].ndps; }
    const size_t o_acc = 0, o_cnt = o_acc + (size_t)K * 4, o_off = o_cnt + (size_t)K * 4, o_cur = o_off + (size_t)K * 4, o_an = o_cur + (size_t)K * 4,
                 o_nd = (o_an + (size_t)K * 4 + 63) & ~(size_t)63, o_sub = (o_nd + (size_t)K * sizeof(NodeDev) + 63) & ~(size_t)63,
                 o_res = (o_sub + (size_t)K * sizeof(SubJob) + 63) & ~(size_t)63, o_tiles = (o_res + (size_t)K * sizeof(NodeRes) + 63) & ~(size_t)63;
    const size_t max_tiles = (size_t)S / TILE_A + (size_t)K + 2;
    if ((rc = own_reserve(ctx, r->lvl, o_tiles + max_tiles * sizeof(Tile) + 64))) return rc;
    if ((rc = own_reserve(ctx, r->list, (size_t)std::max(1, S) * 4))) return rc;
    uint8_t* dl = (uint8_t*)r->lvl.p;
    int32_t *d_acc = (int32_t*)(dl + o_acc), *d_cnt = (int32_t*)(dl + o_cnt), *d_off = (int32_t*)(dl + o_off), *d_cur = (int32_t*)(dl + o_cur), *d_an = (int32_t*)(dl + o_an);
    std::vector<uint8_t> blob(o_tiles, 0);
    memcpy(blob.data() + o_acc, acc.data(), (size_t)K * 4);
    if (r->n_acc) memcpy(blob.data() + o_an, r->acc_nodes.data(), (size_t)r->n_acc * 4);
    memcpy(blob.data() + o_nd, nd.data(), (size_t)K * sizeof(NodeDev));
    if (!r->subs.empty()) memcpy(blob.data() + o_sub, r->subs.data(), r->subs.size() * sizeof(SubJob));
    ABW_HIP(ctx, hipMemcpyAsync(dl, blob.data(), blob.size(), hipMemcpyHostToDevice, st));
    const int32_t* slot = d_scaf_slot + r->scaf_first;
    std::vector<int32_t> cnt((size_t)K, 0), off((size_t)K + 1, 0);
    if (S > 0 && r->n_acc) {
        hipLaunchKernelGGL(stree_count_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, slot, S, (const int32_t*)d_acc, K, d_cnt);
        ABW_LAUNCH_CHECK(ctx);
        ABW_HIP(ctx, hipMemcpyAsync(cnt.data(), d_cnt, (size_t)K * 4, hipMemcpyDeviceToHost, st));
    }
    ABW_HIP(ctx, hipStreamSynchronize(st));                    // the counts are here, and blob may go
    std::vector<Tile> tiles;
    for (int32_t k = 0; k < K; k++) {
        off[k + 1] = off[k] + cnt[k];
        for (int32_t a = off[k]; a < off[k + 1]; a += TILE_A) tiles.push_back(Tile{k, a, std::min(off[k + 1], a + TILE_A)});
    }
    const Tile* d_tiles = (const Tile*)(dl + o_tiles);
    if (!tiles.empty()) {
        ABW_HIP(ctx, hipMemcpyAsync(d_off, off.data(), (size_t)K * 4, hipMemcpyHostToDevice, st));
        ABW_HIP(ctx, hipMemcpyAsync(dl + o_tiles, tiles.data(), tiles.size() * sizeof(Tile), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(stree_scatter_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, slot, S, (const int32_t*)d_acc, K, (const int32_t*)d_off, d_cur, (int32_t*)r->list.p);
        ABW_LAUNCH_CHECK(ctx);
    }
    const unsigned gy = (unsigned)((TBn + 1023) / 1024);
    {
        ABW_PROF(ctx, "sep_accum");
        if (r->n_acc) {
            hipLaunchKernelGGL(tree_zero_kernel, dim3((unsigned)r->n_acc, gy), dim3(256), 0, st, (const int32_t*)d_an, TBn, c_st, c_bp, c_tot);
            ABW_LAUNCH_CHECK(ctx);
        }
        if (!tiles.empty() && r->Tr.n_groups > 0) {
            if (H.packed) hipLaunchKernelGGL(tree_accum_kernel<true>, dim3((unsigned)tiles.size(), (unsigned)r->Tr.n_groups), dim3(256), 0, st, r->Tr, d_tiles, (const int32_t*)r->list.p, c_st, c_bp, c_tot);
            else hipLaunchKernelGGL(tree_accum_kernel<false>, dim3((unsigned)tiles.size(), (unsigned)r->Tr.n_groups), dim3(256), 0, st, r->Tr, d_tiles, (const int32_t*)r->list.p, c_st, c_bp, c_tot);
            ABW_LAUNCH_CHECK(ctx);
        }
    }
    // the SCG lists of every node from this shard's carriers
    const int64_t scg_n = (int64_t)K * D * s.n_scg;
    if ((rc = own_reserve(ctx, r->scg, (size_t)std::max<int64_t>(1, scg_n) * 8))) return rc;
    int32_t* d_scg_lo = (int32_t*)r->scg.p; int32_t* d_scg_hi = d_scg_lo + scg_n;
    if (scg_n > 0 && (int64_t)K * s.n_scg <= SCG_LEVEL_LDS) {
        hipLaunchKernelGGL(tree_scg_lds_kernel, dim3((unsigned)D), dim3(1024), 0, st, r->Tr, H.d_carriers, H.n_carriers, slot, K, d_scg_lo, d_scg_hi);
        ABW_LAUNCH_CHECK(ctx);
    } else if (scg_n > 0) {
        hipLaunchKernelGGL(tree_scg_reset_kernel, dim3((unsigned)std::min<int64_t>((scg_n + 255) / 256, ctx->num_cu * 8)), dim3(256), 0, st, d_scg_lo, d_scg_hi, scg_n);
        ABW_LAUNCH_CHECK(ctx);
        if (H.n_carriers > 0) {
            hipLaunchKernelGGL(tree_scg_kernel, dim3((unsigned)(((int64_t)H.n_carriers * D + 255) / 256)), dim3(256), 0, st, r->Tr, H.d_carriers, H.n_carriers, slot, K, d_scg_lo, d_scg_hi);
            ABW_LAUNCH_CHECK(ctx);
        }
    }
    // the slices of the accumulated nodes, packed for the exchange
    const int64_t words = 2 * TBn + (TBn + 1) / 2;
    if ((rc = own_reserve(ctx, r->ex, (size_t)std::max<int64_t>(1, (int64_t)r->n_acc * words) * 8))) return rc;
    if (r->n_acc && r->Tr.n_narrow > 0) {
        hipLaunchKernelGGL(stree_pack_kernel, dim3((unsigned)r->n_acc, (unsigned)r->Tr.n_narrow), dim3(256), 0, st, r->Tr, (const int32_t*)d_an, (const int64_t*)r->canon.p, words,
                           c_st, c_bp, c_tot, (unsigned long long*)r->ex.p, 0);
        ABW_LAUNCH_CHECK(ctx);
    }
    ex->d_hist = (int64_t*)r->ex.p; ex->hist_count = (uint64_t)((int64_t)r->n_acc * words);
    ex->d_scg_lo = d_scg_lo; ex->d_scg_hi = d_scg_hi; ex->scg_count = (uint64_t)scg_n;
    r->phase = 1;
    return ABW_OK;
}

// After the exchange buffers hold the SUM / MIN / MAX over the shards: this level's best narrow candidate of every node.
int abw_sep_stree_eval(abw_ctx* ctx, int32_t min_ndps, abw_tree_winner* best, int64_t cap)
{
    if (!ctx) return ABW_ERR_ARG;
    STreeRun* r = stree_of(ctx, "abw_sep_stree_eval", false);
    if (!r) return ABW_ERR_STATE;
    if (r->phase != 1) return abw_fail(ctx, ABW_ERR_STATE, "abw_sep_stree_eval: no accumulated level (abw_sep_stree_accum)");
    if (!best || cap < r->K || min_ndps < 0) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_stree_eval: bad argument");
    abw_device_guard g(ctx->device);
    abw_sep_state& s = ctx->sep;
    hipStream_t st = ctx->stream;
    const int32_t K = r->K, D = s.D;
    const int64_t TBn = r->TBn;
    int rc;
    unsigned long long* c_st = (unsigned long long*)r->pool[r->cur].p;
    unsigned long long* c_bp = c_st + (int64_t)K * TBn;
    uint32_t* c_tot = (uint32_t*)(c_bp + (int64_t)K * TBn);
    const int64_t Kp = (int64_t)r->pool_nodes[r->cur ^ 1];
    const unsigned long long* p_st = (const unsigned long long*)r->pool[r->cur ^ 1].p;
    const unsigned long long* p_bp = p_st ? p_st + Kp * TBn : nullptr;
    const uint32_t* p_tot = p_st ? (const uint32_t*)(p_bp + Kp * TBn) : nullptr;
    uint8_t* dl = (uint8_t*)r->lvl.p;
    const size_t o_an = (size_t)K * 16, o_nd = (o_an + (size_t)K * 4 + 63) & ~(size_t)63, o_sub = (o_nd + (size_t)K * sizeof(NodeDev) + 63) & ~(size_t)63,
                 o_res = (o_sub + (size_t)K * sizeof(SubJob) + 63) & ~(size_t)63;
    const int32_t* d_an = (const int32_t*)(dl + o_an);
    const NodeDev* d_nodes = (const NodeDev*)(dl + o_nd);
    const SubJob* d_subs = (const SubJob*)(dl + o_sub);
    NodeRes* d_res = (NodeRes*)(dl + o_res);
    const int64_t words = 2 * TBn + (TBn + 1) / 2;
    const unsigned gy = (unsigned)((TBn + 1023) / 1024);
    const int64_t scg_n = (int64_t)K * D * s.n_scg;
    const int32_t* d_scg_lo = (const int32_t*)r->scg.p; const int32_t* d_scg_hi = d_scg_lo + scg_n;
    if ((rc = own_reserve(ctx, r->cand, (size_t)K * D * sizeof(Cand) + 64))) return rc;
    Cand* d_cand = (Cand*)r->cand.p;
    {
        ABW_PROF(ctx, "sep_accum");
        if (r->n_acc && r->Tr.n_narrow > 0) {
            hipLaunchKernelGGL(stree_pack_kernel, dim3((unsigned)r->n_acc, (unsigned)r->Tr.n_narrow), dim3(256), 0, st, r->Tr, d_an, (const int64_t*)r->canon.p, words,
                               c_st, c_bp, c_tot, (unsigned long long*)r->ex.p, 1);
            ABW_LAUNCH_CHECK(ctx);
        }
        if (!r->subs.empty()) {
            hipLaunchKernelGGL(tree_sub_kernel, dim3((unsigned)r->subs.size(), gy), dim3(256), 0, st, d_subs, TBn, c_st, c_bp, c_tot, p_st, p_bp, p_tot);
            ABW_LAUNCH_CHECK(ctx);
        }
        if (r->Tr.n_narrow > 0) {
            hipLaunchKernelGGL(tree_eval_narrow_kernel, dim3((unsigned)K, (unsigned)r->Tr.n_narrow), dim3(256), 0, st, r->Tr, d_nodes, min_ndps, (const unsigned long long*)c_st,
                               (const unsigned long long*)c_bp, (const uint32_t*)c_tot, d_scg_lo, d_scg_hi, d_cand);
            ABW_LAUNCH_CHECK(ctx);
        }
        hipLaunchKernelGGL(tree_winner_kernel, dim3((unsigned)K), dim3(64), 0, st, r->Tr, d_nodes, (const Cand*)d_cand, d_res);
        ABW_LAUNCH_CHECK(ctx);
    }
    r->h_res.resize((size_t)K);
    ABW_HIP(ctx, hipMemcpyAsync(r->h_res.data(), d_res, (size_t)K * sizeof(NodeRes), hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    for (int32_t k = 0; k < K; k++) {
        const GlobalRes& w = r->h_res[k].win;
        abw_tree_winner& b = best[k];
        memset(&b, 0, sizeof(b));
        b.found = w.found; b.dimension = w.found ? w.dim : -1; b.key = w.found ? w.bin : -1;      // narrow columns: the bin is the key
        b.legal = w.legal; b.sensitivity = w.sens; b.specificity = w.spec;
        b.global_dimension = w.found ? w.dim + 1 : -1;
    }
    r->phase = 2;
    return ABW_OK;
}

// winners[k].dimension >= 0: node k was won by THIS table's column `dimension` (as eval returned it; the same on every shard);
// the shard's scaffolds in those nodes get their sides in d_side (the walking context's array, all scaffolds of the table).
int abw_sep_stree_assign(abw_ctx* ctx, const abw_tree_winner* winners, int64_t n, const int32_t* d_scaf_slot, uint8_t* d_side)
{
    if (!ctx) return ABW_ERR_ARG;
    STreeRun* r = stree_of(ctx, "abw_sep_stree_assign", false);
    if (!r) return ABW_ERR_STATE;
    if (r->phase != 2) return abw_fail(ctx, ABW_ERR_STATE, "abw_sep_stree_assign: no evaluated level (abw_sep_stree_eval)");
    if (!winners || n != r->K || !d_scaf_slot || !d_side) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_stree_assign: bad argument");
    abw_device_guard g(ctx->device);
    hipStream_t st = ctx->stream;
    const int32_t K = r->K;
    bool any = false;
    for (int32_t k = 0; k < K; k++) {
        const abw_tree_winner& w = winners[k];
        GlobalRes& mine = r->h_res[k].win;
        const bool is_mine = w.found && w.dimension >= 0 && mine.found && w.dimension == mine.dim && w.key == mine.bin;
        if (w.found && w.dimension >= 0 && !is_mine)
            return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_stree_assign: node %d: the winner named as this table's is not the one abw_sep_stree_eval returned", k);
        if (!(is_mine && w.legal)) mine.legal = 0;
        any |= mine.legal != 0;
    }
    if (any && r->Tr.S > 0) {
        uint8_t* dl = (uint8_t*)r->lvl.p;
        const size_t o_an = (size_t)K * 16, o_nd = (o_an + (size_t)K * 4 + 63) & ~(size_t)63, o_sub = (o_nd + (size_t)K * sizeof(NodeDev) + 63) & ~(size_t)63,
                     o_res = (o_sub + (size_t)K * sizeof(SubJob) + 63) & ~(size_t)63;
        NodeRes* d_res = (NodeRes*)(dl + o_res);
        ABW_HIP(ctx, hipMemcpyAsync(d_res, r->h_res.data(), (size_t)K * sizeof(NodeRes), hipMemcpyHostToDevice, st));
        ABW_PROF(ctx, "sep_eval");
        hipLaunchKernelGGL(stree_assign_kernel, dim3((unsigned)((r->Tr.S + 255) / 256)), dim3(256), 0, st, r->Tr, d_scaf_slot + r->scaf_first, K, (const NodeRes*)d_res, d_side + r->scaf_first);
        ABW_LAUNCH_CHECK(ctx);
        ABW_HIP(ctx, hipStreamSynchronize(st));                // h_res may change at the next level
    }
    r->cur ^= 1;
    r->phase = 0;
    return ABW_OK;
}

}  // extern "C"
