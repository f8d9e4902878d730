// Stage 2 — the cluster separator: ClusterSeparator::separate() for one node of the recursion.
//
// Reference behaviour restated here (reference = CK7/abawaca):
//   sweep          src/ClusterSeparatorBySensitivitySpecificity.cpp:91-161 (+ ClusterStats :6-76)
//   tie-breaks     src/ClusterSeparatorBySensitivitySpecificity.h:25-35, src/ClusterSeparator.cpp:11-16
//   gate           src/ClusterQuality.cpp:96-136, src/SCGdb.cpp:21-38
//   finalisation   src/ClusterSeparator.cpp:25-135
//
// The reference sorts every dimension and walks the thresholds while moving dps between two
// clusters.  All quantities it looks at after threshold t are COUNTS, so the sweep has an exact
// integer formulation with no sort of the N points:
//   t_s      = the (floor(n_s/2)+1)-th smallest value of scaffold s   (scaffold s flips to cluster1
//              exactly when t reaches t_s: "2*n1_s > n_s", ...Specificity.cpp:125)
//   size1(t) = #{p : v_p <= t}                tot1(t) = #{p : t_s(p) <= t}
//   TP1(t)   = #{p : max(v_p, t_s(p)) <= t}   TP2(t)  = N - #{p : min(v_p, t_s(p)) <= t}
//            and #{min <= t} = size1 + tot1 - TP1 because {max,min} = {v,t_s} as multisets
//   gate     : per SCG g, lo_g/hi_g = min/max t_s over the node's scaffolds carrying g;
//              |G1| = #{lo_g <= t}, |G2| = #{hi_g > t}, |G1 n G2| = #{lo_g <= t < hi_g};
//              bp1(t) = sum of len_s over t_s <= t.
// Kernels: (1) histograms -- narrow dimensions: sep_accum, one thread per (scaffold, dimension), order
// statistic + updates of a workgroup-private LDS histogram; wide dimensions: sep_wide_ts + sep_wide_hist, a
// walk over the dps in (static) bin order with run-length reduction in registers, no scatter;
// (2) chunked scan of the histograms and evaluation of every threshold; (3) per-scaffold counts for the
// final assignment.
#include "abw_internal.h"
#include "abw_sep_common.h"
#include <string.h>
#include <algorithm>
#include <thread>
#include <limits.h>

namespace {



// Narrow dimensions (keys in [0, NARROW_T): every signature column), in groups of up to 8 (NGroup, abw_sep_common.h).
// ---- the static tables of the groups, built once per loaded table ------------------------------------------------------
// flip value of every scaffold in every narrow dimension, into its group's [scaffold][8] table (nslot = group * 8 + lane).
// grid (ceil(S / 256), narrow dimensions in nsort order)
__global__ __launch_bounds__(256) void sep_static_ts_kernel(SepDev P, const int32_t* __restrict__ nsort, const int32_t* __restrict__ nslot,
                                                           int32_t S, uint16_t* __restrict__ ts8)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    const int d = nsort[blockIdx.y], slot = nslot[blockIdx.y];
    const int32_t* __restrict__ col = P.bins + (int64_t)d * P.N;
    const int f = P.scaf_first[s], n = P.scaf_ndps[s], k = n >> 1;      // 0-based index of the flip value
    int ts, gt = 0;                                      // gt: its dps above the flip value
    if (n <= 8) {
        int x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = (i < n) ? col[f + i] : INT_MAX;
        ABW_SORT8(x);
        ts = (k == 1) ? x[1] : (k == 2) ? x[2] : (k == 3) ? x[3] : x[4];
#pragma unroll
        for (int i = 0; i < 8; i++) gt += (i < n && x[i] > ts);
    } else {
        ts = select_kth_mem(col + f, n, k, P.T[d]);
        for (int i = 0; i < n; i++) gt += col[f + i] > ts;
    }
    ts8[((int64_t)(slot >> 3) * S + s) * 8 + (slot & 7)] = (uint16_t)(ts | (min(gt, 63) << 10));
}

// the re-coded narrow columns.  grid (ceil(N / 256), groups)
__global__ __launch_bounds__(256) void sep_static_codes_kernel(SepDev P, const NGroup* __restrict__ groups, const int32_t* __restrict__ dp2scaf,
                                                              int32_t S, const uint4* __restrict__ ts8, uint4* __restrict__ nk16)
{
    const int64_t dp = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (dp >= P.N) return;
    const int j = blockIdx.y;
    const NGroup G = groups[j];
    const uint4 t = ts8[(int64_t)j * S + dp2scaf[dp]];
    const uint32_t tw[4] = {t.x, t.y, t.z, t.w};
    uint32_t cw[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 8; q++) {
        if (q < G.cnt) {
            const uint32_t v = (uint32_t)P.bins[(int64_t)G.dim[q] * P.N + dp];
            const uint32_t ts = (tw[q >> 1] >> (16 * (q & 1))) & 0x3FFu;
            cw[q >> 1] |= (v | (v > ts ? 0x8000u : 0u)) << (16 * (q & 1));
        }
    }
    nk16[(int64_t)j * P.N + dp] = make_uint4(cw[0], cw[1], cw[2], cw[3]);
}

__global__ void sep_static_srec_kernel(SepDev P, int32_t S, int4* __restrict__ srec)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const unsigned long long len = (unsigned long long)P.scaf_len[s];
    srec[s] = make_int4(P.scaf_first[s], P.scaf_ndps[s], (int)(uint32_t)len, (int)(uint32_t)(len >> 32));
}

// ---- the node's histograms of the narrow dimensions.  grid (ceil(m / TILE_A), groups) ----------------------------------
template <bool PACKED>
__global__ __launch_bounds__(256) void sep_accum_kernel(SepDev P, const NGroup* __restrict__ groups, const uint4* __restrict__ nk16,
                                                       const uint4* __restrict__ ts8, const int4* __restrict__ srec, int32_t S,
                                                       const int32_t* __restrict__ node_scaf, int32_t m)
{
    const int j = blockIdx.y;
    const NGroup G = groups[j];
    const int begin = blockIdx.x * TILE_A;
    accum_group_tile<PACKED>(G, nk16 + (int64_t)j * P.N, ts8 + (int64_t)j * S, srec, node_scaf, begin, min(m, begin + TILE_A), P, nullptr, nullptr,
                     [&](int b, unsigned long long a, uint32_t c3, unsigned long long e) {
                         int dim = G.dim[0], lo = 0;                    // the bin's dimension: selects, no indexed copy of G
#pragma unroll
                         for (int i = 1; i < 8; i++) if (i < G.cnt && b >= G.loff[i]) { dim = G.dim[i]; lo = G.loff[i]; }
                         const int64_t at = P.Toff[dim] + (b - lo);
                         if (a) atomicAdd(&P.h_st[at], a);
                         if (c3) atomicAdd(&P.h_tot[at], c3);
                         if (e) atomicAdd(&P.h_bp[at], e);
                     });
}

// ---- the node's SCG lists: per dimension and SCG the smallest / largest flip value over the node's scaffolds that carry it
// (what ClusterQuality::is_split_better's SCGdb::get_scg2count calls come to, ref src/ClusterQuality.cpp:96-136, SCGdb.cpp:21-38).
// The flip values are static, only a few scaffolds carry SCGs, and every (dimension, SCG) is the target of one update per
// carrier: the carriers are kept by SCG as well, and one wave per (dimension, SCG) takes the minimum / maximum over them and
// writes the pair once -- no atomics (as global atomics of the accumulation kernels they were a third of the root node's time).
// One wave per (dimension, SCG) walks the SCG's carriers (member = NULL: every scaffold is in the node).  wext = NULL: narrow
// dimensions only (the wide ones' own kernels fill their lists).  grid (ceil(D x n_scg / 4))
__global__ __launch_bounds__(256) void sep_scg_kernel(SepDev P, int32_t D, const int32_t* __restrict__ dim_slot, const uint16_t* __restrict__ ts8,
                                                     const uint4* __restrict__ wext, int32_t S, const int32_t* __restrict__ g2c_off,
                                                     const int32_t* __restrict__ g2c, const uint8_t* __restrict__ member)
{
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pair >= D * P.n_scg) return;
    const int d = pair / P.n_scg, g = pair % P.n_scg;
    const int slot = dim_slot[d];
    if (slot < 0 && !wext) return;
    int lo = INT_MAX, hi = -1;
    for (int i = g2c_off[g] + lane; i < g2c_off[g + 1]; i += 64) {
        const int s = g2c[i];
        if (member && !member[s]) continue;
        const int ts = slot >= 0 ? (int)(ts8[((int64_t)(slot >> 3) * S + s) * 8 + (slot & 7)] & 0x3FFu) : (int)wext[(int64_t)(-slot - 1) * S + s].w;
        lo = min(lo, ts); hi = max(hi, ts);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
    if (lane == 0) { P.scg_lo[(int64_t)d * P.n_scg + g] = lo; P.scg_hi[(int64_t)d * P.n_scg + g] = hi; }
}

__global__ void sep_member_kernel(const int32_t* __restrict__ list, int32_t m, uint8_t* __restrict__ member)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) member[list[i]] = 1;
}

// ---- wide dimensions (coverage: about as many distinct keys as dps) -----------------------------------
// Scattering N dps into ~N bins costs one device-scope atomic per dp (measured r01: ~23 G atomics/s, 0.52 ms
// for 10 columns of 0.8 M dps, twice the time of the 179 narrow columns).  Instead the table keeps, per wide
// dimension, the dps SORTED by bin (perm / sbin, built once at load): walking the node in that order makes
// every histogram update land on the walker's own bin, so a wave reduces runs of equal bins in registers and
// writes each bin once with a plain store -- no atomics except for runs cut by a wave boundary.
//   pass A (sep_wide_ts):   per (scaffold, dimension) the flip value t_s and the number of dps below it; SCG lo/hi.
//   pass B (sep_wide_hist): per sorted position: size += 1, tp1 += [v >= t_s]; the scaffold's "owner" also carries
//                           what the scaffold puts on bin t_s: tp1 credits of the dps below it, tot += n_s,
//                           bp += len_s.  Owner = first dp of the scaffold that sits on t_s; the dps of a scaffold
//                           are consecutive dp ids, so those on one bin are ADJACENT in (bin, dp) order and the
//                           owner is the position whose predecessor is another scaffold or another bin.
// The only random access left is the 4-byte gather of t_s by scaffold (S x 4 B per dimension: L2-resident).
struct WideDev {
    const int32_t* sscaf;     // [Dw][N] scaffold of the dp at each sorted position (ascending (bin, dp))
    const int32_t* sbin;      // [Dw][N] its bin
    int32_t* ts;              // [Dw][S] flip value of scaffold s in the current node, -1 = not in the node
    int32_t* below;           // [Dw][S] dps of s below t_s
    int32_t S;
};

__global__ __launch_bounds__(256) void sep_wide_ts_kernel(SepDev P, WideDev W, const int32_t* __restrict__ dimlist,
                                                         const int32_t* __restrict__ node_scaf, int32_t m)
{
    const int ls = blockIdx.x * 256 + threadIdx.x;
    if (ls >= m) return;
    const int w = blockIdx.y;
    const int d = dimlist[w];
    const int T = P.T[d];
    const int32_t* __restrict__ col = P.bins + (int64_t)d * P.N;
    const int s = node_scaf ? node_scaf[ls] : ls;
    const int f = P.scaf_first[s];
    const int n = P.scaf_ndps[s];
    const int k = n >> 1;
    int ts, below = 0;
    if (n <= 8) {
        int v[8], x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { v[i] = (i < n) ? col[f + i] : INT_MAX; x[i] = v[i]; }
#define ABW_CX(a, b) { const int lo_ = min(x[a], x[b]), hi_ = max(x[a], x[b]); x[a] = lo_; x[b] = hi_; }
        ABW_CX(0, 1) ABW_CX(2, 3) ABW_CX(4, 5) ABW_CX(6, 7)
        ABW_CX(0, 2) ABW_CX(1, 3) ABW_CX(4, 6) ABW_CX(5, 7)
        ABW_CX(1, 2) ABW_CX(5, 6)
        ABW_CX(0, 4) ABW_CX(1, 5) ABW_CX(2, 6) ABW_CX(3, 7)
        ABW_CX(2, 4) ABW_CX(3, 5)
        ABW_CX(1, 2) ABW_CX(3, 4) ABW_CX(5, 6)
#undef ABW_CX
        ts = (k == 1) ? x[1] : (k == 2) ? x[2] : (k == 3) ? x[3] : x[4];
#pragma unroll
        for (int i = 0; i < 8; i++) below += v[i] < ts;
    } else {
        ts = select_kth_mem(col + f, n, k, T);
        for (int i = 0; i < n; i++) below += col[f + i] < ts;
    }
    W.ts[(int64_t)w * W.S + s] = ts;
    W.below[(int64_t)w * W.S + s] = below;
    if (P.n_scg > 0) {
        for (int e = P.scg_off[s]; e < P.scg_off[s + 1]; e++) {
            const int g = P.scg_ids[e];
            atomicMin(&P.scg_lo[(int64_t)d * P.n_scg + g], ts);
            atomicMax(&P.scg_hi[(int64_t)d * P.n_scg + g], ts);
        }
    }
}

// Small nodes: the sorted walk costs the whole table whatever the node holds, so a node with few dps scatters them
// instead -- one thread per (scaffold, wide dimension), one device-scope atomic per dp (size and tp1 share a 64-bit
// word) plus what lands on bin t_s.  Cost proportional to the node (~23 G atomics/s).
__global__ __launch_bounds__(256) void sep_wide_scatter_kernel(SepDev P, const int32_t* __restrict__ dimlist,
                                                              const int32_t* __restrict__ node_scaf, int32_t m)
{
    const int ls = blockIdx.x * 256 + threadIdx.x;
    const int d = dimlist[blockIdx.y];
    int vmin = INT_MAX, vmax = -1;
    if (ls < m) {
        const int T = P.T[d];
        const int64_t toff = P.Toff[d];
        const int32_t* __restrict__ col = P.bins + (int64_t)d * P.N;
        const int s = node_scaf ? node_scaf[ls] : ls;
        const int f = P.scaf_first[s];
        const int n = P.scaf_ndps[s];
        constexpr unsigned long long BOTH = 1ull | (1ull << 32);
        unsigned long long at_ts = 0;                          // everything that lands on bin t_s, in one add
        int ts;
        if (n <= 8) {                                          // values in registers, flip value by the sorting network
            int v[8], x[8];
#pragma unroll
            for (int i = 0; i < 8; i++) { v[i] = (i < n) ? col[f + i] : INT_MAX; x[i] = v[i]; }
#define ABW_CX(a, b) { const int lo_ = min(x[a], x[b]), hi_ = max(x[a], x[b]); x[a] = lo_; x[b] = hi_; }
            ABW_CX(0, 1) ABW_CX(2, 3) ABW_CX(4, 5) ABW_CX(6, 7)
            ABW_CX(0, 2) ABW_CX(1, 3) ABW_CX(4, 6) ABW_CX(5, 7)
            ABW_CX(1, 2) ABW_CX(5, 6)
            ABW_CX(0, 4) ABW_CX(1, 5) ABW_CX(2, 6) ABW_CX(3, 7)
            ABW_CX(2, 4) ABW_CX(3, 5)
            ABW_CX(1, 2) ABW_CX(3, 4) ABW_CX(5, 6)
#undef ABW_CX
            const int k = n >> 1;
            ts = (k == 1) ? x[1] : (k == 2) ? x[2] : (k == 3) ? x[3] : x[4];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (i < n) {
                    vmin = min(vmin, v[i]); vmax = max(vmax, v[i]);
                    if (v[i] > ts) atomicAdd(&P.h_st[toff + v[i]], BOTH);
                    else if (v[i] < ts) { atomicAdd(&P.h_st[toff + v[i]], 1ull); at_ts += 1ull << 32; }
                    else at_ts += BOTH;
                }
            }
        } else {
            ts = select_kth_mem(col + f, n, n >> 1, T);
            for (int i = 0; i < n; i++) {
                const int v = col[f + i];
                vmin = min(vmin, v); vmax = max(vmax, v);
                if (v > ts) atomicAdd(&P.h_st[toff + v], BOTH);
                else if (v < ts) { atomicAdd(&P.h_st[toff + v], 1ull); at_ts += 1ull << 32; }
                else at_ts += BOTH;
            }
        }
        atomicAdd(&P.h_st[toff + ts], at_ts);
        atomicAdd(&P.h_tot[toff + ts], (uint32_t)n);
        atomicAdd(&P.h_bp[toff + ts], (unsigned long long)P.scaf_len[s]);
        if (P.n_scg > 0) {
            for (int e = P.scg_off[s]; e < P.scg_off[s + 1]; e++) {
                const int g = P.scg_ids[e];
                atomicMin(&P.scg_lo[(int64_t)d * P.n_scg + g], ts);
                atomicMax(&P.scg_hi[(int64_t)d * P.n_scg + g], ts);
            }
        }
    }
    // the bins this node touches in dimension d: only they are evaluated and cleared afterwards
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { vmin = min(vmin, __shfl_xor(vmin, o, 64)); vmax = max(vmax, __shfl_xor(vmax, o, 64)); }
    if ((threadIdx.x & 63) == 0 && vmax >= 0) { atomicMin(&P.range[2 * d], vmin); atomicMax(&P.range[2 * d + 1], vmax); }
}

// Between nodes the histograms are all zero: a node clears what the node before it touched (its recorded ranges)
// instead of the whole block, then the ranges start over -- full for the dimensions in `full` mode (narrow dimensions,
// the sorted walk, sharded tables: every bin may receive something), empty for dimensions that scatter.
__global__ __launch_bounds__(256) void sep_clear_kernel(SepDev P, const ChunkDesc* __restrict__ chunks)
{
    const ChunkDesc c = chunks[blockIdx.x];
    const int lo = P.range[2 * c.dim], hi = P.range[2 * c.dim + 1];
    if (hi < lo || c.start + c.len - 1 < lo || c.start > hi) return;
    const int64_t base = P.Toff[c.dim] + c.start;
    for (int t = threadIdx.x; t < c.len; t += 256) { P.h_st[base + t] = 0; P.h_tot[base + t] = 0; P.h_bp[base + t] = 0; }
}

__global__ void sep_range_reset_kernel(SepDev P, int32_t D, const int32_t* __restrict__ widelist, int32_t n_wide, int wide_full)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    // the SCG lo / hi lists of the node start empty (lo = INT_MAX, hi = -1)
    for (int64_t i = d; i < (int64_t)D * P.n_scg; i += (int64_t)gridDim.x * blockDim.x) { P.scg_lo[i] = INT_MAX; P.scg_hi[i] = -1; }
    if (d >= D) return;
    bool wide = false;
    for (int w = 0; w < n_wide; w++) wide |= widelist[w] == d;
    const bool full = !wide || wide_full;
    P.range[2 * d] = full ? 0 : INT_MAX;
    P.range[2 * d + 1] = full ? P.T[d] - 1 : -1;
}

constexpr int WIDE_TILES = 4;        // 64-position tiles per wave: their loads are issued together (the kernel is
                                     // a chain of dependent loads; one tile per wave left it latency-bound)

__global__ __launch_bounds__(256) void sep_wide_hist_kernel(SepDev P, WideDev W, const int32_t* __restrict__ dimlist)
{
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.y;
    const int d = dimlist[w];
    const int64_t N = P.N;
    const int64_t wave0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 * WIDE_TILES);
    const int32_t* __restrict__ sbin = W.sbin + (int64_t)w * N;
    const int32_t* __restrict__ sscaf = W.sscaf + (int64_t)w * N;
    const int32_t* __restrict__ tsd = W.ts + (int64_t)w * W.S;
    const int32_t* __restrict__ belowd = W.below + (int64_t)w * W.S;
    if (wave0 >= N) return;
    int v[WIDE_TILES], sc[WIDE_TILES], vprev[WIDE_TILES], scprev[WIDE_TILES], vnext[WIDE_TILES], t[WIDE_TILES];
    // predecessor / successor in sorted order: neighbours inside the wave's span come by shuffle, the two
    // positions outside it are read by lane 0 / lane 63
#pragma unroll
    for (int k = 0; k < WIDE_TILES; k++) {
        const int64_t i = wave0 + 64 * k + lane;
        v[k] = -1; sc[k] = -1;
        if (i < N) { v[k] = sbin[i]; sc[k] = sscaf[i]; }
    }
    int v_before = -2, sc_before = -2, v_after = -2;
    if (lane == 0 && wave0 > 0) { v_before = sbin[wave0 - 1]; sc_before = sscaf[wave0 - 1]; }
    if (lane == 63 && wave0 + 64 * WIDE_TILES < N) v_after = sbin[wave0 + 64 * WIDE_TILES];
#pragma unroll
    for (int k = 0; k < WIDE_TILES; k++) t[k] = (sc[k] >= 0) ? tsd[sc[k]] : -1;
#pragma unroll
    for (int k = 0; k < WIDE_TILES; k++) {
        const int up_v = __shfl_up(v[k], 1, 64), up_s = __shfl_up(sc[k], 1, 64);
        const int last_v = (k > 0) ? __shfl(v[k > 0 ? k - 1 : 0], 63, 64) : v_before;      // lane 0's predecessor
        const int last_s = (k > 0) ? __shfl(sc[k > 0 ? k - 1 : 0], 63, 64) : sc_before;
        vprev[k] = lane == 0 ? last_v : up_v;
        scprev[k] = lane == 0 ? last_s : up_s;
        const int dn_v = __shfl_down(v[k], 1, 64);
        const int first_v = (k + 1 < WIDE_TILES) ? __shfl(v[k + 1 < WIDE_TILES ? k + 1 : k], 0, 64) : v_after;
        vnext[k] = lane == 63 ? first_v : dn_v;
    }
    unsigned long long st[WIDE_TILES], bp[WIDE_TILES];
    uint32_t tot[WIDE_TILES];
#pragma unroll
    for (int k = 0; k < WIDE_TILES; k++) {
        st[k] = 0; bp[k] = 0; tot[k] = 0;
        if (t[k] >= 0) {
            st[k] = 1ull | ((unsigned long long)(v[k] >= t[k]) << 32);
            if (v[k] == t[k] && (sc[k] != scprev[k] || v[k] != vprev[k])) {          // the scaffold's owner
                st[k] += (unsigned long long)(uint32_t)belowd[sc[k]] << 32;
                tot[k] = (uint32_t)P.scaf_ndps[sc[k]]; bp[k] = (unsigned long long)P.scaf_len[sc[k]];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < WIDE_TILES; k++) {
        // runs of equal bins inside the tile
        const bool head = lane == 0 || v[k] != vprev[k];
        const bool tail = lane == 63 || v[k] != vnext[k];
        bool f = head;
        unsigned long long a = st[k], c = bp[k];
        uint32_t b3 = tot[k];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long ya = __shfl_up(a, o, 64), yc = __shfl_up(c, o, 64);
            const uint32_t yb = __shfl_up(b3, o, 64);
            const int yf = __shfl_up((int)f, o, 64);
            if (lane >= o && !f) { a += ya; c += yc; b3 += yb; f = yf != 0; }
        }
        const unsigned long long heads = __ballot(head);
        const int headlane = 63 - __clzll((long long)(heads & (~0ull >> (63 - lane))));
        // does the tile's first / last run go on in the neighbouring tile?
        const int first_open = __shfl((int)(v[k] == vprev[k]), 0, 64);
        const bool open_tail = (lane == 63) && (v[k] == vnext[k]);
        if (v[k] >= 0 && tail) {
            const int64_t bin = P.Toff[d] + v[k];
            if ((headlane > 0 || !first_open) && !open_tail) {
                // write-through stores (system scope).  Measured r01: with ordinary stores the dirty histogram
                // lines push the t_s table out of L2 and the gathers above go to HBM: 0.246 ms vs 0.140 ms
                // (no stores at all: 0.089 ms; "nt" stores: 0.276 ms)
                __hip_atomic_store(&P.h_st[bin], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&P.h_tot[bin], b3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&P.h_bp[bin], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                if (a) atomicAdd(&P.h_st[bin], a);
                if (b3) atomicAdd(&P.h_tot[bin], b3);
                if (c) atomicAdd(&P.h_bp[bin], c);
            }
        }
    }
}

// ---- chunked scan + evaluation ----------------------------------------------------------------
__device__ __forceinline__ Sums wave_incl_scan(Sums x, int lane)
{
    (void)lane;
    return Sums{wave_prefix_u64(x.size), wave_prefix_u64(x.tp1), wave_prefix_u64(x.tot), wave_prefix_u64(x.bp)};
}

// phase A: per-chunk totals
__global__ __launch_bounds__(256) void sep_chunk_sums_kernel(SepDev P, const ChunkDesc* __restrict__ chunks, Sums* __restrict__ chunk_sums)
{
    __shared__ Sums wsum[4];
    const ChunkDesc c = chunks[blockIdx.x];
    if (c.start + c.len - 1 < P.range[2 * c.dim] || c.start > P.range[2 * c.dim + 1]) {      // untouched by this node: all zero
        if (threadIdx.x == 0) chunk_sums[blockIdx.x] = Sums{0, 0, 0, 0};
        return;
    }
    const int64_t base = P.Toff[c.dim] + c.start;
    Sums acc = {0, 0, 0, 0};
    for (int t = threadIdx.x; t < c.len; t += 256) {
        const unsigned long long st = P.h_st[base + t];
        acc.size += (uint32_t)st; acc.tp1 += st >> 32; acc.tot += P.h_tot[base + t]; acc.bp += P.h_bp[base + t];
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    acc = wave_incl_scan(acc, lane);
    if (lane == 63) wsum[wid] = acc;
    __syncthreads();
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = sums_add(sums_add(wsum[0], wsum[1]), sums_add(wsum[2], wsum[3]));
}

// phase B: exclusive scan of the chunk totals within each dimension (chunks of a dimension are consecutive);
// one wave per dimension, 64 chunks at a time
__global__ __launch_bounds__(64) void sep_chunk_scan_kernel(const int32_t* __restrict__ dim_first_chunk, const int32_t* __restrict__ dim_nchunks,
                                                           int32_t D, Sums* __restrict__ chunk_sums, Sums* __restrict__ dim_total)
{
    const int d = blockIdx.x;
    const int lane = threadIdx.x;
    if (d >= D) return;
    const int f = dim_first_chunk[d], n = dim_nchunks[d];
    Sums run = {0, 0, 0, 0};
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        Sums v = {0, 0, 0, 0};
        if (i < n) v = chunk_sums[f + i];
        const Sums x = wave_incl_scan(v, lane);
        if (i < n) {
            Sums e = sums_add(run, x);
            e.size -= v.size; e.tp1 -= v.tp1; e.tot -= v.tot; e.bp -= v.bp;        // exclusive
            chunk_sums[f + i] = e;
        }
        Sums last;
        last.size = __shfl(x.size, 63, 64); last.tp1 = __shfl(x.tp1, 63, 64); last.tot = __shfl(x.tot, 63, 64); last.bp = __shfl(x.bp, 63, 64);
        run = sums_add(run, last);
    }
    if (lane == 0) dim_total[d] = run;
}

// phase C: rescan each chunk with its carry-in, evaluate every threshold, keep the chunk's best
__global__ __launch_bounds__(256) void sep_eval_kernel(SepDev P, const ChunkDesc* __restrict__ chunks,
                                                      const Sums* __restrict__ chunk_carry, const Sums* __restrict__ dim_total,
                                                      int32_t min_ndps, Cand* __restrict__ chunk_best)
{
    __shared__ Sums wsum[4];
    __shared__ Sums carry_s;
    __shared__ int32_t s_lo[MAX_SCG], s_hi[MAX_SCG];
    __shared__ int32_t s_ng;
    __shared__ Cand wbest[4];
    const ChunkDesc c = chunks[blockIdx.x];
    const int d = c.dim;
    if (c.start + c.len - 1 < P.range[2 * d] || c.start > P.range[2 * d + 1]) {             // no value of the node here: no threshold
        if (threadIdx.x == 0) chunk_best[blockIdx.x] = Cand{0.0, 0, 0, 0.0, 0.0, 0};
        return;
    }
    const int64_t base = P.Toff[d] + c.start;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const Sums total = dim_total[d];
    const long long n = (long long)total.size;
    if (threadIdx.x == 0) { carry_s = chunk_carry[blockIdx.x]; s_ng = 0; }
    __syncthreads();
    // SCGs present in the node: lo != INT_MAX.  compact them.
    for (int g = threadIdx.x; g < P.n_scg; g += 256) {
        const int lo = P.scg_lo[(int64_t)d * P.n_scg + g];
        if (lo != INT_MAX) {
            const int slot = atomicAdd(&s_ng, 1);
            s_lo[slot] = lo; s_hi[slot] = P.scg_hi[(int64_t)d * P.n_scg + g];
        }
    }
    __syncthreads();
    const int ng = s_ng;
    Cand best = {0.0, 0, 0, 0.0, 0.0, 0};
    for (int t0 = 0; t0 < c.len; t0 += 256) {
        const int t = t0 + threadIdx.x;
        Sums v = {0, 0, 0, 0};
        uint32_t here = 0;
        if (t < c.len) {
            const unsigned long long st = P.h_st[base + t];
            here = (uint32_t)st;
            v.size = here; v.tp1 = st >> 32; v.tot = P.h_tot[base + t]; v.bp = P.h_bp[base + t];
        }
        Sums x = wave_incl_scan(v, lane);
        if (lane == 63) wsum[wid] = x;
        __syncthreads();
        Sums off = carry_s;
        for (int k = 0; k < wid; k++) off = sums_add(off, wsum[k]);
        x = sums_add(x, off);
        __syncthreads();
        if (threadIdx.x == 255) carry_s = x;
        if (here) {                                                    // only values present in the node are thresholds
            const long long size1 = (long long)x.size, tp1 = (long long)x.tp1, tot1 = (long long)x.tot;
            const long long size2 = n - size1, tot2 = n - tot1;
            const long long tp2 = n - (size1 + tot1 - tp1);
            if (size1 >= min_ndps && size2 >= min_ndps) {              // ref ...Specificity.cpp:151
                const bool small1 = size1 < size2;                     // ref :155 (tie -> cluster2)
                const long long TP = small1 ? tp1 : tp2, TOT = small1 ? tot1 : tot2, SZ = small1 ? size1 : size2;
                const double sens = (double)(int)TP / (double)(int)TOT;   // ref .h:75 (int fields)
                const double spec = (double)(int)TP / (double)(int)SZ;    // ref .h:74
                const double prod = sens * spec;
                if (prod == prod) {                                    // NaN is never better (ref .h:28)
                    Cand cnd = {prod, c.start + t, 1, sens, spec, size1};
                    if (cand_better(cnd, best)) {
                        // gate, ref ClusterQuality.cpp:96-136
                        const int tt = c.start + t;
                        int cl = 0, chh = 0;
                        for (int g = 0; g < ng; g++) { cl += s_lo[g] <= tt; chh += s_hi[g] <= tt; }
                        const int G1 = cl, G2 = ng - chh, both = cl - chh;
                        bool pass;
                        if (G1 == 0 || G2 == 0) pass = (x.bp >= 500000ull) && (total.bp - x.bp >= 500000ull);
                        else pass = !((double)both / (double)(G1 < G2 ? G1 : G2) < 0.2);
                        if (pass) best = cnd;
                    }
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { Cand other = cand_shfl_xor(best, o); if (cand_better(other, best)) best = other; }
    if (lane == 0) wbest[wid] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        Cand b = wbest[0];
        for (int k = 1; k < 4; k++) if (cand_better(wbest[k], b)) b = wbest[k];
        chunk_best[blockIdx.x] = b;
    }
}

// phase D: best chunk of each dimension; one wave per dimension
__global__ __launch_bounds__(64) void sep_dimbest_kernel(const int32_t* __restrict__ dim_first_chunk, const int32_t* __restrict__ dim_nchunks,
                                                        int32_t D, const Cand* __restrict__ chunk_best, const Sums* __restrict__ dim_total,
                                                        DimBestDev* __restrict__ out)
{
    const int d = blockIdx.x;
    const int lane = threadIdx.x;
    if (d >= D) return;
    Cand b = {0.0, 0, 0, 0.0, 0.0, 0};
    const int f = dim_first_chunk[d], n = dim_nchunks[d];
    for (int i = lane; i < n; i += 64) if (cand_better(chunk_best[f + i], b)) b = chunk_best[f + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { Cand other = cand_shfl_xor(b, o); if (cand_better(other, b)) b = other; }
    if (lane == 0) {
        out[d].found = b.found; out[d].bin = b.bin; out[d].sens = b.sens; out[d].spec = b.spec; out[d].prod = b.prod;
        out[d].size1 = b.size1; out[d].n = (long long)dim_total[d].size;
    }
}

// The global winner over the dimensions: ClusteringResult::is_better (ref ...Specificity.h:25-30 +
// ClusterSeparator.cpp:11-16) applied in dimension order from the reset best_separation (dimension = (size_t)-1,
// value = -1, sens = spec = 0): a larger product wins, equal products keep the smaller dimension -- and a
// dimension whose winner has product 0 still replaces the reset state.  Then is_legal (ref .h:31-35).
// In dimension order that rule picks, among the dimensions that found a winner, the largest product and on equal
// products the smallest dimension: a strict order, so one wave reduces it (lanes stride over the dimensions).
__global__ __launch_bounds__(64) void sep_global_kernel(const DimBestDev* __restrict__ db, int32_t D, GlobalRes* __restrict__ out,
                                                       unsigned long long* __restrict__ totals)
{
    const int lane = threadIdx.x;
    if (lane < 2 && totals) totals[lane] = 0;             // sep_assign adds to them
    double bp = -1.0;                                      // products are >= 0: -1 = nothing found yet
    int bd = -1;
    for (int d = lane; d < D; d += 64) {
        if (!db[d].found) continue;
        const double a = db[d].sens * db[d].spec;
        if (a > bp) { bp = a; bd = d; }                    // d ascends within a lane: equal products keep the smaller d
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double op = __shfl_xor(bp, o, 64);
        const int od = __shfl_xor(bd, o, 64);
        if (od >= 0 && (bd < 0 || op > bp || (op == bp && od < bd))) { bp = op; bd = od; }
    }
    if (lane != 0) return;
    const int gd = bd;
    const double gsens = gd >= 0 ? db[gd].sens : 0.0, gspec = gd >= 0 ? db[gd].spec : 0.0;
    GlobalRes r;
    r.found = gd >= 0; r.dim = gd; r.bin = gd >= 0 ? db[gd].bin : -1;
    r.sens = gsens; r.spec = gspec;
    r.legal = (gd >= 0) && (gsens >= 0.8) && (gspec >= 0.8) && (gsens + gspec >= 1.6) && (gsens * gspec >= 0.8);
    r.size1 = gd >= 0 ? db[gd].size1 : 0; r.n = gd >= 0 ? db[gd].n : 0;
    *out = r;
}

// dst (+)= src over the three exchange buffers of the scaffold-sharded separator (abw_sep_shard_combine)
__global__ __launch_bounds__(256) void sep_combine_kernel(int64_t* __restrict__ h, const int64_t* __restrict__ sh, uint64_t nh, int32_t* __restrict__ lo,
                                                         const int32_t* __restrict__ slo, int32_t* __restrict__ hi, const int32_t* __restrict__ shi,
                                                         uint64_t ng)
{
    const uint64_t step = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nh; i += step) h[i] += sh[i];
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < ng; i += step) { lo[i] = min(lo[i], slo[i]); hi[i] = max(hi[i], shi[i]); }
}


// ref ClusterSeparator.cpp:35-53,87-122 on the device, one thread per scaffold of the node: the dps with bin <= t in
// the winning dimension are the raw "<=" side; cluster1 is the raw side with fewer dps (tie: the "<=" side); a scaffold
// goes to cluster1 iff at least half of its dps are there; raw flags are written as 1 / 2.
// n = dps of the whole node and le = those on the "<=" side (size1 of the winning threshold, known from
// the histogram scan; both are totals over ALL shards when the node is sharded).
// totals: [0], [1] = dps of this shard's scaffolds assigned to cluster1 / cluster2 (out; zeroed by sep_global_kernel)
__global__ __launch_bounds__(256) void sep_assign_kernel(SepDev P, const int32_t* __restrict__ node_scaf,
                                                        const int32_t* __restrict__ node_dpoff, int32_t m, const GlobalRes* __restrict__ win,
                                                        uint8_t* __restrict__ raw, uint8_t* __restrict__ side,
                                                        unsigned long long* __restrict__ totals)
{
    __shared__ unsigned long long b1, b2;
    if (!win->legal) return;                              // uniform: no legal winner, nothing to reconstruct (ref :27-28)
    const long long n = win->n, le = win->size1;
    const int d = win->dim, tbin = win->bin;
    if (threadIdx.x == 0) { b1 = 0; b2 = 0; }
    __syncthreads();
    const bool swap = n - le < le;
    const int ls = blockIdx.x * 256 + threadIdx.x;
    if (ls < m) {
        const int s = node_scaf ? node_scaf[ls] : ls;
        const int f = P.scaf_first[s], tot = P.scaf_ndps[s];
        const int64_t o = node_dpoff ? node_dpoff[ls] : f;
        const int32_t* col = P.bins + (int64_t)d * P.N;
        int c = 0;
        for (int i = 0; i < tot; i++) {
            const bool le_i = col[f + i] <= tbin;
            c += le_i;
            raw[o + i] = (le_i != swap) ? 1 : 2;          // ref :87-90
        }
        const int c1 = swap ? tot - c : c;
        const bool to1 = c1 > 0 && 2 * (long long)c1 >= tot;
        side[ls] = to1 ? 1 : 2;
        atomicAdd(to1 ? &b1 : &b2, (unsigned long long)tot);
    }
    __syncthreads();
    if (threadIdx.x == 0) { if (b1) atomicAdd(&totals[0], b1); if (b2) atomicAdd(&totals[1], b2); }
}

// abw_sep_node_assign: the node's dps on the "<=" side of a given threshold (what the histogram scan knows as size1 of the
// winner), added into win->size1.  One thread per scaffold of the node.
__global__ __launch_bounds__(256) void sep_count_le_kernel(SepDev P, const int32_t* __restrict__ node_scaf, int32_t m, GlobalRes* __restrict__ win)
{
    __shared__ unsigned long long b;
    if (threadIdx.x == 0) b = 0;
    __syncthreads();
    const int ls = blockIdx.x * 256 + threadIdx.x;
    int c = 0;
    if (ls < m) {
        const int s = node_scaf ? node_scaf[ls] : ls;
        const int f = P.scaf_first[s], tot = P.scaf_ndps[s];
        const int32_t* col = P.bins + (int64_t)win->dim * P.N;
        const int tbin = win->bin;
        for (int i = 0; i < tot; i++) c += col[f + i] <= tbin;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&b, (unsigned long long)c);
    __syncthreads();
    if (threadIdx.x == 0 && b) atomicAdd(reinterpret_cast<unsigned long long*>(&win->size1), b);
}

}  // namespace

static abw_sep_extra* g_extra_of(abw_ctx* ctx);

// The context keeps the host-side extras in a side table keyed by the context pointer (keeps
// abw_internal.h free of kernel-only types).
#include <map>
#include <mutex>
static std::map<abw_ctx*, abw_sep_extra*> g_extras;
static std::mutex g_extras_mtx;
SepHost& abw_sep_host(abw_ctx* ctx) { return g_extra_of(ctx)->h; }

static abw_sep_extra* g_extra_of(abw_ctx* ctx)
{
    std::lock_guard<std::mutex> l(g_extras_mtx);
    auto it = g_extras.find(ctx);
    if (it != g_extras.end()) return it->second;
    abw_sep_extra* e = new abw_sep_extra();
    g_extras[ctx] = e;
    return e;
}

extern "C" void abw_sep_free(abw_ctx* ctx)
{
    abw_sep_state& s = ctx->sep;
    void* ptrs[] = {s.d_keys, s.d_bins, s.d_scaf_first, s.d_scaf_ndps, s.d_scaf_len, s.d_scg_off, s.d_scg_ids,
                    s.d_T, s.d_Toff, s.d_binval, s.d_binval_off};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    s.d_keys = s.d_bins = s.d_scaf_first = s.d_scaf_ndps = s.d_scg_off = s.d_scg_ids = s.d_T = s.d_binval = nullptr;
    s.d_scaf_len = s.d_Toff = s.d_binval_off = nullptr;
    abw_release(s.hist); abw_release(s.scg_lohi); abw_release(s.node_scaf);
    abw_release(s.dimbest);
    s.loaded = false;
    abw_sep_extra* e = nullptr;
    {
        std::lock_guard<std::mutex> l(g_extras_mtx);
        auto it = g_extras.find(ctx);
        if (it != g_extras.end()) { e = it->second; g_extras.erase(it); }
    }
    if (e) {
        void* q[] = {e->h.d_narrow, e->h.d_wide, e->h.d_dim_first_chunk, e->h.d_dim_nchunks, e->h.d_chunks,
                     e->h.d_chunk_sums, e->h.d_dim_total, e->h.d_chunk_best, e->h.d_sscaf, e->h.d_sbin, e->h.d_sdp,
                     e->h.d_wts, e->h.d_wbelow, e->h.d_range, e->h.d_groups, e->h.d_nsort, e->h.d_toffn, e->h.d_dim_slot, e->h.d_ts8, e->h.d_nk16, e->h.d_srec, e->h.d_wword, e->h.d_wext, e->h.d_wpack, e->h.d_posof, e->h.d_sinfo, e->h.d_carriers, e->h.d_nw, e->h.d_g2c_off, e->h.d_g2c, e->h.d_member};
        for (void* p : q) if (p) (void)hipFree(p);
        abw_tree_run_free(e->h.tree_run);
        abw_stree_run_free(e->h.stree_run);
        delete e;
    }
}

template <typename T>
static int upload(abw_ctx* ctx, T** dst, const T* src, size_t n)
{
    *dst = nullptr;
    if (n == 0) n = 1;
    ABW_HIP(ctx, hipMalloc((void**)dst, n * sizeof(T)));
    if (src) ABW_HIP(ctx, hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return ABW_OK;
}

static int sep_load_upload(abw_ctx* ctx, std::vector<int32_t>& st_first, std::vector<int32_t>& st_ndps, std::vector<int32_t>& st_T,
                           std::vector<int32_t>& st_span, std::vector<int64_t>& st_Toff, std::vector<int64_t>& st_binval_off, std::vector<std::vector<int32_t>>& uniq,
                           std::vector<int32_t>& bins, std::vector<char>& is_wide, const int32_t* keys, int32_t D, int64_t N,
                           const int32_t* dp2scaf, const int64_t* scaf_len, int32_t S, const int32_t* scg_off, const int32_t* scg_ids,
                           int32_t n_scg, bool sharded);

// ---- the column store built on the device -----------------------------------------------------------------------------
// What the host build below does with 16 threads (classify every column, sort the wide ones, rank their keys, order their
// dps by (bin, dp)) as kernels over the keys where they are: 1.9 s -> tens of milliseconds at 4 M datapoints x 199 columns.
namespace {

// smallest and largest key of every column: mm[2 d], mm[2 d + 1] (start: INT_MAX, INT_MIN); grid (any, D)
__global__ __launch_bounds__(256) void load_minmax_kernel(const int32_t* __restrict__ keys, int64_t N, int32_t* __restrict__ mm)
{
    const int32_t* __restrict__ col = keys + (int64_t)blockIdx.y * N;
    int32_t mn = INT_MAX, mx = INT_MIN;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) { const int32_t v = col[i]; mn = min(mn, v); mx = max(mx, v); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = min(mn, __shfl_xor(mn, o, 64)); mx = max(mx, __shfl_xor(mx, o, 64)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(&mm[2 * blockIdx.y], mn); atomicMax(&mm[2 * blockIdx.y + 1], mx); }
}

// a wide column as unsigned sort keys: key - smallest key
__global__ __launch_bounds__(256) void load_bias_kernel(const int32_t* __restrict__ col, int64_t N, int32_t mn, uint32_t* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N) out[i] = (uint32_t)col[i] - (uint32_t)mn;
}

__global__ __launch_bounds__(256) void load_heads_kernel(const uint32_t* __restrict__ ks, int64_t N, uint32_t* __restrict__ head)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N) head[i] = (i == 0 || ks[i] != ks[i - 1]) ? 1u : 0u;
}

// sorted position i of a wide column (keys ks ascending, equal keys in dp order; perm = the dp; before = distinct keys in
// front of it): its bin = rank of its key, the (bin, dp)-ordered tables, the dp's bin, the bin's key
__global__ __launch_bounds__(256) void load_rank_kernel(const uint32_t* __restrict__ ks, const uint32_t* __restrict__ perm, const uint64_t* __restrict__ before,
                                                       int64_t N, int32_t mn, const int32_t* __restrict__ dp2scaf, int32_t* __restrict__ sbin,
                                                       int32_t* __restrict__ sdp, int32_t* __restrict__ sscaf, int32_t* __restrict__ bins_col,
                                                       int32_t* __restrict__ binval)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const bool head = i == 0 || ks[i] != ks[i - 1];
    const int32_t rank = (int32_t)(before[i] + (head ? 1u : 0u)) - 1;
    const uint32_t dp = perm[i];
    sbin[i] = rank; sdp[i] = (int32_t)dp; sscaf[i] = dp2scaf[dp];
    bins_col[dp] = rank;
    if (head) binval[rank] = (int32_t)(ks[i] + (uint32_t)mn);
}

}  // namespace

// abw_sep_narrow_bins: bins the NEXT load gives a narrow column at least (tables that share a histogram layout)
static int32_t narrow_bins_floor(abw_ctx* ctx, int32_t d, int32_t D)
{
    return ((int32_t)ctx->sep_min_bins.size() == D) ? std::min(ctx->sep_min_bins[(size_t)d], (int32_t)NARROW_T) : 0;
}

// keys: [D][N] on the host or on the device.  Fills T / span / is_wide / uniq (the distinct keys of every wide column) and
// leaves keys, bins and the sorted tables of the wide columns on the device (ctx->sep.d_keys, d_bins; H.d_sscaf, d_sbin, d_sdp).
static int sep_device_build(abw_ctx* ctx, const int32_t* keys, bool keys_on_device, int32_t D, int64_t N, const int32_t* dp2scaf,
                            std::vector<int32_t>& h_T, std::vector<int32_t>& h_span, std::vector<char>& is_wide,
                            std::vector<std::vector<int32_t>>& uniq)
{
    abw_sep_state& s = ctx->sep;
    SepHost& H = g_extra_of(ctx)->h;
    hipStream_t st = ctx->stream;
    int rc;
    ABW_HIP(ctx, hipMalloc((void**)&s.d_keys, (size_t)D * N * 4));
    ABW_HIP(ctx, hipMemcpyAsync(s.d_keys, keys, (size_t)D * N * 4, keys_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    if ((rc = abw_reserve(ctx, ctx->scratch[6], (size_t)D * 8))) return rc;
    int32_t* d_mm = (int32_t*)ctx->scratch[6].p;
    std::vector<int32_t> mm((size_t)2 * D);
    for (int32_t d = 0; d < D; d++) { mm[2 * d] = INT_MAX; mm[2 * d + 1] = INT_MIN; }
    ABW_HIP(ctx, hipMemcpyAsync(d_mm, mm.data(), mm.size() * 4, hipMemcpyHostToDevice, st));
    const unsigned gx = (unsigned)std::min<int64_t>((N + 255) / 256, std::max(1, ctx->num_cu * 8 / std::max(1, D) + 1));
    hipLaunchKernelGGL(load_minmax_kernel, dim3(gx, (unsigned)D), dim3(256), 0, st, (const int32_t*)s.d_keys, N, d_mm);
    ABW_LAUNCH_CHECK(ctx);
    ABW_HIP(ctx, hipMemcpyAsync(mm.data(), d_mm, mm.size() * 4, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    size_t n_wide = 0;
    for (int32_t d = 0; d < D; d++) {
        const int32_t mn = mm[2 * d], mx = mm[2 * d + 1];
        if (mn >= 0 && mx < NARROW_T) { h_T[d] = std::max(mx + 1, narrow_bins_floor(ctx, d, D)); h_span[d] = mx + 1; }
        else { is_wide[d] = 1; n_wide++; }
    }
    if (n_wide == 0) return ABW_OK;
    ABW_HIP(ctx, hipMalloc((void**)&s.d_bins, (size_t)D * N * 4));
    ABW_HIP(ctx, hipMemcpyAsync(s.d_bins, s.d_keys, (size_t)D * N * 4, hipMemcpyDeviceToDevice, st));    // narrow columns: bin = key
    ABW_HIP(ctx, hipMalloc((void**)&H.d_sscaf, n_wide * (size_t)N * 4));
    ABW_HIP(ctx, hipMalloc((void**)&H.d_sbin, n_wide * (size_t)N * 4));
    ABW_HIP(ctx, hipMalloc((void**)&H.d_sdp, n_wide * (size_t)N * 4));
    auto up16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
    const size_t b_n = up16((size_t)N * 4);
    // scratch[0]: biased keys | heads | bin keys (u32[N] each) | dp2scaf i32[N];  scratch[1]: heads before u64[N + 1];  scratch[2]: the sort's work
    if ((rc = abw_reserve(ctx, ctx->scratch[0], 4 * b_n))) return rc;
    if ((rc = abw_reserve(ctx, ctx->scratch[1], ((size_t)N + 2) * 8))) return rc;
    if ((rc = abw_reserve(ctx, ctx->scratch[2], abw_sort_work_bytes((uint64_t)N)))) return rc;
    uint8_t* base = (uint8_t*)ctx->scratch[0].p;
    uint32_t* d_biased = (uint32_t*)base;
    uint32_t* d_head = (uint32_t*)(base + b_n);
    int32_t* d_binval = (int32_t*)(base + 2 * b_n);
    int32_t* d_dp2scaf = (int32_t*)(base + 3 * b_n);
    uint64_t* d_before = (uint64_t*)ctx->scratch[1].p;
    ABW_HIP(ctx, hipMemcpyAsync(d_dp2scaf, dp2scaf, (size_t)N * 4, hipMemcpyHostToDevice, st));
    const unsigned gn = (unsigned)((N + 255) / 256);
    size_t w = 0;
    for (int32_t d = 0; d < D; d++) {
        if (!is_wide[d]) continue;
        const int32_t mn = mm[2 * d], mx = mm[2 * d + 1];
        const uint64_t range = (uint64_t)((int64_t)mx - (int64_t)mn);
        int bits = 1;
        while (bits < 32 && (range >> bits) != 0) bits++;
        const int32_t* col = s.d_keys + (int64_t)d * N;
        hipLaunchKernelGGL(load_bias_kernel, dim3(gn), dim3(256), 0, st, col, N, mn, d_biased);
        ABW_LAUNCH_CHECK(ctx);
        const uint32_t *ks = nullptr, *perm = nullptr;
        if ((rc = abw_stable_sort_u32(ctx, d_biased, nullptr, (uint64_t)N, bits, (uint8_t*)ctx->scratch[2].p, &ks, &perm))) return rc;
        hipLaunchKernelGGL(load_heads_kernel, dim3(gn), dim3(256), 0, st, ks, N, d_head);
        ABW_LAUNCH_CHECK(ctx);
        if ((rc = abw_device_scan_u32(ctx, d_head, (uint64_t)N, d_before))) return rc;
        hipLaunchKernelGGL(load_rank_kernel, dim3(gn), dim3(256), 0, st, ks, perm, (const uint64_t*)d_before, N, mn, (const int32_t*)d_dp2scaf,
                           H.d_sbin + w * (size_t)N, H.d_sdp + w * (size_t)N, H.d_sscaf + w * (size_t)N, s.d_bins + (int64_t)d * N, d_binval);
        ABW_LAUNCH_CHECK(ctx);
        uint64_t T = 0;
        ABW_HIP(ctx, hipMemcpyAsync(&T, d_before + N, 8, hipMemcpyDeviceToHost, st));
        ABW_HIP(ctx, hipStreamSynchronize(st));
        if (T == 0 || T > (uint64_t)N) return abw_fail(ctx, ABW_ERR_HIP, "abw_sep_load: %llu distinct keys in dimension %d of %lld datapoints", (unsigned long long)T, d, (long long)N);
        h_T[d] = (int32_t)T;
        uniq[d].resize((size_t)T);
        ABW_HIP(ctx, hipMemcpy(uniq[d].data(), d_binval, (size_t)T * 4, hipMemcpyDeviceToHost));
        w++;
    }
    return ABW_OK;
}

// dict_off == NULL: the table is the whole problem, the bin layout follows from it (narrow dimension:
// T = max key + 1; wide: the column's own distinct keys).  dict_off != NULL: the table is ONE SHARD of a
// scaffold-sharded problem and every shard must lay its histograms out identically: a dimension with an
// empty dictionary is narrow with T = NARROW_T, one with a dictionary (the sorted distinct keys of the
// column over ALL shards) is wide with T = its length.
static int sep_load_impl(abw_ctx* ctx, const char* who, const int32_t* keys, int32_t D, int64_t N, const int32_t* dp2scaf,
                         const int64_t* scaf_len, int32_t S, const int32_t* scg_off, const int32_t* scg_ids,
                         int32_t n_scg, const int32_t* dict_vals, const int64_t* dict_off, bool keys_on_device = false)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!keys || !dp2scaf || !scaf_len || D <= 0 || N <= 0 || S <= 0 || N > INT_MAX - 16)
        return abw_fail(ctx, ABW_ERR_ARG, "%s: bad argument", who);
    if (n_scg < 0 || n_scg > MAX_SCG) return abw_fail(ctx, ABW_ERR_ARG, "%s: n_scg must be in [0,%d]", who, MAX_SCG);
    if (n_scg > 0 && (!scg_off || !scg_ids)) return abw_fail(ctx, ABW_ERR_ARG, "%s: null SCG table", who);
    if (dict_off) {
        if (dict_off[0] != 0) return abw_fail(ctx, ABW_ERR_ARG, "%s: dict_off[0] != 0", who);
        for (int32_t d = 0; d < D; d++) {
            if (dict_off[d + 1] < dict_off[d] || dict_off[d + 1] - dict_off[d] > INT_MAX - 16) return abw_fail(ctx, ABW_ERR_ARG, "%s: dict_off not monotone", who);
            if (dict_off[d + 1] > dict_off[d] && !dict_vals) return abw_fail(ctx, ABW_ERR_ARG, "%s: null dictionary", who);
            for (int64_t i = dict_off[d] + 1; i < dict_off[d + 1]; i++)
                if (dict_vals[i] <= dict_vals[i - 1]) return abw_fail(ctx, ABW_ERR_ARG, "%s: dictionary of dimension %d is not strictly ascending", who, d);
        }
    }
    abw_device_guard g(ctx->device);
    ABW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // Everything that can reject the input is checked BEFORE the resident table is touched: a refused reload leaves
    // the previous table usable.  The new layout is built in locals and moved in once it is known to be good.
    struct Staged { std::vector<int32_t> h_scaf_first, h_scaf_ndps, h_T, h_span; std::vector<int64_t> h_Toff, h_binval_off; } s;

    // scaffold layout (ref: ScafDpData.cpp:91-99 numbers dps scaffold by scaffold)
    s.h_scaf_first.assign(S, 0);
    s.h_scaf_ndps.assign(S, 0);
    {
        int32_t prev = -1;
        for (int64_t p = 0; p < N; p++) {
            const int32_t sc = dp2scaf[p];
            if (sc < 0 || sc >= S || (sc != prev && sc != prev + 1))
                return abw_fail(ctx, ABW_ERR_ARG, "%s: dp2scaf must list scaffolds 0..S-1 contiguously and ascending (dp %lld)", who, (long long)p);
            if (sc != prev) s.h_scaf_first[sc] = (int32_t)p;
            s.h_scaf_ndps[sc]++;
            prev = sc;
        }
        if (prev != S - 1) return abw_fail(ctx, ABW_ERR_ARG, "%s: dp2scaf covers %d scaffolds, S = %d", who, prev + 1, S);
        for (int32_t sc = 0; sc < S; sc++)
            if (s.h_scaf_ndps[sc] < 2)
                return abw_fail(ctx, ABW_ERR_ARG, "%s: scaffold %d has %d dp(s); the reference drops such scaffolds (ScafDpData.cpp:92-93)", who, sc, s.h_scaf_ndps[sc]);
    }
    if (n_scg > 0) {
        if (scg_off[0] != 0) return abw_fail(ctx, ABW_ERR_ARG, "%s: scg_off[0] != 0", who);
        for (int32_t sc = 0; sc < S; sc++) if (scg_off[sc + 1] < scg_off[sc]) return abw_fail(ctx, ABW_ERR_ARG, "%s: scg_off not monotone", who);
        for (int32_t e = 0; e < scg_off[S]; e++) if (scg_ids[e] < 0 || scg_ids[e] >= n_scg) return abw_fail(ctx, ABW_ERR_ARG, "%s: scg id out of range", who);
    }

    // bins: narrow dimensions use the key as bin; wide ones the rank among the column's distinct keys
    s.h_T.assign(D, 0);
    s.h_span.assign(D, 0);                               // narrow dimensions: largest key of THIS table + 1 (<= T)
    s.h_Toff.assign(D + 1, 0);
    s.h_binval_off.assign(D, -1);
    std::vector<std::vector<int32_t>> uniq(D);
    std::vector<int32_t> bins;
    std::vector<char> is_wide(D, 0);
    std::vector<int32_t> bad_dim(D, 0);
    // A shard's layout comes from the shared dictionaries: built on the host, as is any table when ABW_SEP_LOAD_HOST is set
    // (the older path, kept as the check of the device build: tests/test_gpu_separator_load.py).
    const char* force_host = getenv("ABW_SEP_LOAD_HOST");
    const bool host_build = dict_off != nullptr || (force_host && force_host[0] == '1');
    std::vector<int32_t> from_device;
    if (host_build && keys_on_device) {
        from_device.resize((size_t)D * N);
        ABW_HIP(ctx, hipMemcpy(from_device.data(), keys, (size_t)D * N * 4, hipMemcpyDeviceToHost));
        keys = from_device.data();
        keys_on_device = false;
    }
    if (!host_build) {
        abw_sep_free(ctx);                               // nothing below rejects the input; failures from here on are the device's
        const int rc = sep_device_build(ctx, keys, keys_on_device, D, N, dp2scaf, s.h_T, s.h_span, is_wide, uniq);
        if (rc) { const std::string msg = ctx->err; abw_sep_free(ctx); ctx->err = msg; return rc; }
        Staged staged = std::move(s);
        return sep_load_upload(ctx, staged.h_scaf_first, staged.h_scaf_ndps, staged.h_T, staged.h_span, staged.h_Toff, staged.h_binval_off, uniq, bins, is_wide,
                               nullptr, D, N, dp2scaf, scaf_len, S, scg_off, scg_ids, n_scg, false);
    }
    {
        unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        auto classify_dim = [&](unsigned t) {
            for (int32_t d = t; d < D; d += nt) {
                const int32_t* col = keys + (int64_t)d * N;
                int32_t mn = INT_MAX, mx = INT_MIN;
                for (int64_t p = 0; p < N; p++) { mn = std::min(mn, col[p]); mx = std::max(mx, col[p]); }
                if (dict_off) {
                    if (dict_off[d + 1] == dict_off[d]) {
                        if (mn < 0 || mx >= NARROW_T) bad_dim[d] = 1;
                        s.h_T[d] = NARROW_T;
                        s.h_span[d] = std::max(1, std::min(mx, NARROW_T - 1) + 1);
                    } else {
                        is_wide[d] = 1;
                        uniq[d].assign(dict_vals + dict_off[d], dict_vals + dict_off[d + 1]);
                        s.h_T[d] = (int32_t)uniq[d].size();
                    }
                    continue;
                }
                if (mn >= 0 && mx < NARROW_T) { s.h_T[d] = std::max(mx + 1, narrow_bins_floor(ctx, d, D)); s.h_span[d] = mx + 1; continue; }
                is_wide[d] = 1;
                std::vector<int32_t>& u = uniq[d];
                u.assign(col, col + N);
                std::sort(u.begin(), u.end());
                u.erase(std::unique(u.begin(), u.end()), u.end());
                s.h_T[d] = (int32_t)u.size();
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(classify_dim, t);
        for (auto& x : th) x.join();
        bool any_wide = false;
        for (int32_t d = 0; d < D; d++) any_wide |= is_wide[d];
        if (any_wide) {
            bins.assign(keys, keys + (int64_t)D * N);
            auto rank_dim = [&](unsigned t) {
                for (int32_t d = t; d < D; d += nt) {
                    if (!is_wide[d]) continue;
                    const std::vector<int32_t>& u = uniq[d];
                    int32_t* col = bins.data() + (int64_t)d * N;
                    for (int64_t p = 0; p < N; p++) {
                        const auto it = std::lower_bound(u.begin(), u.end(), col[p]);
                        if (it == u.end() || *it != col[p]) { bad_dim[d] = 2; break; }
                        col[p] = (int32_t)(it - u.begin());
                    }
                }
            };
            std::vector<std::thread> th2;
            for (unsigned t = 0; t < nt; t++) th2.emplace_back(rank_dim, t);
            for (auto& x : th2) x.join();
        }
        for (int32_t d = 0; d < D; d++) {
            if (bad_dim[d] == 1) return abw_fail(ctx, ABW_ERR_ARG, "%s: dimension %d has no dictionary but keys outside [0,%d)", who, d, NARROW_T);
            if (bad_dim[d] == 2) return abw_fail(ctx, ABW_ERR_ARG, "%s: a key of dimension %d is missing from its dictionary", who, d);
        }
    }
    // the input is good: drop the previous table, move the staged layout in
    abw_sep_free(ctx);
    Staged staged = std::move(s);
    return sep_load_upload(ctx, staged.h_scaf_first, staged.h_scaf_ndps, staged.h_T, staged.h_span, staged.h_Toff, staged.h_binval_off, uniq, bins, is_wide,
                           keys, D, N, dp2scaf, scaf_len, S, scg_off, scg_ids, n_scg, dict_off != nullptr);
}

static SepDev sep_params(abw_sep_state& s, SepHost& H);

// The groups of narrow dimensions and their static tables (NGroup, abw_sep_common.h).  span[d]: largest key of dimension d
// in this table + 1 (the LDS layout follows the table; the histogram layout T may be wider when shards share it).
static int sep_build_narrow(abw_ctx* ctx, const std::vector<int32_t>& span, const int32_t* dp2scaf)
{
    abw_sep_state& s = ctx->sep;
    SepHost& H = g_extra_of(ctx)->h;
    const int32_t n_narrow = (int32_t)H.narrow_dims.size(), D = s.D, S = s.S;
    const int64_t N = s.N;
    H.nsort.assign(H.narrow_dims.begin(), H.narrow_dims.end());
    std::stable_sort(H.nsort.begin(), H.nsort.end(), [&](int32_t a, int32_t b) { return span[a] < span[b]; });
    H.nslot.assign((size_t)n_narrow, 0);
    H.toffn.assign((size_t)n_narrow + 1, 0);
    H.groups.clear();
    for (int32_t i = 0; i < n_narrow; i++) {
        const int32_t d = H.nsort[i], w = span[d];
        if (H.groups.empty() || H.groups.back().cnt == 8 || H.groups.back().total + w > GRP_BINS) {
            NGroup g;
            memset(&g, 0, sizeof(g));
            g.goff = H.toffn[i];
            H.groups.push_back(g);
        }
        NGroup& g = H.groups.back();
        g.loff[g.cnt] = g.total;
        g.toff[g.cnt] = (int32_t)(H.toffn[i] - g.goff);
        g.dim[g.cnt] = d;
        H.nslot[i] = (int32_t)(H.groups.size() - 1) * 8 + g.cnt;
        g.cnt++; g.total += w;
        H.toffn[i + 1] = H.toffn[i] + s.h_T[d];
    }
    std::vector<int32_t> slot((size_t)D, 0);
    for (int32_t i = 0; i < n_narrow; i++) slot[H.nsort[i]] = H.nslot[i];
    for (size_t w = 0; w < H.wide_dims.size(); w++) slot[H.wide_dims[w]] = -(int32_t)w - 1;
    int rc;
    if ((rc = upload(ctx, &H.d_dim_slot, slot.data(), (size_t)D))) return rc;
    if (n_narrow == 0) return ABW_OK;
    const size_t n_groups = H.groups.size();
    int32_t *d_nslot = nullptr, *d_dp2scaf = nullptr;
    if ((rc = upload(ctx, &H.d_groups, H.groups.data(), n_groups))) return rc;
    if ((rc = upload(ctx, &H.d_nsort, H.nsort.data(), (size_t)n_narrow))) return rc;
    if ((rc = upload(ctx, &H.d_toffn, H.toffn.data(), (size_t)n_narrow + 1))) return rc;
    ABW_HIP(ctx, hipMalloc(&H.d_ts8, n_groups * (size_t)S * 16));
    ABW_HIP(ctx, hipMalloc(&H.d_nk16, n_groups * (size_t)N * 16));
    ABW_HIP(ctx, hipMalloc(&H.d_srec, (size_t)S * 16));
    rc = upload(ctx, &d_nslot, H.nslot.data(), (size_t)n_narrow);
    if (!rc) rc = upload(ctx, &d_dp2scaf, dp2scaf, (size_t)N);
    if (!rc) {
        hipStream_t st = ctx->stream;
        SepDev P = sep_params(s, H);
        hipError_t e = hipMemsetAsync(H.d_ts8, 0, n_groups * (size_t)S * 16, st);               // the unused lanes of a group
        hipLaunchKernelGGL(sep_static_ts_kernel, dim3((unsigned)((S + 255) / 256), (unsigned)n_narrow), dim3(256), 0, st, P, H.d_nsort, d_nslot, S,
                           (uint16_t*)H.d_ts8);
        hipLaunchKernelGGL(sep_static_codes_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)n_groups), dim3(256), 0, st, P, H.d_groups, d_dp2scaf, S,
                           (const uint4*)H.d_ts8, (uint4*)H.d_nk16);
        hipLaunchKernelGGL(sep_static_srec_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, P, S, (int4*)H.d_srec);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = abw_fail(ctx, ABW_ERR_HIP, "abw_sep_load: building the grouped narrow columns: %s", hipGetErrorString(e));
    }
    if (d_nslot) (void)hipFree(d_nslot);
    if (d_dp2scaf) (void)hipFree(d_dp2scaf);
    return rc;
}

// second half of sep_load_impl: builds the chunk tables and uploads.  On any failure the half-built table is freed.
static int sep_load_upload_impl(abw_ctx* ctx, std::vector<int32_t>& st_first, std::vector<int32_t>& st_ndps, std::vector<int32_t>& st_T,
                                std::vector<int32_t>& st_span, std::vector<int64_t>& st_Toff, std::vector<int64_t>& st_binval_off, std::vector<std::vector<int32_t>>& uniq,
                                std::vector<int32_t>& bins, std::vector<char>& is_wide, const int32_t* keys, int32_t D, int64_t N,
                                const int32_t* dp2scaf, const int64_t* scaf_len, int32_t S, const int32_t* scg_off, const int32_t* scg_ids,
                                int32_t n_scg, bool sharded)
{
    abw_sep_state& s = ctx->sep;
    SepHost& H = g_extra_of(ctx)->h;
    s.h_scaf_first = std::move(st_first); s.h_scaf_ndps = std::move(st_ndps); s.h_T = std::move(st_T);
    s.h_Toff = std::move(st_Toff); s.h_binval_off = std::move(st_binval_off);
    s.h_binval.clear();
    H.narrow_dims.clear(); H.wide_dims.clear(); H.chunks.clear();
    H.dim_first_chunk.assign(D, 0); H.dim_nchunks.assign(D, 0);
    int64_t tb = 0;
    for (int32_t d = 0; d < D; d++) {
        s.h_Toff[d] = tb;
        tb += s.h_T[d];
        if (is_wide[d]) {
            H.wide_dims.push_back(d);
            s.h_binval_off[d] = (int64_t)s.h_binval.size();
            s.h_binval.insert(s.h_binval.end(), uniq[d].begin(), uniq[d].end());
        } else H.narrow_dims.push_back(d);
        H.dim_first_chunk[d] = (int32_t)H.chunks.size();
        for (int32_t st = 0; st < s.h_T[d]; st += SCAN_CH) {
            ChunkDesc c = {d, st, std::min(SCAN_CH, s.h_T[d] - st), H.dim_first_chunk[d]};
            H.chunks.push_back(c);
        }
        H.dim_nchunks[d] = (int32_t)H.chunks.size() - H.dim_first_chunk[d];
    }
    s.h_Toff[D] = tb;
    s.TB = tb;
    s.D = D; s.N = N; s.S = S; s.n_scg = n_scg;

    int rc;
    const bool built = keys == nullptr;                  // sep_device_build has left keys, bins and the sorted wide tables on the device
    if (!built) {
        if ((rc = upload(ctx, &s.d_keys, keys, (size_t)D * N))) return rc;
        if (!bins.empty()) { if ((rc = upload(ctx, &s.d_bins, bins.data(), (size_t)D * N))) return rc; }
    }
    if ((rc = upload(ctx, &s.d_scaf_first, s.h_scaf_first.data(), (size_t)S))) return rc;
    if ((rc = upload(ctx, &s.d_scaf_ndps, s.h_scaf_ndps.data(), (size_t)S))) return rc;
    if ((rc = upload(ctx, &s.d_scaf_len, scaf_len, (size_t)S))) return rc;
    if (n_scg > 0) {
        if ((rc = upload(ctx, &s.d_scg_off, scg_off, (size_t)S + 1))) return rc;
        if ((rc = upload(ctx, &s.d_scg_ids, scg_ids, (size_t)scg_off[S]))) return rc;
        std::vector<int32_t> carriers;                    // the scaffolds that carry an SCG: all that the SCG lists of a node depend on
        for (int32_t sc = 0; sc < S; sc++) if (scg_off[sc + 1] > scg_off[sc]) carriers.push_back(sc);
        H.n_carriers = (int32_t)carriers.size();
        if ((rc = upload(ctx, &H.d_carriers, carriers.data(), carriers.size()))) return rc;
        // ... and the same pairs by SCG: g2c_off[g] .. g2c_off[g + 1] = the scaffolds that carry SCG g
        std::vector<int32_t> goff((size_t)n_scg + 1, 0), g2c((size_t)scg_off[S]);
        for (int32_t e = 0; e < scg_off[S]; e++) goff[(size_t)scg_ids[e] + 1]++;
        for (int32_t g = 0; g < n_scg; g++) goff[(size_t)g + 1] += goff[g];
        {
            std::vector<int32_t> at(goff.begin(), goff.end() - 1);
            for (int32_t sc = 0; sc < S; sc++) for (int32_t e = scg_off[sc]; e < scg_off[sc + 1]; e++) g2c[(size_t)at[scg_ids[e]]++] = sc;
        }
        if ((rc = upload(ctx, &H.d_g2c_off, goff.data(), goff.size()))) return rc;
        if ((rc = upload(ctx, &H.d_g2c, g2c.data(), g2c.size()))) return rc;
    }
    if ((rc = upload<uint8_t>(ctx, &H.d_member, nullptr, (size_t)S + 64))) return rc;
    if ((rc = upload(ctx, &s.d_T, s.h_T.data(), (size_t)D))) return rc;
    if ((rc = upload(ctx, &s.d_Toff, s.h_Toff.data(), (size_t)D + 1))) return rc;
    if ((rc = upload(ctx, &H.d_narrow, H.narrow_dims.data(), H.narrow_dims.size()))) return rc;
    if ((rc = upload(ctx, &H.d_wide, H.wide_dims.data(), H.wide_dims.size()))) return rc;
    if ((rc = upload(ctx, &H.d_dim_first_chunk, H.dim_first_chunk.data(), (size_t)D))) return rc;
    if ((rc = upload(ctx, &H.d_dim_nchunks, H.dim_nchunks.data(), (size_t)D))) return rc;
    if ((rc = upload(ctx, &H.d_chunks, H.chunks.data(), H.chunks.size()))) return rc;
    if ((rc = upload<Sums>(ctx, &H.d_chunk_sums, nullptr, H.chunks.size()))) return rc;
    if ((rc = upload<Sums>(ctx, &H.d_dim_total, nullptr, (size_t)D))) return rc;
    if ((rc = upload<Cand>(ctx, &H.d_chunk_best, nullptr, H.chunks.size()))) return rc;
    if (!H.wide_dims.empty() && built) {
        const size_t Dw = H.wide_dims.size();
        if ((rc = upload<int32_t>(ctx, &H.d_wts, nullptr, Dw * (size_t)S))) return rc;
        if ((rc = upload<int32_t>(ctx, &H.d_wbelow, nullptr, Dw * (size_t)S))) return rc;
    } else if (!H.wide_dims.empty()) {
        // dps of every wide dimension in ascending (bin, dp) order: counting sort on the dense ranks
        const size_t Dw = H.wide_dims.size();
        std::vector<int32_t> sscaf(Dw * (size_t)N), sbin(Dw * (size_t)N), sdp(Dw * (size_t)N);
        unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        auto sort_dim = [&](unsigned t) {
            std::vector<int64_t> start;
            for (size_t w = t; w < Dw; w += nt) {
                const int32_t d = H.wide_dims[w];
                const int32_t* col = bins.data() + (int64_t)d * N;
                start.assign((size_t)s.h_T[d] + 1, 0);
                for (int64_t p = 0; p < N; p++) start[(size_t)col[p] + 1]++;
                for (int32_t b = 0; b < s.h_T[d]; b++) start[(size_t)b + 1] += start[b];
                int32_t* pw = sscaf.data() + w * (size_t)N;
                int32_t* sw = sbin.data() + w * (size_t)N;
                int32_t* dw = sdp.data() + w * (size_t)N;
                for (int64_t p = 0; p < N; p++) { const int64_t at = start[col[p]]++; pw[at] = dp2scaf[p]; sw[at] = col[p]; dw[at] = (int32_t)p; }
            }
        };
        std::vector<std::thread> th3;
        for (unsigned t = 0; t < nt; t++) th3.emplace_back(sort_dim, t);
        for (auto& x : th3) x.join();
        if ((rc = upload(ctx, &H.d_sscaf, sscaf.data(), sscaf.size()))) return rc;
        if ((rc = upload(ctx, &H.d_sbin, sbin.data(), sbin.size()))) return rc;
        if ((rc = upload(ctx, &H.d_sdp, sdp.data(), sdp.size()))) return rc;
        if ((rc = upload<int32_t>(ctx, &H.d_wts, nullptr, Dw * (size_t)S))) return rc;
        if ((rc = upload<int32_t>(ctx, &H.d_wbelow, nullptr, Dw * (size_t)S))) return rc;
    }
    if ((rc = abw_reserve(ctx, s.hist, (size_t)tb * (4 * 3 + 8) + 64))) return rc;
    ABW_HIP(ctx, hipMemset(s.hist.p, 0, (size_t)tb * (4 * 3 + 8) + 64));          // all zero between nodes from here on
    {
        std::vector<int32_t> none((size_t)2 * D);
        for (int32_t d = 0; d < D; d++) { none[2 * d] = INT_MAX; none[2 * d + 1] = -1; }   // nothing touched yet
        if ((rc = upload(ctx, &H.d_range, none.data(), none.size()))) return rc;
    }
    H.sharded = sharded;
    H.packed = true;
    for (int32_t sc = 0; sc < S; sc++) if (s.h_scaf_ndps[sc] > PACK_MAX_NDPS) { H.packed = false; break; }
    if (getenv("ABW_SEP_UNPACKED")) H.packed = false;      // the tests run both forms of the accumulation
    if ((rc = sep_build_narrow(ctx, st_span, dp2scaf))) return rc;
    if ((rc = abw_reserve(ctx, s.scg_lohi, (size_t)std::max<int64_t>(1, (int64_t)D * n_scg) * 8))) return rc;
    if ((rc = abw_reserve(ctx, s.dimbest, (size_t)D * sizeof(DimBestDev) + sizeof(GlobalRes)))) return rc;
    s.loaded = true;
    s.node_open = false;
    return ABW_OK;
}

static int sep_load_upload(abw_ctx* ctx, std::vector<int32_t>& st_first, std::vector<int32_t>& st_ndps, std::vector<int32_t>& st_T,
                           std::vector<int32_t>& st_span, std::vector<int64_t>& st_Toff, std::vector<int64_t>& st_binval_off, std::vector<std::vector<int32_t>>& uniq,
                           std::vector<int32_t>& bins, std::vector<char>& is_wide, const int32_t* keys, int32_t D, int64_t N,
                           const int32_t* dp2scaf, const int64_t* scaf_len, int32_t S, const int32_t* scg_off, const int32_t* scg_ids,
                           int32_t n_scg, bool sharded)
{
    const int rc = sep_load_upload_impl(ctx, st_first, st_ndps, st_T, st_span, st_Toff, st_binval_off, uniq, bins, is_wide, keys, D, N, dp2scaf,
                                        scaf_len, S, scg_off, scg_ids, n_scg, sharded);
    if (rc) { const std::string msg = ctx->err; abw_sep_free(ctx); ctx->err = msg; }      // no half-populated table behind an error
    return rc;
}

static SepDev sep_params(abw_sep_state& s, SepHost& H)
{
    SepDev P;
    P.bins = s.d_bins ? s.d_bins : s.d_keys; P.N = s.N;
    P.scaf_first = s.d_scaf_first; P.scaf_ndps = s.d_scaf_ndps; P.scaf_len = s.d_scaf_len;
    P.scg_off = s.d_scg_off; P.scg_ids = s.d_scg_ids; P.n_scg = s.n_scg;
    P.T = s.d_T; P.Toff = s.d_Toff;
    P.h_st = (unsigned long long*)s.hist.p; P.h_bp = P.h_st + s.TB; P.h_tot = (uint32_t*)(P.h_bp + s.TB);
    P.scg_lo = (int32_t*)s.scg_lohi.p; P.scg_hi = P.scg_lo + (int64_t)s.D * s.n_scg;
    P.range = H.d_range;
    return P;
}

// The node's scaffold ids and the offsets of their dps in the node's own numbering go up in one copy (s.node_scaf: ids[m] then
// dp offsets[m + 1]); *n_out = the node's dps.  all: every scaffold, nothing to upload.
static int sep_upload_node(abw_ctx* ctx, const char* who, bool all, const int32_t* scaf_ids, int32_t m, int64_t* n_out)
{
    abw_sep_state& s = ctx->sep;
    hipStream_t st = ctx->stream;
    if (m < 0 || (!all && m > 0 && !scaf_ids)) return abw_fail(ctx, ABW_ERR_ARG, "%s: bad scaffold list", who);
    int64_t n = 0;
    int rc;
    if (!all && m > 0) {
        // ids and dp offsets go up in one copy from pinned memory (the block is sized for the results of any node of this
        // table, too, so that the evaluation phase never has to grow it while this copy is in flight)
        const size_t out_bytes = (((size_t)s.D * sizeof(DimBestDev) + sizeof(GlobalRes) + 15) & ~(size_t)15) + 16 + (((size_t)s.S + 15) & ~(size_t)15) +
                                 (size_t)s.N + 16;
        if ((rc = abw_reserve_pinned(ctx, std::max(out_bytes, ((size_t)2 * m + 1) * 4)))) return rc;
        int32_t* h_ids = (int32_t*)ctx->pinned;
        int32_t* dpoff = h_ids + m;
        memcpy(h_ids, scaf_ids, (size_t)m * 4);
        // duplicate check in O(m): a stamp per scaffold, a new stamp value per call
        if (s.h_stamp.size() != (size_t)s.S) { s.h_stamp.assign((size_t)s.S, 0); s.stamp = 0; }
        if (++s.stamp == 0) { std::fill(s.h_stamp.begin(), s.h_stamp.end(), 0u); s.stamp = 1; }
        for (int32_t i = 0; i < m; i++) {
            const int32_t sc = scaf_ids[i];
            if (sc < 0 || sc >= s.S) return abw_fail(ctx, ABW_ERR_ARG, "%s: scaffold id %d out of range", who, sc);
            if (s.h_stamp[sc] == s.stamp) return abw_fail(ctx, ABW_ERR_ARG, "%s: duplicate scaffold id", who);
            s.h_stamp[sc] = s.stamp;
            dpoff[i] = (int32_t)n;
            n += s.h_scaf_ndps[sc];
        }
        dpoff[m] = (int32_t)n;
        if ((rc = abw_reserve(ctx, s.node_scaf, ((size_t)2 * m + 1) * 4))) return rc;      // ids[m] then dp offsets[m + 1]
        ABW_HIP(ctx, hipMemcpyAsync(s.node_scaf.p, h_ids, ((size_t)2 * m + 1) * 4, hipMemcpyHostToDevice, st));
    } else if (all) n = s.N;
    *n_out = n;
    return ABW_OK;
}

// Phase 1: the node's scaffolds held by this context -> histograms and SCG lo/hi lists.
// all = every scaffold of the table (scaf_ids ignored); otherwise m >= 0 ids (an empty shard of a node is fine).
static int sep_accum_phase(abw_ctx* ctx, const char* who, bool all, const int32_t* scaf_ids, int32_t m)
{
    abw_sep_state& s = ctx->sep;
    SepHost& H = g_extra_of(ctx)->h;
    hipStream_t st = ctx->stream;
    s.node_open = false;
    if (all) m = s.S;
    int64_t n = 0;
    int rc;
    if ((rc = sep_upload_node(ctx, who, all, scaf_ids, m, &n))) return rc;
    const int32_t* d_node_scaf = all ? nullptr : (const int32_t*)s.node_scaf.p;

    SepDev P = sep_params(s, H);
    // wide dimensions: small nodes scatter their dps into the histograms (cost ~ node); big ones evaluate prefix sums along
    // the table's static (bin, dp) order and keep no histogram at all (abw_tree.hip: node_wide_*); shards, whose histograms
    // are what they exchange, take the sorted walk into the histograms
    const bool scatter = !H.sharded && n * 6 < s.N;
    const bool scan = !scatter && m > 0 && abw_node_wide_usable(ctx);
    H.node_scan = scan;
    hipLaunchKernelGGL(sep_clear_kernel, dim3((unsigned)H.chunks.size()), dim3(256), 0, st, P, H.d_chunks);      // what the node before touched
    ABW_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(sep_range_reset_kernel, dim3((s.D + 63) / 64), dim3(64), 0, st, P, s.D, H.d_wide, (int32_t)H.wide_dims.size(), (scatter || scan) ? 0 : 1);
    ABW_LAUNCH_CHECK(ctx);
    if (m > 0) {
        const unsigned gx = (unsigned)((m + TILE_A - 1) / TILE_A);
        ABW_PROF(ctx, "sep_accum");
        const uint8_t* d_member = nullptr;                 // a byte per scaffold: in the node (NULL: all are)
        if (!all) {
            ABW_HIP(ctx, hipMemsetAsync(H.d_member, 0, (size_t)s.S, st));
            hipLaunchKernelGGL(sep_member_kernel, dim3((unsigned)std::min<int64_t>((m + 255) / 256, ctx->num_cu * 8)), dim3(256), 0, st, d_node_scaf, m, H.d_member);
            ABW_LAUNCH_CHECK(ctx);
            d_member = H.d_member;
        }
        if (s.n_scg > 0) {
            // narrow dimensions always; the wide ones when their flip values are static tables (the scan), else their own kernels do it
            if (scan && (rc = abw_tree_static_ensure(ctx))) return rc;
            hipLaunchKernelGGL(sep_scg_kernel, dim3((unsigned)(((int64_t)s.D * s.n_scg + 3) / 4)), dim3(256), 0, st, P, s.D, (const int32_t*)H.d_dim_slot,
                               (const uint16_t*)H.d_ts8, scan ? (const uint4*)H.d_wext : (const uint4*)nullptr, s.S, (const int32_t*)H.d_g2c_off,
                               (const int32_t*)H.d_g2c, d_member);
            ABW_LAUNCH_CHECK(ctx);
        }
        if (!H.narrow_dims.empty()) {
            if (H.packed) hipLaunchKernelGGL(sep_accum_kernel<true>, dim3(gx, (unsigned)H.groups.size()), dim3(256), 0, st, P, H.d_groups, (const uint4*)H.d_nk16,
                                             (const uint4*)H.d_ts8, (const int4*)H.d_srec, s.S, d_node_scaf, m);
            else hipLaunchKernelGGL(sep_accum_kernel<false>, dim3(gx, (unsigned)H.groups.size()), dim3(256), 0, st, P, H.d_groups, (const uint4*)H.d_nk16,
                                    (const uint4*)H.d_ts8, (const int4*)H.d_srec, s.S, d_node_scaf, m);
            ABW_LAUNCH_CHECK(ctx);
        }
        if (!H.wide_dims.empty()) {
            const unsigned Dw = (unsigned)H.wide_dims.size();
            if (scatter) {
                hipLaunchKernelGGL(sep_wide_scatter_kernel, dim3((unsigned)((m + 255) / 256), Dw), dim3(256), 0, st, P, H.d_wide, d_node_scaf, m);
                ABW_LAUNCH_CHECK(ctx);
            } else if (scan) {
                if ((rc = abw_node_wide_accum(ctx, d_member))) return rc;
            } else {
            WideDev W = {H.d_sscaf, H.d_sbin, H.d_wts, H.d_wbelow, s.S};
            ABW_HIP(ctx, hipMemsetAsync(H.d_wts, 0xFF, (size_t)Dw * (size_t)s.S * 4, st));      // -1: not in the node
            hipLaunchKernelGGL(sep_wide_ts_kernel, dim3((unsigned)((m + 255) / 256), Dw), dim3(256), 0, st, P, W, H.d_wide, d_node_scaf, m);
            ABW_LAUNCH_CHECK(ctx);
            hipLaunchKernelGGL(sep_wide_hist_kernel, dim3((unsigned)((s.N + 256 * WIDE_TILES - 1) / (256 * WIDE_TILES)), Dw), dim3(256), 0, st, P, W, H.d_wide);
            ABW_LAUNCH_CHECK(ctx);
            }
        }
    }
    s.node_open = true; s.node_all = all; s.node_m = m; s.node_n = n;
    return ABW_OK;
}

// Phase 2: scan + evaluation of the (combined) histograms, the global winner, and this context's part of
// the finalisation.  out->ndps1/2 = dps of THIS context's scaffolds that went to cluster1/2; out->split is
// left 0: the size check needs the totals over all shards (sep_decide).
static int sep_eval_phase(abw_ctx* ctx, const char* who, int32_t min_ndps, abw_split* out, abw_dimbest* per_dim,
                          uint8_t* scaf_side, uint8_t* dp_raw_side)
{
    abw_sep_state& s = ctx->sep;
    SepHost& H = g_extra_of(ctx)->h;
    hipStream_t st = ctx->stream;
    if (!s.node_open) return abw_fail(ctx, ABW_ERR_STATE, "%s: no node accumulated", who);
    s.node_open = false;
    const int32_t m = s.node_m;
    const int64_t n = s.node_n;
    const int32_t* d_node_scaf = s.node_all ? nullptr : (const int32_t*)s.node_scaf.p;
    const int32_t* d_node_dpoff = s.node_all ? nullptr : (const int32_t*)s.node_scaf.p + s.node_m;
    memset(out, 0, sizeof(*out));
    out->dimension = -1; out->key = -1;
    SepDev P = sep_params(s, H);
    int rc;

    const unsigned nch = (unsigned)H.chunks.size();
    // everything that goes back to the host lies in ONE device block, so that one copy brings it:
    // winners of the dimensions | global winner | 2 totals | scaffold sides | raw flags of the dps
    const size_t b_db = (size_t)s.D * sizeof(DimBestDev) + sizeof(GlobalRes);
    const size_t o_tot = (b_db + 15) & ~(size_t)15, o_side = o_tot + 16, o_raw = o_side + (((size_t)m + 15) & ~(size_t)15);
    if ((rc = abw_reserve(ctx, s.dimbest, o_raw + (size_t)n + 16))) return rc;
    uint8_t* d_back = (uint8_t*)s.dimbest.p;
    DimBestDev* d_dimbest = (DimBestDev*)d_back;
    GlobalRes* d_win = (GlobalRes*)(d_dimbest + s.D);
    unsigned long long* d_tot = (unsigned long long*)(d_back + o_tot);
    uint8_t* d_side = d_back + o_side;
    uint8_t* d_raw = d_back + o_raw;
    {
    ABW_PROF(ctx, "sep_eval");
    hipLaunchKernelGGL(sep_chunk_sums_kernel, dim3(nch), dim3(256), 0, st, P, H.d_chunks, H.d_chunk_sums);
    ABW_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(sep_chunk_scan_kernel, dim3(s.D), dim3(64), 0, st, H.d_dim_first_chunk, H.d_dim_nchunks, s.D, H.d_chunk_sums, H.d_dim_total);
    ABW_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(sep_eval_kernel, dim3(nch), dim3(256), 0, st, P, H.d_chunks, H.d_chunk_sums, H.d_dim_total, min_ndps, H.d_chunk_best);
    ABW_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(sep_dimbest_kernel, dim3(s.D), dim3(64), 0, st, H.d_dim_first_chunk, H.d_dim_nchunks, s.D, H.d_chunk_best, H.d_dim_total, d_dimbest);
    ABW_LAUNCH_CHECK(ctx);
    }
    if (H.node_scan) {
        // the wide dimensions' thresholds along the sorted positions (their histogram ranges are empty: the chunk kernels
        // above skipped them), with the narrow dimensions' best product as the bar
        ABW_PROF(ctx, "sep_wide");
        if ((rc = abw_node_wide_eval(ctx, s.node_all ? (const uint8_t*)nullptr : (const uint8_t*)H.d_member, per_dim != nullptr, min_ndps, P.scg_lo, P.scg_hi, d_dimbest))) return rc;
    }
    // the global winner and this context's part of the finalisation follow on the device without a host round trip
    hipLaunchKernelGGL(sep_global_kernel, dim3(1), dim3(64), 0, st, d_dimbest, s.D, d_win, d_tot);
    ABW_LAUNCH_CHECK(ctx);
    if (m > 0) {
        hipLaunchKernelGGL(sep_assign_kernel, dim3((m + 255) / 256), dim3(256), 0, st, P, d_node_scaf, d_node_dpoff, m, d_win, d_raw, d_side, d_tot);
        ABW_LAUNCH_CHECK(ctx);
    }
    const bool want_side = scaf_side && m > 0, want_raw = dp_raw_side && n > 0;
    const size_t back_bytes = want_raw ? o_raw + (size_t)n : want_side ? o_side + (size_t)m : o_side;
    if ((rc = abw_reserve_pinned(ctx, o_raw + (size_t)n + 16))) return rc;
    uint8_t* hp = (uint8_t*)ctx->pinned;
    ABW_HIP(ctx, hipMemcpyAsync(hp, d_back, back_bytes, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    const DimBestDev* db = (const DimBestDev*)hp;
    const GlobalRes win = *(const GlobalRes*)(hp + (size_t)s.D * sizeof(DimBestDev));

    auto key_of = [&](int32_t d, int32_t bin) -> int32_t {
        return s.h_binval_off[d] < 0 ? bin : s.h_binval[(size_t)s.h_binval_off[d] + bin];
    };
    if (per_dim) {
        for (int32_t d = 0; d < s.D; d++) {
            const bool f = db[d].found != 0;
            per_dim[d].found = f; per_dim[d].key = f ? key_of(d, db[d].bin) : -1;
            per_dim[d].sensitivity = f ? db[d].sens : 0.0; per_dim[d].specificity = f ? db[d].spec : 0.0;
        }
    }
    if (scaf_side && m > 0) memset(scaf_side, 0, (size_t)m);
    if (dp_raw_side && n > 0) memset(dp_raw_side, 0, (size_t)n);
    if (!win.found) return ABW_OK;
    out->dimension = win.dim + 1;
    out->key = key_of(win.dim, win.bin);
    out->sensitivity = win.sens; out->specificity = win.spec;
    out->legal = win.legal;                                           // is_legal, ref ...Specificity.h:31-35
    if (!out->legal) return ABW_OK;                                   // reconstruct_best_clusters returns early (ref ClusterSeparator.cpp:27-28)
    if (m == 0) return ABW_OK;                                        // this shard holds nothing of the node
    if (want_side) memcpy(scaf_side, hp + o_side, (size_t)m);
    if (want_raw) memcpy(dp_raw_side, hp + o_raw, (size_t)n);
    const unsigned long long* tot2 = (const unsigned long long*)(hp + o_tot);
    out->ndps1 = (int64_t)tot2[0]; out->ndps2 = (int64_t)tot2[1];
    return ABW_OK;
}

extern "C" {

int abw_sep_load(abw_ctx* ctx, const int32_t* keys, int32_t D, int64_t N, const int32_t* dp2scaf,
                 const int64_t* scaf_len, int32_t S, const int32_t* scg_off, const int32_t* scg_ids,
                 int32_t n_scg)
{
    const int rc = sep_load_impl(ctx, "abw_sep_load", keys, D, N, dp2scaf, scaf_len, S, scg_off, scg_ids, n_scg, nullptr, nullptr);
    if (ctx) ctx->sep_min_bins.clear();
    return rc;
}

// Tables whose histograms are added up bin by bin (the shards of abw_sep_stree_*) need the same number of bins in every narrow
// column: bins[d] = the largest key of column d over ALL of them + 1.  Applies to the next abw_sep_load / abw_sep_load_dev of
// this context only (a column with more keys than that in its own table keeps its own count).  bins = NULL: forget it.
int abw_sep_narrow_bins(abw_ctx* ctx, const int32_t* bins, int32_t D)
{
    if (!ctx) return ABW_ERR_ARG;
    if (bins && D <= 0) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_narrow_bins: bad argument");
    ctx->sep_min_bins.clear();
    if (bins) ctx->sep_min_bins.assign(bins, bins + D);
    return ABW_OK;
}

int abw_sep_load_dev(abw_ctx* ctx, const int32_t* d_keys, int32_t D, int64_t N, const int32_t* dp2scaf,
                     const int64_t* scaf_len, int32_t S, const int32_t* scg_off, const int32_t* scg_ids,
                     int32_t n_scg)
{
    const int rc = sep_load_impl(ctx, "abw_sep_load_dev", d_keys, D, N, dp2scaf, scaf_len, S, scg_off, scg_ids, n_scg, nullptr, nullptr, true);
    if (ctx) ctx->sep_min_bins.clear();
    return rc;
}

int abw_sep_shard_load(abw_ctx* ctx, const int32_t* keys, int32_t D, int64_t N, const int32_t* dp2scaf,
                       const int64_t* scaf_len, int32_t S, const int32_t* scg_off, const int32_t* scg_ids,
                       int32_t n_scg, const int32_t* dict_vals, const int64_t* dict_off)
{
    if (ctx && !dict_off) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_shard_load: null dict_off");
    return sep_load_impl(ctx, "abw_sep_shard_load", keys, D, N, dp2scaf, scaf_len, S, scg_off, scg_ids, n_scg, dict_vals, dict_off);
}

const int32_t* abw_sep_dev_keys(const abw_ctx* ctx) { return (ctx && ctx->sep.loaded) ? ctx->sep.d_keys : nullptr; }

// ref ClusterSeparator.cpp:125-132: both clusters need cluster_ndps_threshold dps, else everything is reset
int abw_sep_shard_decide(abw_split* out, int64_t ndps1_total, int64_t ndps2_total, int32_t min_ndps)
{
    if (!out) return ABW_ERR_ARG;
    if (out->dimension < 0 || !out->legal) { out->split = 0; out->ndps1 = out->ndps2 = 0; return ABW_OK; }
    if (ndps1_total >= min_ndps && ndps2_total >= min_ndps) {
        out->split = 1; out->ndps1 = ndps1_total; out->ndps2 = ndps2_total;
    } else {
        out->split = 0; out->ndps1 = out->ndps2 = 0;
        out->dimension = -1; out->key = -1; out->sensitivity = 0; out->specificity = 0;   // best_separation->reset(), ref :130
    }
    return ABW_OK;
}

int abw_sep_node(abw_ctx* ctx, const int32_t* scaf_ids, int32_t n_scaf, int32_t min_ndps,
                 abw_split* out, abw_dimbest* per_dim, uint8_t* scaf_side, uint8_t* dp_raw_side)
{
    if (!ctx) return ABW_ERR_ARG;
    abw_sep_state& s = ctx->sep;
    if (!s.loaded) return abw_fail(ctx, ABW_ERR_STATE, "abw_sep_node: call abw_sep_load first");
    if (!out) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_node: null out");
    if (scaf_ids && n_scaf <= 0) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_node: empty node");
    abw_device_guard g(ctx->device);
    int rc;
    if ((rc = sep_accum_phase(ctx, "abw_sep_node", scaf_ids == nullptr, scaf_ids, n_scaf))) return rc;
    const int32_t m = s.node_m;
    if ((rc = sep_eval_phase(ctx, "abw_sep_node", min_ndps, out, per_dim, scaf_side, dp_raw_side))) return rc;
    const int32_t legal = out->legal;
    abw_sep_shard_decide(out, out->ndps1, out->ndps2, min_ndps);
    if (legal && !out->split && scaf_side) memset(scaf_side, 0, (size_t)m);
    return ABW_OK;
}

// The finalisation of a node whose winner is known already (ref ClusterSeparator.cpp:35-53,87-132 given best_separation): a
// caller that ran the whole recursion in one call (abw_sep_tree) and now needs a node's raw sets and scaffold sides.
int abw_sep_node_assign(abw_ctx* ctx, const int32_t* scaf_ids, int32_t n_scaf, int32_t dimension, int32_t key, int32_t min_ndps,
                        abw_split* out, uint8_t* scaf_side, uint8_t* dp_raw_side)
{
    if (!ctx) return ABW_ERR_ARG;
    abw_sep_state& s = ctx->sep;
    if (!s.loaded) return abw_fail(ctx, ABW_ERR_STATE, "abw_sep_node_assign: call abw_sep_load first");
    SepHost& H = g_extra_of(ctx)->h;
    if (H.sharded) return abw_fail(ctx, ABW_ERR_STATE, "abw_sep_node_assign: the table is one shard of a scaffold-sharded problem");
    if (!out || dimension < 1 || dimension > s.D) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_node_assign: bad argument");
    if (scaf_ids && n_scaf <= 0) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_node_assign: empty node");
    const int32_t d = dimension - 1;
    int32_t bin = key;
    if (s.h_binval_off[d] >= 0) {                              // a wide dimension: the key's rank among the column's distinct keys
        const int32_t* v = s.h_binval.data() + s.h_binval_off[d];
        const int32_t* e = std::lower_bound(v, v + s.h_T[d], key);
        if (e == v + s.h_T[d] || *e != key) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_node_assign: dimension %d holds no key %d", dimension, key);
        bin = (int32_t)(e - v);
    } else if (key < 0 || key >= s.h_T[d]) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_node_assign: dimension %d holds no key %d", dimension, key);
    abw_device_guard g(ctx->device);
    hipStream_t st = ctx->stream;
    s.node_open = false;
    const bool all = scaf_ids == nullptr;
    const int32_t m = all ? s.S : n_scaf;
    int64_t n = 0;
    int rc;
    if ((rc = sep_upload_node(ctx, "abw_sep_node_assign", all, scaf_ids, m, &n))) return rc;
    const int32_t* d_node_scaf = all ? nullptr : (const int32_t*)s.node_scaf.p;
    const int32_t* d_node_dpoff = all ? nullptr : (const int32_t*)s.node_scaf.p + m;
    memset(out, 0, sizeof(*out));
    SepDev P = sep_params(s, H);
    // the same block layout as the evaluation phase: global winner | 2 totals | scaffold sides | raw flags
    const size_t b_db = (size_t)s.D * sizeof(DimBestDev) + sizeof(GlobalRes);
    const size_t o_tot = (b_db + 15) & ~(size_t)15, o_side = o_tot + 16, o_raw = o_side + (((size_t)m + 15) & ~(size_t)15);
    if ((rc = abw_reserve(ctx, s.dimbest, o_raw + (size_t)n + 16))) return rc;
    if ((rc = abw_reserve_pinned(ctx, o_raw + (size_t)n + 16 + sizeof(GlobalRes) + 64))) return rc;
    uint8_t* d_back = (uint8_t*)s.dimbest.p;
    GlobalRes* d_win = (GlobalRes*)(d_back + (size_t)s.D * sizeof(DimBestDev));
    unsigned long long* d_tot = (unsigned long long*)(d_back + o_tot);
    uint8_t* hp = (uint8_t*)ctx->pinned;
    GlobalRes* h_win = (GlobalRes*)(hp + ((o_raw + (size_t)n + 16 + 15) & ~(size_t)15));
    ABW_HIP(ctx, hipStreamSynchronize(st));                    // the upload of the ids has left the pinned block
    h_win->found = 1; h_win->dim = d; h_win->bin = bin; h_win->legal = 1; h_win->sens = 0; h_win->spec = 0; h_win->size1 = 0; h_win->n = n;
    ABW_HIP(ctx, hipMemcpyAsync(d_win, h_win, sizeof(GlobalRes), hipMemcpyHostToDevice, st));
    ABW_HIP(ctx, hipMemsetAsync(d_tot, 0, 16, st));
    hipLaunchKernelGGL(sep_count_le_kernel, dim3((m + 255) / 256), dim3(256), 0, st, P, d_node_scaf, m, d_win);
    ABW_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(sep_assign_kernel, dim3((m + 255) / 256), dim3(256), 0, st, P, d_node_scaf, d_node_dpoff, m, d_win, d_back + o_raw, d_back + o_side, d_tot);
    ABW_LAUNCH_CHECK(ctx);
    const size_t back_bytes = dp_raw_side ? o_raw + (size_t)n : o_side + (size_t)m;
    ABW_HIP(ctx, hipMemcpyAsync(hp + o_tot, d_back + o_tot, back_bytes - o_tot, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    if (scaf_side) memcpy(scaf_side, hp + o_side, (size_t)m);
    if (dp_raw_side) memcpy(dp_raw_side, hp + o_raw, (size_t)n);
    const unsigned long long* tot2 = (const unsigned long long*)(hp + o_tot);
    out->dimension = dimension; out->key = key; out->legal = 1;
    abw_sep_shard_decide(out, (int64_t)tot2[0], (int64_t)tot2[1], min_ndps);
    if (!out->split && scaf_side) memset(scaf_side, 0, (size_t)m);
    return ABW_OK;
}

int abw_sep_shard_accum(abw_ctx* ctx, const int32_t* scaf_ids, int32_t n_scaf, abw_sep_exchange* ex)
{
    if (!ctx) return ABW_ERR_ARG;
    abw_sep_state& s = ctx->sep;
    if (!s.loaded) return abw_fail(ctx, ABW_ERR_STATE, "abw_sep_shard_accum: call abw_sep_shard_load first");
    if (!ex) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_shard_accum: null exchange");
    abw_device_guard g(ctx->device);
    int rc;
    if ((rc = sep_accum_phase(ctx, "abw_sep_shard_accum", n_scaf < 0, scaf_ids, n_scaf))) return rc;
    ex->d_hist = (int64_t*)s.hist.p;
    ex->hist_count = ((uint64_t)s.TB * 20 + 7) / 8;
    ex->d_scg_lo = (int32_t*)s.scg_lohi.p;
    ex->d_scg_hi = ex->d_scg_lo + (int64_t)s.D * s.n_scg;
    ex->scg_count = (uint64_t)s.D * (uint64_t)s.n_scg;
    return ABW_OK;
}

// All-reduce of the exchange buffers over the contexts of ONE process (a context per device of the node, or several on
// one device): direct reduce-scatter + all-gather with peer copies.  Context k owns the k-th slice of every buffer: it
// pulls that slice from every other context into a staging block and folds it in (integer sum / min / max: any order
// gives the same bits), then every context pulls the finished slices from their owners.  Each owner works on its own
// stream, so on an xGMI mesh all links carry traffic at once.
int abw_sep_shard_combine(abw_ctx* const* ctxs, int32_t n_ctx)
{
    if (!ctxs || n_ctx <= 0) return ABW_ERR_ARG;
    for (int32_t k = 0; k < n_ctx; k++) if (!ctxs[k]) return ABW_ERR_ARG;
    abw_ctx* c0 = ctxs[0];
    for (int32_t k = 0; k < n_ctx; k++) {
        const abw_sep_state& s = ctxs[k]->sep;
        if (!s.loaded || !s.node_open) return abw_fail(ctxs[k], ABW_ERR_STATE, "abw_sep_shard_combine: context %d has no node accumulated (abw_sep_shard_accum)", k);
        if (s.TB != c0->sep.TB || s.D != c0->sep.D || s.n_scg != c0->sep.n_scg)
            return abw_fail(ctxs[k], ABW_ERR_ARG, "abw_sep_shard_combine: context %d was loaded with another bin layout", k);
        for (int32_t j = 0; j < k; j++) if (ctxs[j] == ctxs[k]) return abw_fail(ctxs[k], ABW_ERR_ARG, "abw_sep_shard_combine: context %d listed twice", k);
    }
    if (n_ctx == 1) return ABW_OK;
    const uint64_t H = ((uint64_t)c0->sep.TB * 20 + 7) / 8, C = (uint64_t)c0->sep.D * (uint64_t)c0->sep.n_scg;
    auto cut = [&](uint64_t total, int32_t k) { return total * (uint64_t)k / (uint64_t)n_ctx; };
    auto peer_copy = [&](abw_ctx* to, void* dst, abw_ctx* from, const void* src, size_t bytes, hipStream_t st) -> hipError_t {
        if (bytes == 0) return hipSuccess;
        if (to->device == from->device) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
        return hipMemcpyPeerAsync(dst, to->device, src, from->device, bytes, st);
    };
    for (int32_t k = 0; k < n_ctx; k++) {                     // the accumulations have finished
        abw_device_guard g(ctxs[k]->device);
        ABW_HIP(ctxs[k], hipStreamSynchronize(ctxs[k]->stream));
    }
    for (int32_t k = 0; k < n_ctx; k++) {                     // reduce-scatter
        abw_ctx* ck = ctxs[k];
        abw_device_guard g(ck->device);
        const uint64_t h0 = cut(H, k), h1 = cut(H, k + 1), g0 = cut(C, k), g1 = cut(C, k + 1);
        const size_t hb = (size_t)(h1 - h0) * 8, gb = (size_t)(g1 - g0) * 4, hb_pad = (hb + 63) & ~(size_t)63, gb_pad = (gb + 63) & ~(size_t)63;
        int rc;
        if ((rc = abw_reserve(ck, ck->scratch[0], hb_pad + 2 * gb_pad + 64))) return rc;
        int64_t* st_h = (int64_t*)ck->scratch[0].p;
        int32_t* st_lo = (int32_t*)((uint8_t*)ck->scratch[0].p + hb_pad);
        int32_t* st_hi = (int32_t*)((uint8_t*)ck->scratch[0].p + hb_pad + gb_pad);
        int64_t* my_h = (int64_t*)ck->sep.hist.p + h0;
        int32_t* my_lo = (int32_t*)ck->sep.scg_lohi.p + g0;
        int32_t* my_hi = (int32_t*)ck->sep.scg_lohi.p + C + g0;
        for (int32_t i = 0; i < n_ctx; i++) {
            if (i == k) continue;
            abw_ctx* ci = ctxs[i];
            ABW_HIP(ck, peer_copy(ck, st_h, ci, (const int64_t*)ci->sep.hist.p + h0, hb, ck->stream));
            ABW_HIP(ck, peer_copy(ck, st_lo, ci, (const int32_t*)ci->sep.scg_lohi.p + g0, gb, ck->stream));
            ABW_HIP(ck, peer_copy(ck, st_hi, ci, (const int32_t*)ci->sep.scg_lohi.p + C + g0, gb, ck->stream));
            const uint64_t work = std::max<uint64_t>(h1 - h0, g1 - g0);
            if (work > 0) {
                hipLaunchKernelGGL(sep_combine_kernel, dim3((unsigned)std::min<uint64_t>((work + 255) / 256, (uint64_t)ck->num_cu * 8)), dim3(256), 0, ck->stream,
                                   my_h, (const int64_t*)st_h, h1 - h0, my_lo, (const int32_t*)st_lo, my_hi, (const int32_t*)st_hi, g1 - g0);
                ABW_LAUNCH_CHECK(ck);
            }
        }
    }
    for (int32_t k = 0; k < n_ctx; k++) {
        abw_device_guard g(ctxs[k]->device);
        ABW_HIP(ctxs[k], hipStreamSynchronize(ctxs[k]->stream));
    }
    for (int32_t k = 0; k < n_ctx; k++) {                     // all-gather: everybody takes slice k from its owner
        abw_ctx* ck = ctxs[k];
        abw_device_guard g(ck->device);
        const uint64_t h0 = cut(H, k), h1 = cut(H, k + 1), g0 = cut(C, k), g1 = cut(C, k + 1);
        for (int32_t i = 0; i < n_ctx; i++) {
            if (i == k) continue;
            abw_ctx* ci = ctxs[i];
            ABW_HIP(ck, peer_copy(ci, (int64_t*)ci->sep.hist.p + h0, ck, (const int64_t*)ck->sep.hist.p + h0, (size_t)(h1 - h0) * 8, ck->stream));
            ABW_HIP(ck, peer_copy(ci, (int32_t*)ci->sep.scg_lohi.p + g0, ck, (const int32_t*)ck->sep.scg_lohi.p + g0, (size_t)(g1 - g0) * 4, ck->stream));
            ABW_HIP(ck, peer_copy(ci, (int32_t*)ci->sep.scg_lohi.p + C + g0, ck, (const int32_t*)ck->sep.scg_lohi.p + C + g0, (size_t)(g1 - g0) * 4, ck->stream));
        }
    }
    for (int32_t k = 0; k < n_ctx; k++) {
        abw_device_guard g(ctxs[k]->device);
        ABW_HIP(ctxs[k], hipStreamSynchronize(ctxs[k]->stream));
    }
    return ABW_OK;
}

int abw_sep_shard_eval(abw_ctx* ctx, int32_t min_ndps, abw_split* out, abw_dimbest* per_dim,
                       uint8_t* scaf_side, uint8_t* dp_raw_side)
{
    if (!ctx) return ABW_ERR_ARG;
    abw_sep_state& s = ctx->sep;
    if (!s.loaded) return abw_fail(ctx, ABW_ERR_STATE, "abw_sep_shard_eval: call abw_sep_shard_load first");
    if (!out) return abw_fail(ctx, ABW_ERR_ARG, "abw_sep_shard_eval: null out");
    abw_device_guard g(ctx->device);
    return sep_eval_phase(ctx, "abw_sep_shard_eval", min_ndps, out, per_dim, scaf_side, dp_raw_side);
}

}  // extern "C"
