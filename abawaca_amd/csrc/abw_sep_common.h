// Definitions shared by the separator's translation units (abw_sep.hip: one node at a time; abw_tree.hip: the whole
// recursion level by level): histogram layout, candidate records and their order, the flip-value selection.
#pragma once
#include "abw_internal.h"
#include <limits.h>
#include <vector>

constexpr int NARROW_T = 1024;      // dimensions with keys in [0, NARROW_T) use the key itself as bin
constexpr int SCAN_CH = 2048;       // bins per scan chunk
#ifndef ABW_ACC_SPB
#define ABW_ACC_SPB 2048
#endif
constexpr int ACC_SPB = ABW_ACC_SPB;       // scaffolds per workgroup in sep_accum
constexpr int MAX_SCG = 4096;

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
    // abw_sep_tree's static tables, built at its first call on this table (abw_tree.hip): a scaffold's flip value in a
    // dimension depends on the scaffold's own dps only, so it is the same in every node of the recursion
    void* d_ts8 = nullptr;             // abw_sep_tree's static tables (abw_tree.hip): [groups][S][8] x 16 bit flip values,
    void* d_nk16 = nullptr;            //   [groups][N][8] x 16 bit re-coded narrow columns,
    void* d_srec = nullptr;            //   [S] x 16 bytes: first dp, dps, length
    uint32_t* d_wword = nullptr;       // [n_wide][N] per sorted position: scaffold | (v >= t_s) << 31 | (owner) << 30
    uint32_t* d_wext = nullptr;        // [n_wide][S][4] per scaffold: dps below t_s, dps, min(length, 500000), bin of t_s
    uint32_t* d_wpack = nullptr;       // [n_wide][S] the first three of d_wext in one word (scaffolds of <= 127 dps), else all ones
    int32_t* d_carriers = nullptr;     // scaffolds that carry at least one SCG
    int32_t n_carriers = 0;
    bool sharded = false;              // table loaded with a shared bin layout (abw_sep_shard_load)
};


struct abw_sep_extra { SepHost h; };
SepHost& abw_sep_host(abw_ctx* ctx);     // abw_sep.hip

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

// ---- narrow dimensions: histograms of one tile of scaffolds in one dimension --------------------------------------
constexpr int ACC_LDS_WORDS = NARROW_T + 32;     // copies * (T | 1) <= ACC_LDS_WORDS

// One thread per scaffold of the tile [ls_begin, ls_end) (positions in node_scaf, or scaffold ids themselves when it is
// NULL), 256 threads striding over it: the flip value by a sorting network, then per dp size[v] += 1 and
// tp1[max(v, t_s)] += 1 (the dps at or below the flip value are credited to bin t_s in one add), per scaffold
// tot[t_s] += n and bp[t_s] += len -- all into workgroup-private LDS histograms (several copies, see sep_accum_kernel),
// flushed with one atomic per non-empty bin to g_st / g_tot / g_bp (the dimension's slice of the node's histogram).
// scg_lo / scg_hi: the dimension's [n_scg] slice of the node's per-SCG min / max flip values.
static __device__ __forceinline__ void accum_tile(const SepDev& P, int d, const int32_t* __restrict__ node_scaf, int ls_begin, int ls_end,
                                                  unsigned long long* __restrict__ g_st, uint32_t* __restrict__ g_tot,
                                                  unsigned long long* __restrict__ g_bp, int32_t* __restrict__ scg_lo, int32_t* __restrict__ scg_hi,
                                                  uint32_t* s_size, uint32_t* s_tp1, uint32_t* s_tot, unsigned long long* s_bp)
{
    const int T = P.T[d];
    const int32_t* __restrict__ col = P.bins + (int64_t)d * P.N;
    const int Tp = T | 1;
    int copies = 1;
    while (copies < 32 && 2 * copies * Tp <= ACC_LDS_WORDS) copies <<= 1;
    for (int t = threadIdx.x; t < copies * Tp; t += 256) { s_size[t] = 0; s_tp1[t] = 0; s_tot[t] = 0; s_bp[t] = 0; }
    __syncthreads();
    const int mine = (threadIdx.x & (copies - 1)) * Tp;
    uint32_t* my_size = s_size + mine;
    uint32_t* my_tp1 = s_tp1 + mine;
    uint32_t* my_tot = s_tot + mine;
    unsigned long long* my_bp = s_bp + mine;

    // The loads of a scaffold are a dependent chain (list -> scaffold table -> column values); the next scaffold's are
    // issued before this one is processed.
    struct Item { int s, f, n; long long len; int v[8]; };
    auto fetch = [&](int ls, Item& it) {
        it.s = node_scaf ? node_scaf[ls] : ls;
        it.f = P.scaf_first[it.s];
        it.n = P.scaf_ndps[it.s];
        it.len = P.scaf_len[it.s];
#pragma unroll
        for (int i = 0; i < 8; i++) it.v[i] = (i < it.n) ? col[it.f + i] : INT_MAX;      // the first 8 values (all, for n <= 8)
    };
    Item cur, nxt;
    int ls = ls_begin + threadIdx.x;
    if (ls < ls_end) fetch(ls, cur);
    for (; ls < ls_end; ls += 256) {
        if (ls + 256 < ls_end) fetch(ls + 256, nxt);
        const int s = cur.s, f = cur.f, n = cur.n;
        const long long slen = cur.len;
        const int k = n >> 1;                              // 0-based index of the flip value
        int ts;
        uint32_t tp_ts = 0;                                // dps at or below the flip value: tp1 credits bin t_s
        if (n <= 8) {
            int w[8];
#pragma unroll
            for (int i = 0; i < 8; i++) w[i] = cur.v[i];
            ABW_SORT8(w);                                  // the tail is padded with INT_MAX
            ts = (k == 1) ? w[1] : (k == 2) ? w[2] : (k == 3) ? w[3] : w[4];
            // one LDS atomic per DISTINCT value of the scaffold (its segments often share a 3-decimal value): runs of the
            // sorted copy; the padding (INT_MAX) ends the last run
            uint32_t run = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (i < n) {
                    run++;
                    const int nextv = (i < 7) ? w[i + 1] : INT_MAX;
                    if (nextv != w[i]) {
                        atomicAdd(&my_size[w[i]], run);
                        if (w[i] > ts) atomicAdd(&my_tp1[w[i]], run); else tp_ts += run;
                        run = 0;
                    }
                }
            }
        } else {
            ts = select_kth_mem(col + f, n, k, T);
            for (int i = 0; i < n; i++) {
                const int v = col[f + i];
                atomicAdd(&my_size[v], 1u);
                if (v > ts) atomicAdd(&my_tp1[v], 1u); else tp_ts++;
            }
        }
        atomicAdd(&my_tp1[ts], tp_ts);
        atomicAdd(&my_tot[ts], (uint32_t)n);
        atomicAdd(&my_bp[ts], (unsigned long long)slen);
        if (P.n_scg > 0) {
            for (int e = P.scg_off[s]; e < P.scg_off[s + 1]; e++) {
                const int g = P.scg_ids[e];
                atomicMin(&scg_lo[g], ts);
                atomicMax(&scg_hi[g], ts);
            }
        }
        cur = nxt;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 256) {
        unsigned long long a = 0, e = 0;
        uint32_t c3 = 0;
        for (int c = 0; c < copies; c++) {
            const int at = c * Tp + t;
            a += (unsigned long long)s_size[at] | ((unsigned long long)s_tp1[at] << 32);
            c3 += s_tot[at];
            e += s_bp[at];
        }
        if (a) atomicAdd(&g_st[t], a);
        if (c3) atomicAdd(&g_tot[t], c3);
        if (e) atomicAdd(&g_bp[t], e);
    }
}
