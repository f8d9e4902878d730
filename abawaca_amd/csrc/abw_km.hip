// Extension — batched k-means restarts (k = 2) over the dp feature table.
//
// The reference contains NO k-means; this stage is the north-star's extension and
// its semantics are fixed by DESIGN.md "K-means" and restated on the CPU in oracle/abw_oracle.c
// (abw_or_kmeans).  Only the RNG has a reference counterpart: TinyMT32 with the parameter triple of
// reference src/misc.cpp:29-31 (tinymt32.c:62-76, tinymt32.h:70-107).
//
// Kernel design (gfx950): both halves of a Lloyd iteration are contractions and run on the f32
// MFMA (v_mfma_f32_32x32x2_f32: exact fp32, result == a k-ordered fmaf chain, so the CPU oracle
// reproduces every bit):
//   scores   X[p][rc] = -0.5|c_rc|^2 + sum_j x[p][j] c[rc][j]          (points x (restarts*2))
//   sums     S[rc][j] = sum_p member[p][rc] x[p][j]                    ((restarts*2) x dims)
// One workgroup = 4 waves = 64 restarts; a wave owns 32 score columns (16 restarts x 2 clusters).
// A 32-point tile of x is staged once in LDS and feeds both products; the score accumulator of the
// first product IS the A operand of the second (column on the lane, rows in registers), so
// memberships never leave registers.  fp32 partial sums run over one chunk of points in point
// order, one slab per chunk; slabs are added in fp64 in chunk order by km_update_kernel.
#include "abw_internal.h"
#include <type_traits>
#include <string.h>
#include <vector>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr uint32_t TMT_MAT1 = 0x8f7011eeu, TMT_MAT2 = 0xfc78ff1fu, TMT_TMAT = 0x3793fdffu;   // ref misc.cpp:29-31

struct TinyMT { uint32_t s[4]; };

__host__ __device__ inline void tmt_next_state(TinyMT& t)          // ref tinymt32.h:70-86
{
    uint32_t y = t.s[3];
    uint32_t x = (t.s[0] & 0x7fffffffu) ^ t.s[1] ^ t.s[2];
    x ^= (x << 1);
    y ^= (y >> 1) ^ x;
    t.s[0] = t.s[1];
    t.s[1] = t.s[2];
    t.s[2] = x ^ (y << 10);
    t.s[3] = y;
    const uint32_t mask = (uint32_t)(-(int32_t)(y & 1));
    t.s[1] ^= mask & TMT_MAT1;
    t.s[2] ^= mask & TMT_MAT2;
}

__host__ __device__ inline void tmt_init(TinyMT& t, uint32_t seed)  // ref tinymt32.c:62-76
{
    t.s[0] = seed; t.s[1] = TMT_MAT1; t.s[2] = TMT_MAT2; t.s[3] = TMT_TMAT;
    for (uint32_t i = 1; i < 8; i++) t.s[i & 3] ^= i + 1812433253u * (t.s[(i - 1) & 3] ^ (t.s[(i - 1) & 3] >> 30));
    if ((t.s[0] & 0x7fffffffu) == 0 && t.s[1] == 0 && t.s[2] == 0 && t.s[3] == 0) { t.s[0] = 'T'; t.s[1] = 'I'; t.s[2] = 'N'; t.s[3] = 'Y'; }
    for (int i = 0; i < 8; i++) tmt_next_state(t);
}

__host__ __device__ inline uint32_t tmt_next(TinyMT& t)             // ref tinymt32.h:94-107,168-171
{
    tmt_next_state(t);
    uint32_t t0 = t.s[3];
    const uint32_t t1 = t.s[0] + (t.s[2] >> 8);
    t0 ^= t1;
    t0 ^= (uint32_t)(-(int32_t)(t1 & 1)) & TMT_TMAT;
    return t0;
}

__global__ void tmt_selftest_kernel(uint32_t seed, int n, uint32_t* out)
{
    if (blockIdx.x || threadIdx.x) return;
    TinyMT t;
    tmt_init(t, seed);
    for (int i = 0; i < n; i++) out[i] = tmt_next(t);
}

// centroid buffers: cent [Rp*2][W] fp32 (W = DT*32, zero beyond d), hneg [Rp*2]
__global__ void km_init_kernel(const float* __restrict__ x, int64_t N, int d, int W, int R, int Rp, uint32_t base_seed,
                               float* __restrict__ cent, float* __restrict__ hneg)
{
    const int r = blockIdx.x;
    __shared__ int64_t idx[2];
    if (threadIdx.x == 0) {
        int64_t i0 = 0, i1 = 1 % N;
        if (r < R) {
            TinyMT t;
            tmt_init(t, base_seed + (uint32_t)r);
            i0 = (int64_t)(tmt_next(t) % (uint64_t)N);
            i1 = i0;
            for (int tries = 0; tries < 64 && i1 == i0; tries++) i1 = (int64_t)(tmt_next(t) % (uint64_t)N);
        }
        idx[0] = i0; idx[1] = i1;
    }
    __syncthreads();
    for (int c = 0; c < 2; c++)
        for (int j = threadIdx.x; j < W; j += blockDim.x)
            cent[((int64_t)r * 2 + c) * W + j] = (j < d) ? x[(int64_t)j * N + idx[c]] : 0.0f;
    __syncthreads();
    if (threadIdx.x < 2) {
        const float* c = cent + ((int64_t)r * 2 + threadIdx.x) * W;
        float n2 = 0.0f;
        for (int j = 0; j < d; j++) n2 = fmaf(c[j], c[j], n2);
        hneg[r * 2 + threadIdx.x] = -(0.5f * n2);
    }
}

constexpr int XS = 33;   // LDS row stride (floats) of the staged x tile: odd -> conflict-free both ways

// row i of the score tile holds the point whose position in the fp32 summation order is rowpoint(i)
__device__ __forceinline__ int rowpoint(int i) { return 2 * ((i & 3) + 4 * (i >> 3)) + ((i >> 2) & 1); }

// lane l <-> lane l^1 exchange inside a quad (DPP quad_perm [1,0,3,2]); no LDS crossbar involved
// Score-product MFMAs with the accumulator pinned to VGPRs: its consumer is VALU code (membership),
// and left to itself the allocator parks it in the AGPRs of the sums accumulators, paying 64 copies
// per tile.  Inline asm is invisible to the hazard recogniser: mfma_v_done() supplies the 18 wait
// states a 16-pass MFMA result needs before VALU may read it.
__device__ __forceinline__ void mfma_v_first(f32x16& c, float a, float b)
{
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_v(f32x16& c, float a, float b)
{
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_v_done(f32x16& c0, f32x16& c1)
{
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(c0), "+v"(c1));
}

__device__ __forceinline__ float swap_pair(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}

#ifdef ABW_KM_TIMING
__device__ unsigned long long km_tim[10];
#define KM_T(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define KM_T(i) do { } while (0)
#endif
template <int DT, bool ASSIGN>
__global__ __launch_bounds__(256) void km_step_kernel(const float* __restrict__ x, int64_t N, int d,
                                                     const float* __restrict__ cent, const float* __restrict__ hneg,
                                                     int Rp, int chunk, int64_t nchunks,
                                                     float* __restrict__ slabs, uint8_t* __restrict__ assign, int R)
{
    constexpr int W = DT * 32;
    constexpr int NBUF = (DT <= 6) ? 2 : 1;                  // x-tile buffers that fit beside the centroids in 160 KiB
    constexpr int NS = 16 * DT;                              // k-steps of the score product
    // centroid operand of the score product, [wave][j][col]: lanes 0-31 read one 128-B row, conflict-free.
    // Row d holds -0.5|c|^2: it meets the all-ones row d of the x tile, so the score chain ends with it.
    __shared__ float cs[4 * W * 32];
    __shared__ float xs[NBUF][W * XS];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int rc0 = blockIdx.y * 128 + wid * 32;             // first score column of this wave
    // a wave past the last column (only when Rp*2 is not a multiple of 128) recomputes the last
    // tile of columns and stores nothing: keeps every wave on one straight-line MFMA stream
    const bool wave_active = rc0 < Rp * 2;
    const int rc0c = wave_active ? rc0 : Rp * 2 - 32;
    const int rc = rc0c + col;
    {
        float* mine = cs + wid * W * 32;
        for (int i = lane; i < W * 32; i += 64) {
            const int j = i >> 5, c = i & 31;
            mine[i] = (j == d) ? hneg[rc0c + c] : cent[(int64_t)(rc0c + c) * W + j];
        }
    }
    const float* bop = cs + wid * W * 32 + half * 32 + col;  // + 64*s for k-step s
    const int rp = rowpoint(col);
    const bool is_c1 = col & 1;
    const float thr = is_c1 ? 0.0f : __int_as_float(0x80000001);
    const int sp = threadIdx.x & 31;                         // staging: this thread's point within the tile
    const int sj = threadIdx.x >> 5;                         //          and its first row
    // rows >= d of both buffers never change: row d all ones (count column / meets -0.5|c|^2), the rest zero
    for (int b = 0; b < NBUF; b++)
        for (int i = threadIdx.x; i < (W - d) * 32; i += 256)
            xs[b][(d + (i >> 5)) * XS + (i & 31)] = (i < 32) ? 1.0f : 0.0f;

#ifdef ABW_KM_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long r0_ = __builtin_amdgcn_s_memrealtime();
#endif
    // Staging of the next tile: thread (sp, sj) moves rows sj, sj+8, ... of point sp.  The global loads
    // and LDS stores are issued a few at a time between the MFMA batches of the current tile (all four
    // waves share one texture-address unit: 96 loads issued back to back stall the waves ~1400 cycles).
    float stg[DT * 4];
    const float* ldp = x;                                    // set per tile: x + clamped point
    auto load_k = [&](int k) { stg[k] = ldp[(int64_t)min(sj + 8 * k, d - 1) * N]; };   // unconditional (clamped row)
    // Rows >= d must keep their constants, so a store aimed at one is bent into the padding column
    // (32) of this thread's last real row instead of being branched around: one v_min per store.
    const int trash = 8 * XS * max(0, (d - 1 - sj) >> 3) + 32 - sp;
    const int first = (sj < d) ? 0 : trash;
    float* stq = xs[0] + sj * XS + sp;                       // set per tile: destination buffer
    auto store_k = [&](int k) { stq[k == 0 ? first : min(8 * k * XS, trash)] = stg[k]; };

    for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const int64_t p_begin = ch * chunk;
        const int64_t p_end = min(N, p_begin + chunk);
        int cur = 0;
        KM_T(0);
        __syncthreads();                                      // previous chunk's last tile fully consumed
        ldp = x + min(p_begin + sp, N - 1);                   // a point past N re-reads the last one; its membership is 0
        stq = xs[0] + sj * XS + sp;
#pragma unroll
        for (int k = 0; k < DT * 4; k++) load_k(k);
#pragma unroll
        for (int k = 0; k < DT * 4; k++) store_k(k);
        __syncthreads();
        f32x16 acc2[DT];
#pragma unroll
        for (int t = 0; t < DT; t++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc2[t][e] = 0.0f;
        KM_T(1);

        for (int64_t tile = p_begin; tile < p_end; tile += 32) {
            const bool have_next = tile + 32 < p_end;
            // two buffers: the next tile is fetched and stored while this one is computed; on the last
            // tile of a chunk the same code re-fetches the current tile into the idle buffer (harmless)
            ldp = x + min((have_next ? tile + 32 : tile) + sp, N - 1);
            stq = xs[cur ^ 1] + sj * XS + sp;
            const float* xt = xs[cur];
            const float* sop = xt + col * XS + half;         // sums operand: + 32*XS*t + 2*e
            float bb[2][DT];
            KM_T(2);
            // ---- scores: X[p][rc] = sum_j x[p][j] c[rc][j] - 0.5|c|^2; operands read 8 k-steps ahead -------
            // Two chains: k-steps alternate between them, so that no MFMA waits for the one before it
            // (a dependent MFMA that does not follow its producer back to back issues ~13 cycles late).
            f32x16 acc1, acc1b;
            {
                // One wave per SIMD: whatever is issued between two MFMAs has to fit into the 64 cycles
                // the first one runs, so the next batch's LDS reads and the staging loads are dealt out
                // one or two per MFMA instead of in a block (a block of ~16 costs a bubble per batch).
                const float* aop = xt + half * XS + rp;      // + 2*XS*s for k-step s
                float a[2][8], b[2][8];
#pragma unroll
                for (int u = 0; u < 8; u++) { a[0][u] = aop[2 * XS * u]; b[0][u] = bop[64 * u]; }
#pragma unroll
                for (int bt = 0; bt < NS / 8; bt++) {
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int s1 = 8 * (bt + 1) + (u & ~1);          // k-step pair fetched behind this MFMA
                        if (bt + 1 < NS / 8) {
                            if (!(u & 1)) { a[(bt + 1) & 1][u] = aop[2 * XS * s1]; a[(bt + 1) & 1][u + 1] = aop[2 * XS * (s1 + 1)]; }
                            else          { b[(bt + 1) & 1][u - 1] = bop[64 * s1]; b[(bt + 1) & 1][u] = bop[64 * (s1 + 1)]; }
                        } else if (u < DT) {
                            bb[0][u] = sop[32 * XS * u];                 // first operands of the sums product
                        }
                        if (NBUF == 2 && u == 2) load_k(2 * bt);
                        if (NBUF == 2 && u == 6) load_k(2 * bt + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        if (bt == 0 && u < 2) mfma_v_first((u & 1) ? acc1b : acc1, a[0][u], b[0][u]);
                        else                  mfma_v((u & 1) ? acc1b : acc1, a[bt & 1][u], b[bt & 1][u]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            // ---- memberships (tie -> cluster 0), straight from the accumulator layout:
            //      register e of lane (col, half) is point q = 2e + half of the tile.
            //      score = chain0 + chain1.  cluster-1 column: member iff mine > other; cluster-0 column:
            //      member iff mine >= other.  fp32 subtraction (denormals on) is zero only for equal
            //      operands, so both read "mine - other > thr" with thr = 0 resp. the largest negative number.
            // ---- sums: S[rc][j] += sum_p mem[p][rc] x[p][j]; operands read one point pair ahead; on the
            //      common path membership e+1 is computed in the shadow of the MFMAs of point pair e.
            KM_T(3);
            const int nv = (int)min((int64_t)32, N - tile);                       // real points in this tile
            float mem[16], sc = 0.0f, df = 0.0f;
            auto member = [&](int e) {                                           // from df of point pair e
                const bool valid = 2 * e + half < nv;
                if (ASSIGN) {
                    // a cluster-0 lane holds score0 - score1: cluster 1 wins iff that is negative
                    if (wave_active && !is_c1 && valid && (rc >> 1) < R)
                        assign[(int64_t)(rc >> 1) * N + tile + 2 * e + half] = (df < 0.0f) ? 1 : 0;
                }
                return (valid && df > thr) ? 1.0f : 0.0f;
            };
            mfma_v_done(acc1, acc1b);
            sc = acc1[0] + acc1b[0];
            df = sc - swap_pair(sc);
            mem[0] = member(0);
            KM_T(4);
#pragma unroll
            for (int e = 0; e < 16; e++) {
#pragma unroll
                for (int t = 0; t < DT; t++) {
                    if (e + 1 < 16) bb[(e + 1) & 1][t] = sop[32 * XS * t + 2 * (e + 1)];
                    if (NBUF == 2 && t == (DT > 1 ? 1 : 0) && 2 * e < DT * 4) store_k(2 * e);
                    if (NBUF == 2 && t == (DT > 3 ? 3 : DT - 1) && 2 * e + 1 < DT * 4) store_k(2 * e + 1);
                    if (e + 1 < 16) {
                        if (t == 0) sc = acc1[e + 1] + acc1b[e + 1];
                        if (t == (DT >= 5 ? 2 : 0)) df = sc - swap_pair(sc);
                        if (t == (DT >= 5 ? 4 : 0)) mem[e + 1] = member(e + 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(mem[e], bb[e & 1][t], acc2[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            KM_T(5);
            if (NBUF == 2) {
                __syncthreads();                             // next tile stored by all; this one read by all
                cur ^= 1;
            } else if (have_next) {
                __syncthreads();                             // everyone done reading the only buffer
                ldp = x + min(tile + 32 + sp, N - 1);
                stq = xs[0] + sj * XS + sp;
#pragma unroll
                for (int k = 0; k < DT * 4; k++) load_k(k);
#pragma unroll
                for (int k = 0; k < DT * 4; k++) store_k(k);
                __syncthreads();
            }
            KM_T(6);
        }
        if (wave_active) {
            // slab[ch][rc_row][j]: C/D layout col = lane&31 -> j, row = (e&3) + 8(e>>2) + 4*half -> rc
            float* slab = slabs + ((int64_t)ch * Rp * 2 + rc0) * W;
#pragma unroll
            for (int t = 0; t < DT; t++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
                    slab[(int64_t)row * W + t * 32 + col] = acc2[t][e];
                }
        }
    }
#ifdef ABW_KM_TIMING
    KM_T(7);
    if (threadIdx.x == 0 && blockIdx.x == 7)
    {   for (int i = 0; i < 8; i++) atomicAdd(&km_tim[i], tacc[i]);
        atomicAdd(&km_tim[8], __builtin_amdgcn_s_memrealtime() - r0_); }
#endif
}
#ifdef ABW_KM_TIMING
extern "C" void abw_dbg_km_timing(unsigned long long* out)
{
    unsigned long long z[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    hipMemcpyFromSymbol(out, HIP_SYMBOL(km_tim), sizeof(z));
    hipMemcpyToSymbol(HIP_SYMBOL(km_tim), z, sizeof(z));
}
#endif

constexpr int KM_GROUP = 32;   // chunk slabs per fp64 group partial

// Level 1 of the slab reduction: group g adds its KM_GROUP consecutive chunk slabs in fp64 in chunk
// order.  Elementwise over [Rp*2][W]; the loads of one thread are independent, only its adds chain.
__global__ __launch_bounds__(256) void km_group_kernel(const float* __restrict__ slabs, int64_t nchunks, int64_t elems,
                                                      double* __restrict__ gsum)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= elems) return;
    const int64_t g = blockIdx.y;
    const int64_t c0 = g * KM_GROUP;
    const int n = (int)min((int64_t)KM_GROUP, nchunks - c0);
    const float* p = slabs + c0 * elems + e;
    float v[KM_GROUP];
#pragma unroll
    for (int c = 0; c < KM_GROUP; c++) v[c] = (c < n) ? p[(int64_t)c * elems] : 0.0f;
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < KM_GROUP; c++) if (c < n) s += (double)v[c];
    gsum[g * elems + e] = s;
}

// Level 2: one workgroup per score column rc adds the group partials in group order, then either
// moves the centroid (update = 1) or contributes the column's part of the inertia (update = 0).
__global__ void km_update_kernel(const double* __restrict__ gsum, int64_t ngroups, int Rp, int W, int d,
                                 float* __restrict__ cent, float* __restrict__ hneg, int update,
                                 double* __restrict__ col_term)
{
    extern __shared__ double sh[];            // [W] sums, then [W] floats of the centroid
    const int rc = blockIdx.x;
    float* cs = (float*)(sh + W);
    for (int j = threadIdx.x; j < W; j += blockDim.x) {
        double s = 0.0;
        const double* p = gsum + (int64_t)rc * W + j;
        const int64_t stride = (int64_t)Rp * 2 * W;
        for (int64_t c = 0; c < ngroups; c++) s += p[c * stride];
        sh[j] = s;
    }
    __syncthreads();
    const double cnt = sh[d];
    for (int j = threadIdx.x; j < W; j += blockDim.x) {
        float c = cent[(int64_t)rc * W + j];
        if (update && j < d && cnt > 0.0) { c = (float)(sh[j] / cnt); cent[(int64_t)rc * W + j] = c; }
        cs[j] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (update) {
            float n2 = 0.0f;
            for (int j = 0; j < d; j++) n2 = fmaf(cs[j], cs[j], n2);
            hneg[rc] = -(0.5f * n2);
        } else {
            // sum_{p in cluster} |x_p - c|^2 = [sum |x_p|^2] - 2 c.S + n |c|^2 ; the bracket is added per restart
            double dot = 0.0, n2 = 0.0;
            for (int j = 0; j < d; j++) { dot += (double)cs[j] * sh[j]; n2 += (double)cs[j] * (double)cs[j]; }
            col_term[rc] = cnt * n2 - 2.0 * dot;
        }
    }
}

// sum_p x[j][p]^2 per dimension: 256 strided fp64 partials, then a fixed butterfly
__global__ __launch_bounds__(256) void km_x2_kernel(const float* __restrict__ x, int64_t N, double* __restrict__ x2_dim)
{
    __shared__ double w[4];
    const int j = blockIdx.x;
    const float* row = x + (int64_t)j * N;
    double acc = 0.0;
    int64_t p = threadIdx.x;
    for (; p + 7 * 256 < N; p += 8 * 256) {                  // 8 loads in flight; the adds keep their order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = row[p + u * 256];
#pragma unroll
        for (int u = 0; u < 8; u++) { const double w2 = (double)v[u]; acc += w2 * w2; }
    }
    for (; p < N; p += 256) { const double v = (double)row[p]; acc += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) x2_dim[j] = (w[0] + w[1]) + (w[2] + w[3]);
}

__global__ void km_inertia_kernel(const double* __restrict__ x2_dim, int d, const double* __restrict__ col_term, int R,
                                  double* __restrict__ inertia)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    double x2 = 0.0;
    for (int j = 0; j < d; j++) x2 += x2_dim[j];
    inertia[r] = (x2 + col_term[2 * r]) + col_term[2 * r + 1];
}

__global__ void km_copy_centroids_kernel(const float* __restrict__ cent, int W, int d, int R, float* __restrict__ out)
{
    const int64_t n = (int64_t)R * 2 * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t rc = i / d; const int j = (int)(i % d);
        out[i] = cent[rc * W + j];
    }
}

__global__ void km_features_kernel(const int32_t* __restrict__ keys, int64_t n, int64_t N, const float* __restrict__ scale,
                                   float* __restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float s = scale ? scale[i / N] : 1.0f;
        out[i] = (float)keys[i] * s;
    }
}

template <int DT>
void launch_step(abw_ctx* ctx, dim3 grid, const float* x, int64_t N, int d, const float* cent, const float* hneg, int Rp,
                 int chunk, int64_t nchunks, float* slabs, uint8_t* assign, int R)
{
    if (assign)
        hipLaunchKernelGGL((km_step_kernel<DT, true>), grid, dim3(256), 0, ctx->stream, x, N, d, cent, hneg, Rp, chunk, nchunks, slabs, assign, R);
    else
        hipLaunchKernelGGL((km_step_kernel<DT, false>), grid, dim3(256), 0, ctx->stream, x, N, d, cent, hneg, Rp, chunk, nchunks, slabs, assign, R);
}

}  // namespace

extern "C" {

int abw_km_tinymt_selftest(abw_ctx* ctx, uint32_t seed, int32_t n, uint32_t* out)
{
    if (!ctx || !out || n <= 0) return ABW_ERR_ARG;
    abw_device_guard g(ctx->device);
    int rc = abw_reserve(ctx, ctx->scratch[0], (size_t)n * 4);
    if (rc) return rc;
    hipLaunchKernelGGL(tmt_selftest_kernel, dim3(1), dim3(64), 0, ctx->stream, seed, n, (uint32_t*)ctx->scratch[0].p);
    ABW_LAUNCH_CHECK(ctx);
    ABW_HIP(ctx, hipMemcpyAsync(out, ctx->scratch[0].p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    ABW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ABW_OK;
}

int abw_km_features_from_keys(abw_ctx* ctx, const int32_t* d_keys, int32_t D, int64_t N, const float* scale, float* d_x)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!d_keys || !d_x || D <= 0 || N <= 0) return abw_fail(ctx, ABW_ERR_ARG, "abw_km_features_from_keys: bad argument");
    abw_device_guard g(ctx->device);
    float* d_scale = nullptr;
    if (scale) {
        int rc = abw_reserve(ctx, ctx->scratch[0], (size_t)D * 4);
        if (rc) return rc;
        d_scale = (float*)ctx->scratch[0].p;
        ABW_HIP(ctx, hipMemcpyAsync(d_scale, scale, (size_t)D * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    hipLaunchKernelGGL(km_features_kernel, dim3(ctx->num_cu * 8), dim3(256), 0, ctx->stream, d_keys, (int64_t)D * N, N, d_scale, d_x);
    ABW_LAUNCH_CHECK(ctx);
    if (scale) ABW_HIP(ctx, hipStreamSynchronize(ctx->stream));   // `scale` is the caller's memory
    return ABW_OK;
}

int abw_km_run_dev(abw_ctx* ctx, const float* d_x, int64_t N, int32_t d, int32_t R, uint32_t base_seed,
                   int32_t iters, int32_t chunk, uint8_t* d_assign, float* d_centroids, double* d_inertia)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!d_x || N < 2 || d < 1 || R < 1 || iters < 0) return abw_fail(ctx, ABW_ERR_ARG, "abw_km_run_dev: bad argument");
    if (chunk == 0) chunk = 512;
    if (chunk < 32 || (chunk & 31)) return abw_fail(ctx, ABW_ERR_ARG, "abw_km_run_dev: chunk must be a positive multiple of 32");
    const int DT = (d + 1 + 31) / 32;                       // dims + the count column, in 32-wide tiles
    if (DT > 7) return abw_fail(ctx, ABW_ERR_ARG, "abw_km_run_dev: d = %d not supported (max 222)", d);
    abw_device_guard g(ctx->device);
    const int W = DT * 32;
    const int Rp = (R + 15) & ~15;
    const int64_t nchunks = (N + chunk - 1) / chunk;
    int rc;
    // scratch[1]: centroids + hneg + col_term + x2 ; scratch[2]: slabs
    const size_t cent_bytes = (size_t)Rp * 2 * W * 4, hneg_bytes = (size_t)Rp * 2 * 4;
    const size_t term_bytes = (size_t)Rp * 2 * 8, x2_bytes = (size_t)d * 8;
    if ((rc = abw_reserve(ctx, ctx->scratch[1], cent_bytes + hneg_bytes + term_bytes + x2_bytes + 64))) return rc;
    const int64_t ngroups = (nchunks + KM_GROUP - 1) / KM_GROUP;
    const int64_t elems = (int64_t)Rp * 2 * W;
    if ((rc = abw_reserve(ctx, ctx->scratch[2], (size_t)nchunks * elems * 4))) return rc;
    if ((rc = abw_reserve(ctx, ctx->scratch[5], (size_t)ngroups * elems * 8))) return rc;
    double* gsum = (double*)ctx->scratch[5].p;
    uint8_t* base = (uint8_t*)ctx->scratch[1].p;
    double* col_term = (double*)base;
    double* x2_dim = (double*)(base + term_bytes);
    float* cent = (float*)(base + term_bytes + x2_bytes);
    float* hneg = (float*)(base + term_bytes + x2_bytes + cent_bytes);
    float* slabs = (float*)ctx->scratch[2].p;
    hipStream_t st = ctx->stream;

    hipLaunchKernelGGL(km_init_kernel, dim3(Rp), dim3(256), 0, st, d_x, N, d, W, R, Rp, base_seed, cent, hneg);
    ABW_LAUNCH_CHECK(ctx);
    const unsigned gy = (unsigned)((Rp * 2 + 127) / 128);
    unsigned gx = (unsigned)std::min<int64_t>(nchunks, (int64_t)ctx->num_cu * 8);
    const dim3 grid(gx, gy);
    const size_t upd_lds = (size_t)W * 8 + (size_t)W * 4;
    for (int it = 0; it <= iters; it++) {
        const bool last = it == iters;
        uint8_t* asg = last ? d_assign : nullptr;
        {
        ABW_PROF(ctx, "km_step");
        switch (DT) {
            case 1: launch_step<1>(ctx, grid, d_x, N, d, cent, hneg, Rp, chunk, nchunks, slabs, asg, R); break;
            case 2: launch_step<2>(ctx, grid, d_x, N, d, cent, hneg, Rp, chunk, nchunks, slabs, asg, R); break;
            case 3: launch_step<3>(ctx, grid, d_x, N, d, cent, hneg, Rp, chunk, nchunks, slabs, asg, R); break;
            case 4: launch_step<4>(ctx, grid, d_x, N, d, cent, hneg, Rp, chunk, nchunks, slabs, asg, R); break;
            case 5: launch_step<5>(ctx, grid, d_x, N, d, cent, hneg, Rp, chunk, nchunks, slabs, asg, R); break;
            case 6: launch_step<6>(ctx, grid, d_x, N, d, cent, hneg, Rp, chunk, nchunks, slabs, asg, R); break;
            default: launch_step<7>(ctx, grid, d_x, N, d, cent, hneg, Rp, chunk, nchunks, slabs, asg, R); break;
        }
        }
        ABW_LAUNCH_CHECK(ctx);
        if (last && !d_inertia) break;
        ABW_PROF(ctx, "km_update");
        hipLaunchKernelGGL(km_group_kernel, dim3((unsigned)((elems + 255) / 256), (unsigned)ngroups), dim3(256), 0, st, slabs, nchunks, elems, gsum);
        ABW_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(km_update_kernel, dim3(Rp * 2), dim3(W), upd_lds, st, gsum, ngroups, Rp, W, d, cent, hneg, last ? 0 : 1, col_term);
        ABW_LAUNCH_CHECK(ctx);
    }
    if (d_inertia) {
        hipLaunchKernelGGL(km_x2_kernel, dim3(d), dim3(256), 0, st, d_x, N, x2_dim);
        ABW_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(km_inertia_kernel, dim3((R + 63) / 64), dim3(64), 0, st, x2_dim, d, col_term, R, d_inertia);
        ABW_LAUNCH_CHECK(ctx);
    }
    if (d_centroids) {
        hipLaunchKernelGGL(km_copy_centroids_kernel, dim3(64), dim3(256), 0, st, cent, W, d, R, d_centroids);
        ABW_LAUNCH_CHECK(ctx);
    }
    return ABW_OK;
}

int abw_km_run(abw_ctx* ctx, const float* x, int64_t N, int32_t d, int32_t R, uint32_t base_seed,
               int32_t iters, int32_t chunk, uint8_t* assign, float* centroids, double* inertia)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!x || N < 2 || d < 1 || R < 1) return abw_fail(ctx, ABW_ERR_ARG, "abw_km_run: bad argument");
    abw_device_guard g(ctx->device);
    int rc;
    const size_t xb = (size_t)N * d * 4, ab = (size_t)R * N, cb = (size_t)R * 2 * d * 4, ib = (size_t)R * 8;
    if ((rc = abw_reserve(ctx, ctx->scratch[3], xb))) return rc;
    if ((rc = abw_reserve(ctx, ctx->scratch[4], ab + cb + ib + 64))) return rc;
    float* d_x = (float*)ctx->scratch[3].p;
    double* d_in = (double*)ctx->scratch[4].p;
    float* d_c = (float*)((uint8_t*)d_in + ib);
    uint8_t* d_a = (uint8_t*)d_c + cb;
    ABW_HIP(ctx, hipMemcpyAsync(d_x, x, xb, hipMemcpyHostToDevice, ctx->stream));
    rc = abw_km_run_dev(ctx, d_x, N, d, R, base_seed, iters, chunk, assign ? d_a : nullptr, centroids ? d_c : nullptr, inertia ? d_in : nullptr);
    if (rc) return rc;
    if (assign) ABW_HIP(ctx, hipMemcpyAsync(assign, d_a, ab, hipMemcpyDeviceToHost, ctx->stream));
    if (centroids) ABW_HIP(ctx, hipMemcpyAsync(centroids, d_c, cb, hipMemcpyDeviceToHost, ctx->stream));
    if (inertia) ABW_HIP(ctx, hipMemcpyAsync(inertia, d_in, ib, hipMemcpyDeviceToHost, ctx->stream));
    ABW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ABW_OK;
}

}  // extern "C"
