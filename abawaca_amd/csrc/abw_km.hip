// Extension — batched k-means restarts (k = 2) over the dp feature table.
//
// The reference contains NO k-means; this stage is the north-star's extension and
// its semantics are fixed by DESIGN.md "K-means" and restated on the CPU in oracle/abw_oracle.c
// (abw_or_kmeans).  Only the RNG has a reference counterpart: TinyMT32 with the parameter triple of
// reference src/misc.cpp:29-31 (tinymt32.c:62-76, tinymt32.h:70-107).
//
// Kernel design (gfx950).  With k = 2 a Lloyd iteration needs, per restart, ONE dot product per point
// and the sums of ONE cluster:
//   discriminant  g[p][r] = sum_j x[p][j] D[r][j] + e[r],  D = c1 - c0,  e = 0.5|c0|^2 - 0.5|c1|^2
//                 point p joins cluster 1 iff g > 0 (tie -> cluster 0)
//   sums          S1[r][j] = sum_p [g[p][r] > 0] x[p][j]  (j = d: the count);  S0 = T - S1 with the
//                 restart-independent totals T[j] = sum_p x[p][j] computed once per call.
// Both are contractions and run on the f32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32, result == a
// k-ordered fmaf chain, so the CPU oracle reproduces every bit).  A wave owns 32 restarts; a
// workgroup of 4 waves is CW column tiles x PT = 4/CW point lanes, each point lane walking its own
// chunk of points.  A 32-point tile of x is staged once in LDS and feeds both products; the
// accumulator of the first product IS the A operand of the second (restart on the lane, points in
// registers), so memberships never leave registers.  fp32 partial sums run over one chunk of points
// in point order, one slab per chunk; slabs are added in fp64 in chunk order (km_group / km_update).
#include "abw_internal.h"
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte access at 4-byte alignment
typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1))); // 16-byte access at any alignment

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

// Buffers: cent [Rp*2][W] fp32 centroids (W = DT*32, zero beyond d); dT [W][Rp] fp32, the
// discriminant operand, dimension-major: row j < d holds D[.][j], row d holds e[.], rows > d zero.
__device__ inline void km_discriminant(const float* c0, const float* c1, int d, int W, int Rp, int r, float* dT, int tid, int nthr)
{
    for (int j = tid; j < W; j += nthr)
        if (j != d) dT[(int64_t)j * Rp + r] = (j < d) ? c1[j] - c0[j] : 0.0f;
    if (tid == 0) {
        float n0 = 0.0f, n1 = 0.0f;
        for (int j = 0; j < d; j++) { n0 = fmaf(c0[j], c0[j], n0); n1 = fmaf(c1[j], c1[j], n1); }
        dT[(int64_t)d * Rp + r] = 0.5f * n0 - 0.5f * n1;
    }
}

// x holds points [first, first + N) of N_total; a centroid whose seed point lives in another shard is
// left all-zero bits here (the shards' buffers are combined by integer addition, see abw_km_shard_begin)
__global__ void km_init_kernel(const float* __restrict__ x, int64_t N, int64_t first, int64_t N_total, int d, int W, int R,
                               uint32_t base_seed, float* __restrict__ cent)
{
    const int r = blockIdx.x;
    __shared__ int64_t idx[2];
    if (threadIdx.x == 0) {
        int64_t i0 = 0, i1 = 1 % N_total;
        if (r < R) {
            TinyMT t;
            tmt_init(t, base_seed + (uint32_t)r);
            i0 = (int64_t)(tmt_next(t) % (uint64_t)N_total);
            i1 = i0;
            for (int tries = 0; tries < 64 && i1 == i0; tries++) i1 = (int64_t)(tmt_next(t) % (uint64_t)N_total);
        }
        idx[0] = i0 - first; idx[1] = i1 - first;
    }
    __syncthreads();
    for (int c = 0; c < 2; c++) {
        const bool mine = idx[c] >= 0 && idx[c] < N;
        for (int j = threadIdx.x; j < W; j += blockDim.x)
            cent[((int64_t)r * 2 + c) * W + j] = (mine && j < d) ? x[(int64_t)j * N + idx[c]] : 0.0f;
    }
}

__global__ void km_discriminant_kernel(const float* __restrict__ cent, int d, int W, int Rp, float* __restrict__ dT)
{
    const int r = blockIdx.x;
    km_discriminant(cent + (int64_t)r * 2 * W, cent + ((int64_t)r * 2 + 1) * W, d, W, Rp, r, dT, threadIdx.x, blockDim.x);
}

constexpr int XS = 33;   // LDS row stride (floats) of the staged x tile: odd -> conflict-free both ways

// row i of the discriminant tile holds the point whose position in the fp32 summation order is rowpoint(i)
__device__ __forceinline__ int rowpoint(int i) { return 2 * ((i & 3) + 4 * (i >> 3)) + ((i >> 2) & 1); }

// Discriminant MFMAs with the accumulator pinned to VGPRs: its consumer is VALU code (membership),
// and left to itself the allocator parks it in the AGPRs of the sums accumulators, paying 64 copies
// per tile.  Inline asm is invisible to the hazard recogniser: mfma_v_done() supplies the 18 wait
// states a 16-pass MFMA result needs before VALU may read it.
// b: the discriminant operand D[r][j] of this lane, held in an accumulation register for the whole launch (KM_BREG)
__device__ __forceinline__ void mfma_v_first(f32x16& c, float a, float b)
{
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_v(f32x16& c, float a, float b)
{
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_v_done(f32x16& c0, f32x16& c1)
{
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(c0), "+v"(c1));
}

// bit i of the low half -> bit 2i
__device__ __forceinline__ uint32_t spread16(uint32_t v)
{
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

#ifdef ABW_KM_TIMING
__device__ unsigned long long km_tim[10];
#define KM_T(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define KM_T(i) do { } while (0)
#endif

template <int DT, int CW> struct KmCfg {
    static constexpr int W = DT * 32;
    static constexpr int PT = 4 / CW;                        // point lanes (chunks walked side by side)
    // One staged x tile: row j = sr + 32 kk of the tile lies at LDS row sr * DTP + kk, so the DT rows a staging thread fills
    // are neighbours and its stores need no address arithmetic (an instruction between two MFMAs costs the pipe ~14
    // cycles); DTP odd keeps the second product's reads -- row col * DTP + t for lane col -- on 32 different banks.
    static constexpr int DTP = DT | 1;
    static constexpr int TILE_FLOATS = 32 * DTP * XS;
    // two buffers per point lane when they fit in 160 KiB, else one (the discriminant operand lives in registers)
    static constexpr int NBUF = (PT * 2 * TILE_FLOATS * 4 <= 160 * 1024) ? 2 : 1;
};

template <int DT, int CW, bool ASSIGN, bool OFF32>
__global__ __launch_bounds__(256) void km_step_kernel(const float* __restrict__ x, int64_t N, int d,
                                                     const float* __restrict__ dT, int Rp, int chunk, int64_t nchunks,
                                                     float* __restrict__ slabs, uint8_t* __restrict__ assign, int R)
{
    using C = KmCfg<DT, CW>;
    constexpr int W = C::W, PT = C::PT, NBUF = C::NBUF, DTP = C::DTP;
    constexpr int NS = 16 * DT;                              // k-steps of the discriminant product = MFMAs of either product
    __shared__ float xs[PT * NBUF][C::TILE_FLOATS];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int ct = wid % CW, pl = wid / CW;                  // this wave's column tile and point lane
    const int r0 = ((int)blockIdx.y * CW + ct) * 32;         // first restart of this wave
    // a wave past the last restart tile recomputes the last one and stores nothing: keeps every
    // wave on one straight-line MFMA stream and inside the workgroup barriers
    const bool wave_cols = r0 < Rp;
    // the discriminant operand of this wave's restarts: k-step s needs D[r0 + col][2s + half] -- one register per k-step
    float breg[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) breg[s] = dT[(int64_t)(2 * s + half) * Rp + min(r0, Rp - 32) + col];
    const int rp = rowpoint(col);
    const int ns_used = (d + 2) >> 1;                        // k-steps of the first product that touch a row <= d
    const int sq = threadIdx.x & 7;                          // staging: this thread's point quad within a tile
    const int sr = threadIdx.x >> 3;                         //          and its first row (then every 32nd)

#ifdef ABW_KM_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long r0_ = __builtin_amdgcn_s_memrealtime();
#endif
    // Staging of the next tiles (one per point lane): thread (sq, sr) moves points 4sq..4sq+3 of rows
    // sr, sr+32, ... of every lane's tile with one 16-byte load each (rows of x start at any 4-byte
    // boundary, so the loads are unaligned ones).  Loads and LDS stores are issued one at a time
    // between the MFMAs of the current tile: the four waves share one texture-address unit, and a
    // block of loads issued back to back stalls them for its whole length.
    // Rows >= d exist only in LDS: row d is all ones (the count column, and it meets e[r] in the
    // discriminant), rows above are zero; only the last row group can hold them.
    constexpr int NL = PT * DT;                              // staging loads (and stores) per thread per step
    f32x4u stg[NL];
    const float* ldp[PT];                                    // per point lane: x + first point of the quad being fetched
    // per point lane: LDS byte address of destination buffer + (sr, 4sq), each in a register of its own (left to itself the
    // compiler keeps one and adds the lanes' constant distance in front of every store: a vector add between two MFMAs)
    typedef __attribute__((address_space(3))) float lds_float;
    uint32_t stq[PT];
    // OFF32 (every byte offset inside a tile's d rows fits 32 bits: (d - 1) N + 31 < 2^30, what launch_step_cw checks): the
    // tile's first point is a uniform address -- a raw buffer descriptor in scalar registers -- and a thread's offsets to its
    // rows never change, so a staging load is ONE instruction.  With 64-bit pointers every load also paid a 64-bit vector add,
    // and those twelve adds per tile were all that kept the first product off the matrix pipe's rate (6 546 -> 6 121 cycles
    // per tile, the pipe's 6 144 less the first tile's share; experiments/README.md).
    uint32_t voff[DT];
#pragma unroll
    for (int kk = 0; kk < DT; kk++) voff[kk] = OFF32 ? (uint32_t)(((int64_t)min(sr + 32 * kk, d - 1) * N + 4 * sq) * 4) : 0u;
    auto load_k = [&](int k) {
        if (OFF32) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ldp[k / DT], 0, 0xFFFFFFFFu, 0x00027000);
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff[k % DT], 0, 0);
            stg[k] = f32x4u{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
        } else {
            stg[k] = *(const f32x4u*)(ldp[k / DT] + (int64_t)min(sr + 32 * (k % DT), d - 1) * N);
        }
    };
    auto store_k = [&](int k) {
        const int kk = k % DT;
        f32x4u v = stg[k];
        if (kk == DT - 1) {
            const int row = sr + 32 * kk;
            const float cst = (row == d) ? 1.0f : 0.0f;
            if (row >= d) v = f32x4u{cst, cst, cst, cst};
        }
        lds_float* q = (lds_float*)(size_t)stq[k / DT] + kk * XS;
        q[0] = v[0]; q[1] = v[1]; q[2] = v[2]; q[3] = v[3];
    };
    // A tile that would run past its chunk's end (only the last tile of the data can) is moved back to
    // end there: every load stays inside x, and the points in front of the original start -- already
    // consumed by the previous tile -- are masked out like points of no chunk at all.
    auto tile_origin = [&](int64_t t0, int64_t end) { return min(t0, end - 32); };

    const int64_t nsteps_total = (nchunks + PT - 1) / PT;    // chunk groups: point lane q of group g walks chunk g*PT + q
    for (int64_t cg = blockIdx.x; cg < nsteps_total; cg += gridDim.x) {
        int64_t pb[PT], pe[PT];
#pragma unroll
        for (int q = 0; q < PT; q++) {
            pb[q] = min(N, (cg * PT + q) * chunk);           // a lane without a chunk gets an empty range at N
            pe[q] = min(N, pb[q] + chunk);
        }
        const int64_t ch = cg * PT + pl;                      // this wave's chunk
        const int64_t p_begin = pb[pl], p_end = pe[pl];
        const int ntiles = (int)((pe[0] - pb[0] + 31) >> 5);  // lane 0 always has the longest chunk of the group
        int cur = 0;
        KM_T(0);
        __syncthreads();                                      // previous group's last tiles fully consumed
#pragma unroll
        for (int q = 0; q < PT; q++) {
            ldp[q] = x + tile_origin(pb[q], pe[q]) + (OFF32 ? 0 : 4 * sq);
            stq[q] = (uint32_t)(size_t)(lds_float*)(xs[q * NBUF] + sr * DTP * XS + 4 * sq);
            asm volatile("" : "+v"(stq[q]));
        }
#pragma unroll
        for (int k = 0; k < NL; k++) load_k(k);
#pragma unroll
        for (int k = 0; k < NL; k++) store_k(k);
        __syncthreads();
        f32x16 acc2[DT];
#pragma unroll
        for (int t = 0; t < DT; t++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc2[t][e] = 0.0f;
        KM_T(1);

        for (int ti = 0; ti < ntiles; ti++) {
            const bool have_next = ti + 1 < ntiles;
            const int64_t tile = p_begin + 32 * (int64_t)ti;  // this wave's tile
            // two buffers: the next tiles are fetched and stored while these are computed; on the last
            // step of a group the same code re-fetches the current tiles into the idle buffers (harmless)
#pragma unroll
            for (int q = 0; q < PT; q++) {
                ldp[q] = x + tile_origin(pb[q] + 32 * (int64_t)(have_next ? ti + 1 : ti), pe[q]) + (OFF32 ? 0 : 4 * sq);
                stq[q] = (uint32_t)(size_t)(lds_float*)(xs[q * NBUF + (NBUF == 2 ? (cur ^ 1) : 0)] + sr * DTP * XS + 4 * sq);
                asm volatile("" : "+v"(stq[q]));
            }
            const float* xt = xs[pl * NBUF + (NBUF == 2 ? cur : 0)];
            const float* sop = xt + col * DTP * XS + half;   // sums operand: + XS*t + 2*e
            float bb[2][DT];
            KM_T(2);
            // ---- discriminant: g[p][r] = sum_j x[p][j] D[r][j] + e[r]; operands read 8 k-steps ahead.
            // Two chains: k-steps alternate between them, so that no MFMA waits for the one before it
            // (a dependent MFMA that does not follow its producer back to back issues ~13 cycles late).
            f32x16 acc1, acc1b;
            {
                // One wave per SIMD: whatever is issued between two MFMAs has to fit into the 64 cycles
                // the first one runs, so the next batch's LDS reads and the staging loads are dealt out
                // one or two per MFMA instead of in a block (a block of ~16 costs a bubble per batch).
                const float* aop = xt + half * DTP * XS + rp; // + arow(s) for k-step s: dims 2s, 2s+1 = LDS rows ((2s)%32 + half) DTP + (2s)/32
                auto arow = [](int s_) { return (((2 * s_) % 32) * DTP + (2 * s_) / 32) * XS; };
                float a[2][8];
#pragma unroll
                for (int u = 0; u < 8; u++) a[0][u] = aop[arow(u)];
#pragma unroll
                for (int bt = 0; bt < NS / 8; bt++) {
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int m = 8 * bt + u;                        // index of this MFMA
                        if (bt + 1 < NS / 8) {
                            a[(bt + 1) & 1][u] = aop[arow(8 * (bt + 1) + u)];            // one operand read per MFMA
                        } else if (u < DT) {
                            bb[0][u] = sop[XS * u];                      // first operands of the sums product
                        }
                        if ((m * PT) % 16 == 0) load_k(m * PT / 16);      // next tiles -> registers, also with one buffer
                        __builtin_amdgcn_sched_barrier(0);
                        // k-step m multiplies rows 2m, 2m+1 of the tile: past row d (the ones row that meets e[r]) both operands are
                        // zero, and the last row group's k-steps beyond it are skipped (d = 199: 12 of 112; uniform branches)
                        if (bt == 0 && u < 2)                          mfma_v_first((u & 1) ? acc1b : acc1, a[0][u], breg[m]);
                        else if (m < 16 * (DT - 1) || m < ns_used)     mfma_v((u & 1) ? acc1b : acc1, a[bt & 1][u], breg[m]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            // ---- memberships, straight from the accumulator layout: register e of lane (col, half) is
            //      point q = 2e + half of the tile, restart r0 + col.  g = chain0 + chain1; member of
            //      cluster 1 iff g > 0.
            // ---- sums: S1[r][j] += sum_p mem[p][r] x[p][j]; operands read one point pair ahead;
            //      the sixteen memberships are computed in a block in front of it.
            KM_T(3);
            const int64_t tile0 = tile_origin(tile, p_end);                       // where the staged tile really starts
            const int skip = (int)min((int64_t)32, tile - tile0);                 // its leading points that are not ours
            float mem[16];
            uint32_t abits = 0;                               // ASSIGN: bit e = point 2e+half of the tile is in cluster 1
            auto member = [&](int e) {
                const bool valid = 2 * e + half >= skip;
                const bool in1 = acc1[e] + acc1b[e] > 0.0f;
                if (ASSIGN) abits |= in1 ? (1u << e) : 0u;
                return (valid && in1) ? 1.0f : 0.0f;
            };
            mfma_v_done(acc1, acc1b);
            // all sixteen at once, before the second product: between its MFMAs every one of these instructions costs the
            // matrix pipe ~14 cycles (760 cycles per tile), here their issue slots (600) -- experiments/README.md, round 3
            // ... and a tile whose 32 points are all ours (every tile but a chunk's last) does not ask for validity: three
            // instructions per point pair instead of six.  The branch is around this block only (around the loop of the second
            // product as well it made the allocator spill 300 registers).
            if (__builtin_amdgcn_readfirstlane(skip) == 0) {
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const bool in1 = acc1[e] + acc1b[e] > 0.0f;
                    if (ASSIGN) abits |= in1 ? (1u << e) : 0u;
                    mem[e] = in1 ? 1.0f : 0.0f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; e++) mem[e] = member(e);
            }
            KM_T(4);
#pragma unroll
            for (int e = 0; e < 16; e++) {
#pragma unroll
                for (int t = 0; t < DT; t++) {
                    const int m = e * DT + t;
                    if (e + 1 < 16) bb[(e + 1) & 1][t] = sop[XS * t + 2 * (e + 1)];
                    if (NBUF == 2 && (m * PT) % 16 == 0) store_k(m * PT / 16);
                    __builtin_amdgcn_sched_barrier(0);
                    acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(mem[e], bb[e & 1][t], acc2[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (ASSIGN) {
                // One 16-byte store per lane: lane (col, half) writes points 16*half .. +15 of restart
                // r0+col.  Points of the tile that are not ours (skip) are written too: whoever owns
                // them computes the same discriminant from the same inputs, so the bytes are identical.
                const uint32_t other = (uint32_t)__shfl_xor((int)abits, 32, 64);
                const uint32_t M = spread16(half ? other : abits) | (spread16(half ? abits : other) << 1);
                const uint32_t m16 = (M >> (16 * half)) & 0xFFFFu;
                u32x4u out;
#pragma unroll
                for (int w4 = 0; w4 < 4; w4++) out[w4] = (((m16 >> (4 * w4)) & 0xFu) * 0x00204081u) & 0x01010101u;
                if (wave_cols && r0 + col < R)
                    *(u32x4u*)(assign + (int64_t)(r0 + col) * N + tile0 + 16 * half) = out;
            }
            KM_T(5);
            if (NBUF == 2) {
                __syncthreads();                             // next tiles stored by all; these read by all
                cur ^= 1;
            } else if (have_next) {
                // one buffer per point lane: the next tiles were fetched into registers behind the discriminant
                // MFMAs; only their LDS stores wait for the readers of the current ones
                __syncthreads();                             // everyone done reading the only buffers
#pragma unroll
                for (int k = 0; k < NL; k++) store_k(k);
                __syncthreads();
            }
            KM_T(6);
        }
        if (wave_cols && ch < nchunks) {
            // slab[ch][r][j]: C/D layout col = lane&31 -> j, row = (e&3) + 8(e>>2) + 4*half -> restart
            float* slab = slabs + ((int64_t)ch * Rp + r0) * W;
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
    if (threadIdx.x == 0 && blockIdx.x == 7) {
        for (int i = 0; i < 8; i++) atomicAdd(&km_tim[i], tacc[i]);
        atomicAdd(&km_tim[8], __builtin_amdgcn_s_memrealtime() - r0_);
    }
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
// order.  Elementwise over [Rp][W]; the loads of one thread are independent, only its adds chain.
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

// Level 2a: the group partials added in group order -> this shard's sums [Rp][W]
__global__ __launch_bounds__(256) void km_sum_groups_kernel(const double* __restrict__ gsum, int64_t ngroups, int64_t elems,
                                                           double* __restrict__ sums)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= elems) return;
    double s = 0.0;
    for (int64_t g = 0; g < ngroups; g++) s += gsum[g * elems + e];
    sums[e] = s;
}

// Level 2b: one workgroup per restart takes the (all-shard) sums of cluster 1, derives
// cluster 0 from the totals, then either moves both centroids and rebuilds the discriminant row
// (update = 1) or contributes the restart's two inertia terms (update = 0).
__global__ void km_update_kernel(const double* __restrict__ sums, int64_t nparts, const double* __restrict__ tot,
                                 int Rp, int W, int d, float* __restrict__ cent, float* __restrict__ dT, int update,
                                 double* __restrict__ col_term)
{
    extern __shared__ double sh[];            // [2][W] sums of cluster 0 / 1, then [2][W] floats of the centroids
    const int r = blockIdx.x;
    double* s0 = sh; double* s1 = sh + W;
    float* c0 = (float*)(sh + 2 * W); float* c1 = c0 + W;
    for (int j = threadIdx.x; j < W; j += blockDim.x) {
        double s = 0.0;                       // nparts = 1: combined sums; > 1: this run's own group partials, in order
        for (int64_t g = 0; g < nparts; g++) s += sums[g * (int64_t)Rp * W + (int64_t)r * W + j];
        s1[j] = s;
        s0[j] = (j <= d) ? tot[j] - s : 0.0;
    }
    __syncthreads();
    const double n0 = s0[d], n1 = s1[d];
    for (int j = threadIdx.x; j < W; j += blockDim.x) {
        float v0 = cent[((int64_t)r * 2) * W + j], v1 = cent[((int64_t)r * 2 + 1) * W + j];
        if (update && j < d) {                // an empty cluster keeps its centroid
            if (n0 > 0.0) { v0 = (float)(s0[j] / n0); cent[((int64_t)r * 2) * W + j] = v0; }
            if (n1 > 0.0) { v1 = (float)(s1[j] / n1); cent[((int64_t)r * 2 + 1) * W + j] = v1; }
        }
        c0[j] = v0; c1[j] = v1;
    }
    __syncthreads();
    if (update) {
        km_discriminant(c0, c1, d, W, Rp, r, dT, threadIdx.x, blockDim.x);
    } else if (threadIdx.x < 2) {
        // sum_{p in cluster} |x_p - c|^2 = [sum |x_p|^2] - 2 c.S + n |c|^2 ; the bracket is added per restart
        const float* c = threadIdx.x ? c1 : c0;
        const double* s = threadIdx.x ? s1 : s0;
        double dot = 0.0, n2 = 0.0;
        for (int j = 0; j < d; j++) { dot += (double)c[j] * s[j]; n2 += (double)c[j] * (double)c[j]; }
        col_term[2 * r + threadIdx.x] = s[d] * n2 - 2.0 * dot;
    }
}

// Per dimension j, once per call: T[j] = sum_p x[j][p] and sum_p x[j][p]^2, each as 256 strided fp64
// partials followed by a fixed butterfly: xor steps inside each group of 64 partials (one wave per
// group, so all CUs stream), then (w0 + w1) + (w2 + w3).  tot[d] = N (the count column of the totals).
__global__ __launch_bounds__(64) void km_moments_kernel(const float* __restrict__ x, int64_t N, double* __restrict__ wpart)
{
    const int j = blockIdx.x, g = blockIdx.y;
    const float* row = x + (int64_t)j * N;
    double a1 = 0.0, a2 = 0.0;
    int64_t p = g * 64 + threadIdx.x;
    for (; p + 31 * 256 < N; p += 32 * 256) {                // 32 loads in flight; the adds keep their order
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; u++) v[u] = row[p + u * 256];
#pragma unroll
        for (int u = 0; u < 32; u++) { const double w2 = (double)v[u]; a1 += w2; a2 += w2 * w2; }
    }
    for (; p < N; p += 256) { const double v = (double)row[p]; a1 += v; a2 += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64); }
    if (threadIdx.x == 0) { wpart[(j * 4 + g) * 2] = a1; wpart[(j * 4 + g) * 2 + 1] = a2; }
}

__global__ void km_moments_finish_kernel(const double* __restrict__ wpart, int64_t N, int d, double* __restrict__ tot, double* __restrict__ x2_dim)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) tot[d] = (double)N;
    if (j >= d) return;
    const double* w = wpart + (int64_t)j * 8;
    tot[j] = (w[0] + w[2]) + (w[4] + w[6]);
    x2_dim[j] = (w[1] + w[3]) + (w[5] + w[7]);
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

struct StepArgs {
    const float* x; int64_t N; int d; const float* dT; int Rp; int chunk; int64_t nchunks; float* slabs; uint8_t* assign; int R;
};

template <int DT, int CW>
void launch_step_cw(abw_ctx* ctx, const StepArgs& a)
{
    constexpr int PT = 4 / CW;
    const unsigned gy = (unsigned)((a.Rp / 32 + CW - 1) / CW);
    const unsigned gx = (unsigned)std::min<int64_t>((a.nchunks + PT - 1) / PT, (int64_t)ctx->num_cu * 8);
    const dim3 grid(gx, gy);
    // byte offsets inside a tile's rows in 32 bits (the staging's one-instruction loads); ABW_KM_PTR64=1 forces the other path (tests)
    static const bool force64 = getenv("ABW_KM_PTR64") != nullptr;
    const bool off32 = !force64 && (int64_t)(a.d - 1) * a.N + 31 < (1ll << 30);
    if (a.assign && off32)
        hipLaunchKernelGGL((km_step_kernel<DT, CW, true, true>), grid, dim3(256), 0, ctx->stream, a.x, a.N, a.d, a.dT, a.Rp, a.chunk, a.nchunks, a.slabs, a.assign, a.R);
    else if (a.assign)
        hipLaunchKernelGGL((km_step_kernel<DT, CW, true, false>), grid, dim3(256), 0, ctx->stream, a.x, a.N, a.d, a.dT, a.Rp, a.chunk, a.nchunks, a.slabs, a.assign, a.R);
    else if (off32)
        hipLaunchKernelGGL((km_step_kernel<DT, CW, false, true>), grid, dim3(256), 0, ctx->stream, a.x, a.N, a.d, a.dT, a.Rp, a.chunk, a.nchunks, a.slabs, a.assign, a.R);
    else
        hipLaunchKernelGGL((km_step_kernel<DT, CW, false, false>), grid, dim3(256), 0, ctx->stream, a.x, a.N, a.d, a.dT, a.Rp, a.chunk, a.nchunks, a.slabs, a.assign, a.R);
}

// Column tiles per workgroup: as many as there are restart tiles (up to 4), the remaining waves
// become point lanes, so that all four SIMDs work whatever R is.
template <int DT>
void launch_step(abw_ctx* ctx, const StepArgs& a)
{
    const int tiles = a.Rp / 32;
    if (tiles >= 3)      launch_step_cw<DT, 4>(ctx, a);
    else if (tiles == 2) launch_step_cw<DT, 2>(ctx, a);
    else                 launch_step_cw<DT, 1>(ctx, a);
}

}  // namespace

__global__ __launch_bounds__(256) void km_sum_ordered_kernel(const double* __restrict__ parts, int32_t n_parts, int64_t n, double* out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double acc = parts[i];
    for (int32_t p = 1; p < n_parts; p++) acc += parts[(int64_t)p * n + i];
    out[i] = acc;
}

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

int32_t abw_km_chunk_for(int32_t num_cu, int64_t N, int32_t R)
{
    if (num_cu < 1 || N < 1 || R < 1) return 0;
    // workgroups that run side by side: one per compute unit (the step kernel takes ~150 KiB of LDS);
    // each walks PT chunks at once and covers CW restart tiles (see launch_step)
    const int tiles_r = (R + 31) / 32;
    const int cw = tiles_r >= 3 ? 4 : tiles_r;
    const int pt = 4 / cw, gy = (tiles_r + cw - 1) / cw;
    const int64_t slots = std::max<int64_t>(1, (int64_t)num_cu / gy) * pt;
    const int64_t tiles = (N + 31) / 32;
    for (int64_t rounds = 1;; rounds++) {
        const int64_t per = (tiles + slots * rounds - 1) / (slots * rounds);
        if (per <= 64) return (int32_t)(32 * std::max<int64_t>(1, per));
    }
}

int32_t abw_km_auto_chunk(const abw_ctx* ctx, int64_t N, int32_t R)
{
    return ctx ? abw_km_chunk_for(ctx->num_cu, N, R) : 0;
}

int abw_km_set_arith(abw_ctx* ctx, int32_t arith)
{
    if (!ctx) return ABW_ERR_ARG;
    if (arith != ABW_KM_ARITH_F32 && arith != ABW_KM_ARITH_BF16X3) return abw_fail(ctx, ABW_ERR_ARG, "abw_km_set_arith: unknown arithmetic %d", arith);
    ctx->km_arith = arith;
    return ABW_OK;
}

int abw_km_shard_begin(abw_ctx* ctx, const float* d_x, int64_t N_local, int64_t first_point, int64_t N_total, int32_t d,
                       int32_t R, uint32_t base_seed, int32_t chunk, abw_km_exchange* ex)
{
    if (!ctx) return ABW_ERR_ARG;
    ctx->km.active = false;
    if (!d_x || !ex || N_local < 32 || d < 1 || R < 1 || first_point < 0 || first_point + N_local > N_total)
        return abw_fail(ctx, ABW_ERR_ARG, "abw_km_shard_begin: bad argument (needs d_x, >= 32 local points inside [0, N_total), d >= 1, R >= 1)");
    if (chunk == 0) chunk = abw_km_auto_chunk(ctx, N_local, R);
    if (chunk < 32 || (chunk & 31)) return abw_fail(ctx, ABW_ERR_ARG, "abw_km_shard_begin: chunk must be a positive multiple of 32");
    const int DT = (d + 1 + 31) / 32;                       // dims + the count column, in 32-wide tiles
    if (DT > 7) return abw_fail(ctx, ABW_ERR_ARG, "abw_km_shard_begin: d = %d not supported (max 223)", d);
    abw_device_guard g(ctx->device);
    abw_km_state& k = ctx->km;
    k.x = d_x; k.N = N_local; k.first = first_point; k.N_total = N_total; k.d = d; k.R = R; k.seed = base_seed; k.chunk = chunk;
    k.DT = DT; k.W = DT * 32; k.Rp = (R + 31) & ~31;
    k.nchunks = (N_local + chunk - 1) / chunk;
    k.ngroups = (k.nchunks + KM_GROUP - 1) / KM_GROUP;
    k.elems = (int64_t)k.Rp * k.W;
    int rc;
    // The run's state lives in buffers of its own (ctx->km_buf), not in the shared scratch slots: control returns to the
    // caller between the phases, and any other entry point may grow (= free) a scratch slot in between.
    // km_buf[0]: col_term + moments (totals, squares) + sums + centroids + discriminant operand ;
    // km_buf[1]: slabs ; km_buf[2]: fp64 group sums
    const size_t cent_bytes = (size_t)k.Rp * 2 * k.W * 4, dt_bytes = (size_t)k.Rp * k.W * 4;
    const size_t term_bytes = (size_t)k.Rp * 2 * 8, mom_bytes = (size_t)(k.W + d) * 8, sums_bytes = (size_t)k.elems * 8;
    if ((rc = abw_reserve(ctx, ctx->km_buf[0], term_bytes + mom_bytes + sums_bytes + cent_bytes + dt_bytes + 64))) return rc;
    if ((rc = abw_reserve(ctx, ctx->km_buf[1], (size_t)k.nchunks * k.elems * 4))) return rc;
    if ((rc = abw_reserve(ctx, ctx->km_buf[2], (size_t)k.ngroups * k.elems * 8))) return rc;
    uint8_t* base = (uint8_t*)ctx->km_buf[0].p;
    k.col_term = (double*)base;
    k.moments = (double*)(base + term_bytes);
    k.sums = (double*)(base + term_bytes + mom_bytes);
    k.cent = (float*)(base + term_bytes + mom_bytes + sums_bytes);
    k.dT = (float*)(base + term_bytes + mom_bytes + sums_bytes + cent_bytes);
    k.slabs = (float*)ctx->km_buf[1].p;
    k.gsum = (double*)ctx->km_buf[2].p;
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(km_init_kernel, dim3(k.Rp), dim3(256), 0, st, d_x, N_local, first_point, N_total, d, k.W, R, base_seed, k.cent);
    ABW_LAUNCH_CHECK(ctx);
    {
        ABW_PROF(ctx, "km_moments");
        ABW_HIP(ctx, hipMemsetAsync(k.moments, 0, mom_bytes, st));
        double* wpart = k.gsum;                             // 8 doubles per dimension; the group sums are not in use yet
        hipLaunchKernelGGL(km_moments_kernel, dim3(d, 4), dim3(64), 0, st, d_x, N_local, wpart);
        ABW_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(km_moments_finish_kernel, dim3((d + 63) / 64), dim3(64), 0, st, wpart, N_local, d, k.moments, k.moments + k.W);
        ABW_LAUNCH_CHECK(ctx);
    }
    ex->d_init = k.cent; ex->init_count = (uint64_t)k.Rp * 2 * k.W;
    ex->d_moments = k.moments; ex->moments_count = (uint64_t)(k.W + d);
    ex->d_sums = k.sums; ex->sums_count = (uint64_t)k.elems;
    k.single = false;
    k.active = true;
    return ABW_OK;
}

int abw_km_shard_ready(abw_ctx* ctx)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!ctx->km.active) return abw_fail(ctx, ABW_ERR_STATE, "abw_km_shard_ready: no k-means run in progress");
    abw_device_guard g(ctx->device);
    abw_km_state& k = ctx->km;
    hipLaunchKernelGGL(km_discriminant_kernel, dim3(k.Rp), dim3(256), 0, ctx->stream, k.cent, k.d, k.W, k.Rp, k.dT);
    ABW_LAUNCH_CHECK(ctx);
    return ABW_OK;
}

int abw_km_shard_step(abw_ctx* ctx, uint8_t* d_assign)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!ctx->km.active) return abw_fail(ctx, ABW_ERR_STATE, "abw_km_shard_step: no k-means run in progress");
    abw_device_guard g(ctx->device);
    abw_km_state& k = ctx->km;
    const StepArgs a = { k.x, k.N, k.d, k.dT, k.Rp, k.chunk, k.nchunks, k.slabs, d_assign, k.R };
    {
        // the split-bf16 arithmetic runs the passes that only feed the centroid update; the final assign pass is the exact one
        const bool want_split = ctx->km_arith == ABW_KM_ARITH_BF16X3 && !d_assign && abw_km_b3_has(k.d, k.Rp);
        ABW_PROF(ctx, want_split ? "km_step_b3" : "km_step");
        const bool split = want_split && abw_km_step_b3(ctx, k.x, k.N, k.d, k.dT, k.Rp, k.chunk, k.nchunks, k.slabs);
        if (!split) switch (k.DT) {
            case 1: launch_step<1>(ctx, a); break;
            case 2: launch_step<2>(ctx, a); break;
            case 3: launch_step<3>(ctx, a); break;
            case 4: launch_step<4>(ctx, a); break;
            case 5: launch_step<5>(ctx, a); break;
            case 6: launch_step<6>(ctx, a); break;
            default: launch_step<7>(ctx, a); break;
        }
    }
    ABW_LAUNCH_CHECK(ctx);
    ABW_PROF(ctx, "km_update");
    hipLaunchKernelGGL(km_group_kernel, dim3((unsigned)((k.elems + 255) / 256), (unsigned)k.ngroups), dim3(256), 0, ctx->stream,
                       k.slabs, k.nchunks, k.elems, k.gsum);
    ABW_LAUNCH_CHECK(ctx);
    if (!k.single) {                          // a lone shard skips the copy: its update reads the group partials
        hipLaunchKernelGGL(km_sum_groups_kernel, dim3((unsigned)((k.elems + 255) / 256)), dim3(256), 0, ctx->stream, k.gsum, k.ngroups, k.elems, k.sums);
        ABW_LAUNCH_CHECK(ctx);
    }
    return ABW_OK;
}

int abw_km_shard_update(abw_ctx* ctx, int32_t last)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!ctx->km.active) return abw_fail(ctx, ABW_ERR_STATE, "abw_km_shard_update: no k-means run in progress");
    abw_device_guard g(ctx->device);
    abw_km_state& k = ctx->km;
    ABW_PROF(ctx, "km_update");
    const size_t upd_lds = (size_t)k.W * 16 + (size_t)k.W * 8;
    hipLaunchKernelGGL(km_update_kernel, dim3(k.Rp), dim3(k.W), upd_lds, ctx->stream, k.single ? k.gsum : k.sums, k.single ? k.ngroups : (int64_t)1,
                       k.moments, k.Rp, k.W, k.d, k.cent, k.dT,
                       last ? 0 : 1, k.col_term);
    ABW_LAUNCH_CHECK(ctx);
    return ABW_OK;
}

// The combine step of the point-sharded run for a caller that holds the shards' buffers side by side on the device
// (d_parts: [n_parts][n] fp64, e.g. the output of one all-gather): d_out[i] = ((part 0 + part 1) + part 2) + ..., the order
// abawaca_amd/multi.py adds them in, so every rank holds identical bits.  On the context's stream; d_out may be a part.
int abw_km_sum_ordered(abw_ctx* ctx, const double* d_parts, int32_t n_parts, int64_t n, double* d_out)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!d_parts || !d_out || n_parts < 1 || n < 0) return abw_fail(ctx, ABW_ERR_ARG, "abw_km_sum_ordered: bad argument");
    if (n == 0) return ABW_OK;
    abw_device_guard g(ctx->device);
    hipLaunchKernelGGL(km_sum_ordered_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_parts, n_parts, n, d_out);
    ABW_LAUNCH_CHECK(ctx);
    return ABW_OK;
}

int abw_km_shard_finish(abw_ctx* ctx, float* d_centroids, double* d_inertia)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!ctx->km.active) return abw_fail(ctx, ABW_ERR_STATE, "abw_km_shard_finish: no k-means run in progress");
    abw_device_guard g(ctx->device);
    abw_km_state& k = ctx->km;
    hipStream_t st = ctx->stream;
    if (d_inertia) {
        hipLaunchKernelGGL(km_inertia_kernel, dim3((k.R + 63) / 64), dim3(64), 0, st, k.moments + k.W, k.d, k.col_term, k.R, d_inertia);
        ABW_LAUNCH_CHECK(ctx);
    }
    if (d_centroids) {
        hipLaunchKernelGGL(km_copy_centroids_kernel, dim3(64), dim3(256), 0, st, k.cent, k.W, k.d, k.R, d_centroids);
        ABW_LAUNCH_CHECK(ctx);
    }
    k.active = false;
    return ABW_OK;
}

int abw_km_run_dev(abw_ctx* ctx, const float* d_x, int64_t N, int32_t d, int32_t R, uint32_t base_seed,
                   int32_t iters, int32_t chunk, uint8_t* d_assign, float* d_centroids, double* d_inertia)
{
    if (!ctx) return ABW_ERR_ARG;
    if (iters < 0) return abw_fail(ctx, ABW_ERR_ARG, "abw_km_run_dev: iters < 0");
    abw_km_exchange ex;
    int rc = abw_km_shard_begin(ctx, d_x, N, 0, N, d, R, base_seed, chunk, &ex);     // one shard: nothing to exchange
    if (rc) return rc;
    ctx->km.single = true;
    if ((rc = abw_km_shard_ready(ctx))) return rc;
    for (int it = 0; it <= iters; it++) {
        const bool last = it == iters;
        if ((rc = abw_km_shard_step(ctx, last ? d_assign : nullptr))) return rc;
        if (last && !d_inertia) break;
        if ((rc = abw_km_shard_update(ctx, last ? 1 : 0))) return rc;
    }
    return abw_km_shard_finish(ctx, d_centroids, d_inertia);
}

int abw_km_run(abw_ctx* ctx, const float* x, int64_t N, int32_t d, int32_t R, uint32_t base_seed,
               int32_t iters, int32_t chunk, uint8_t* assign, float* centroids, double* inertia)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!x || N < 32 || d < 1 || R < 1) return abw_fail(ctx, ABW_ERR_ARG, "abw_km_run: bad argument (needs x, N >= 32 points, d >= 1, R >= 1)");
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
