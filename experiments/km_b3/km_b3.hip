// Experiment (round 3): the k-means step with x and D split into three bf16 planes on
// v_mfma_f32_32x32x16_bf16 -- is it >= 1.5x the exact fp32 kernel (km_step_kernel<6,2>: 0.351 ms at configs[1])?
//
//   x = hi + mid + lo exactly (24 mantissa bits = 8 + 8 + 8, each plane the truncated top half of what is left);
//   g = x.D is taken as the six plane products of weight >= 2^-16 (hh, hm, mh, mm, hl, lh), fp32 accumulation;
//   S1 = mem^T.x is three products (memberships are 0/1, exact in bf16).
//
// One image of the staged tile serves both products: [plane][j][32 points] bf16 with 64-byte rows whose 8-byte units are
// XOR-swizzled; the discriminant's A operand comes out of it with ds_read_b64_tr_b16, the sums' B operand with ds_read_b64.
//
// Standalone: builds its own inputs, checks a few chunks against fp64 on the host, times the launch.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize km_b3.hip -o km_b3 && ./km_b3 [N d R chunk]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

#define SB() __builtin_amdgcn_sched_barrier(0)
#ifdef KMT
__device__ unsigned long long km_tim[8];
#define KM_T(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define KM_T(i) do { } while (0)
#endif
union Frag { bf16x8 v; s16x4 h[2]; uint32_t w[4]; uint2 d2[2]; uint4 q; };

// h, m, l: words whose UPPER halves are the three bf16 planes of v (the lower halves are whatever is left: hi2 drops them)
__device__ __forceinline__ void split3(float v, uint32_t& h, uint32_t& m, uint32_t& l)
{
    h = __float_as_uint(v);
    const float r1 = v - __uint_as_float(h & 0xFFFF0000u);    // built with -fno-slp-vectorize: v_pk_add_f32 beside MFMAs is slow
    m = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(m & 0xFFFF0000u);
    l = __float_as_uint(r2);
}
// the upper halves of two words as one: a's in the low half, b's in the high half (v_perm_b32)
__device__ __forceinline__ uint32_t hi2(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// XOR applied to the 16-byte chunk index (0..3) of row j of the x image
__device__ __forceinline__ int sigma(int j) { return (j >> 2) & 3; }

template <int DT, int CW>
__global__ __launch_bounds__(256) void km_step_b3_kernel(const float* __restrict__ x, int64_t N, int d,
                                                         const float* __restrict__ dT, int Rp, int chunk, int64_t nchunks,
                                                         float* __restrict__ slabs)
{
    constexpr int W = 32 * DT, NS = 2 * DT;                   // k-steps (of 16 dims) of the discriminant = slots of the sums
    constexpr int XPLANE = W * 64;                            // bytes of one plane of one staged tile
    constexpr int XTILE = 3 * XPLANE;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr int PT = 4 / CW;                                // point lanes (chunks walked side by side)
    uint8_t* ximg = lds;                                      // [PT point lanes][2 buffers][3][W][64]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int ct = wid % CW, pl = wid / CW;
    const int r0 = ((int)blockIdx.y * CW + ct) * 32;
    const bool wave_cols = r0 < Rp;                           // a wave past the last restart tile recomputes the last one and stores nothing
    const int r0c = min(r0, Rp - 32);

    // ---- D of this wave's 32 restarts, in registers for the whole launch: lane (r, half) holds D[r][16 s + 8 half + i]
    Frag dreg[3][NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        uint32_t h[8], m[8], l[8];
#pragma unroll
        for (int i = 0; i < 8; i++) split3(dT[(int64_t)(16 * s + 8 * half + i) * Rp + r0c + col], h[i], m[i], l[i]);
#pragma unroll
        for (int w = 0; w < 4; w++) {
            dreg[0][s].w[w] = hi2(h[2 * w], h[2 * w + 1]);
            dreg[1][s].w[w] = hi2(m[2 * w], m[2 * w + 1]);
            dreg[2][s].w[w] = hi2(l[2 * w], l[2 * w + 1]);
        }
    }

    // ---- staging: thread (sq, sr) moves points 4sq..4sq+3 of rows sr, sr+32, ... of both point lanes' tiles
    const int sq = threadIdx.x & 7, sr = threadIdx.x >> 3;
    constexpr int NL = PT * DT;
    f32x4u stg[NL];
    const float* ldp[PT];
    // the tile's first point as a raw buffer descriptor in scalar registers + the thread's constant 32-bit row offsets: a
    // staging load is one instruction (the launcher refuses tables whose row offsets need more than 32 bits)
    uint32_t voff[DT];
#pragma unroll
    for (int kk = 0; kk < DT; kk++) voff[kk] = (uint32_t)(((int64_t)min(sr + 32 * kk, d - 1) * N + 4 * sq) * 4);
    auto load_k = [&](int k) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ldp[k / DT], 0, 0xFFFFFFFFu, 0x00027000);
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff[k % DT], 0, 0);
        stg[k] = f32x4u{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
    };
    auto store_k = [&](int k, int buf) {
        const int kk = k % DT, row = sr + 32 * kk;
        f32x4u v = stg[k];
        if (kk == DT - 1 && row >= d) { const float c = (row == d) ? 1.0f : 0.0f; v = f32x4u{c, c, c, c}; }
        uint32_t h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; e++) split3(v[e], h[e], m[e], l[e]);
        uint8_t* q = ximg + ((k / DT) * 2 + buf) * XTILE + row * 64 + (((sq >> 1) ^ sigma(row)) << 4) + ((sq & 1) << 3);
        *(uint2*)(q) = make_uint2(hi2(h[0], h[1]), hi2(h[2], h[3]));
        *(uint2*)(q + XPLANE) = make_uint2(hi2(m[0], m[1]), hi2(m[2], m[3]));
        *(uint2*)(q + 2 * XPLANE) = make_uint2(hi2(l[0], l[1]), hi2(l[2], l[3]));
    };
    auto tile_origin = [&](int64_t t0, int64_t end) { return min(t0, end - 32); };

    // ---- per-lane LDS addresses
    const int li = lane & 15, g1 = (lane >> 4) & 1;
    const int tr_row = 8 * half + (li >> 2);                  // + 16 s + 4 t
    // slot q = 16 s2 + 8 h + i of a row holds tile point q and is row (i&3) + 8(2 s2 + (i>>2)) + 4h of the discriminant tile:
    // the address lane (q', pp) of a transposed read hands columns 4pp..4pp+3 the unit 4 g1 + 2(pp&1) + (pp>>1)
    const int tr_unit = 4 * g1 + 2 * (li & 1) + ((li >> 1) & 1);
    const int sg = sigma(col);

#ifdef KMT
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long k0 = tprev, rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int64_t ngroups = (nchunks + PT - 1) / PT;
    for (int64_t cg = blockIdx.x; cg < ngroups; cg += gridDim.x) {
        int64_t pb[PT], pe[PT];
#pragma unroll
        for (int q = 0; q < PT; q++) { pb[q] = min(N, (cg * PT + q) * chunk); pe[q] = min(N, pb[q] + chunk); }
        const int64_t ch = cg * PT + pl;
        const int64_t p_begin = pb[pl], p_end = pe[pl];
        const int ntiles = (int)((pe[0] - pb[0] + 31) >> 5);
        auto point_at = [&](int q, int t) { return x + tile_origin(pb[q] + 32 * (int64_t)min(t, ntiles - 1), pe[q]); };
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PT; q++) ldp[q] = point_at(q, 0);
#pragma unroll
        for (int k = 0; k < NL; k++) load_k(k);
#pragma unroll
        for (int k = 0; k < NL; k++) store_k(k, 0);
#pragma unroll
        for (int q = 0; q < PT; q++) ldp[q] = point_at(q, 1);
#pragma unroll
        for (int k = 0; k < NL; k++) load_k(k);               // tile 1 waits in registers
        __syncthreads();
        f32x16 acc2[DT];
#pragma unroll
        for (int t = 0; t < DT; t++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc2[t][e] = 0.0f;
        int cur = 0;

        for (int ti = 0; ti < ntiles; ti++) {
            const int64_t tile = p_begin + 32 * (int64_t)ti;
            const uint8_t* xw = ximg + (pl * 2 + cur) * XTILE;
            // ---- discriminant: x fragments one k-step ahead
            auto read_xa = [&](Frag* xa, int s) {
#pragma unroll
                for (int P = 0; P < 3; P++)
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        const int sgm = (2 * half + t) & 3;                       // sigma of rows 16s + 8half + 4t + q'
                        const uint8_t* a = xw + P * XPLANE + (16 * s + 4 * t + tr_row) * 64 + (((tr_unit >> 1) ^ sgm) << 4) + ((tr_unit & 1) << 3);
                        xa[P].h[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)a);
                    }
            };
            auto read_xb = [&](Frag* xb, int m) {
                const int t = m >> 1, s2 = m & 1;
                const uint8_t* rowp = xw + (32 * t + col) * 64;
#pragma unroll
                for (int P = 0; P < 3; P++) xb[P].q = *(const uint4*)(rowp + P * XPLANE + (((2 * s2 + half) ^ sg) << 4));
            };
            auto read_xa1 = [&](Frag* xa, int s, int P) {
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int sgm = (2 * half + t) & 3;                           // sigma of rows 16s + 8half + 4t + q'
                    const uint8_t* a = xw + P * XPLANE + (16 * s + 4 * t + tr_row) * 64 + (((tr_unit >> 1) ^ sgm) << 4) + ((tr_unit & 1) << 3);
                    xa[P].h[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)a);
                }
            };
            // sums: MFMA n = (group gi = n / DT, column tile t = n % DT), group gi = (point half s2 = gi / 3, plane P = 2 - gi % 3);
            // its x fragment is read three MFMAs ahead into a ring of four
            constexpr int NM = 6 * DT;
#ifndef RB
#define RB 4
#endif
            Frag xb[RB];
            auto read_xbn = [&](int n) {
                if (n >= NM) return;
                const int gi = n / DT, t = n % DT, s2 = gi / 3, P = 2 - gi % 3;
                xb[n % RB].q = *(const uint4*)(xw + (32 * t + col) * 64 + P * XPLANE + (((2 * s2 + half) ^ sg) << 4));
            };
            Frag xa[2][3];
            f32x16 S0, S1;
#pragma unroll
            for (int e = 0; e < 16; e++) { S0[e] = 0.0f; S1[e] = 0.0f; }
            read_xa(xa[0], 0);
            SB();                                             // the first fragments are on their way: now the address arithmetic
            const float* nxt[PT];
#pragma unroll
            for (int q = 0; q < PT; q++) nxt[q] = point_at(q, ti + 2);
            KM_T(0);
            // ---- discriminant: two chains; between two MFMAs one sixth of the k-step's other work: the x fragments of the
            //      next k-step, and staging load s of tile ti+1 split into planes, stored, and re-issued for tile ti+2
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const Frag* a = xa[s & 1];
                Frag* nx = xa[(s + 1) & 1];
#ifdef NOSTAGE
                const bool st = false;
#else
                const bool st = s < NL;                                           // this k-step carries staging load s
#endif
                const int kk = s % DT, row = sr + 32 * kk;
                f32x4u v = stg[st ? s : 0];
                if (kk == DT - 1 && row >= d) { const float c = (row == d) ? 1.0f : 0.0f; v = f32x4u{c, c, c, c}; }
                uint32_t h[4], m[4], l[4];
                SB();
                S0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2].v, dreg[0][s].v, S0, 0, 0, 0);
                SB();
                if (s + 1 < NS) read_xa1(nx, s + 1, 0); else { read_xbn(0); if (RB > 4) read_xbn(3); }
                if (st) split3(v[0], h[0], m[0], l[0]);
                SB();
                S1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, dreg[2][s].v, S1, 0, 0, 0);
                SB();
                if (s + 1 < NS) read_xa1(nx, s + 1, 1); else { read_xbn(1); if (RB > 5) read_xbn(4); }
                if (st) split3(v[1], h[1], m[1], l[1]);
                SB();
                S0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, dreg[1][s].v, S0, 0, 0, 0);
                SB();
                if (s + 1 < NS) read_xa1(nx, s + 1, 2); else { read_xbn(2); if (RB > 6) read_xbn(5); if (RB > 7) read_xbn(6); }
                if (st) split3(v[2], h[2], m[2], l[2]);
                SB();
                S1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, dreg[0][s].v, S1, 0, 0, 0);
                SB();
                if (st) split3(v[3], h[3], m[3], l[3]);
                uint8_t* q = ximg + (((st ? s : 0) / DT) * 2 + (cur ^ 1)) * XTILE + row * 64 + (((sq >> 1) ^ sigma(row)) << 4) + ((sq & 1) << 3);
                SB();
                S0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, dreg[1][s].v, S0, 0, 0, 0);
                SB();
                if (st) {
                    *(uint2*)(q) = make_uint2(hi2(h[0], h[1]), hi2(h[2], h[3]));
                    *(uint2*)(q + XPLANE) = make_uint2(hi2(m[0], m[1]), hi2(m[2], m[3]));
                }
                SB();
                S1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, dreg[0][s].v, S1, 0, 0, 0);
                SB();
                if (st) {
                    *(uint2*)(q + 2 * XPLANE) = make_uint2(hi2(l[0], l[1]), hi2(l[2], l[3]));
                    ldp[s / DT] = nxt[s / DT];
                    load_k(s);
                }
                SB();
            }
            KM_T(1);
            // ---- memberships: register e of lane (col, half) is tile point 16(e>>3) + 8 half + (e&7)
            const int64_t tile0 = tile_origin(tile, p_end);
            const int skip = (int)min((int64_t)32, tile - tile0);
            Frag mf[2];
            auto members = [&](int s2) {
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    uint32_t v = 0;
#pragma unroll
                    for (int b = 0; b < 2; b++) {
                        const int e = 8 * s2 + 2 * w + b;
                        const int pos = 16 * (e >> 3) + 8 * half + (e & 7);
                        if (pos >= skip && S0[e] + S1[e] > 0.0f) v |= 0x3F80u << (16 * b);
                    }
                    mf[s2].w[w] = v;
                }
            };
            members(0);
            KM_T(2);
            // ---- sums: six groups (point half s2 x plane), each one MFMA per column tile: no MFMA waits for the one before
#pragma unroll
            for (int n = 0; n < NM; n++) {
                SB();
#ifndef NOSUMREAD
                read_xbn(n + RB - 1);
#endif
                if (n == 0) members(1);
                SB();
                acc2[n % DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf[n / (3 * DT)].v, xb[n % RB].v, acc2[n % DT], 0, 0, 0);
            }
            // the discriminant's accumulators stay allocated until here: left to itself the allocator gives them the registers
            // of two sums accumulators and copies those away and back around every tile (64 v_accvgpr_mov)
            asm volatile("; S0, S1 live to here %0 %1" :: "a"(S0), "a"(S1));
            SB();
            KM_T(3);
            __syncthreads();
            KM_T(4);
            cur ^= 1;
        }
#ifdef KMT
        if (blockIdx.x == 7 && lane == 0) {
            for (int i = 0; i < 5; i++) atomicAdd(&km_tim[i], tacc[i]);
            atomicAdd(&km_tim[5], __builtin_amdgcn_s_memtime() - k0);
            atomicAdd(&km_tim[6], __builtin_amdgcn_s_memrealtime() - rt0);
        }
#endif
        if (wave_cols && ch < nchunks) {
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
}

static uint64_t rng_state = 88172645463325252ull;
static inline double urand() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (rng_state >> 11) * (1.0 / 9007199254740992.0); }
static inline double nrand() { double u = urand(), v = urand(); return sqrt(-2.0 * log(u + 1e-300)) * cos(6.283185307179586 * v); }

template <int DT, int CW>
static float run(const float* dx, int64_t N, int d, const float* ddT, int Rp, int chunk, int64_t nchunks, float* dslabs, int reps)
{
    constexpr int W = 32 * DT;
    constexpr int PT = 4 / CW;
    const size_t lds = (size_t)PT * 2 * 3 * W * 64;
    CK(hipFuncSetAttribute((const void*)km_step_b3_kernel<DT, CW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid((unsigned)std::min<int64_t>((nchunks + PT - 1) / PT, 256 * 8), (unsigned)((Rp / 32 + CW - 1) / CW));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((km_step_b3_kernel<DT, CW>), grid, dim3(256), lds, 0, dx, N, d, ddT, Rp, chunk, nchunks, dslabs);
    for (int i = 0; i < 10 * reps; i++) hipLaunchKernelGGL((km_step_b3_kernel<DT, CW>), grid, dim3(256), lds, 0, dx, N, d, ddT, Rp, chunk, nchunks, dslabs);   // clocks up
    CK(hipDeviceSynchronize());
#ifdef KMT
    { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(km_tim), z, sizeof(z))); }
#endif
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((km_step_b3_kernel<DT, CW>), grid, dim3(256), lds, 0, dx, N, d, ddT, Rp, chunk, nchunks, dslabs);
    CK(hipEventRecord(b, 0));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    printf("lds %zu B, grid %u x %u\n", lds, grid.x, grid.y);
    return ms / reps;
}

int main(int argc, char** argv)
{
    const int64_t N = argc > 1 ? atoll(argv[1]) : 799859;
    const int d = argc > 2 ? atoi(argv[2]) : 189;
    const int R = argc > 3 ? atoi(argv[3]) : 64;
    const int chunk = argc > 4 ? atoi(argv[4]) : 1568;
    const int DT = (d + 1 + 31) / 32, W = 32 * DT, Rp = (R + 31) / 32 * 32;
    const int64_t nchunks = (N + chunk - 1) / chunk;
    printf("N %lld d %d R %d chunk %d -> W %d, %lld chunks\n", (long long)N, d, R, chunk, W, (long long)nchunks);
    // points: two planted groups per restart direction would be overkill; z-scored noise + a planted shift on a few dims
    std::vector<float> x((size_t)d * N), dT((size_t)W * Rp, 0.0f);
    for (int64_t p = 0; p < N; p++) {
        const int grp = (int)(urand() * 8);
        for (int j = 0; j < d; j++) x[(size_t)j * N + p] = (float)(nrand() + ((j % 8) == grp ? 1.5 : 0.0));
    }
    // D = c1 - c0 of two random points, e = (|c0|^2 - |c1|^2) / 2
    for (int r = 0; r < R; r++) {
        const int64_t a = (int64_t)(urand() * N), b = (int64_t)(urand() * N);
        double e = 0;
        for (int j = 0; j < d; j++) {
            const float c0 = x[(size_t)j * N + a], c1 = x[(size_t)j * N + b];
            dT[(size_t)j * Rp + r] = c1 - c0;
            e += 0.5 * ((double)c0 * c0 - (double)c1 * c1);
        }
        dT[(size_t)d * Rp + r] = (float)e;
    }
    float *dx, *ddT, *dslabs;
    CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&ddT, dT.size() * 4)); CK(hipMalloc(&dslabs, (size_t)nchunks * Rp * W * 4));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ddT, dT.data(), dT.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dslabs, 0xFF, (size_t)nchunks * Rp * W * 4));
    float ms = 0;
    const int tiles_r = Rp / 32;
    if (DT == 6 && tiles_r == 2) ms = run<6, 2>(dx, N, d, ddT, Rp, chunk, nchunks, dslabs, 20);
    else if (DT == 6 && tiles_r >= 3) ms = run<6, 4>(dx, N, d, ddT, Rp, chunk, nchunks, dslabs, 20);
    else if (DT == 7 && tiles_r >= 3) ms = run<7, 4>(dx, N, d, ddT, Rp, chunk, nchunks, dslabs, 20);
    else { fprintf(stderr, "DT %d not built\n", DT); return 1; }
    printf("km_step_b3: %.4f ms per launch\n", ms);
#ifdef KMT
    { unsigned long long t[8]; CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(km_tim), sizeof(t)));
      const double per = 1.0 / (20.0 * 4 * (Rp / 32 > 2 ? (Rp / 128.0) : 1.0) * ((chunk + 31) / 32));   // launches x waves x tiles
      printf("memtime ticks per tile: top %.0f, discriminant %.0f, members %.0f, sums %.0f, barrier %.0f\n", t[0] * per, t[1] * per, t[2] * per, t[3] * per, t[4] * per);
      printf("loop in all: %.0f ticks per tile; clock %.1f MHz (memtime / memrealtime at 100 MHz)\n", t[5] * per, 100.0 * t[5] / t[6]); }
#endif

    // ---- check: first, a middle and the last chunk, every restart, against fp64
    std::vector<float> slab((size_t)Rp * W);
    int64_t bad_count = 0, near = 0, units = 0; double worst = 0;
    const int64_t which[3] = {0, nchunks / 2, nchunks - 1};
    for (int w = 0; w < 3; w++) {
        const int64_t ch = which[w], pb = ch * chunk, pe = std::min<int64_t>(N, pb + chunk);
        CK(hipMemcpy(slab.data(), dslabs + (size_t)ch * Rp * W, slab.size() * 4, hipMemcpyDeviceToHost));
        for (int r = 0; r < R; r++) {
            std::vector<double> sum(d + 1, 0.0), sum_alt(d + 1, 0.0);
            int ambiguous = 0;
            for (int64_t p = pb; p < pe; p++) {
                double g = dT[(size_t)d * Rp + r], mag = fabs(g);
                for (int j = 0; j < d; j++) { const double t = (double)x[(size_t)j * N + p] * dT[(size_t)j * Rp + r]; g += t; mag += fabs(t); }
                units++;
                if (fabs(g) < 1e-6 * mag) { ambiguous++; near++; continue; }
                if (g > 0) { for (int j = 0; j < d; j++) sum[j] += x[(size_t)j * N + p]; sum[d] += 1; }
            }
            if (ambiguous) continue;
            if (slab[(size_t)r * W + d] != (float)sum[d]) bad_count++;
            for (int j = 0; j < d; j++) {
                const double err = fabs(slab[(size_t)r * W + j] - sum[j]) / (fabs(sum[j]) + sqrt(sum[d] + 1));
                worst = std::max(worst, err);
            }
        }
    }
    printf("check: %lld units, %lld near ties skipped, %lld (chunk, restart) counts differ, worst sum error %.3g (relative to |sum| + sqrt(n))\n",
           (long long)units, (long long)near, (long long)bad_count, worst);
    return bad_count ? 2 : 0;
}
