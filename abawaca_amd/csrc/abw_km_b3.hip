// The k-means step on three bf16 planes (abw_km_set_arith(ctx, ABW_KM_ARITH_BF16X3)): the same two contractions as
// km_step_kernel (abw_km.hip) on v_mfma_f32_32x32x16_bf16 instead of v_mfma_f32_32x32x2_f32.
//
//   x = hi + mid + lo exactly (24 mantissa bits = 8 + 8 + 8; each plane is the truncated top half of what is left), D alike;
//   g = x.D + e is taken as the six plane products of weight >= 2^-16 (lh, hl, mm, mh, hm, hh), fp32 accumulation, two chains;
//   S1 = mem^T.x is three products per plane (memberships are 0/1, exact in bf16), fp32 accumulation.
// The MFMA's accumulation order is the hardware's, so the CPU restatement cannot reproduce the bits: this arithmetic is held to
// the documented tolerance of the independent Lloyd implementation (tests/test_gpu_kmeans_lloyd.py), not to bit
// parity, and it is opt-in.  The final assign pass, the fp64 slab reduction, the centroid update and the inertia are the exact
// path's.  (No counterpart in the reference, like the rest of the k-means extension.)
//
// One image of the staged tile serves both products: [plane][j][32 points] bf16, 64-byte rows, 16-byte chunks XOR-swizzled by
// (j >> 2) & 3.  The discriminant's A operand comes out of it with ds_read_b64_tr_b16 (conflict-free: 4 rows x 64 B per half
// wave), the sums' B operand with ds_read_b128 (16 lanes x 4 banks, all distinct).  D of the wave's 32 restarts stays in
// registers (24 DT of them).  One wave per SIMD: between two MFMAs sits one sixth of a k-step's other work -- the next k-step's
// fragments, and one staging load of the next tile split into planes, stored into the other buffer and re-issued for the tile
// after it -- because a wave issues in order and a block of VALU behind a dependent MFMA chain would not overlap it.
// Built with -fno-slp-vectorize (Makefile): v_pk_add_f32 beside MFMAs costs more than the two v_sub_f32 it replaces.
#include "abw_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
#define SB() __builtin_amdgcn_sched_barrier(0)

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
            constexpr int RB = 4;
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
            // ---- discriminant: two chains; between two MFMAs one sixth of the k-step's other work: the x fragments of the
            //      next k-step, and staging load s of tile ti+1 split into planes, stored, and re-issued for tile ti+2
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const Frag* a = xa[s & 1];
                Frag* nx = xa[(s + 1) & 1];
                const bool st = s < NL;                                           // this k-step carries staging load s
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
            // ---- sums: six groups (point half s2 x plane), each one MFMA per column tile: no MFMA waits for the one before
#pragma unroll
            for (int n = 0; n < NM; n++) {
                SB();
                read_xbn(n + RB - 1);
                if (n == 0) members(1);
                SB();
                acc2[n % DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf[n / (3 * DT)].v, xb[n % RB].v, acc2[n % DT], 0, 0, 0);
            }
            // the discriminant's accumulators stay allocated until here: left to itself the allocator gives them the registers
            // of two sums accumulators and copies those away and back around every tile (64 v_accvgpr_mov)
            asm volatile("; S0, S1 live to here %0 %1" :: "a"(S0), "a"(S1));
            SB();
            __syncthreads();
            cur ^= 1;
        }
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

template <int DT, int CW>
int launch_b3(abw_ctx* ctx, const float* x, int64_t N, int d, const float* dT, int Rp, int chunk, int64_t nchunks, float* slabs)
{
    constexpr int PT = 4 / CW, W = 32 * DT;
    constexpr size_t lds = (size_t)PT * 2 * 3 * W * 64;
    static_assert(lds <= 160 * 1024, "two buffers per point lane must fit the LDS");
    // more than 64 KiB of dynamic LDS has to be asked for, per function and device
    if (hipFuncSetAttribute((const void*)km_step_b3_kernel<DT, CW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 0;
    const dim3 grid((unsigned)std::min<int64_t>((nchunks + PT - 1) / PT, (int64_t)ctx->num_cu * 8), (unsigned)((Rp / 32 + CW - 1) / CW));
    hipLaunchKernelGGL((km_step_b3_kernel<DT, CW>), grid, dim3(256), lds, ctx->stream, x, N, d, dT, Rp, chunk, nchunks, slabs);
    return 1;
}

}  // namespace

// whether abw_km_step_b3 has a kernel for d dimensions and Rp (padded) restarts
int abw_km_b3_has(int d, int Rp)
{
    const int DT = (d + 1 + 31) / 32, tiles = Rp / 32;
    return (Rp & 31) == 0 && ((tiles == 2 && (DT == 5 || DT == 6)) || (tiles >= 3 && DT >= 5 && DT <= 7));
}

// 1: launched; 0: no kernel of this arithmetic for the shape (the caller runs the exact one).  Shapes: 2 restart tiles with
// 128 <= d + 1 <= 192 (two point lanes, 144 KiB), 3 or more restart tiles with 128 <= d + 1 <= 224.
int abw_km_step_b3(abw_ctx* ctx, const float* x, int64_t N, int d, const float* dT, int Rp, int chunk, int64_t nchunks, float* slabs)
{
    const int DT = (d + 1 + 31) / 32, tiles = Rp / 32;
    if (N < 32 || chunk < 32 || (chunk & 31) || !abw_km_b3_has(d, Rp)) return 0;
    if ((int64_t)(d - 1) * N + 31 >= (1ll << 30)) return 0;      // row offsets beyond 32 bits: the exact kernel has a path for them
    if (tiles == 2) {
        if (DT == 5) return launch_b3<5, 2>(ctx, x, N, d, dT, Rp, chunk, nchunks, slabs);
        if (DT == 6) return launch_b3<6, 2>(ctx, x, N, d, dT, Rp, chunk, nchunks, slabs);
    } else if (tiles >= 3) {
        if (DT == 5) return launch_b3<5, 4>(ctx, x, N, d, dT, Rp, chunk, nchunks, slabs);
        if (DT == 6) return launch_b3<6, 4>(ctx, x, N, d, dT, Rp, chunk, nchunks, slabs);
        if (DT == 7) return launch_b3<7, 4>(ctx, x, N, d, dT, Rp, chunk, nchunks, slabs);
    }
    return 0;
}
