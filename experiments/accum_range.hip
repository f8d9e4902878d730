// Where the time of the narrow accumulation goes at the ROOT (tiles taken as ranges of scaffolds: the dp loop + the scaffold loop of
// accum_group_tile), on synthetic k-mer-like codes.  Variants switch parts off.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I abawaca_amd/csrc -I include experiments/accum_range.hip -o experiments/accum_range
#include "abw_sep_common.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// MODE bits: 1 = dp-loop size atomics, 2 = dp-loop tp1 atomics, 4 = scaffold loop, 8 = flush, 16 = zeroing
template <int MODE, int PACK = 0>
__global__ __launch_bounds__(256) void k_range(const NGroup* groups, const uint4* nk16, const uint4* ts8, const int4* srec, int S, int64_t N, int tile,
                                               unsigned long long* h_st, unsigned long long* h_bp, uint32_t* h_tot, uint32_t* sink)
{
    __shared__ uint32_t s_size[GRP_BINS];
    __shared__ uint32_t s_tp1[GRP_BINS];
    __shared__ uint32_t s_tot[GRP_BINS];
    __shared__ unsigned long long s_bp[(PACK & 1) ? 1 : GRP_BINS];
    __shared__ uint32_t s_bp32[(PACK & 1) ? GRP_BINS : 1];
    const int j = blockIdx.y;
    const NGroup G = groups[j];
    const uint4* codes = nk16 + (int64_t)j * N;
    const uint4* tsv = ts8 + (int64_t)j * S;
    const int begin = blockIdx.x * tile, end = min(S, begin + tile);
    int lg = 0;
    while (lg < 4 && (G.total << (lg + 1)) <= GRP_BINS) lg++;
    const int copies = 1 << lg, mine = threadIdx.x & (copies - 1);
    if (MODE & 16) for (int i = threadIdx.x; i < (G.total << lg); i += 256) { s_size[i] = 0; if (!(PACK & 4)) s_tp1[i] = 0; s_tot[i] = 0; if (PACK & 1) s_bp32[i] = 0; else s_bp[i] = 0; }
    __syncthreads();
    uint32_t acc = 0;
    const int4 r0 = srec[begin], r1 = srec[end - 1];
    for (int dp = r0.x + (int)threadIdx.x; dp < r1.x + r1.y; dp += 256) {
        const uint4 c = codes[dp];
        const uint32_t cw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (q < G.cnt) {
                const uint32_t code = (cw[q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
                const int at = ((G.loff[q] + (int)(code & 0x7FFFu)) << lg) + mine;
                if (PACK & 4) atomicAdd(&s_size[at], 1u + ((code & 0x8000u) << 1));
                else {
                if (MODE & 1) atomicAdd(&s_size[at], 1u); else acc += at;
                if (code & 0x8000u) { if (MODE & 2) atomicAdd(&s_tp1[at], 1u); else acc ^= at; }
                }
            }
        }
    }
    if (MODE & 4)
    for (int s = begin + (int)threadIdx.x; s < end; s += 256) {
        const int4 rec = srec[s];
        const uint4 t8 = tsv[s];
        const int n = rec.y;
        const uint32_t tw[4] = {t8.x, t8.y, t8.z, t8.w};
        const unsigned long long len = (unsigned long long)(uint32_t)rec.z;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (q < G.cnt) {
                const uint32_t e = (tw[q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
                const int ts = (int)(e & 0x3FFu);
                const uint32_t gt = e >> 10;
                const int at = ((G.loff[q] + ts) << lg) + mine;
                if (PACK & 2) atomicAdd(&s_tot[at], (uint32_t)n | (((uint32_t)n - gt) << 16));
                else { atomicAdd(&s_tp1[at], (uint32_t)n - gt); atomicAdd(&s_tot[at], (uint32_t)n); }
                if (PACK & 1) atomicAdd(&s_bp32[at], (uint32_t)len); else atomicAdd(&s_bp[at], len);
            }
        }
    }
    __syncthreads();
    if (MODE & 8)
    for (int b = threadIdx.x; b < G.total; b += 256) {
        unsigned long long a = 0, e = 0;
        uint32_t c3 = 0;
        for (int c = 0; c < copies; c++) {
            const int i = (b << lg) + ((c + b) & (copies - 1));
            if (PACK & 4) { const uint32_t w = s_size[i]; a += (unsigned long long)(w & 0xFFFFu) | ((unsigned long long)(w >> 16) << 32); }
            else a += (unsigned long long)s_size[i] | ((unsigned long long)s_tp1[i] << 32);
            if (PACK & 2) { const uint32_t w = s_tot[i]; c3 += w & 0xFFFFu; a += (unsigned long long)(w >> 16) << 32; } else c3 += s_tot[i];
            e += (PACK & 1) ? (unsigned long long)s_bp32[i] : s_bp[i];
        }
        if (a) atomicAdd(&h_st[G.goff + b], a);
        if (c3) atomicAdd(&h_tot[G.goff + b], c3);
        if (e) atomicAdd(&h_bp[G.goff + b], e);
    }
    if (acc == 0x12345u) sink[0] = acc;
}

// a plain streaming read of the same bytes
__global__ __launch_bounds__(256) void k_stream(const uint4* p, int64_t n, uint32_t* sink)
{
    uint32_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const uint4 c = p[i]; acc += c.x ^ c.y ^ c.z ^ c.w; }
    if (acc == 0x12345u) sink[0] = acc;
}

int main(int argc, char** argv)
{
    const int S = argc > 1 ? atoi(argv[1]) : 200000;
    std::mt19937 rng(1);
    std::vector<int4> srec(S);
    int64_t N = 0;
    for (int s = 0; s < S; s++) { const int n = 2 + rng() % 5; srec[s] = make_int4((int)N, n, 9000, 0); N += n; }
    std::vector<NGroup> groups;
    int64_t goff = 0;
    auto add = [&](int cnt, int span) { NGroup g = NGroup(); g.cnt = cnt; for (int q = 0; q < cnt; q++) { g.loff[q] = g.total; g.toff[q] = g.total; g.total += span; } g.goff = goff; goff += g.total; groups.push_back(g); };
    for (int i = 0; i < 17; i++) add(8, 14);
    for (int i = 0; i < 4; i++) add(8, 40);
    for (int i = 0; i < 2; i++) add(5, 130);
    for (int i = 0; i < 1; i++) add(1, 700);
    const int ng = (int)groups.size();
    std::vector<uint16_t> nk((size_t)ng * N * 8, 0), ts((size_t)ng * S * 8, 0);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int j = 0; j < ng; j++)
        for (int q = 0; q < groups[j].cnt; q++) {
            const int span = (q + 1 < groups[j].cnt ? groups[j].loff[q + 1] : groups[j].total) - groups[j].loff[q];
            for (int s = 0; s < S; s++) {
                int v[8], x[8];
                for (int i = 0; i < srec[s].y; i++) { v[i] = std::min(span - 1, std::max(0, (int)(span * 0.5f + span * 0.12f * nd(rng)))); x[i] = v[i]; }
                std::sort(x, x + srec[s].y);
                const int t = x[srec[s].y >> 1];
                int gt = 0;
                for (int i = 0; i < srec[s].y; i++) gt += v[i] > t;
                ts[((size_t)j * S + s) * 8 + q] = (uint16_t)(t | (gt << 10));
                for (int i = 0; i < srec[s].y; i++) nk[((size_t)j * N + srec[s].x + i) * 8 + q] = (uint16_t)(v[i] | (v[i] > t ? 0x8000 : 0));
            }
        }
    NGroup* d_groups; uint4 *d_nk, *d_ts; int4* d_srec; unsigned long long *d_st, *d_bp; uint32_t *d_tot, *d_sink;
    CK(hipMalloc(&d_groups, ng * sizeof(NGroup))); CK(hipMemcpy(d_groups, groups.data(), ng * sizeof(NGroup), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_nk, nk.size() * 2)); CK(hipMemcpy(d_nk, nk.data(), nk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_ts, ts.size() * 2)); CK(hipMemcpy(d_ts, ts.data(), ts.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_srec, S * 16)); CK(hipMemcpy(d_srec, srec.data(), S * 16, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_st, goff * 8)); CK(hipMalloc(&d_bp, goff * 8)); CK(hipMalloc(&d_tot, goff * 4)); CK(hipMalloc(&d_sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) best = std::min(best, ms);
        }
        printf("%-60s %8.1f us\n", name, best * 1e3);
    };
    printf("%d scaffolds, %lld dps, %d groups, %.0f MB of codes\n", S, (long long)N, ng, N * 16.0 * ng / 1e6);
    timeit("stream read of the codes", [&] { hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, d_nk, (int64_t)N * ng, d_sink); });
    for (int tile : {512, 1024, 2048, 4096}) {
        const dim3 grid((S + tile - 1) / tile, ng);
        char nm[128];
#define RUN(M, what) snprintf(nm, sizeof nm, "tile %4d %s", tile, what); timeit(nm, [&] { hipLaunchKernelGGL(k_range<M>, grid, dim3(256), 0, 0, d_groups, d_nk, d_ts, d_srec, S, N, tile, d_st, d_bp, d_tot, d_sink); });
        RUN(31, "full");
        if (tile == 1024) {
            RUN(16 + 8 + 4 + 1, "no tp1 atomics in the dp loop");
            RUN(16 + 8 + 4, "no atomics in the dp loop");
            RUN(16 + 8 + 3, "no scaffold loop");
            RUN(16 + 8, "loads + zero + flush only");
            RUN(16 + 7, "no flush");
            RUN(0, "loads only");
#define RUNP(P, what) snprintf(nm, sizeof nm, "tile %4d %s", tile, what); timeit(nm, [&] { hipLaunchKernelGGL((k_range<31, P>), grid, dim3(256), 0, 0, d_groups, d_nk, d_ts, d_srec, S, N, tile, d_st, d_bp, d_tot, d_sink); });
            RUNP(1, "full, bp as 32-bit");
            RUNP(3, "full, bp 32-bit, tot | credit in one word");
            RUNP(7, "full, all packed (1 atomic per dp, 2 per scaffold)");
            RUNP(4, "full, only the dp loop packed");
        }
    }
    return 0;
}
