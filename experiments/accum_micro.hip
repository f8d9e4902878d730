// Micro-benchmark of the narrow accumulation (accum_group_tile) on synthetic k-mer-like data: which part costs what.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I abawaca_amd/csrc -I include experiments/accum_micro.hip -o experiments/accum_micro
#include "abw_sep_common.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>   // 0 full, 1 no LDS atomics, 2 no flush, 3 loads only (1 + 2)
__global__ __launch_bounds__(256) void k_accum(const NGroup* groups, const uint4* nk16, const uint4* ts8, const int4* srec, int S, int64_t N,
                                               const int32_t* list, int m, int tile, unsigned long long* h_st, unsigned long long* h_bp, uint32_t* h_tot,
                                               int64_t TBn, uint32_t* sink)
{
    const int j = blockIdx.y;
    const NGroup G = groups[j];
    const int begin = blockIdx.x * tile, end = min(m, begin + tile);
    SepDev P; P.n_scg = 0;
    if (MODE == 0 || MODE == 2) {
        accum_group_tile<false>(G, nk16 + (int64_t)j * N, ts8 + (int64_t)j * S, srec, list, begin, end, P, nullptr, nullptr,
                         [&](int b, unsigned long long a, uint32_t c3, unsigned long long e) {
                             if (MODE == 2) { if (a == 0xdeadbeefull) sink[0] = c3; return; }
                             if (a) atomicAdd(&h_st[G.goff + b], a);
                             if (c3) atomicAdd(&h_tot[G.goff + b], c3);
                             if (e) atomicAdd(&h_bp[G.goff + b], e);
                         });
    } else {
        const uint4* codes = nk16 + (int64_t)j * N;
        const uint4* tsv = ts8 + (int64_t)j * S;
        uint32_t acc = 0;
        for (int ls = begin + threadIdx.x; ls < end; ls += 256) {
            const int s = list ? list[ls] : ls;
            const int4 rec = srec[s];
            const uint4 t8 = tsv[s];
            for (int i = 0; i < rec.y; i++) { const uint4 c = codes[rec.x + i]; acc += c.x ^ c.y ^ c.z ^ c.w; }
            acc += t8.x ^ t8.w;
        }
        if (acc == 0x12345u) sink[0] = acc;
    }
}

int main(int argc, char** argv)
{
    const int S = argc > 1 ? atoi(argv[1]) : 200000;
    const int stride = argc > 2 ? atoi(argv[2]) : 1;
    const int tile = argc > 3 ? atoi(argv[3]) : TILE_A;           // node = every stride-th scaffold (scattered nodes)
    std::mt19937 rng(1);
    std::vector<int4> srec(S);
    int64_t N = 0;
    for (int s = 0; s < S; s++) { const int n = 2 + rng() % 5; srec[s] = make_int4((int)N, n, 9000, 0); N += n; }
    // 25 groups: 17 of 8 tetramers (span 14), 4 of 8 trimers (span 40), 2 of 5 dimers (span 130), 2 single monomers (span 700)
    std::vector<NGroup> groups;
    int64_t goff = 0;
    auto add = [&](int cnt, int span) { NGroup g = NGroup(); g.cnt = cnt; for (int q = 0; q < cnt; q++) { g.loff[q] = g.total; g.toff[q] = g.total; g.total += span; } g.goff = goff; goff += g.total; groups.push_back(g); };
    for (int i = 0; i < 17; i++) add(8, 14);
    for (int i = 0; i < 4; i++) add(8, 40);
    for (int i = 0; i < 2; i++) add(5, 130);
    for (int i = 0; i < 2; i++) add(1, 700);
    const int ng = (int)groups.size();
    std::vector<uint16_t> nk((size_t)ng * N * 8, 0), ts((size_t)ng * S * 8, 0);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int j = 0; j < ng; j++)
        for (int q = 0; q < groups[j].cnt; q++) {
            const int span = (q + 1 < groups[j].cnt ? groups[j].loff[q + 1] : groups[j].total) - groups[j].loff[q];
            for (int s = 0; s < S; s++) {
                const int t = std::min(span - 1, std::max(0, (int)(span * 0.5f + span * 0.12f * nd(rng))));
                ts[((size_t)j * S + s) * 8 + q] = (uint16_t)t;
                for (int i = 0; i < srec[s].y; i++) {
                    const int v = std::min(span - 1, std::max(0, (int)(span * 0.5f + span * 0.12f * nd(rng))));
                    nk[((size_t)j * N + srec[s].x + i) * 8 + q] = (uint16_t)(v | (v > t ? 0x8000 : 0));
                }
            }
        }
    std::vector<int32_t> list;
    for (int s = 0; s < S; s += stride) list.push_back(s);
    const int m = (int)list.size();
    NGroup* d_groups; uint4 *d_nk, *d_ts; int4* d_srec; int32_t* d_list; unsigned long long *d_st, *d_bp; uint32_t *d_tot, *d_sink;
    CK(hipMalloc(&d_groups, ng * sizeof(NGroup))); CK(hipMemcpy(d_groups, groups.data(), ng * sizeof(NGroup), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_nk, nk.size() * 2)); CK(hipMemcpy(d_nk, nk.data(), nk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_ts, ts.size() * 2)); CK(hipMemcpy(d_ts, ts.data(), ts.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_srec, S * 16)); CK(hipMemcpy(d_srec, srec.data(), S * 16, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_list, m * 4)); CK(hipMemcpy(d_list, list.data(), m * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_st, goff * 8)); CK(hipMalloc(&d_bp, goff * 8)); CK(hipMalloc(&d_tot, goff * 4)); CK(hipMalloc(&d_sink, 64));
    CK(hipMemset(d_st, 0, goff * 8)); CK(hipMemset(d_bp, 0, goff * 8)); CK(hipMemset(d_tot, 0, goff * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid((m + tile - 1) / tile, ng);
    const char* names[4] = {"full", "no LDS atomics (loads only)", "no flush", "loads only"};
    for (int use_list = (stride == 1 ? 0 : 1); use_list < 2; use_list++)
    for (int mode : {0, 2, 3}) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_accum<0>, grid, dim3(256), 0, 0, d_groups, d_nk, d_ts, d_srec, S, N, (use_list ? d_list : nullptr), m, tile, d_st, d_bp, d_tot, goff, d_sink);
            if (mode == 2) hipLaunchKernelGGL(k_accum<2>, grid, dim3(256), 0, 0, d_groups, d_nk, d_ts, d_srec, S, N, (use_list ? d_list : nullptr), m, tile, d_st, d_bp, d_tot, goff, d_sink);
            if (mode == 3) hipLaunchKernelGGL(k_accum<3>, grid, dim3(256), 0, 0, d_groups, d_nk, d_ts, d_srec, S, N, (use_list ? d_list : nullptr), m, tile, d_st, d_bp, d_tot, goff, d_sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) best = std::min(best, ms);
        }
        int64_t dps = 0; for (int s : list) dps += srec[s].y;
        printf("%s tile %4d %-30s %8.1f us   (%d scaffolds of %d, %lld dps, %d groups, %.2f TB/s of 16-byte codes)\n", use_list ? "list " : "range", tile, names[mode], best * 1e3, m, S, (long long)dps, ng,
               dps * 16.0 * ng / (best * 1e-3) / 1e12);
    }
    return 0;
}
