// Diagnostic: fp32 MFMA chains fed from LDS with explicit software prefetch (mimics km_step).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* stamps, int iters)
{
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (float)((i * 7) % 13) * 0.125f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc1, acc2[6];
    for (int e = 0; e < 16; e++) acc1[e] = 0.f;
    for (int t = 0; t < 6; t++) for (int e = 0; e < 16; e++) acc2[t][e] = 0.f;
    float ra = lane * 0.01f, rb = 1.0f - lane * 0.003f;
    const float* aop = lds + lane;
    const float* bop = lds + 8192 + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 2) {             // dependent chain, register operands
#pragma unroll
            for (int s = 0; s < 96; s++) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ra, rb, acc1, 0, 0, 0);
        }
        if (MODE == 3 || MODE == 4) { // dependent chain, both operands from LDS, prefetch PF batches of 8 ahead
            constexpr int PF = (MODE == 3) ? 1 : 2;
            float a[PF + 1][8], b[PF + 1][8];
#pragma unroll
            for (int q = 0; q < PF; q++)
#pragma unroll
                for (int u = 0; u < 8; u++) { a[q][u] = aop[64 * (8 * q + u)]; b[q][u] = bop[64 * (8 * q + u)]; }
#pragma unroll
            for (int bt = 0; bt < 12; bt++) {
                if (bt + PF < 12) {
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        a[(bt + PF) % (PF + 1)][u] = aop[64 * (8 * (bt + PF) + u)];
                        b[(bt + PF) % (PF + 1)][u] = bop[64 * (8 * (bt + PF) + u)];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; u++)
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[bt % (PF + 1)][u], b[bt % (PF + 1)][u], acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE == 5) {             // 6 accumulators, b from LDS one step ahead (like the sums product)
            float bb[2][6];
#pragma unroll
            for (int t = 0; t < 6; t++) bb[0][t] = aop[64 * t];
#pragma unroll
            for (int e = 0; e < 16; e++) {
                if (e + 1 < 16) {
#pragma unroll
                    for (int t = 0; t < 6; t++) bb[(e + 1) & 1][t] = aop[64 * (6 * (e + 1) + t)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 6; t++) acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra, bb[e & 1][t], acc2[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE == 6) {             // like 3 but the waits placed by hand: one lgkmcnt per batch
            float a[2][8], b[2][8];
#pragma unroll
            for (int u = 0; u < 8; u++) { a[0][u] = aop[64 * u]; b[0][u] = bop[64 * u]; }
#pragma unroll
            for (int bt = 0; bt < 12; bt++) {
                if (bt + 1 < 12) {
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        a[(bt + 1) & 1][u] = aop[64 * (8 * (bt + 1) + u)];
                        b[(bt + 1) & 1][u] = bop[64 * (8 * (bt + 1) + u)];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int u = 0; u < 8; u++)
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[bt & 1][u], b[bt & 1][u], acc1, 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int e = 0; e < 16; e++) s += acc1[e];
    for (int t = 0; t < 6; t++) for (int e = 0; e < 16; e++) s += acc2[t][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE>
void run(const char* name)
{
    const int blocks = 256, iters = 2000;
    float* out; unsigned long long* st;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&st, blocks * 16);
    for (int rep = 0; rep < 2; rep++) {
        for (int q = 0; q < 10; q++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, st, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 2);
        hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
        printf("%s rep %d: clock %.0f MHz, %.1f cycles/MFMA\n", name, rep, (double)h[0] / (double)h[1] * 100.0, (double)h[0] / (iters * 96.0));
    }
}

int main()
{
    run<2>("chain regs         ");
    run<3>("chain lds pf1      ");
    run<4>("chain lds pf2      ");
    run<5>("6acc  lds pf1      ");
    run<6>("chain lds pf1 prio ");
    return 0;
}
