// Diagnostic (not part of the library): what fp32-MFMA rate and shader clock does this MI355X hold
// (a) with operands in registers, (b) with both operands read from LDS every MFMA?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* stamps, int iters)
{
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)((i * 7) % 13) * 0.125f;
    __syncthreads();
    f32x16 acc[6];
    for (int t = 0; t < 6; t++) for (int e = 0; e < 16; e++) acc[t][e] = 0.f;
    const int lane = threadIdx.x & 63;
    float a = lane * 0.01f, b = 1.0f - lane * 0.003f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int t = 0; t < 6; t++) {
                float bb = b;
                if (MODE == 1) bb = lds[((it + u) * 6 + t) * 64 % 8128 + lane];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[t], 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int t = 0; t < 6; t++) for (int e = 0; e < 16; e++) s += acc[t][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE>
void run(const char* name)
{
    const int blocks = 256, iters = 4000;
    float* out; unsigned long long* st;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&st, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        for (int q = 0; q < 20; q++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, st, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 2);
        hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
        double flops = 20.0 * blocks * 4 * (double)iters * 96 * 32 * 32 * 2 * 2;
        double clk = (double)h[0] / (double)h[1] * 100.0;   // MHz
        double cyc_per_mfma = (double)h[0] / (iters * 96.0);
        printf("%s rep %d: %.1f TFLOP/s, block0 shader clock %.0f MHz, %.1f cycles/MFMA\n", name, rep, flops / (ms * 1e-3) / 1e12, clk, cyc_per_mfma);
    }
}

int main() { run<0>("regs"); run<1>("lds "); return 0; }
