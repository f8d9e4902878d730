// Does the k-means step's read pattern cost HBM bandwidth?  x is [d][N] fp32; a 32-point tile is d segments of 128 B,
// one per row, rows N*4 bytes apart.  (a) that pattern, one workgroup per chunk of consecutive tiles, thread (sq, sr)
// reading 16 B of row sr + 32k as in km_step_kernel; (b) the same bytes as one contiguous stream per workgroup.
// build: hipcc --offload-arch=gfx950 -O3 -o experiments/xread_pattern experiments/xread_pattern.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef f4 f4u __attribute__((aligned(4)));

__global__ __launch_bounds__(256) void tiled(const float* __restrict__ x, int64_t N, int d, int tiles_per_wg, float* out)
{
    const int sq = threadIdx.x & 7, sr = threadIdx.x >> 3;
    const int64_t p0 = (int64_t)blockIdx.x * tiles_per_wg * 32;
    f4 acc = {0, 0, 0, 0};
    for (int t = 0; t < tiles_per_wg; t++) {
        const int64_t p = p0 + 32 * (int64_t)t + 4 * sq;
        if (p + 4 > N) break;
        for (int r = sr; r < d; r += 32) acc += *(const f4u*)(x + (int64_t)r * N + p);
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = acc.x;
}
__global__ __launch_bounds__(256) void stream(const float* __restrict__ x, int64_t total, int64_t per_wg, float* out)
{
    const int64_t b = (int64_t)blockIdx.x * per_wg, e = b + per_wg < total ? b + per_wg : total;
    f4 acc = {0, 0, 0, 0};
    for (int64_t i = b + 4 * threadIdx.x; i + 4 <= e; i += 1024) acc += *(const f4*)(x + i);
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = acc.x;
}
int main()
{
    const int64_t N = 3999552; const int d = 199;
    float *x, *out; hipMalloc(&x, N * d * 4); hipMalloc(&out, 64); hipMemset(x, 0, N * d * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int tiles : {62, 16, 4}) {
        const int wgs = (int)((N / 32 + tiles - 1) / tiles);
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a); hipLaunchKernelGGL(tiled, dim3(wgs), dim3(256), 0, 0, x, N, d, tiles, out); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("tiled  %3d tiles/wg %6d wgs: %.3f ms  %.2f TB/s\n", tiles, wgs, ms, N * d * 4.0 / ms / 1e9);
        }
    }
    for (int wgs : {2048, 8192}) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a); hipLaunchKernelGGL(stream, dim3(wgs), dim3(256), 0, 0, x, N * d, (N * d / wgs + 3) & ~3ll, out); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("stream %6d wgs: %.3f ms  %.2f TB/s\n", wgs, ms, N * d * 4.0 / ms / 1e9);
        }
    }
    return 0;
}
