// K-means separator on a RESIDENT feature table (SURVEY §8(f) row N2 for the k-means extension):
// abw_km_load puts the fp32 features of every datapoint on the GPU once; abw_km_node runs one node of the
// recursion given only the node's scaffold list -- gather of the node's columns, the batched restarts
// (abw_km_run_dev), the best restart, and the finalisation every separator of the reference shares
// (ref: src/ClusterSeparator.cpp:49-53 smaller raw side = cluster1, :94-101 a scaffold goes where at least half
// of its dps are, :103-122 dps follow their scaffold, :125-132 both children >= cluster_ndps_threshold).
// The reference has no k-means; the SCG gate (ClusterQuality::is_split_better) stays with the caller.
#include "abw_internal.h"
#include <string.h>
#include <algorithm>

namespace {

// global dp index of every dp of the node, scaffold by scaffold in list order
__global__ __launch_bounds__(256) void kmn_dps_kernel(const int32_t* __restrict__ node_scaf, const int32_t* __restrict__ node_dpoff,
                                                     const int32_t* __restrict__ scaf_first, const int32_t* __restrict__ scaf_ndps,
                                                     int32_t m, int32_t* __restrict__ node_dps)
{
    const int ls = blockIdx.x * 256 + threadIdx.x;
    if (ls >= m) return;
    const int s = node_scaf[ls];
    const int f = scaf_first[s], n = scaf_ndps[s], o = node_dpoff[ls];
    for (int i = 0; i < n; i++) node_dps[o + i] = f + i;
}

// x_node[j][i] = x[j][node_dps[i]]
__global__ __launch_bounds__(256) void kmn_gather_kernel(const float* __restrict__ x, int64_t N, const int32_t* __restrict__ node_dps,
                                                        int64_t n, float* __restrict__ x_node)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int j = blockIdx.y;
    x_node[(int64_t)j * n + i] = x[(int64_t)j * N + node_dps[i]];
}

// best restart = smallest inertia, ties to the earlier one; and the number of points it puts in cluster 1
__global__ __launch_bounds__(256) void kmn_best_kernel(const double* __restrict__ inertia, int R, const uint8_t* __restrict__ assign,
                                                      int64_t n, unsigned long long* __restrict__ res)   // res[0] = best, res[1] = ones
{
    __shared__ int best_s;
    __shared__ unsigned long long ones_s;
    if (threadIdx.x == 0) {
        int b = 0;
        for (int r = 1; r < R; r++) if (inertia[r] < inertia[b]) b = r;
        best_s = b; ones_s = 0;
    }
    __syncthreads();
    const uint8_t* a = assign + (int64_t)best_s * n;
    unsigned long long c = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) c += a[i];
    atomicAdd(&ones_s, c);
    __syncthreads();
    if (threadIdx.x == 0) { res[0] = (unsigned long long)best_s; res[1] = ones_s; }
}

// per scaffold of the node: dps on the cluster-1 side, the scaffold's side, raw flags 1/2, totals
// tot[0], tot[1] = dps of the scaffolds sent to cluster1 / cluster2; tot[2] = dps that agree with their scaffold
__global__ __launch_bounds__(256) void kmn_assign_kernel(const int32_t* __restrict__ node_scaf, const int32_t* __restrict__ node_dpoff,
                                                        const int32_t* __restrict__ scaf_first, const int32_t* __restrict__ scaf_ndps,
                                                        int32_t m, int64_t n, const uint8_t* __restrict__ assign,
                                                        const unsigned long long* __restrict__ res, uint8_t* __restrict__ raw,
                                                        uint8_t* __restrict__ side, unsigned long long* __restrict__ tot)
{
    __shared__ unsigned long long b[3];
    if (threadIdx.x < 3) b[threadIdx.x] = 0;
    __syncthreads();
    const long long ones = (long long)res[1];
    const uint8_t first = (ones < (long long)n - ones) ? 1 : 0;         // the library's side that becomes cluster1 (smaller; tie: side 0)
    const uint8_t* a = assign + (int64_t)res[0] * n;
    const int ls = blockIdx.x * 256 + threadIdx.x;
    if (ls < m) {
        const int s = node_scaf ? node_scaf[ls] : ls;
        const int cnt = scaf_ndps[s];
        const int64_t o = node_dpoff ? node_dpoff[ls] : scaf_first[s];
        int k = 0;
        for (int i = 0; i < cnt; i++) { const bool in1 = a[o + i] == first; k += in1; raw[o + i] = in1 ? 1 : 2; }
        const bool to1 = 2 * (long long)k >= cnt;                       // ref :95 (k = 0 never qualifies: cnt >= 2)
        side[ls] = to1 ? 1 : 2;
        atomicAdd(&b[to1 ? 0 : 1], (unsigned long long)cnt);
        atomicAdd(&b[2], (unsigned long long)(to1 ? k : cnt - k));
    }
    __syncthreads();
    if (threadIdx.x < 3 && b[threadIdx.x]) atomicAdd(&tot[threadIdx.x], b[threadIdx.x]);
}

}  // namespace

extern "C" void abw_kmstore_free(abw_ctx* ctx)
{
    abw_km_store& k = ctx->kmst;
    if (k.d_x) (void)hipFree(k.d_x);
    if (k.d_scaf_first) (void)hipFree(k.d_scaf_first);
    if (k.d_scaf_ndps) (void)hipFree(k.d_scaf_ndps);
    k = abw_km_store();
}

extern "C" {

int abw_km_load(abw_ctx* ctx, const float* x, int32_t D, int64_t N, const int32_t* dp2scaf, int32_t S)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!x || !dp2scaf || D < 1 || D > 223 || N < 1 || S < 1 || N > INT32_MAX - 16)
        return abw_fail(ctx, ABW_ERR_ARG, "abw_km_load: bad argument (needs x, dp2scaf, 1 <= D <= 223)");
    abw_device_guard g(ctx->device);
    ABW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    abw_kmstore_free(ctx);
    abw_km_store& k = ctx->kmst;
    k.h_scaf_first.assign(S, 0); k.h_scaf_ndps.assign(S, 0);
    int32_t prev = -1;
    for (int64_t p = 0; p < N; p++) {
        const int32_t sc = dp2scaf[p];
        if (sc < 0 || sc >= S || (sc != prev && sc != prev + 1))
            return abw_fail(ctx, ABW_ERR_ARG, "abw_km_load: dp2scaf must list scaffolds 0..S-1 contiguously and ascending (dp %lld)", (long long)p);
        if (sc != prev) k.h_scaf_first[sc] = (int32_t)p;
        k.h_scaf_ndps[sc]++;
        prev = sc;
    }
    if (prev != S - 1) return abw_fail(ctx, ABW_ERR_ARG, "abw_km_load: dp2scaf covers %d scaffolds, S = %d", prev + 1, S);
    ABW_HIP(ctx, hipMalloc((void**)&k.d_x, (size_t)D * N * 4));
    ABW_HIP(ctx, hipMalloc((void**)&k.d_scaf_first, (size_t)S * 4));
    ABW_HIP(ctx, hipMalloc((void**)&k.d_scaf_ndps, (size_t)S * 4));
    ABW_HIP(ctx, hipMemcpy(k.d_x, x, (size_t)D * N * 4, hipMemcpyHostToDevice));
    ABW_HIP(ctx, hipMemcpy(k.d_scaf_first, k.h_scaf_first.data(), (size_t)S * 4, hipMemcpyHostToDevice));
    ABW_HIP(ctx, hipMemcpy(k.d_scaf_ndps, k.h_scaf_ndps.data(), (size_t)S * 4, hipMemcpyHostToDevice));
    k.D = D; k.N = N; k.S = S; k.loaded = true;
    return ABW_OK;
}

int abw_km_node(abw_ctx* ctx, const int32_t* scaf_ids, int32_t n_scaf, int32_t R, uint32_t base_seed, int32_t iters,
                int32_t chunk, int32_t min_ndps, abw_km_split* out, uint8_t* scaf_side, uint8_t* dp_raw_side)
{
    if (!ctx) return ABW_ERR_ARG;
    abw_km_store& k = ctx->kmst;
    if (!k.loaded) return abw_fail(ctx, ABW_ERR_STATE, "abw_km_node: call abw_km_load first");
    if (!out || R < 1 || iters < 0 || min_ndps < 0 || chunk < 0 || (chunk & 31)) return abw_fail(ctx, ABW_ERR_ARG, "abw_km_node: bad argument (chunk: 0 or a multiple of 32)");
    if (scaf_ids && n_scaf <= 0) return abw_fail(ctx, ABW_ERR_ARG, "abw_km_node: empty node");
    abw_device_guard g(ctx->device);
    hipStream_t st = ctx->stream;
    memset(out, 0, sizeof(*out));
    out->best_restart = -1;
    const int32_t m = scaf_ids ? n_scaf : k.S;
    int64_t n = 0;
    int rc;
    std::vector<int32_t>& dpoff = k.h_node_dpoff;
    if (scaf_ids) {
        dpoff.resize((size_t)m + 1);
        for (int32_t i = 0; i < m; i++) {
            const int32_t sc = scaf_ids[i];
            if (sc < 0 || sc >= k.S) return abw_fail(ctx, ABW_ERR_ARG, "abw_km_node: scaffold id %d out of range", sc);
            if (i > 0 && sc <= scaf_ids[i - 1]) return abw_fail(ctx, ABW_ERR_ARG, "abw_km_node: scaffold ids must be strictly ascending");
            dpoff[i] = (int32_t)n;
            n += k.h_scaf_ndps[sc];
        }
        dpoff[m] = (int32_t)n;
    } else n = k.N;
    out->ndps = n;
    if (scaf_side) memset(scaf_side, 0, (size_t)m);
    if (dp_raw_side) memset(dp_raw_side, 0, (size_t)n);
    if (n < 2 * (int64_t)min_ndps || n < 32) return ABW_OK;             // two children of min_ndps dps cannot exist

    // scratch[6]: node_scaf | node_dpoff | node_dps ; scratch[3]: x_node ; scratch[4]: inertia | res | tot | assign | raw | side
    const size_t ib = (size_t)R * 8, ab = (size_t)R * n;
    if ((rc = abw_reserve(ctx, ctx->scratch[4], ib + 64 + ab + (size_t)n + (size_t)m + 64))) return rc;
    double* d_in = (double*)ctx->scratch[4].p;
    unsigned long long* d_res = (unsigned long long*)((uint8_t*)d_in + ib);      // [0..1] best/ones, [2..4] totals
    uint8_t* d_assign = (uint8_t*)d_res + 64;
    uint8_t* d_raw = d_assign + ab;
    uint8_t* d_side = d_raw + n;
    const float* d_xn = k.d_x;
    const int32_t *d_node_scaf = nullptr, *d_node_dpoff = nullptr;
    if (scaf_ids) {
        k.h_node_ids.assign(scaf_ids, scaf_ids + m);
        if ((rc = abw_reserve(ctx, ctx->scratch[6], ((size_t)2 * m + 1 + (size_t)n) * 4))) return rc;
        if ((rc = abw_reserve(ctx, ctx->scratch[3], (size_t)k.D * n * 4))) return rc;
        int32_t* p_scaf = (int32_t*)ctx->scratch[6].p;
        int32_t* p_dpoff = p_scaf + m;
        int32_t* p_dps = p_dpoff + m + 1;
        ABW_HIP(ctx, hipMemcpyAsync(p_scaf, k.h_node_ids.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
        ABW_HIP(ctx, hipMemcpyAsync(p_dpoff, dpoff.data(), ((size_t)m + 1) * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(kmn_dps_kernel, dim3((m + 255) / 256), dim3(256), 0, st, p_scaf, p_dpoff, k.d_scaf_first, k.d_scaf_ndps, m, p_dps);
        ABW_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(kmn_gather_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)k.D), dim3(256), 0, st, k.d_x, k.N, p_dps, n, (float*)ctx->scratch[3].p);
        ABW_LAUNCH_CHECK(ctx);
        d_xn = (const float*)ctx->scratch[3].p;
        d_node_scaf = p_scaf; d_node_dpoff = p_dpoff;
    }
    // the chunk bounds the fp32 partial sums, i.e. it is part of the result: 0 = sized for this device (abw_km_auto_chunk),
    // and what was used is reported so that the node can be reproduced on another device
    if (chunk == 0) chunk = abw_km_auto_chunk(ctx, n, R);
    out->chunk = chunk;
    if ((rc = abw_km_run_dev(ctx, d_xn, n, k.D, R, base_seed, iters, chunk, d_assign, nullptr, d_in))) return rc;
    ABW_HIP(ctx, hipMemsetAsync(d_res, 0, 64, st));
    hipLaunchKernelGGL(kmn_best_kernel, dim3(1), dim3(256), 0, st, d_in, R, d_assign, n, d_res);
    ABW_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(kmn_assign_kernel, dim3((m + 255) / 256), dim3(256), 0, st, d_node_scaf, d_node_dpoff, k.d_scaf_first, k.d_scaf_ndps,
                       m, n, d_assign, d_res, d_raw, d_side, d_res + 2);
    ABW_LAUNCH_CHECK(ctx);
    unsigned long long res[5] = {0, 0, 0, 0, 0};
    std::vector<double> inertia((size_t)R);
    ABW_HIP(ctx, hipMemcpyAsync(res, d_res, sizeof(res), hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipMemcpyAsync(inertia.data(), d_in, ib, hipMemcpyDeviceToHost, st));
    if (scaf_side) ABW_HIP(ctx, hipMemcpyAsync(scaf_side, d_side, (size_t)m, hipMemcpyDeviceToHost, st));
    if (dp_raw_side) ABW_HIP(ctx, hipMemcpyAsync(dp_raw_side, d_raw, (size_t)n, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    const int64_t ones = (int64_t)res[1];
    out->best_restart = (int32_t)res[0];
    out->inertia = inertia[(size_t)res[0]];
    out->raw1 = std::min(ones, n - ones);
    out->ndps1 = (int64_t)res[2]; out->ndps2 = (int64_t)res[3];
    out->agreement = (double)(int64_t)res[4] / (double)n;
    if (out->raw1 < min_ndps) {                                          // no significant separation (ref :83-84)
        out->ndps1 = out->ndps2 = 0; out->agreement = 0;
        if (scaf_side) memset(scaf_side, 0, (size_t)m);
        if (dp_raw_side) memset(dp_raw_side, 0, (size_t)n);
        return ABW_OK;
    }
    out->split = out->ndps1 >= min_ndps && out->ndps2 >= min_ndps;       // ref :125-132; acceptance is the caller's
    return ABW_OK;
}

}  // extern "C"
