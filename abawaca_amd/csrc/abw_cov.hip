// Row N1 — per-segment read coverage of abawaca-build.
//
// Reference semantics (src/abawaca-build.cpp:231-244 Scaf::add_mapped_read, :177-185
// Scaf_segment::add_mapped_read, read parsed by src/ReadMapping.cpp:38-44): a mapping with 0-based
// position p (= SAM POS - 1) and read length l covers [s, e] = [p, p + l - 1]; it is compared AS IS with
// the 1-based inclusive segment coordinates (start, end) of its scaffold -- the reference's own
// off-by-one, kept -- and every segment with s <= end and e >= start receives
//     double(min(e, end) - max(s, start) + 1) / double(end - start + 1)
// added to its fp64 accumulator IN THE ORDER OF THE SAM FILE.  fp64 addition does not commute in the
// last bit and abawaca.lrn prints int(1000 * x), so the order is part of the result: contributions are
// emitted in read order, brought together per segment WITHOUT changing the order inside a segment, and added by
// one thread per segment in that order (cov_runs marks where every segment's run begins and ends).
// Bringing together: a coordinate-sorted SAM file (the usual input) emits its pairs already grouped by segment --
// detected by one pass over the keys, nothing is moved then.  Otherwise a stable least-significant-digit radix
// sort on the segment index, up to 10 bits per pass (two passes up to a million segments), written for wave64: a wave owns a
// tile of 4096 consecutive pairs and takes them 64 at a time in order; the lanes that hold the same digit find each other
// with one ballot per digit bit, their rank among
// themselves is a population count below the lane, and a per-wave table of running destinations (LDS, started from
// the scanned per-tile digit counts) gives every pair its place -- no atomics on the way, so equal keys keep their order.
// The filter (unmapped / > 15 SNPs / secondary, ref :546) stays with the caller's SAM parser.
#include "abw_internal.h"
#include <string.h>
#include <cstring>

namespace {

// first segment of every scaffold (segments arrive grouped by scaffold, ascending start): seg_first[s] ..
// seg_first[s + 1]; scaffolds without segments get an empty range
__global__ void cov_seg_first_kernel(const uint32_t* __restrict__ seg_scaf, uint64_t n_seg, uint64_t n_scaf, uint64_t* __restrict__ seg_first)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_seg) return;
    const uint64_t prev = (i == 0) ? 0 : (uint64_t)seg_scaf[i - 1] + 1;       // scaffolds up to prev-1 end before segment i
    const uint64_t next = (i == n_seg) ? n_scaf + 1 : (uint64_t)seg_scaf[i] + 1;
    for (uint64_t s = prev; s < next && s <= n_scaf; s++) seg_first[s] = i;    // scaffolds prev..next-1 start at segment i
}

// segments [lo, hi) of the read's scaffold that the read touches: first with end >= s (binary search), then
// forward while start <= e
__device__ __forceinline__ void read_span(const uint64_t* __restrict__ seg_start, const uint64_t* __restrict__ seg_end,
                                          uint64_t f0, uint64_t f1, uint64_t s, uint64_t e, uint64_t& lo, uint64_t& hi)
{
    uint64_t a = f0, b = f1;
    while (a < b) { const uint64_t m = (a + b) >> 1; if (seg_end[m] >= s) b = m; else a = m + 1; }
    lo = a;
    hi = a;
    while (hi < f1 && seg_start[hi] <= e) hi++;
}

// pass 1: the segments [lo, lo + count) every read touches
__global__ __launch_bounds__(256) void cov_span_kernel(const uint32_t* __restrict__ read_scaf, const uint64_t* __restrict__ read_pos,
                                                      const uint32_t* __restrict__ read_len, uint64_t n_reads, uint64_t n_scaf,
                                                      const uint64_t* __restrict__ seg_first, const uint64_t* __restrict__ seg_start,
                                                      const uint64_t* __restrict__ seg_end,
                                                      uint32_t* __restrict__ count, uint32_t* __restrict__ first)
{
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t sc = read_scaf[r];
    uint64_t lo = 0, hi = 0;
    const uint64_t s = read_pos[r], l = read_len[r];
    // l = 0 is not special-cased: e = s + l - 1 wraps in 64 bits exactly as the reference's size_t does (ref :233) -- at
    // s = 0 every segment of the scaffold then receives 1.0, elsewhere a contribution of 0.0
    if (sc < n_scaf) read_span(seg_start, seg_end, seg_first[sc], seg_first[sc + 1], s, s + l - 1, lo, hi);
    count[r] = (uint32_t)(hi - lo);
    first[r] = (uint32_t)lo;
}

// pass 2: (segment, contribution) pairs in read order
__global__ __launch_bounds__(256) void cov_pairs_kernel(const uint64_t* __restrict__ read_pos, const uint32_t* __restrict__ read_len,
                                                       uint64_t n_reads, const uint64_t* __restrict__ seg_start,
                                                       const uint64_t* __restrict__ seg_end, const uint32_t* __restrict__ count,
                                                       const uint32_t* __restrict__ first, const uint64_t* __restrict__ offset,
                                                       uint32_t* __restrict__ key, double* __restrict__ val)
{
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t c = count[r];
    if (c == 0) return;
    const uint64_t s = read_pos[r], e = s + read_len[r] - 1;
    uint64_t o = offset[r];
    const uint64_t lo = first[r];
    for (uint64_t g = lo; g < lo + c; g++, o++) {
        const uint64_t st = seg_start[g], en = seg_end[g];
        const uint64_t a = (s < st) ? st : s, b = (e > en) ? en : e;           // ref :181-182
        key[o] = (uint32_t)g;
        val[o] = (double)(b - a + 1) / (double)(en - st + 1);                  // ref :183
    }
}

// ---- the stable radix sort (see the head of the file) ------------------------------------------------------------------
constexpr int SORT_ROUNDS = 128, SORT_TILE = 64 * SORT_ROUNDS;         // pairs per wave
constexpr int SORT_MAX_DB = 9, SORT_MAX_BINS = 1 << SORT_MAX_DB;      // digit bits per pass

// descents[0] = number of keys smaller than their predecessor (0: the pairs are grouped already)
__global__ __launch_bounds__(256) void cov_sorted_kernel(const uint32_t* __restrict__ key, uint64_t n, unsigned long long* __restrict__ descents)
{
    unsigned long long c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i + 1 < n; i += (uint64_t)gridDim.x * 256) c += key[i + 1] < key[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (c && (threadIdx.x & 63) == 0) atomicAdd(descents, c);           // one per wave of a fixed grid
}

// digit counts of every tile, digit-major: tile_hist[digit * n_tiles + tile]
__global__ __launch_bounds__(256) void cov_digit_hist_kernel(const uint32_t* __restrict__ key, uint64_t n, int shift, int db, uint32_t* __restrict__ tile_hist,
                                                            uint64_t n_tiles)
{
    __shared__ uint32_t h[4][SORT_MAX_BINS];
    const int bins = 1 << db;
    const uint32_t dmask = (uint32_t)bins - 1u;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint64_t tile = (uint64_t)blockIdx.x * 4 + wid;
    if (tile >= n_tiles) return;                                        // whole waves leave; nothing below synchronises the workgroup
    for (int d = lane; d < bins; d += 64) h[wid][d] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int r = 0; r < SORT_ROUNDS; r++) {
        const uint64_t i = tile * SORT_TILE + (uint64_t)r * 64 + lane;
        if (i < n) atomicAdd(&h[wid][(key[i] >> shift) & dmask], 1u);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < bins; d += 64) tile_hist[(uint64_t)d * n_tiles + tile] = h[wid][d];
}

// every pair to its place: tile_off = exclusive scan of tile_hist (the first destination of every (digit, tile))
__global__ __launch_bounds__(256) void cov_digit_scatter_kernel(const uint32_t* __restrict__ key_in, const double* __restrict__ val_in, uint64_t n, int shift,
                                                               int db, const uint64_t* __restrict__ tile_off, uint64_t n_tiles,
                                                               uint32_t* __restrict__ key_out, double* __restrict__ val_out)
{
    __shared__ uint64_t next[4][SORT_MAX_BINS];                         // running destination of every digit in this wave's tile
    const int bins = 1 << db;
    const uint32_t dmask = (uint32_t)bins - 1u;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint64_t tile = (uint64_t)blockIdx.x * 4 + wid;
    if (tile >= n_tiles) return;
    for (int d = lane; d < bins; d += 64) next[wid][d] = tile_off[(uint64_t)d * n_tiles + tile];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int r = 0; r < SORT_ROUNDS; r++) {
        const uint64_t i = tile * SORT_TILE + (uint64_t)r * 64 + lane;
        const bool valid = i < n;
        const uint32_t k = valid ? key_in[i] : 0u;
        const double v = valid ? val_in[i] : 0.0;
        const uint32_t d = (k >> shift) & dmask;
        unsigned long long peers = __ballot(valid);                     // the lanes of this round with my digit
        for (int b = 0; b < db; b++) {
            const unsigned long long has = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? has : ~has;
        }
        const uint64_t base = next[wid][d];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();                                // everybody has read before the leaders write
        const int rank = __popcll(peers & below);
        if (valid && rank == 0) next[wid][d] = base + (uint64_t)__popcll(peers);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid) { key_out[base + rank] = k; val_out[base + rank] = v; }
    }
}

// where the run of every segment begins and ends in the grouped pairs
__global__ __launch_bounds__(256) void cov_runs_kernel(const uint32_t* __restrict__ key, uint64_t n_pairs, uint64_t* __restrict__ run_start,
                                                      uint64_t* __restrict__ run_end)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pairs) return;
    const uint32_t g = key[i];
    if (i == 0 || key[i - 1] != g) run_start[g] = i;
    if (i + 1 == n_pairs || key[i + 1] != g) run_end[g] = i + 1;
}

// one thread per segment: its contributions in (stable-sorted = read) order, onto the caller's accumulator
__global__ __launch_bounds__(256) void cov_sum_kernel(const double* __restrict__ val, const uint64_t* __restrict__ run_start,
                                                     const uint64_t* __restrict__ run_end, uint64_t n_seg, double* __restrict__ cov)
{
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_seg) return;
    const uint64_t a = run_start[g], b = run_end[g];
    if (a >= b) return;
    double acc = cov[g];
    for (uint64_t j = a; j < b; j++) acc += val[j];
    cov[g] = acc;
}

}  // namespace

// the scan of abw_sig.hip (two-level, scratch[8])
int abw_device_scan_u32(abw_ctx* ctx, const uint32_t* in, uint64_t n, uint64_t* out);

extern "C" int abw_cov_accumulate_dev(abw_ctx* ctx, const uint32_t* d_seg_scaf, const uint64_t* d_seg_start, const uint64_t* d_seg_end,
                                      uint64_t n_seg, uint64_t n_scaf, const uint32_t* d_read_scaf, const uint64_t* d_read_pos,
                                      const uint32_t* d_read_len, uint64_t n_reads, double* d_cov)
{
    if (!ctx) return ABW_ERR_ARG;
    if (n_reads == 0 || n_seg == 0) return ABW_OK;
    if (!d_seg_scaf || !d_seg_start || !d_seg_end || !d_read_scaf || !d_read_pos || !d_read_len || !d_cov)
        return abw_fail(ctx, ABW_ERR_ARG, "abw_cov_accumulate_dev: null argument");
    if (n_seg >= 0xFFFFFFFFull || n_reads >= 0xFFFFFFFFull) return abw_fail(ctx, ABW_ERR_ARG, "abw_cov_accumulate_dev: more than 2^32-2 segments or reads per call");
    abw_device_guard g(ctx->device);
    hipStream_t st = ctx->stream;
    int rc;
    // scratch[10]: seg_first (n_scaf+2) u64 | offset u64[n_reads+1] | run_start, run_end u64[n_seg] | count, first u32[n_reads]
    const size_t b_first = (n_scaf + 2) * 8, b_off = (n_reads + 1) * 8, b_run = n_seg * 8, b_count = ((n_reads * 4 + 7) & ~(size_t)7);
    if ((rc = abw_reserve(ctx, ctx->scratch[10], b_first + b_off + 2 * b_run + 2 * b_count))) return rc;
    uint64_t* seg_first = (uint64_t*)ctx->scratch[10].p;
    uint64_t* offset = (uint64_t*)((uint8_t*)seg_first + b_first);
    uint64_t* run_start = (uint64_t*)((uint8_t*)offset + b_off);
    uint64_t* run_end = (uint64_t*)((uint8_t*)run_start + b_run);
    uint32_t* count = (uint32_t*)((uint8_t*)run_end + b_run);
    uint32_t* first = (uint32_t*)((uint8_t*)count + b_count);
    ABW_PROF(ctx, "cov_accumulate");
    hipLaunchKernelGGL(cov_seg_first_kernel, dim3((unsigned)((n_seg + 256) / 256)), dim3(256), 0, st, d_seg_scaf, n_seg, n_scaf, seg_first);
    ABW_LAUNCH_CHECK(ctx);
    const unsigned gr = (unsigned)((n_reads + 255) / 256);
    hipLaunchKernelGGL(cov_span_kernel, dim3(gr), dim3(256), 0, st, d_read_scaf, d_read_pos, d_read_len, n_reads, n_scaf, seg_first,
                       d_seg_start, d_seg_end, count, first);
    ABW_LAUNCH_CHECK(ctx);
    if ((rc = abw_device_scan_u32(ctx, count, n_reads, offset))) return rc;
    uint64_t n_pairs = 0;
    ABW_HIP(ctx, hipMemcpyAsync(&n_pairs, offset + n_reads, 8, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    if (n_pairs == 0) return ABW_OK;
    // scratch[11]: key_in | key_out (u32) | val_in | val_out (f64) | tile_hist u32 + tile_off u64 [256][tiles] | flag
    int bits = 1;
    while (((uint64_t)1 << bits) < n_seg) bits++;
    const uint64_t n_tiles = (n_pairs + SORT_TILE - 1) / SORT_TILE;
    const uint64_t bins_max = (uint64_t)SORT_MAX_BINS;
    const size_t b_key = (n_pairs * 4 + 7) & ~(size_t)7, b_val = n_pairs * 8, b_hist = (bins_max * n_tiles * 4 + 7) & ~(size_t)7, b_toff = (bins_max * n_tiles + 1) * 8;
    if ((rc = abw_reserve(ctx, ctx->scratch[11], 2 * b_key + 2 * b_val + b_hist + b_toff + 256))) return rc;
    uint8_t* base = (uint8_t*)ctx->scratch[11].p;
    uint32_t* key_a = (uint32_t*)base; uint32_t* key_b = (uint32_t*)(base + b_key);
    double* val_a = (double*)(base + 2 * b_key); double* val_b = (double*)(base + 2 * b_key + b_val);
    uint32_t* tile_hist = (uint32_t*)(base + 2 * b_key + 2 * b_val);
    uint64_t* tile_off = (uint64_t*)((uint8_t*)tile_hist + b_hist);
    unsigned long long* d_flag = (unsigned long long*)((uint8_t*)tile_off + b_toff);
    ABW_HIP(ctx, hipMemsetAsync(d_flag, 0, 8, st));
    hipLaunchKernelGGL(cov_pairs_kernel, dim3(gr), dim3(256), 0, st, d_read_pos, d_read_len, n_reads, d_seg_start, d_seg_end, count, first,
                       offset, key_a, val_a);
    ABW_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(cov_sorted_kernel, dim3((unsigned)std::min<uint64_t>((n_pairs + 255) / 256, (uint64_t)ctx->num_cu * 8)), dim3(256), 0, st, (const uint32_t*)key_a, n_pairs, d_flag);
    ABW_LAUNCH_CHECK(ctx);
    unsigned long long unsorted = 0;
    ABW_HIP(ctx, hipMemcpyAsync(&unsorted, d_flag, 8, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    const uint32_t* key_out = key_a;
    const double* val_out = val_a;
    if (unsorted) {
        // Nearly grouped pairs (a coordinate-sorted file whose reads straddle segment borders) stay near their places: few wide
        // passes.  Pairs in no order scatter over the whole array, and a (tile, digit) run of a few pairs wastes most of every
        // cache line it is written to: more passes of fewer digits, longer runs.
        const int max_db = (unsorted * 8 > n_pairs) ? 6 : SORT_MAX_DB;
        const int passes = (bits + max_db - 1) / max_db, db = (bits + passes - 1) / passes;      // e.g. 18 bits: 2 passes of 9, or 3 of 6
        const uint64_t bins = (uint64_t)1 << db;
        const unsigned gt = (unsigned)((n_tiles + 3) / 4);
        for (int shift = 0; shift < bits; shift += db) {
            hipLaunchKernelGGL(cov_digit_hist_kernel, dim3(gt), dim3(256), 0, st, (const uint32_t*)key_a, n_pairs, shift, db, tile_hist, n_tiles);
            ABW_LAUNCH_CHECK(ctx);
            if ((rc = abw_device_scan_u32(ctx, tile_hist, bins * n_tiles, tile_off))) return rc;
            hipLaunchKernelGGL(cov_digit_scatter_kernel, dim3(gt), dim3(256), 0, st, (const uint32_t*)key_a, (const double*)val_a, n_pairs, shift, db,
                               (const uint64_t*)tile_off, n_tiles, key_b, val_b);
            ABW_LAUNCH_CHECK(ctx);
            std::swap(key_a, key_b);
            std::swap(val_a, val_b);
        }
        key_out = key_a;
        val_out = val_a;
    }
    ABW_HIP(ctx, hipMemsetAsync(run_start, 0, 2 * b_run, st));                      // start = end = 0: no contributions
    hipLaunchKernelGGL(cov_runs_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, st, key_out, n_pairs, run_start, run_end);
    ABW_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(cov_sum_kernel, dim3((unsigned)((n_seg + 255) / 256)), dim3(256), 0, st, val_out, run_start, run_end, n_seg, d_cov);
    ABW_LAUNCH_CHECK(ctx);
    return ABW_OK;
}

extern "C" int abw_cov_accumulate(abw_ctx* ctx, const uint32_t* seg_scaf, const uint64_t* seg_start, const uint64_t* seg_end,
                                  uint64_t n_seg, uint64_t n_scaf, const uint32_t* read_scaf, const uint64_t* read_pos,
                                  const uint32_t* read_len, uint64_t n_reads, double* cov)
{
    if (!ctx) return ABW_ERR_ARG;
    if (n_seg == 0 || n_reads == 0) return ABW_OK;
    if (!seg_scaf || !seg_start || !seg_end || !read_scaf || !read_pos || !read_len || !cov)
        return abw_fail(ctx, ABW_ERR_ARG, "abw_cov_accumulate: null argument");
    abw_device_guard g(ctx->device);
    hipStream_t st = ctx->stream;
    int rc;
    const size_t bs = n_seg * (4 + 8 + 8 + 8) + 64, br = n_reads * (4 + 8 + 4) + 64;
    if ((rc = abw_reserve(ctx, ctx->scratch[3], bs + br))) return rc;
    uint8_t* p = (uint8_t*)ctx->scratch[3].p;
    uint64_t* d_start = (uint64_t*)p;              p += n_seg * 8;
    uint64_t* d_end = (uint64_t*)p;                p += n_seg * 8;
    double* d_cov = (double*)p;                    p += n_seg * 8;
    uint64_t* d_rpos = (uint64_t*)p;               p += n_reads * 8;
    uint32_t* d_sscaf = (uint32_t*)p;              p += n_seg * 4;
    uint32_t* d_rscaf = (uint32_t*)p;              p += n_reads * 4;
    uint32_t* d_rlen = (uint32_t*)p;
    ABW_HIP(ctx, hipMemcpyAsync(d_start, seg_start, n_seg * 8, hipMemcpyHostToDevice, st));
    ABW_HIP(ctx, hipMemcpyAsync(d_end, seg_end, n_seg * 8, hipMemcpyHostToDevice, st));
    ABW_HIP(ctx, hipMemcpyAsync(d_cov, cov, n_seg * 8, hipMemcpyHostToDevice, st));
    ABW_HIP(ctx, hipMemcpyAsync(d_sscaf, seg_scaf, n_seg * 4, hipMemcpyHostToDevice, st));
    ABW_HIP(ctx, hipMemcpyAsync(d_rpos, read_pos, n_reads * 8, hipMemcpyHostToDevice, st));
    ABW_HIP(ctx, hipMemcpyAsync(d_rscaf, read_scaf, n_reads * 4, hipMemcpyHostToDevice, st));
    ABW_HIP(ctx, hipMemcpyAsync(d_rlen, read_len, n_reads * 4, hipMemcpyHostToDevice, st));
    if ((rc = abw_cov_accumulate_dev(ctx, d_sscaf, d_start, d_end, n_seg, n_scaf, d_rscaf, d_rpos, d_rlen, n_reads, d_cov))) return rc;
    ABW_HIP(ctx, hipMemcpyAsync(cov, d_cov, n_seg * 8, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    return ABW_OK;
}
