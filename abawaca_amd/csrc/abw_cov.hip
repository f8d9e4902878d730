// Row N1 — per-segment read coverage of abawaca-build.
//
// Reference semantics (src/abawaca-build.cpp:231-244 Scaf::add_mapped_read, :177-185
// Scaf_segment::add_mapped_read, read parsed by src/ReadMapping.cpp:38-44): a mapping with 0-based
// position p (= SAM POS - 1) and read length l covers [s, e] = [p, p + l - 1]; it is compared AS IS with
// the 1-based inclusive segment coordinates (start, end) of its scaffold -- the reference's own
// off-by-one, kept -- and every segment with s <= end and e >= start receives
//     double(min(e, end) - max(s, start) + 1) / double(end - start + 1)
// added to its fp64 accumulator IN THE ORDER OF THE SAM FILE.  fp64 addition does not commute in the
// last bit and abawaca.lrn prints int(1000 * x), so the order is part of the result.
//
// How the order is kept.  A read touches the consecutive segments first .. first + count - 1 of its scaffold (cov_span_kernel).
//  * A coordinate-sorted file -- `first` never descends from one read to the next, found by one pass over it -- has the
//    reads grouped by their first segment as they come, and the groups follow each other in the file: segment g hears from
//    the groups g - K + 1 .. g in that order (K = the largest count of the call, at most 4 on this path), from group g - k
//    the reads that touch more than k segments.  Nothing is moved: cov_sum_grouped_kernel walks the reads where they lie.
//  * Otherwise every contribution becomes a pair (segment, numerator) written in read order, the pairs are brought together
//    per segment WITHOUT changing the order inside a segment by a stable least-significant-digit radix sort, and one thread
//    per segment adds numerator / width in that order (cov_sum_pairs_kernel).  The numerator (at most the read's length)
//    travels as 32 bits instead of the 64-bit quotient.
// The sort, written for wave64: a workgroup owns a tile of 4096 consecutive keys, each of its four waves a quarter, taken
// 64 at a time in order.  The lanes of a round that hold the same digit find each other with one ballot per digit bit,
// their rank among themselves is a population count below the lane, and a per-wave table of running positions (LDS,
// started from the counts of the waves before) gives every key its place IN THE TILE, sorted by digit: no atomics on the
// way, so equal keys keep their order.  The tile is then written out from LDS in that order -- neighbouring lanes write
// neighbouring words of a (tile, digit) run -- to the places the scanned per-tile digit counts give.
// The filter (unmapped / > 15 SNPs / secondary, ref :546) stays with the caller's SAM parser.
#include "abw_internal.h"
#include <string.h>
#include <cstring>

namespace {

// first segment of every scaffold (segments arrive grouped by scaffold, ascending start): seg_first[s] ..
// seg_first[s + 1]; scaffolds without segments get an empty range.  Also (start, end) of every segment side by side.
__global__ void cov_seg_table_kernel(const uint32_t* __restrict__ seg_scaf, const uint64_t* __restrict__ seg_start, const uint64_t* __restrict__ seg_end,
                                     uint64_t n_seg, uint64_t n_scaf, uint32_t* __restrict__ seg_first, ulonglong2* __restrict__ se,
                                     unsigned long long* __restrict__ too_wide)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_seg) return;
    if (i < n_seg) {
        se[i] = make_ulonglong2(seg_start[i], seg_end[i]);
        if (seg_end[i] - seg_start[i] >= 0xFFFFFFFFull) atomicOr(too_wide, 1ull);       // a numerator must fit 32 bits (cov_pairs_kernel)
    }
    const uint64_t prev = (i == 0) ? 0 : (uint64_t)seg_scaf[i - 1] + 1;       // scaffolds up to prev-1 end before segment i
    const uint64_t next = (i == n_seg) ? n_scaf + 1 : (uint64_t)seg_scaf[i] + 1;
    for (uint64_t s = prev; s < next && s <= n_scaf; s++) seg_first[s] = (uint32_t)i;    // scaffolds prev..next-1 start at segment i
}

__global__ void cov_seg_range_kernel(const uint32_t* __restrict__ seg_first, uint64_t n_scaf, uint2* __restrict__ seg_range)
{
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_scaf) seg_range[s] = make_uint2(seg_first[s], seg_first[s + 1]);
}

// the segments [first, first + count) every read touches: first with end >= s (binary search), then forward while
// start <= e.  Reads of unknown scaffolds: first = n_seg, count = 0.  kmax[0] = the largest count.
__global__ __launch_bounds__(256) void cov_span_kernel(const uint32_t* __restrict__ read_scaf, const uint64_t* __restrict__ read_pos,
                                                      const uint32_t* __restrict__ read_len, uint64_t n_reads, uint64_t n_scaf, uint64_t n_seg,
                                                      const uint2* __restrict__ seg_range, const ulonglong2* __restrict__ se,
                                                      uint32_t* __restrict__ first, uint32_t* __restrict__ count, uint32_t* __restrict__ kmax)
{
    __shared__ uint32_t s_max;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t cnt = 0;
    if (r < n_reads) {
        const uint64_t sc = read_scaf[r];
        const uint64_t s = read_pos[r], l = read_len[r];
        // l = 0 is not special-cased: e = s + l - 1 wraps in 64 bits exactly as the reference's size_t does (ref :233) -- at
        // s = 0 every segment of the scaffold then receives 1.0, elsewhere a contribution of 0.0
        const uint64_t e = s + l - 1;
        uint64_t lo = n_seg;
        if (sc < n_scaf) {
            const uint2 ff = seg_range[sc];
            const uint32_t f0 = ff.x, f1 = ff.y;
            bool done = false;
            const ulonglong2 w0 = se[0];                                 // the same for all: a scalar load
            const uint64_t width0 = w0.y - w0.x + 1;
            const uint32_t window = (width0 - 1 < 0xFFFFFFFFull) ? (uint32_t)width0 : 0xFFFFFFFFu;
            if (f1 > f0 && l > 0 && s > 0 && s < 0xFFFFFFFFull) {
                // A gather costs the memory pipeline a cycle per lane, whatever its width: the usual read (inside ONE window of
                // a scaffold cut in windows of `window` bases from base 1) is settled by one, the window guessed by division.
                const uint32_t gss = min(f0 + ((uint32_t)s - 1u) / window, f1 - 1u);
                const ulonglong2 w = se[gss];
                if (w.x <= s && s <= w.y) {                              // segments are disjoint and ascending: the first with end >= s
                    uint32_t h = gss + 1;
                    if (e > w.y) while (h < f1 && se[h].x <= e) h++;
                    lo = gss;
                    cnt = h - gss;
                    done = true;
                }
            }
            if (!done) {
                uint32_t a = f0, b = f1;
                while (a < b) { const uint32_t m = a + ((b - a) >> 1); if (se[m].y >= s) b = m; else a = m + 1; }
                uint32_t h = a;
                while (h < f1 && se[h].x <= e) h++;
                lo = a;
                cnt = h - a;
            }
        }
        first[r] = (uint32_t)lo;
        count[r] = cnt;
    }
    uint32_t m = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(&s_max, m);
    __syncthreads();
    if (threadIdx.x == 0 && s_max > __hip_atomic_load(kmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(kmax, s_max);
}

// descents[0..64) summed = number of keys smaller than their predecessor (0: grouped already); 8 keys per thread
__global__ __launch_bounds__(256) void cov_descents_kernel(const uint32_t* __restrict__ key, uint64_t n, const uint64_t* __restrict__ n_dev,
                                                          unsigned long long* __restrict__ descents)
{
    if (n_dev) n = *n_dev;                                              // the grid was sized for an upper bound
    __shared__ uint32_t s_c;
    if (threadIdx.x == 0) s_c = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * 2048;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint64_t i = base + (uint64_t)j * 256 + threadIdx.x;
        if (i + 1 < n) c += key[i + 1] < key[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor((int)c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_c, c);
    __syncthreads();
    if (threadIdx.x == 0 && s_c) atomicAdd(&descents[blockIdx.x & 63], (unsigned long long)s_c);
}

// ---- the stable radix sort (see the head of the file) ------------------------------------------------------------------
constexpr int SORT_ROUNDS = 16, SORT_TILE = 256 * SORT_ROUNDS;         // keys per workgroup: four waves x 16 rounds x 64
constexpr int SORT_MAX_DB = 9, SORT_MAX_BINS = 1 << SORT_MAX_DB;      // digit bits per pass

// digit counts of every tile, digit-major: tile_hist[digit * n_tiles + tile]
__global__ __launch_bounds__(256) void cov_digit_hist_kernel(const uint32_t* __restrict__ key, uint64_t n, int shift, int db, uint32_t* __restrict__ tile_hist,
                                                            uint64_t n_tiles)
{
    __shared__ uint32_t h[SORT_MAX_BINS];
    const int bins = 1 << db;
    const uint32_t dmask = (uint32_t)bins - 1u;
    for (int d = threadIdx.x; d < bins; d += 256) h[d] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * SORT_TILE;
#pragma unroll 4
    for (int r = 0; r < SORT_ROUNDS; r++) {
        const uint64_t i = base + (uint64_t)r * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[(key[i] >> shift) & dmask], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < bins; d += 256) tile_hist[(uint64_t)d * n_tiles + blockIdx.x] = h[d];
}

// every key to its place: tile_off = exclusive scan of tile_hist (the first destination of every (digit, tile)).
// pay_in == NULL: the payload is the key's own index (first pass).
__global__ __launch_bounds__(256) void cov_digit_scatter_kernel(const uint32_t* __restrict__ key_in, const uint32_t* __restrict__ pay_in, uint64_t n, int shift,
                                                               int db, const uint64_t* __restrict__ tile_off, uint64_t n_tiles,
                                                               uint32_t* __restrict__ key_out, uint32_t* __restrict__ pay_out)
{
    __shared__ uint32_t s_key[SORT_TILE], s_pay[SORT_TILE];
    __shared__ uint32_t s_run[4][SORT_MAX_BINS];                        // per wave: digit counts, then running places in the tile
    __shared__ uint32_t s_lstart[SORT_MAX_BINS];                        // first place of every digit in the sorted tile
    __shared__ uint64_t s_gbase[SORT_MAX_BINS];                         // its first destination
    __shared__ uint32_t s_wsum[4];
    const int bins = 1 << db;
    const uint32_t dmask = (uint32_t)bins - 1u;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint64_t tile = blockIdx.x, base = tile * SORT_TILE;
    const uint32_t n_here = (uint32_t)((n - base < (uint64_t)SORT_TILE) ? n - base : (uint64_t)SORT_TILE);
    for (int i = threadIdx.x; i < 4 * SORT_MAX_BINS; i += 256) (&s_run[0][0])[i] = 0;
    for (int d = threadIdx.x; d < bins; d += 256) s_gbase[d] = tile_off[(uint64_t)d * n_tiles + tile];
    __syncthreads();
    uint32_t k[SORT_ROUNDS], p[SORT_ROUNDS];
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; r++) {
        const uint32_t idx = (uint32_t)wid * (64 * SORT_ROUNDS) + (uint32_t)r * 64 + lane;
        const bool valid = idx < n_here;
        k[r] = valid ? key_in[base + idx] : 0u;
        p[r] = valid ? (pay_in ? pay_in[base + idx] : (uint32_t)(base + idx)) : 0u;
        if (valid) atomicAdd(&s_run[wid][(k[r] >> shift) & dmask], 1u);
    }
    __syncthreads();
    // places: digits in order, inside a digit the waves in order (two digits per thread: bins <= 512)
    {
        const int d0 = 2 * threadIdx.x, d1 = d0 + 1;
        uint32_t c0[4], c1[4];
#pragma unroll
        for (int w = 0; w < 4; w++) { c0[w] = d0 < bins ? s_run[w][d0] : 0u; c1[w] = d1 < bins ? s_run[w][d1] : 0u; }
        const uint32_t t0 = c0[0] + c0[1] + c0[2] + c0[3], t1 = c1[0] + c1[1] + c1[2] + c1[3];
        uint32_t incl = t0 + t1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += y; }
        if (lane == 63) s_wsum[wid] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wid; w++) before += s_wsum[w];
        uint32_t at = before + incl - (t0 + t1);
        if (d0 < bins) { s_lstart[d0] = at; for (int w = 0; w < 4; w++) { s_run[w][d0] = at; at += c0[w]; } }
        if (d1 < bins) { s_lstart[d1] = at; for (int w = 0; w < 4; w++) { s_run[w][d1] = at; at += c1[w]; } }
    }
    __syncthreads();
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; r++) {
        const uint32_t idx = (uint32_t)wid * (64 * SORT_ROUNDS) + (uint32_t)r * 64 + lane;
        const bool valid = idx < n_here;
        const uint32_t d = (k[r] >> shift) & dmask;
        unsigned long long peers = __ballot(valid);                     // the lanes of this round with my digit
        for (int b = 0; b < db; b++) {
            const unsigned long long has = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? has : ~has;
        }
        const uint32_t at = s_run[wid][d];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();                                // everybody has read before the leaders write
        const int rank = __popcll(peers & below);
        if (valid && rank == 0) s_run[wid][d] = at + (uint32_t)__popcll(peers);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid) { s_key[at + rank] = k[r]; s_pay[at + rank] = p[r]; }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_here; i += 256) {
        const uint32_t kk = s_key[i];
        const uint32_t d = (kk >> shift) & dmask;
        const uint64_t dest = s_gbase[d] + (i - s_lstart[d]);
        key_out[dest] = kk;
        pay_out[dest] = s_pay[i];
    }
}

// where the group of every key value begins and ends in the grouped keys (values 0 .. n_seg).  With `count`: also the
// window [more_lo, more_hi) of every group that holds its reads touching more than one segment (more_lo starts at all
// ones, more_hi at 0; such reads are few, and in a coordinate-sorted file they sit together at the end of their group).
__global__ __launch_bounds__(256) void cov_runs_kernel(const uint32_t* __restrict__ key, uint64_t n, uint32_t* __restrict__ run_start,
                                                      uint32_t* __restrict__ run_end, const uint32_t* __restrict__ count,
                                                      uint32_t* __restrict__ more_lo, uint32_t* __restrict__ more_hi)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = key[i];
    if (i == 0 || key[i - 1] != g) run_start[g] = (uint32_t)i;
    if (i + 1 == n || key[i + 1] != g) run_end[g] = (uint32_t)(i + 1);
    if (count && count[i] > 1) { atomicMin(&more_lo[g], (uint32_t)i); atomicMax(&more_hi[g], (uint32_t)(i + 1)); }
}

// Reads grouped by `first` as they come (no descents: a coordinate-sorted file).  Segment g hears from the groups g - K + 1
// .. g, and as the groups follow each other in the file, so do their contributions: group g - k adds its reads that touch
// more than k segments (k >= 1: the few reads of the windows above, straight from memory), then its own group.
// A wave owns 64 consecutive segments, whose own groups are one run of reads: the run goes through LDS 1024 reads at a time
// -- loaded side by side, every read's contribution to its first segment computed by the lane that loaded it -- and each
// lane then adds the part of its group that is in LDS, in order.
__device__ __forceinline__ void add_reads(const uint64_t* __restrict__ read_pos, const uint32_t* __restrict__ read_len, const uint32_t* __restrict__ count,
                                          uint32_t a, uint32_t b, uint32_t k, uint64_t st, uint64_t en, double width, double& acc)
{
    for (uint32_t r = a; r < b; r++) {
        if (count[r] > k) {
            const uint64_t s = read_pos[r], e = s + (uint64_t)read_len[r] - 1;
            const uint64_t lo = (s < st) ? st : s, hi = (e > en) ? en : e;                  // ref :181-182
            acc += (double)(hi - lo + 1) / width;                                           // ref :183
        }
    }
}

constexpr int GSUM_CHUNK = 1024;

__global__ __launch_bounds__(64) void cov_sum_grouped_kernel(const ulonglong2* __restrict__ se, const uint64_t* __restrict__ read_pos,
                                                            const uint32_t* __restrict__ read_len, const uint32_t* __restrict__ first,
                                                            const uint32_t* __restrict__ count,
                                                            const uint32_t* __restrict__ run_start, const uint32_t* __restrict__ run_end,
                                                            const uint32_t* __restrict__ more_lo, const uint32_t* __restrict__ more_hi,
                                                            uint64_t n_seg, int K, double* __restrict__ cov)
{
    __shared__ double s_val[GSUM_CHUNK];
    const int lane = threadIdx.x;
    const uint64_t g = (uint64_t)blockIdx.x * 64 + lane;
    const bool live = g < n_seg;
    double acc = 0.0;
    uint32_t a = 0, b = 0;
    if (live) {
        const ulonglong2 seg = se[g];
        const double width = (double)(seg.y - seg.x + 1);
        acc = cov[g];
        a = run_start[g]; b = run_end[g];
        for (int k = K - 1; k >= 1; k--) {
            if (g < (uint64_t)k) continue;
            const uint32_t wa = more_lo[g - k], wb = more_hi[g - k];
            if (wa < wb) add_reads(read_pos, read_len, count, wa, wb, (uint32_t)k, seg.x, seg.y, width, acc);
        }
    }
    uint32_t lo = (a < b) ? a : 0xFFFFFFFFu, hi = (a < b) ? b : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, o, 64));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, o, 64));
    }
    for (uint32_t c0 = lo; c0 < hi; c0 += GSUM_CHUNK) {                 // lo, hi: the same in every lane
#pragma unroll 4
        for (int j = 0; j < GSUM_CHUNK / 64; j++) {
            const uint32_t r = c0 + (uint32_t)j * 64 + lane;
            double v = -1.0;                                            // a read that touches nothing
            if (r < hi && count[r] > 0) {
                const ulonglong2 seg = se[first[r]];
                const uint64_t s = read_pos[r], e = s + (uint64_t)read_len[r] - 1;
                const uint64_t x = (s < seg.x) ? seg.x : s, y = (e > seg.y) ? seg.y : e;    // ref :181-182
                v = (double)(y - x + 1) / (double)(seg.y - seg.x + 1);                      // ref :183
            }
            s_val[j * 64 + lane] = v;
        }
        __syncthreads();
        const uint32_t i0 = max(a, c0), i1 = min(b, c0 + (uint32_t)GSUM_CHUNK);
        for (uint32_t i = i0; i < i1; i += 4) {
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = (i + q < i1) ? s_val[i + q - c0] : -1.0;
#pragma unroll
            for (int q = 0; q < 4; q++) if (v[q] >= 0.0) acc += v[q];
        }
        __syncthreads();
    }
    if (live) cov[g] = acc;
}

// ---- reads in no order: one (segment, read) pair per contribution ---------------------------------------------------------
// in read order: the pairs of read r at offset[r] ..; the payload is the numerator of the contribution (at most the read's
// length; the segment's width for the reference's l = 0, s = 0 case: widths fit 32 bits, checked)
__global__ __launch_bounds__(256) void cov_pairs_kernel(const uint32_t* __restrict__ first, const uint32_t* __restrict__ count,
                                                       const uint64_t* __restrict__ read_pos, const uint32_t* __restrict__ read_len, uint64_t n_reads,
                                                       const ulonglong2* __restrict__ se, const uint64_t* __restrict__ offset,
                                                       uint32_t* __restrict__ key, uint32_t* __restrict__ pay)
{
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t c = count[r];
    if (c == 0) return;
    const uint32_t lo = first[r];
    const uint64_t o = offset[r];
    const uint64_t s = read_pos[r], e = s + (uint64_t)read_len[r] - 1;
    for (uint32_t k = 0; k < c; k++) {
        const ulonglong2 seg = se[lo + k];
        const uint64_t a = (s < seg.x) ? seg.x : s, b = (e > seg.y) ? seg.y : e;           // ref :181-182
        key[o + k] = lo + k;
        pay[o + k] = (uint32_t)(b - a + 1);
    }
}

// one thread per segment: its contributions in (stable-sorted = read) order
__global__ __launch_bounds__(256) void cov_sum_pairs_kernel(const uint32_t* __restrict__ num, const uint32_t* __restrict__ run_start,
                                                           const uint32_t* __restrict__ run_end, const ulonglong2* __restrict__ se, uint64_t n_seg,
                                                           double* __restrict__ cov)
{
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_seg) return;
    const uint32_t a = run_start[g], b = run_end[g];
    if (a >= b) return;
    const ulonglong2 seg = se[g];
    const double width = (double)(seg.y - seg.x + 1);
    double acc = cov[g];
    uint32_t j = a;
    for (; j + 4 <= b; j += 4) {
        const uint32_t n0 = num[j], n1 = num[j + 1], n2 = num[j + 2], n3 = num[j + 3];
        acc += (double)n0 / width;                                                         // ref :183
        acc += (double)n1 / width;
        acc += (double)n2 / width;
        acc += (double)n3 / width;
    }
    for (; j < b; j++) acc += (double)num[j] / width;
    cov[g] = acc;
}

}  // namespace

// the scan of abw_sig.hip (two-level, scratch[8])
int abw_device_scan_u32(abw_ctx* ctx, const uint32_t* in, uint64_t n, uint64_t* out);

// keys (n of them, values below 2^bits) grouped stably: *keys_out = the keys in order, *pay_out = their payloads (pay ==
// NULL: where each came from).  Measured on 20 M keys of 18 bits in no order: 3 passes of 6 bits and 2 of 9 take the same time.
size_t abw_sort_work_bytes(uint64_t n)
{
    const uint64_t n_tiles = (n + SORT_TILE - 1) / SORT_TILE;
    return 4 * ((n * 4 + 15) & ~(size_t)15) + (((uint64_t)SORT_MAX_BINS * n_tiles * 4 + 15) & ~(size_t)15) + ((uint64_t)SORT_MAX_BINS * n_tiles + 2) * 8;
}

int abw_stable_sort_u32(abw_ctx* ctx, const uint32_t* keys, const uint32_t* pay, uint64_t n, int bits, uint8_t* work,
                        const uint32_t** keys_out, const uint32_t** pay_out)
{
    hipStream_t st = ctx->stream;
    int rc;
    const uint64_t n_tiles = (n + SORT_TILE - 1) / SORT_TILE;
    const size_t b_key = (n * 4 + 15) & ~(size_t)15, b_hist = ((uint64_t)SORT_MAX_BINS * n_tiles * 4 + 15) & ~(size_t)15;
    uint32_t* key_a = (uint32_t*)work; uint32_t* key_b = (uint32_t*)(work + b_key);
    uint32_t* pay_a = (uint32_t*)(work + 2 * b_key); uint32_t* pay_b = (uint32_t*)(work + 3 * b_key);
    uint32_t* tile_hist = (uint32_t*)(work + 4 * b_key);
    uint64_t* tile_off = (uint64_t*)(work + 4 * b_key + b_hist);
    const int passes = (bits + SORT_MAX_DB - 1) / SORT_MAX_DB, db = (bits + passes - 1) / passes;      // e.g. 19 bits: 3 passes of 7
    const uint64_t bins = (uint64_t)1 << db;
    const uint32_t* kin = keys;
    const uint32_t* pin = pay;
    for (int shift = 0; shift < bits; shift += db) {
        hipLaunchKernelGGL(cov_digit_hist_kernel, dim3((unsigned)n_tiles), dim3(256), 0, st, kin, n, shift, db, tile_hist, n_tiles);
        ABW_LAUNCH_CHECK(ctx);
        if ((rc = abw_device_scan_u32(ctx, tile_hist, bins * n_tiles, tile_off))) return rc;
        hipLaunchKernelGGL(cov_digit_scatter_kernel, dim3((unsigned)n_tiles), dim3(256), 0, st, kin, pin, n, shift, db,
                           (const uint64_t*)tile_off, n_tiles, key_a, pay_a);
        ABW_LAUNCH_CHECK(ctx);
        kin = key_a; pin = pay_a;
        std::swap(key_a, key_b);
        std::swap(pay_a, pay_b);
    }
    *keys_out = kin;
    *pay_out = pin;
    return ABW_OK;
}

extern "C" int abw_cov_accumulate_dev(abw_ctx* ctx, const uint32_t* d_seg_scaf, const uint64_t* d_seg_start, const uint64_t* d_seg_end,
                                      uint64_t n_seg, uint64_t n_scaf, const uint32_t* d_read_scaf, const uint64_t* d_read_pos,
                                      const uint32_t* d_read_len, uint64_t n_reads, double* d_cov)
{
    if (!ctx) return ABW_ERR_ARG;
    if (n_reads == 0 || n_seg == 0) return ABW_OK;
    if (!d_seg_scaf || !d_seg_start || !d_seg_end || !d_read_scaf || !d_read_pos || !d_read_len || !d_cov)
        return abw_fail(ctx, ABW_ERR_ARG, "abw_cov_accumulate_dev: null argument");
    if (n_seg >= 0xFFFFFFFEull || n_reads >= 0xFFFFFFFFull || n_scaf >= 0xFFFFFFFEull)
        return abw_fail(ctx, ABW_ERR_ARG, "abw_cov_accumulate_dev: more than 2^32-3 segments, scaffolds or reads per call");
    abw_device_guard g(ctx->device);
    hipStream_t st = ctx->stream;
    int rc;
    // scratch[10]: se 16 B [n_seg] | flags u64[72] | seg_first u32[n_scaf+2] | run_start, run_end, more_hi, more_lo u32[n_seg+1] |
    //              first, count u32[n_reads] | seg_range 2 x u32[n_scaf]
    auto up16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
    const size_t b_se = n_seg * 16, b_flags = 72 * 8, b_first = up16((n_scaf + 2) * 4), b_run = up16((n_seg + 1) * 4), b_rfirst = up16(n_reads * 4);
    if ((rc = abw_reserve(ctx, ctx->scratch[10], b_se + b_flags + b_first + 4 * b_run + 2 * b_rfirst + n_scaf * 8))) return rc;
    uint8_t* p = (uint8_t*)ctx->scratch[10].p;
    ulonglong2* se = (ulonglong2*)p;                         p += b_se;
    unsigned long long* d_flags = (unsigned long long*)p;   p += b_flags;     // [0, 64) descents, [64] the largest count, [65] a segment too wide
    uint32_t* seg_first = (uint32_t*)p;                      p += b_first;
    uint32_t* run_start = (uint32_t*)p;                      p += b_run;
    uint32_t* run_end = (uint32_t*)p;                        p += b_run;
    uint32_t* more_hi = (uint32_t*)p;                        p += b_run;
    uint32_t* more_lo = (uint32_t*)p;                        p += b_run;
    uint32_t* first = (uint32_t*)p;                          p += b_rfirst;
    uint32_t* count = (uint32_t*)p;                          p += b_rfirst;
    uint2* seg_range = (uint2*)p;
    ABW_PROF(ctx, "cov_accumulate");
    ABW_HIP(ctx, hipMemsetAsync(d_flags, 0, b_flags, st));
    hipLaunchKernelGGL(cov_seg_table_kernel, dim3((unsigned)((n_seg + 256) / 256)), dim3(256), 0, st, d_seg_scaf, d_seg_start, d_seg_end, n_seg, n_scaf,
                       seg_first, se, d_flags + 65);
    ABW_LAUNCH_CHECK(ctx);
    if (n_scaf) {
        hipLaunchKernelGGL(cov_seg_range_kernel, dim3((unsigned)((n_scaf + 255) / 256)), dim3(256), 0, st, (const uint32_t*)seg_first, n_scaf, seg_range);
        ABW_LAUNCH_CHECK(ctx);
    }
    const unsigned gr = (unsigned)((n_reads + 255) / 256), gs = (unsigned)((n_seg + 255) / 256);
    hipLaunchKernelGGL(cov_span_kernel, dim3(gr), dim3(256), 0, st, d_read_scaf, d_read_pos, d_read_len, n_reads, n_scaf, n_seg,
                       (const uint2*)seg_range, (const ulonglong2*)se, first, count, (uint32_t*)(d_flags + 64));
    ABW_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(cov_descents_kernel, dim3((unsigned)((n_reads + 2047) / 2048)), dim3(256), 0, st, (const uint32_t*)first, n_reads,
                       (const uint64_t*)nullptr, d_flags);
    ABW_LAUNCH_CHECK(ctx);
    if ((rc = abw_reserve_pinned(ctx, 68 * 8))) return rc;
    unsigned long long* h_flags = (unsigned long long*)ctx->pinned;             // [66]: the number of pairs
    ABW_HIP(ctx, hipMemcpyAsync(h_flags, d_flags, 66 * 8, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    if (h_flags[65]) return abw_fail(ctx, ABW_ERR_ARG, "abw_cov_accumulate_dev: a segment of 2^32 bases or more");
    const int K = (int)(uint32_t)h_flags[64];
    if (K == 0) return ABW_OK;                                                  // no read touches a segment
    unsigned long long descents = 0;
    for (int i = 0; i < 64; i++) descents += h_flags[i];
    ABW_HIP(ctx, hipMemsetAsync(run_start, 0, 3 * b_run, st));                  // start = end = 0: an empty group; more_hi = 0
    if (descents == 0 && K <= 4) {
        // a coordinate-sorted file: the reads are grouped by their first segment as they come
        ABW_HIP(ctx, hipMemsetAsync(more_lo, 0xFF, b_run, st));
        hipLaunchKernelGGL(cov_runs_kernel, dim3(gr), dim3(256), 0, st, (const uint32_t*)first, n_reads, run_start, run_end, (const uint32_t*)count, more_lo, more_hi);
        ABW_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(cov_sum_grouped_kernel, dim3((unsigned)((n_seg + 63) / 64)), dim3(64), 0, st, (const ulonglong2*)se, d_read_pos, d_read_len,
                           (const uint32_t*)first, (const uint32_t*)count, (const uint32_t*)run_start, (const uint32_t*)run_end,
                           (const uint32_t*)more_lo, (const uint32_t*)more_hi, n_seg, K, d_cov);
        ABW_LAUNCH_CHECK(ctx);
        return ABW_OK;
    }
    // scratch[11]: offset u64[n_reads+1] | key, pay u32[pairs] | the sort's work.  Up to two segments per read (the usual case) the
    // buffers are sized for 2 x n_reads pairs and the number of pairs comes back together with their descents; otherwise it is
    // fetched first (growing the buffer keeps nothing, so the offsets are then made again).
    const size_t b_off = up16((n_reads + 1) * 8);
    const bool bounded = K <= 2 && 2 * n_reads < 0xFFFFFFFFull;
    uint64_t n_pairs = bounded ? 2 * n_reads : 0;
    if (!bounded) {
        if ((rc = abw_reserve(ctx, ctx->scratch[11], b_off))) return rc;
        if ((rc = abw_device_scan_u32(ctx, count, n_reads, (uint64_t*)ctx->scratch[11].p))) return rc;
        ABW_HIP(ctx, hipMemcpyAsync(h_flags + 66, (uint64_t*)ctx->scratch[11].p + n_reads, 8, hipMemcpyDeviceToHost, st));
        ABW_HIP(ctx, hipStreamSynchronize(st));
        n_pairs = h_flags[66];
        if (n_pairs >= 0xFFFFFFFFull) return abw_fail(ctx, ABW_ERR_ARG, "abw_cov_accumulate_dev: more than 2^32-2 contributions per call");
    }
    const size_t b_key = up16(n_pairs * 4), b_all = b_off + 2 * b_key + abw_sort_work_bytes(n_pairs);
    if (bounded || ctx->scratch[11].cap < b_all) {
        if ((rc = abw_reserve(ctx, ctx->scratch[11], b_all))) return rc;
        if ((rc = abw_device_scan_u32(ctx, count, n_reads, (uint64_t*)ctx->scratch[11].p))) return rc;
    }
    uint64_t* offset = (uint64_t*)ctx->scratch[11].p;
    uint32_t* pkey = (uint32_t*)((uint8_t*)offset + b_off);
    uint32_t* ppay = (uint32_t*)((uint8_t*)pkey + b_key);
    uint8_t* work = (uint8_t*)ppay + b_key;
    hipLaunchKernelGGL(cov_pairs_kernel, dim3(gr), dim3(256), 0, st, (const uint32_t*)first, (const uint32_t*)count, d_read_pos, d_read_len, n_reads,
                       (const ulonglong2*)se, (const uint64_t*)offset, pkey, ppay);
    ABW_LAUNCH_CHECK(ctx);
    const uint32_t* keys = pkey;
    const uint32_t* pays = ppay;
    // one synchronisation: the number of pairs and their descents (with one segment per read the pairs are the reads that touch
    // a segment, in the same order: the descents of the reads stand)
    if (K > 1) {
        ABW_HIP(ctx, hipMemsetAsync(d_flags, 0, 64 * 8, st));
        hipLaunchKernelGGL(cov_descents_kernel, dim3((unsigned)((n_pairs + 2047) / 2048)), dim3(256), 0, st, (const uint32_t*)pkey, n_pairs,
                           (const uint64_t*)(offset + n_reads), d_flags);
        ABW_LAUNCH_CHECK(ctx);
        ABW_HIP(ctx, hipMemcpyAsync(h_flags, d_flags, 64 * 8, hipMemcpyDeviceToHost, st));
    }
    ABW_HIP(ctx, hipMemcpyAsync(h_flags + 66, offset + n_reads, 8, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    if (K > 1) {
        descents = 0;
        for (int i = 0; i < 64; i++) descents += h_flags[i];
    }
    n_pairs = h_flags[66];
    if (n_pairs == 0) return ABW_OK;
    if (descents) {
        int bits = 1;
        while (((uint64_t)1 << bits) < n_seg) bits++;                           // key values 0 .. n_seg - 1
        if ((rc = abw_stable_sort_u32(ctx, pkey, ppay, n_pairs, bits, work, &keys, &pays))) return rc;
    }
    const unsigned gp = (unsigned)((n_pairs + 255) / 256);
    hipLaunchKernelGGL(cov_runs_kernel, dim3(gp), dim3(256), 0, st, keys, n_pairs, run_start, run_end, (const uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr);
    ABW_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(cov_sum_pairs_kernel, dim3(gs), dim3(256), 0, st, pays, (const uint32_t*)run_start, (const uint32_t*)run_end, (const ulonglong2*)se, n_seg, d_cov);
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
