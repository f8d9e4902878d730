// Internal definitions shared by the translation units of libabawaca_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include "../../include/abawaca_hip.h"

#define ABW_ABI_VERSION 2

// Grow-only device buffer owned by a context.
struct abw_devbuf {
    void*  p = nullptr;
    size_t cap = 0;
};

struct abw_sep_state {
    bool      loaded = false;
    int32_t   D = 0;
    int64_t   N = 0;
    int32_t   S = 0;
    int32_t   n_scg = 0;
    int32_t*  d_keys = nullptr;       // [D][N] original keys
    int32_t*  d_bins = nullptr;       // [D][N] bin index per dp (== keys for narrow dims, rank for wide)
    int32_t*  d_scaf_first = nullptr; // [S]
    int32_t*  d_scaf_ndps = nullptr;  // [S]
    int64_t*  d_scaf_len = nullptr;   // [S]
    int32_t*  d_scg_off = nullptr;    // [S+1]
    int32_t*  d_scg_ids = nullptr;
    int32_t*  d_T = nullptr;          // [D] bins per dimension
    int64_t*  d_Toff = nullptr;       // [D+1] offsets into the histogram arrays
    int32_t*  d_binval = nullptr;     // [sum T over wide dims] bin -> key (wide dims only)
    int64_t*  d_binval_off = nullptr; // [D] offset into d_binval, -1 for narrow dims
    std::vector<int32_t> h_T;
    std::vector<int64_t> h_Toff;
    std::vector<int64_t> h_binval_off;
    std::vector<int32_t> h_binval;
    std::vector<int32_t> h_scaf_first, h_scaf_ndps;
    std::vector<uint32_t> h_stamp; uint32_t stamp = 0;   // duplicate check of a node's scaffold list
    int64_t   TB = 0;                 // total bins
    bool      node_open = false, node_all = false;   // a node accumulated and not yet evaluated
    int32_t   node_m = 0;             // its scaffolds held here
    int64_t   node_n = 0;             // and their dps
    // per-node scratch
    abw_devbuf hist;                  // 3 x int32[TB] + int64[TB]
    abw_devbuf scg_lohi;              // 2 x int32[D][n_scg]
    abw_devbuf node_scaf;             // int32[m] scaffold ids, then int32[m+1] dp offsets (one upload)
    abw_devbuf dimbest;               // everything a node returns: winners | global winner | totals | scaffold sides | raw flags
};

struct abw_prof_rec { int id; hipEvent_t a, b; };
struct abw_prof {
    bool on = false;
    uint32_t period = 1;                    // every period-th scope of a name is bracketed (abw_prof_enable(ctx, period))
    std::vector<uint32_t> seen;             // scopes of names[i] met since the last reset
    std::vector<std::string> names;
    std::vector<abw_prof_rec> recs;
    std::vector<hipEvent_t> pool;
};

// k-means run in progress (abw_km_shard_* phase API; abw_km_run_dev drives the same phases itself)
struct abw_km_state {
    bool active = false, single = false;    // single: abw_km_run_dev drives the phases itself, no exchange
    const float* x = nullptr;
    int64_t N = 0, first = 0, N_total = 0, nchunks = 0, ngroups = 0, elems = 0;
    int d = 0, R = 0, Rp = 0, W = 0, DT = 0, chunk = 0;
    uint32_t seed = 0;
    double *col_term = nullptr, *moments = nullptr, *sums = nullptr, *gsum = nullptr;   // moments = tot[W] then x2[d]
    float *cent = nullptr, *dT = nullptr, *slabs = nullptr;
};

// resident feature table of the k-means separator (abw_km_load / abw_km_node)
struct abw_km_store {
    bool loaded = false;
    int32_t D = 0, S = 0;
    int64_t N = 0;
    float* d_x = nullptr;                 // [D][N]
    int32_t *d_scaf_first = nullptr, *d_scaf_ndps = nullptr;
    std::vector<int32_t> h_scaf_first, h_scaf_ndps, h_node_ids, h_node_dpoff;
};

struct abw_sig_pipe;                   // abw_sig.hip: streams, pinned staging ring and events of the pipelined abw_sig_build

struct abw_ctx {
    int         device = 0;
    abw_km_state km;
    abw_km_store kmst;
    abw_prof    prof;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    abw_sep_state sep;
    std::vector<int32_t> sep_min_bins; // abw_sep_narrow_bins: bins per narrow column for the next load (shared histogram layout)
    abw_devbuf  scratch[12];           // general-purpose grow-only scratch (valid only inside one entry-point call)
    uint64_t    scratch_epoch = 0;     // counts the reservations of scratch[] made outside a recursion's own phases: the phase
    bool        scratch_quiet = false; // API of abw_sep_tree keeps its level in scratch[] between calls and checks the count
    abw_devbuf  km_buf[3];             // state of a k-means run in progress (abw_km_shard_*): lives across calls
    int         num_cu = 256;
    int         km_arith = 0;          // ABW_KM_ARITH_*: what abw_km_shard_step's non-final passes compute in (abw_km_set_arith)
    void*       pinned = nullptr;      // grow-only pinned host staging (results of a call leave the GPU in one batch)
    size_t      pinned_cap = 0;
    abw_sig_pipe* sig_pipe = nullptr;
    void*       sig_static = nullptr;  // this device's address of abw_sig.hip's static workspace (abw_sig_segment_static)
    abw_devbuf  sam_names;             // resident hash table of the scaffold names (abw_sam_names_load): slot_hash | name_off | slot_id | blob
    uint64_t    sam_n_scaf = 0, sam_slots = 0;   // sam_slots = 0: none loaded
    abw_devbuf  lrn_table;             // resident [ld][n_cols] doubles of the cluster writer (abw_lrn_table_load[_dev]: transposed at load)
    int64_t     lrn_ld = 0;
    int32_t     lrn_cols = 0;
};

int abw_fail(abw_ctx* ctx, int code, const char* fmt, ...);
int abw_reserve(abw_ctx* ctx, abw_devbuf& b, size_t bytes);
void abw_release(abw_devbuf& b);
int abw_reserve_pinned(abw_ctx* ctx, size_t bytes);

#define ABW_HIP(ctx, call)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return abw_fail((ctx), ABW_ERR_HIP, "%s failed: %s (%s:%d)", #call,                 \
                            hipGetErrorString(e_), __FILE__, __LINE__);                         \
    } while (0)

#define ABW_LAUNCH_CHECK(ctx)                                                                   \
    do {                                                                                        \
        hipError_t e_ = hipGetLastError();                                                      \
        if (e_ != hipSuccess)                                                                   \
            return abw_fail((ctx), ABW_ERR_HIP, "kernel launch failed: %s (%s:%d)",             \
                            hipGetErrorString(e_), __FILE__, __LINE__);                         \
    } while (0)

// Brackets a kernel launch with events when profiling is on (see abw_prof_* in abw_ctx.hip).
struct abw_prof_scope {
    abw_ctx* ctx; hipEvent_t b = nullptr;
    abw_prof_scope(abw_ctx* c, const char* name);
    ~abw_prof_scope();
};
#define ABW_CAT2(a, b) a##b
#define ABW_CAT(a, b) ABW_CAT2(a, b)
#define ABW_PROF(ctx, name) abw_prof_scope ABW_CAT(prof_scope_, __LINE__)((ctx), (name))

// abw_cov.hip: stable least-significant-digit radix sort of n 32-bit keys below 2^bits with 32-bit payloads (pay == NULL: the
// key's own index), on ctx->stream.  work: abw_sort_work_bytes(n) bytes of device memory, 16-byte aligned; the results
// (*keys_out, *pay_out) lie inside it.  Uses ctx->scratch[8] (the scan's temporary).
size_t abw_sort_work_bytes(uint64_t n);
int abw_stable_sort_u32(abw_ctx* ctx, const uint32_t* keys, const uint32_t* pay, uint64_t n, int bits, uint8_t* work,
                        const uint32_t** keys_out, const uint32_t** pay_out);
int abw_device_scan_u32(abw_ctx* ctx, const uint32_t* in, uint64_t n, uint64_t* out);      // abw_sig.hip: exclusive, out[n] = the total

// RAII device guard: every entry point runs with the context's device current.
struct abw_device_guard {
    int prev = -1;
    explicit abw_device_guard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~abw_device_guard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// abw_km_b3.hip: the k-means step on three bf16 planes; 1 = launched on ctx->stream, 0 = no such kernel for this shape
int abw_km_b3_has(int d, int Rp);
int abw_km_step_b3(abw_ctx* ctx, const float* x, int64_t N, int d, const float* dT, int Rp, int chunk, int64_t nchunks, float* slabs);

// ---- wave64 prefix sums in DPP steps (round 3) -------------------------------------------------------------------------
// Rows of 16 lanes: shifts by 1, 2, 4, 8 (a lane without a source adds 0); then lane 15 of a row to the next row and lane 31
// to the upper half.  Six moves and adds per 32 bits, where __shfl_up costs a trip through the LDS crossbar per step --
// the one-wave scans of the recursion (tile sums of a node, partition counts) are chains of exactly these, and the
// parsers and writers of the .lrn files take the field offsets of a row from them.  Every lane of the wave must be active.
template <int CTRL, int ROWS, bool BOUND>
__device__ __forceinline__ uint32_t dpp_from_u32(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROWS, 0xF, BOUND); }
template <int CTRL, int ROWS, bool BOUND>
__device__ __forceinline__ unsigned long long dpp_from_u64(unsigned long long x)
{
    return (unsigned long long)dpp_from_u32<CTRL, ROWS, BOUND>((uint32_t)(x >> 32)) << 32 | dpp_from_u32<CTRL, ROWS, BOUND>((uint32_t)x);
}
__device__ __forceinline__ uint32_t wave_prefix_u32(uint32_t x)                     // inclusive
{
    x += dpp_from_u32<0x111, 0xF, true>(x); x += dpp_from_u32<0x112, 0xF, true>(x); x += dpp_from_u32<0x114, 0xF, true>(x);
    x += dpp_from_u32<0x118, 0xF, true>(x); x += dpp_from_u32<0x142, 0xA, false>(x); x += dpp_from_u32<0x143, 0xC, false>(x);
    return x;
}
__device__ __forceinline__ unsigned long long wave_prefix_u64(unsigned long long x)   // inclusive
{
    x += dpp_from_u64<0x111, 0xF, true>(x); x += dpp_from_u64<0x112, 0xF, true>(x); x += dpp_from_u64<0x114, 0xF, true>(x);
    x += dpp_from_u64<0x118, 0xF, true>(x); x += dpp_from_u64<0x142, 0xA, false>(x); x += dpp_from_u64<0x143, 0xC, false>(x);
    return x;
}
__device__ __forceinline__ unsigned long long wave_last_u64(unsigned long long x)     // lane 63's value, wave-uniform
{
    return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), 63) << 32 | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, 63);
}
__device__ __forceinline__ uint32_t wave_last_u32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)x, 63); }
// the value of the lane above / below (the last / first lane keeps its own, as __shfl_down / __shfl_up by one do)
__device__ __forceinline__ uint32_t lane_above(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x130, 0xF, 0xF, false); }   // wave_shl:1
__device__ __forceinline__ uint32_t lane_below(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x138, 0xF, 0xF, false); }   // wave_shr:1
