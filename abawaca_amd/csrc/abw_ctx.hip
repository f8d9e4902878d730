// Context management for libabawaca_hip.so.
#include "abw_internal.h"
#include <stdarg.h>

int abw_sig_static_init(abw_ctx* ctx);   // abw_sig.hip
void abw_sig_pipe_free(abw_ctx* ctx);    // abw_sig.hip

int abw_fail(abw_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else fprintf(stderr, "abawaca_hip: %s\n", buf);
    return code;
}

int abw_reserve(abw_ctx* ctx, abw_devbuf& b, size_t bytes)
{
    // whoever reserves one of the context's scratch buffers is about to write it (abw_tree.hip: TreeGuard)
    if (!ctx->scratch_quiet && &b >= &ctx->scratch[0] && &b < &ctx->scratch[12]) ctx->scratch_epoch++;
    if (bytes <= b.cap) return ABW_OK;
    if (b.p) { ABW_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return abw_fail(ctx, ABW_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    b.cap = want;
    return ABW_OK;
}

int abw_reserve_pinned(abw_ctx* ctx, size_t bytes)
{
    // the level calls of abw_sep_tree also keep a level's host records in ctx->pinned between calls: whoever reserves the
    // pinned block outside those calls is about to write it from offset 0 (or to grow = free it), and counts like a
    // reservation of scratch[] does
    if (!ctx->scratch_quiet) ctx->scratch_epoch++;
    if (bytes <= ctx->pinned_cap) return ABW_OK;
    if (ctx->pinned) { ABW_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipHostFree(ctx->pinned); ctx->pinned = nullptr; ctx->pinned_cap = 0; }
    const size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault);
    if (e != hipSuccess) { ctx->pinned = nullptr; return abw_fail(ctx, ABW_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
    ctx->pinned_cap = want;
    return ABW_OK;
}

void abw_release(abw_devbuf& b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

static hipEvent_t prof_event(abw_ctx* ctx)
{
    if (!ctx->prof.pool.empty()) { hipEvent_t e = ctx->prof.pool.back(); ctx->prof.pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

abw_prof_scope::abw_prof_scope(abw_ctx* c, const char* name) : ctx(c)
{
    if (!ctx || !ctx->prof.on) return;
    int id = -1;
    for (size_t i = 0; i < ctx->prof.names.size(); i++) if (ctx->prof.names[i] == name) { id = (int)i; break; }
    if (id < 0) { id = (int)ctx->prof.names.size(); ctx->prof.names.push_back(name); ctx->prof.seen.push_back(0); }
    if (ctx->prof.seen[id]++ % ctx->prof.period) return;      // a sampled measurement: this scope of the name is not one of the timed ones
    hipEvent_t a = prof_event(ctx);
    b = prof_event(ctx);
    if (!a || !b) { b = nullptr; return; }
    (void)hipEventRecord(a, ctx->stream);
    ctx->prof.recs.push_back({id, a, b});
}

abw_prof_scope::~abw_prof_scope()
{
    if (b) (void)hipEventRecord(b, ctx->stream);
}

extern "C" {

int abw_prof_enable(abw_ctx* ctx, int on)
{
    if (!ctx) return ABW_ERR_ARG;
    ctx->prof.on = on != 0;
    ctx->prof.period = on > 1 ? (uint32_t)on : 1u;
    return ABW_OK;
}

int abw_prof_reset(abw_ctx* ctx)
{
    if (!ctx) return ABW_ERR_ARG;
    abw_device_guard g(ctx->device);
    ABW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& r : ctx->prof.recs) { ctx->prof.pool.push_back(r.a); ctx->prof.pool.push_back(r.b); }
    ctx->prof.recs.clear();
    for (auto& n : ctx->prof.seen) n = 0;
    return ABW_OK;
}

int abw_prof_read(abw_ctx* ctx, const char* name, uint64_t* launches, double* total_ms)
{
    if (!ctx || !name || !launches || !total_ms) return ABW_ERR_ARG;
    abw_device_guard g(ctx->device);
    ABW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *launches = 0; *total_ms = 0.0;
    for (auto& r : ctx->prof.recs) {
        if (ctx->prof.names[r.id] != name) continue;
        float ms = 0.0f;
        ABW_HIP(ctx, hipEventElapsedTime(&ms, r.a, r.b));
        *launches += 1; *total_ms += ms;
    }
    return ABW_OK;
}

int abw_abi_version(void) { return ABW_ABI_VERSION; }

int abw_device_count(int32_t* n)
{
    if (!n) return ABW_ERR_ARG;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) { *n = 0; return ABW_ERR_HIP; }
    *n = c;
    return ABW_OK;
}

int abw_ctx_create(int device, abw_ctx** out)
{
    if (!out) return ABW_ERR_ARG;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return abw_fail(nullptr, ABW_ERR_HIP, "no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= count)
        return abw_fail(nullptr, ABW_ERR_ARG, "device %d out of range [0,%d)", device, count);
    abw_ctx* ctx = new (std::nothrow) abw_ctx();
    if (!ctx) return ABW_ERR_NOMEM;
    ctx->device = device;
    abw_device_guard g(device);
    e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return abw_fail(nullptr, ABW_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    ctx->stream = ctx->own_stream;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cu = prop.multiProcessorCount;
    if (abw_sig_static_init(ctx) != ABW_OK) {          // once per context, with its device current
        fprintf(stderr, "abawaca_hip: %s\n", ctx->err.c_str());
        (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return ABW_ERR_HIP;
    }
    *out = ctx;
    return ABW_OK;
}

void abw_sep_free(abw_ctx* ctx);   // abw_sep.hip
void abw_kmstore_free(abw_ctx* ctx);   // abw_kmnode.hip

void abw_ctx_destroy(abw_ctx* ctx)
{
    if (!ctx) return;
    abw_device_guard g(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    abw_sep_free(ctx);
    abw_kmstore_free(ctx);
    abw_sig_pipe_free(ctx);
    for (auto& r : ctx->prof.recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto& e : ctx->prof.pool) (void)hipEventDestroy(e);
    for (auto& b : ctx->scratch) abw_release(b);
    for (auto& b : ctx->km_buf) abw_release(b);
    abw_release(ctx->lrn_table);
    abw_release(ctx->sam_names);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int abw_ctx_set_stream(abw_ctx* ctx, void* hip_stream)
{
    if (!ctx) return ABW_ERR_ARG;
    abw_device_guard g(ctx->device);
    ABW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return ABW_OK;
}

int abw_ctx_synchronize(abw_ctx* ctx)
{
    if (!ctx) return ABW_ERR_ARG;
    abw_device_guard g(ctx->device);
    ABW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ABW_OK;
}

const char* abw_last_error(const abw_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

}  // extern "C"
