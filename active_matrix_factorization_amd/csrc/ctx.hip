// ctx.hip - context, memory and stream plumbing of the C-ABI (include/apmf.h).
#include <stdarg.h>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";

void apmf_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- cache of released device blocks ------------------------------------------------
struct apmf_mempool {
    std::unordered_map<void *, size_t> live;                      // block -> rounded size
    std::unordered_map<size_t, std::vector<void *>> free_by_size;
    size_t cached = 0;
    // a block may be released from another thread than the one driving the context (Python's
    // garbage collector runs wherever it is triggered)
    std::mutex mu;
};

namespace {

constexpr size_t POOL_CAP_BYTES = (size_t)16 << 30;   // of 288 GB

// APMF_POISON=1 (debugging / the test-suite): every block handed out, fresh or recycled, and every
// workspace request is filled with 0xFF bytes (NaN as float or double, -1 as int) first, so that a
// kernel reading memory it was supposed to have written itself shows up as NaNs instead of as a
// result that depends on what ran before.
bool poison_on()
{
    static const bool on = getenv("APMF_POISON") != nullptr && getenv("APMF_POISON")[0] == '1';
    return on;
}

size_t pool_round(size_t bytes)
{
    const size_t q = bytes <= ((size_t)1 << 20) ? 512 : ((size_t)1 << 16);
    return (bytes + q - 1) / q * q;
}

// caller holds pool->mu (or owns the context exclusively)
void pool_release(apmf_ctx *ctx)
{
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->pool->free_by_size)
        for (void *p : kv.second) (void)hipFree(p);
    ctx->pool->free_by_size.clear();
    ctx->pool->cached = 0;
}

} // namespace

extern "C" {

int apmf_version(void) { return APMF_VERSION; }

const char *apmf_last_error(void) { return g_err; }

int apmf_grid_ld(int m) { return m <= 0 ? 0 : round_up(m, APMF_GRID_ALIGN); }

int apmf_device_count(int *count)
{
    APMF_REQUIRE(count, "count is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *count = c;
    return APMF_OK;
}

int apmf_ctx_create(int device, apmf_ctx **out)
{
    APMF_REQUIRE(out, "out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        apmf_set_error("no HIP device available (%s): the HIP path cannot run and there is no "
                       "CPU fallback", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return APMF_ENODEV;
    }
    APMF_REQUIRE(device >= 0 && device < count, "device %d out of range [0,%d)", device, count);
    APMF_CHECK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    APMF_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        apmf_set_error("device %d is %s; libapmf_hip is built for gfx950 (MI355X) only", device,
                       prop.gcnArchName);
        return APMF_ENODEV;
    }
    apmf_ctx *c = new (std::nothrow) apmf_ctx();
    if (!c) { apmf_set_error("out of host memory"); return APMF_ENOMEM; }
    memset(c, 0, sizeof(*c));
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    hipError_t se = hipStreamCreateWithFlags(&c->own, hipStreamNonBlocking);
    if (se != hipSuccess) {
        delete c;
        apmf_set_error("hipStreamCreate failed: %s", hipGetErrorString(se));
        return APMF_EHIP;
    }
    c->stream = c->own;
    (void)hipHostMalloc(&c->rd_pin, 4096, hipHostMallocDefault);   // optional: NULL falls back to pageable
    c->pool = new (std::nothrow) apmf_mempool();
    if (!c->pool) { (void)hipStreamDestroy(c->own); delete c; apmf_set_error("out of host memory"); return APMF_ENOMEM; }
    *out = c;
    return APMF_OK;
}

int apmf_ctx_destroy(apmf_ctx *ctx)
{
    if (!ctx) return APMF_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 4; i++)
        if (ctx->ws[i]) (void)hipFree(ctx->ws[i]);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->rd_pin) (void)hipHostFree(ctx->rd_pin);
    if (ctx->prof_ev) {
        for (int i = 0; i < 2 * ctx->prof_cap; i++) (void)hipEventDestroy(ctx->prof_ev[i]);
        delete[] ctx->prof_ev;
    }
    if (ctx->blas && ctx->blas_destroy) ctx->blas_destroy(ctx->blas);
    for (int l = 0; l < 4; l++)
        if (ctx->lane_blas[l] && ctx->blas_destroy) ctx->blas_destroy(ctx->lane_blas[l]);
    pool_release(ctx);
    delete ctx->pool;
    (void)hipStreamDestroy(ctx->own);
    delete ctx;
    return APMF_OK;
}

int apmf_ctx_set_stream(apmf_ctx *ctx, void *hip_stream)
{
    APMF_REQUIRE(ctx, "ctx is NULL");
    // cached blocks are reused in stream order: finish the old stream before switching
    APMF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own;
    return APMF_OK;
}

int apmf_ctx_sync(apmf_ctx *ctx)
{
    APMF_REQUIRE(ctx, "ctx is NULL");
    APMF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return APMF_OK;
}

int apmf_ctx_trim(apmf_ctx *ctx)
{
    APMF_REQUIRE(ctx, "ctx is NULL");
    APMF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    std::lock_guard<std::mutex> lock(ctx->pool->mu);
    pool_release(ctx);
    return APMF_OK;
}

int apmf_profile_begin(apmf_ctx *ctx, int max_records)
{
    APMF_REQUIRE(ctx && max_records >= 1 && max_records <= 4096, "bad profile request");
    if (ctx->prof_ev) {
        for (int i = 0; i < 2 * ctx->prof_cap; i++) (void)hipEventDestroy(ctx->prof_ev[i]);
        delete[] ctx->prof_ev;
        ctx->prof_ev = nullptr;
    }
    ctx->prof_ev = new (std::nothrow) hipEvent_t[2 * (size_t)max_records];
    if (!ctx->prof_ev) { apmf_set_error("out of host memory"); return APMF_ENOMEM; }
    for (int i = 0; i < 2 * max_records; i++) APMF_CHECK_HIP(hipEventCreate(&ctx->prof_ev[i]));
    ctx->prof_cap = max_records;
    ctx->prof_n = 0;
    return APMF_OK;
}

int apmf_profile_end(apmf_ctx *ctx, float *h_ms, int cap, int *h_count)
{
    APMF_REQUIRE(ctx && h_count, "NULL operand");
    *h_count = 0;
    if (!ctx->prof_ev) return APMF_OK;
    APMF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    const int n = ctx->prof_n;
    for (int i = 0; i < n && i < cap; i++) {
        float ms = 0.f;
        APMF_CHECK_HIP(hipEventElapsedTime(&ms, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]));
        if (h_ms) h_ms[i] = ms;
    }
    *h_count = n;
    for (int i = 0; i < 2 * ctx->prof_cap; i++) (void)hipEventDestroy(ctx->prof_ev[i]);
    delete[] ctx->prof_ev;
    ctx->prof_ev = nullptr;
    ctx->prof_cap = ctx->prof_n = 0;
    return APMF_OK;
}

int apmf_malloc(apmf_ctx *ctx, size_t bytes, void **d_out)
{
    APMF_REQUIRE(ctx && d_out, "ctx/d_out is NULL");
    *d_out = nullptr;
    if (bytes == 0) return APMF_OK;
    APMF_CHECK_HIP(hipSetDevice(ctx->device));
    apmf_mempool *pool = ctx->pool;
    std::lock_guard<std::mutex> lock(pool->mu);
    const size_t r = pool_round(bytes);
    auto it = pool->free_by_size.find(r);
    if (it != pool->free_by_size.end() && !it->second.empty()) {
        // Reuse is ordered by the stream: everything that touched the block before it was
        // released was enqueued on ctx->stream, and so is everything that will touch it now.
        *d_out = it->second.back();
        it->second.pop_back();
        pool->cached -= r;
        pool->live[*d_out] = r;
        if (poison_on()) APMF_CHECK_HIP(hipMemsetAsync(*d_out, 0xFF, r, ctx->stream));
        return APMF_OK;
    }
    hipError_t e = hipMalloc(d_out, r);
    if (e != hipSuccess) {      // give the cache back to the driver and try once more
        (void)hipGetLastError();
        pool_release(ctx);
        e = hipMalloc(d_out, r);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *d_out = nullptr;
        apmf_set_error("hipMalloc of %zu bytes failed: %s", r, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? APMF_ENOMEM : APMF_EHIP;
    }
    pool->live[*d_out] = r;
    if (poison_on()) APMF_CHECK_HIP(hipMemsetAsync(*d_out, 0xFF, r, ctx->stream));
    return APMF_OK;
}

int apmf_free(apmf_ctx *ctx, void *d_ptr)
{
    APMF_REQUIRE(ctx, "ctx is NULL");
    if (!d_ptr) return APMF_OK;
    apmf_mempool *pool = ctx->pool;
    std::lock_guard<std::mutex> lock(pool->mu);
    auto it = pool->live.find(d_ptr);
    if (it != pool->live.end()) {
        const size_t r = it->second;
        pool->live.erase(it);
        if (pool->cached + r <= POOL_CAP_BYTES) {
            pool->free_by_size[r].push_back(d_ptr);
            pool->cached += r;
            return APMF_OK;
        }
    }
    APMF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    APMF_CHECK_HIP(hipFree(d_ptr));
    return APMF_OK;
}

int apmf_memset(apmf_ctx *ctx, void *d_ptr, int value, size_t bytes)
{
    APMF_REQUIRE(ctx, "ctx is NULL");
    if (bytes == 0) return APMF_OK;
    APMF_REQUIRE(d_ptr, "d_ptr is NULL");
    APMF_CHECK_HIP(hipMemsetAsync(d_ptr, value, bytes, ctx->stream));
    return APMF_OK;
}

int apmf_h2d(apmf_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    APMF_REQUIRE(ctx, "ctx is NULL");
    if (bytes == 0) return APMF_OK;
    APMF_REQUIRE(d_dst && h_src, "NULL pointer");
    // pageable source: the copy is staged before the call returns, the host buffer is free
    APMF_CHECK_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    APMF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return APMF_OK;
}

int apmf_d2h(apmf_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    APMF_REQUIRE(ctx, "ctx is NULL");
    if (bytes == 0) return APMF_OK;
    APMF_REQUIRE(h_dst && d_src, "NULL pointer");
    if (bytes <= 4096 && ctx->rd_pin) {
        // small read-backs (flags, scalars: several per trial of the searches) land in the
        // context's own pinned buffer: no staging of a pageable destination by the runtime
        APMF_CHECK_HIP(hipMemcpyAsync(ctx->rd_pin, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        APMF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        memcpy(h_dst, ctx->rd_pin, bytes);
        return APMF_OK;
    }
    APMF_CHECK_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    APMF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return APMF_OK;
}

int apmf_d2d(apmf_ctx *ctx, void *d_dst, const void *d_src, size_t bytes)
{
    APMF_REQUIRE(ctx, "ctx is NULL");
    if (bytes == 0) return APMF_OK;
    APMF_REQUIRE(d_dst && d_src, "NULL pointer");
    APMF_CHECK_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return APMF_OK;
}

int apmf_h2d_2d(apmf_ctx *ctx, void *d_dst, size_t dpitch, const void *h_src, size_t spitch,
                size_t width, size_t height)
{
    APMF_REQUIRE(ctx, "ctx is NULL");
    if (width == 0 || height == 0) return APMF_OK;
    APMF_REQUIRE(d_dst && h_src && dpitch >= width && spitch >= width, "bad 2-D copy");
    APMF_CHECK_HIP(hipMemcpy2DAsync(d_dst, dpitch, h_src, spitch, width, height,
                                    hipMemcpyHostToDevice, ctx->stream));
    APMF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return APMF_OK;
}

int apmf_d2h_2d(apmf_ctx *ctx, void *h_dst, size_t dpitch, const void *d_src, size_t spitch,
                size_t width, size_t height)
{
    APMF_REQUIRE(ctx, "ctx is NULL");
    if (width == 0 || height == 0) return APMF_OK;
    APMF_REQUIRE(h_dst && d_src && dpitch >= width && spitch >= width, "bad 2-D copy");
    APMF_CHECK_HIP(hipMemcpy2DAsync(h_dst, dpitch, d_src, spitch, width, height,
                                    hipMemcpyDeviceToHost, ctx->stream));
    APMF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return APMF_OK;
}

int apmf_d2d_2d(apmf_ctx *ctx, void *d_dst, size_t dpitch, const void *d_src, size_t spitch, size_t width,
                size_t height)
{
    APMF_REQUIRE(ctx, "ctx is NULL");
    if (width == 0 || height == 0) return APMF_OK;
    APMF_REQUIRE(d_dst && d_src && dpitch >= width && spitch >= width, "bad 2-D copy");
    APMF_CHECK_HIP(hipMemcpy2DAsync(d_dst, dpitch, d_src, spitch, width, height, hipMemcpyDeviceToDevice,
                                    ctx->stream));
    return APMF_OK;
}

} // extern "C"

int apmf_ws(apmf_ctx *ctx, int slot, size_t bytes, void **out)
{
    if (bytes > ctx->ws_bytes[slot]) {
        // the old block may still be in use by enqueued work
        APMF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->ws[slot]) APMF_CHECK_HIP(hipFree(ctx->ws[slot]));
        ctx->ws[slot] = nullptr;
        ctx->ws_bytes[slot] = 0;
        size_t want = bytes + bytes / 4 + 4096;
        APMF_CHECK_HIP(hipMalloc(&ctx->ws[slot], want));
        ctx->ws_bytes[slot] = want;
    }
    *out = ctx->ws[slot];
    if (poison_on()) APMF_CHECK_HIP(hipMemsetAsync(ctx->ws[slot], 0xFF, bytes, ctx->stream));
    return APMF_OK;
}

int apmf_pinned(apmf_ctx *ctx, size_t bytes, void **out)
{
    if (bytes > ctx->pinned_bytes) {
        APMF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->pinned) APMF_CHECK_HIP(hipHostFree(ctx->pinned));
        ctx->pinned = nullptr;
        ctx->pinned_bytes = 0;
        size_t want = bytes < 4096 ? 4096 : bytes * 2;
        APMF_CHECK_HIP(hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
        ctx->pinned_bytes = want;
    }
    *out = ctx->pinned;
    return APMF_OK;
}

// ---- casts -------------------------------------------------------------------------
template <typename TO, typename FROM>
__global__ __launch_bounds__(256) void k_cast(TO *__restrict__ dst, const FROM *__restrict__ src,
                                              size_t count)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) dst[i] = (TO)src[i];
}

extern "C" int apmf_cast_f64_f32(apmf_ctx *ctx, float *d_dst, const double *d_src, size_t count)
{
    APMF_REQUIRE(ctx, "ctx is NULL");
    if (count == 0) return APMF_OK;
    APMF_REQUIRE(d_dst && d_src, "NULL pointer");
    int blocks = (int)(cdiv64((int64_t)count, 256) < 4096 ? cdiv64((int64_t)count, 256) : 4096);
    hipLaunchKernelGGL((k_cast<float, double>), dim3(blocks), dim3(256), 0, ctx->stream, d_dst,
                       d_src, count);
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}

extern "C" int apmf_cast_f32_f64(apmf_ctx *ctx, double *d_dst, const float *d_src, size_t count)
{
    APMF_REQUIRE(ctx, "ctx is NULL");
    if (count == 0) return APMF_OK;
    APMF_REQUIRE(d_dst && d_src, "NULL pointer");
    int blocks = (int)(cdiv64((int64_t)count, 256) < 4096 ? cdiv64((int64_t)count, 256) : 4096);
    hipLaunchKernelGGL((k_cast<double, float>), dim3(blocks), dim3(256), 0, ctx->stream, d_dst,
                       d_src, count);
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}

// out = x + alpha * y with the product and the sum rounded separately, like NumPy evaluates
// `x + alpha * y` (the trial points of the learning-rate searches, pmf_cy.pyx:271-272)
__global__ __launch_bounds__(256) void k_axpy(double *__restrict__ out, const double *__restrict__ x, double alpha,
                                              const double *__restrict__ y, size_t count)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) out[i] = __dadd_rn(x[i], __dmul_rn(alpha, y[i]));
}

extern "C" int apmf_axpy_f64(apmf_ctx *ctx, double *d_out, const double *d_x, double alpha, const double *d_y,
                             size_t count)
{
    APMF_REQUIRE(ctx, "ctx is NULL");
    if (count == 0) return APMF_OK;
    APMF_REQUIRE(d_out && d_x && d_y, "NULL pointer");
    int blocks = (int)(cdiv64((int64_t)count, 256) < 4096 ? cdiv64((int64_t)count, 256) : 4096);
    hipLaunchKernelGGL(k_axpy, dim3(blocks), dim3(256), 0, ctx->stream, d_out, d_x, alpha, d_y, count);
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}
