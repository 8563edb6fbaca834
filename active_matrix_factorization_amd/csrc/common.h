// common.h - context, error plumbing and small device helpers shared by the
// libapmf_hip translation units.  gfx950 (MI355X, wave64) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/apmf.h"

#define APMF_VERSION 100

struct apmf_ctx {
    int device;
    hipStream_t own;      // created with the ctx
    hipStream_t stream;   // the stream work is enqueued on (own or caller's)
    // grow-only device scratch (per-launch packs and reduction partials)
    void *ws[4];
    size_t ws_bytes[4];
    // pinned host scratch for small read-backs
    void *pinned;
    size_t pinned_bytes;
    int num_cu;
    // optional event pairs around the dominant kernels (apmf_profile_begin/end)
    hipEvent_t *prof_ev;   // 2 * prof_cap events
    int prof_cap, prof_n;
    // size-bucketed cache of freed device blocks (ctx.hip): the Python classes allocate and
    // release many same-sized arrays per step, and hipMalloc / hipFree synchronise the device
    struct apmf_mempool *pool;
    // rocBLAS handle of libapmf_dense.so (one per context: contexts may live on different
    // threads), with the function that destroys it - libapmf_hip.so itself has no BLAS dependency
    void *blas;
    void (*blas_destroy)(void *);
    // further handles for the slow lanes of k6_batch.hip (created at a lane's first pass, kept for the
    // context's life: creating one costs tens of milliseconds - seconds when the library is cold)
    void *lane_blas[4];
    // 4 KB of pinned host memory for small read-backs (apmf_d2h)
    void *rd_pin;
};

// bracket a kernel launch with events when profiling is on
struct ProfScope {
    apmf_ctx *c;
    int slot;
    explicit ProfScope(apmf_ctx *ctx) : c(ctx), slot(-1)
    {
        if (c->prof_ev && c->prof_n < c->prof_cap) {
            slot = c->prof_n++;
            (void)hipEventRecord(c->prof_ev[2 * slot], c->stream);
        }
    }
    ~ProfScope()
    {
        if (slot >= 0) (void)hipEventRecord(c->prof_ev[2 * slot + 1], c->stream);
    }
};

// device-resident ratings: CSR by user (index 0) and by item (index 1), stable order
struct apmf_ratings {
    int64_t N;
    int n, m;
    int64_t *ptr[2];   // (rows+1)
    int32_t *idx[2];   // (N) index on the other side
    double *val[2];    // (N)
    int32_t *order;    // (n+m) rows by descending nnz; entries >= n are item rows + n
    void *desc;        // (n+m) RowDesc records in that order (k1_pmf.hip)
    int max_len;       // longest row of either side
    int has_empty[2];  // some row on that side has no rating
    void *block;       // single allocation backing everything above
    void *h_desc;      // host copy of desc (malloc): the batched re-fits derive per-hypothesis row orders from it
};

void apmf_set_error(const char *fmt, ...);

#define APMF_CHECK_HIP(expr)                                                           \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            apmf_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                           __FILE__, __LINE__);                                        \
            return _e == hipErrorOutOfMemory ? APMF_ENOMEM : APMF_EHIP;                \
        }                                                                              \
    } while (0)

#define APMF_REQUIRE(cond, ...)                                                        \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            apmf_set_error(__VA_ARGS__);                                               \
            return APMF_EINVAL;                                                        \
        }                                                                              \
    } while (0)

#define APMF_TRY(expr)                                                                 \
    do {                                                                               \
        int _rc = (expr);                                                              \
        if (_rc != APMF_OK) return _rc;                                                \
    } while (0)

#define APMF_CHECK_LAUNCH() APMF_CHECK_HIP(hipGetLastError())

// device scratch slot `slot` of at least `bytes` (contents undefined)
int apmf_ws(apmf_ctx *ctx, int slot, size_t bytes, void **out);
int apmf_pinned(apmf_ctx *ctx, size_t bytes, void **out);
// libapmf_dense.so internals shared between dense.hip and k5_klgrad.hip (not part of include/apmf.h)
extern "C" int apmf_dense_logdet_keep_factor(apmf_ctx *ctx, const double *d_mat, int N, double *d_L,
                                             double *h_logdet, int *h_have_factor);
extern "C" int apmf_dense_inverse_from_factor(apmf_ctx *ctx, const double *d_L, int N, double *d_inv);
extern "C" int apmf_dense_certify_and_factor(apmf_ctx *ctx, double *d_mat, int N, double min_eig, double *d_L,
                                             double *h_logdet, int *h_certified, int *h_have_factor);

static inline int round_up(int x, int a) { return (x + a - 1) / a * a; }
static inline int64_t cdiv64(int64_t x, int64_t a) { return (x + a - 1) / a; }

// ---- device helpers -------------------------------------------------------------
#ifdef __HIPCC__
// A pointer READ FROM DEVICE MEMORY (a job list, a table of per-hypothesis matrices) is a generic pointer to the
// compiler, and loads / stores through it become FLAT instructions.  Those count on lgkmcnt as well as on vmcnt, so
// every wait for an LDS read (s_waitcnt lgkmcnt(0) before each MFMA operand) also drains the global loads in
// flight: a prefetch of the next tile issued before the products of this one is waited out before the first of them.
// Everything these kernels address is hipMalloc'ed memory: say so at the access (address space 1 = global), and the
// loads are global_load_*, on vmcnt alone.  (Casting the pointer and back does not survive instruction combining,
// and an assume on is_shared / is_private is not picked up: the cast has to sit at the access.)
template <typename T>
__device__ __forceinline__ auto gmem(T *p) { return (T __attribute__((address_space(1))) *)p; }

__device__ __forceinline__ int wave_id_uniform()
{
    // threadIdx.x >> 6 is wave-uniform; say so, so that row-indexed operands become SGPRs
    return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// (value, index) arg-extremum merge: sign = +1 max / -1 min; ties -> lower index;
// idx < 0 means "empty"
template <typename T>
__device__ __forceinline__ void argext_merge(T &bv, long long &bi, T ov, long long oi, int sign)
{
    if (oi < 0) return;
    bool take;
    if (bi < 0) take = true;
    else if (sign > 0) take = (ov > bv) || (ov == bv && oi < bi);
    else take = (ov < bv) || (ov == bv && oi < bi);
    if (take) { bv = ov; bi = oi; }
}

template <typename T>
__device__ __forceinline__ void wave_argext(T &bv, long long &bi, int sign)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T ov = __shfl_xor(bv, o, 64);
        long long oi = __shfl_xor(bi, o, 64);
        argext_merge(bv, bi, ov, oi, sign);
    }
}
#endif
