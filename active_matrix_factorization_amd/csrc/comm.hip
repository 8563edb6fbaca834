// comm.hip - libapmf_comm.so: the exchange steps of the sharded hot path as
// RCCL collectives on the context's stream, device pointers only.
//
// What is replaced: the reference hands work to a forked multiprocessing.Pool and gets results
// back by pickling whole model objects through pipes (active_pmf.py:739-770,
// bayes_pmf.py:385-422).  Here there is one process per GPU; what crosses between them is
//   * row blocks of U / V (and of the freshly sampled Gibbs side, and diag(Sigma)) - an
//     all-gather of equal-sized blocks into the full array every rank keeps resident,
//   * the three float64 sums of the log-likelihood - an all-reduce, so that every rank takes the
//     same accept / reject decision,
//   * the per-hypothesis results of a sharded lookahead / the (value, flat index) pair of a
//     sharded arg-extremum - a small all-gather.
// Rendezvous: rank 0 draws the ncclUniqueId (apmf_comm_unique_id) and the host side ships its 128
// bytes to the other ranks over a plain TCP socket (sharding.py); no MPI, no PyTorch.
// A library of its own so that libapmf_hip.so keeps no RCCL dependency.
#include <rccl/rccl.h>

#include "common.h"

struct apmf_comm {
    ncclComm_t nccl;
    int rank, world, device;
    int *d_flag;      // barrier scratch
};

#define APMF_CHECK_NCCL(expr)                                                              \
    do {                                                                                   \
        ncclResult_t _r = (expr);                                                          \
        if (_r != ncclSuccess) {                                                           \
            apmf_set_error("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_r), __FILE__, \
                           __LINE__);                                                      \
            return APMF_EHIP;                                                              \
        }                                                                                  \
    } while (0)

extern "C" {

int apmf_comm_unique_id(void *h_id128)
{
    APMF_REQUIRE(h_id128, "NULL operand");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId id;
    APMF_CHECK_NCCL(ncclGetUniqueId(&id));
    memcpy(h_id128, &id, sizeof(id));
    return APMF_OK;
}

int apmf_comm_create(apmf_ctx *ctx, int rank, int world, const void *h_id128, apmf_comm **out)
{
    APMF_REQUIRE(ctx && out && h_id128, "NULL operand");
    *out = nullptr;
    APMF_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank %d of %d", rank, world);
    APMF_CHECK_HIP(hipSetDevice(ctx->device));
    apmf_comm *c = new (std::nothrow) apmf_comm();
    if (!c) { apmf_set_error("out of host memory"); return APMF_ENOMEM; }
    c->rank = rank;
    c->world = world;
    c->device = ctx->device;
    c->d_flag = nullptr;
    ncclUniqueId id;
    memcpy(&id, h_id128, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) {
        apmf_set_error("ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, ctx->device,
                       ncclGetErrorString(r));
        delete c;
        return APMF_EHIP;
    }
    if (hipMalloc((void **)&c->d_flag, 64) != hipSuccess) {
        (void)ncclCommDestroy(c->nccl);
        delete c;
        apmf_set_error("hipMalloc failed");
        return APMF_ENOMEM;
    }
    (void)hipMemset(c->d_flag, 0, 64);
    *out = c;
    return APMF_OK;
}

int apmf_comm_destroy(apmf_comm *c)
{
    if (!c) return APMF_OK;
    (void)hipSetDevice(c->device);
    if (c->d_flag) (void)hipFree(c->d_flag);
    (void)ncclCommDestroy(c->nccl);
    delete c;
    return APMF_OK;
}

int apmf_comm_rank(const apmf_comm *c, int *rank, int *world)
{
    APMF_REQUIRE(c, "NULL comm");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return APMF_OK;
}

// d_recv (world * bytes_per_rank) <- every rank's d_send (bytes_per_rank), in rank order
int apmf_comm_allgather(apmf_ctx *ctx, apmf_comm *c, const void *d_send, void *d_recv, size_t bytes_per_rank)
{
    APMF_REQUIRE(ctx && c, "NULL operand");
    if (bytes_per_rank == 0) return APMF_OK;
    APMF_REQUIRE(d_send && d_recv, "NULL buffer");
    APMF_CHECK_NCCL(ncclAllGather(d_send, d_recv, bytes_per_rank, ncclInt8, c->nccl, ctx->stream));
    return APMF_OK;
}

// `count` all-gathers as ONE group (ncclGroupStart / End: RCCL aggregates them into a single launch):
// d_recv[k] (world * bytes[k]) <- every rank's d_send[k] (bytes[k]).  The exchange step of a sharded fit_lls
// trial - U and V blocks of both speculative branches plus the log-likelihood partial sums - is one such
// group: one collective per trial instead of two all-gathers and an all-reduce.
int apmf_comm_allgather_multi(apmf_ctx *ctx, apmf_comm *c, int count, const void *const *d_send, void *const *d_recv,
                              const size_t *bytes_per_rank)
{
    APMF_REQUIRE(ctx && c, "NULL operand");
    APMF_REQUIRE(count >= 0 && count <= 16, "bad group size %d (0..16)", count);
    if (count == 0) return APMF_OK;
    APMF_REQUIRE(d_send && d_recv && bytes_per_rank, "NULL operand");
    for (int k = 0; k < count; k++)
        APMF_REQUIRE(bytes_per_rank[k] == 0 || (d_send[k] && d_recv[k]), "NULL buffer in group entry %d", k);
    APMF_CHECK_NCCL(ncclGroupStart());
    ncclResult_t first = ncclSuccess;
    for (int k = 0; k < count; k++) {
        if (bytes_per_rank[k] == 0) continue;
        ncclResult_t r = ncclAllGather(d_send[k], d_recv[k], bytes_per_rank[k], ncclInt8, c->nccl, ctx->stream);
        if (r != ncclSuccess && first == ncclSuccess) first = r;
    }
    ncclResult_t end = ncclGroupEnd();        // always closed, even after a failed enqueue
    if (first != ncclSuccess || end != ncclSuccess) {
        apmf_set_error("grouped ncclAllGather failed: %s", ncclGetErrorString(first != ncclSuccess ? first : end));
        return APMF_EHIP;
    }
    return APMF_OK;
}

// RCCL's version code (e.g. 22106), for the one-line diagnostic a rank prints when it joins
int apmf_comm_version(int *version)
{
    APMF_REQUIRE(version, "NULL operand");
    APMF_CHECK_NCCL(ncclGetVersion(version));
    return APMF_OK;
}

// d_recv[i] <- sum over ranks of d_send[i]   (in place allowed)
int apmf_comm_allreduce_f64(apmf_ctx *ctx, apmf_comm *c, const double *d_send, double *d_recv, size_t count)
{
    APMF_REQUIRE(ctx && c, "NULL operand");
    if (count == 0) return APMF_OK;
    APMF_REQUIRE(d_send && d_recv, "NULL buffer");
    APMF_CHECK_NCCL(ncclAllReduce(d_send, d_recv, count, ncclDouble, ncclSum, c->nccl, ctx->stream));
    return APMF_OK;
}

// every rank's stream has reached this point (and the host has waited for it)
int apmf_comm_barrier(apmf_ctx *ctx, apmf_comm *c)
{
    APMF_REQUIRE(ctx && c, "NULL operand");
    APMF_CHECK_NCCL(ncclAllReduce(c->d_flag, c->d_flag + 8, 1, ncclInt32, ncclSum, c->nccl, ctx->stream));
    APMF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return APMF_OK;
}

} // extern "C"
