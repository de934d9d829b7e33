// Multi-GPU reassembly of the scored callset: RCCL all-gather over xGMI.
//
// One process per GPU.  Every rank scores a contiguous, equal-count (+-1) slice of the sorted
// callset; the per-variant result record (tree_score f32, filter u8, flags u8) is gathered
// as three column all-gathers issued as one RCCL group, in place (each rank's slice sits at
// rank*shard_cap of the gather buffers), so rank-order concatenation == callset order.
// RCCL is bound lazily with dlopen so single-GPU use never loads it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include "ugvc_device.hpp"

namespace ugvc {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static Rccl g_rccl;

static int load_rccl() {
    if (g_rccl.h) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return fail(std::string("cannot load RCCL: ") + dlerror());
#define UGVC_SYM(field, name)                                                          \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));          \
    if (!g_rccl.field) return fail(std::string("RCCL symbol missing: ") + name)
    UGVC_SYM(GetUniqueId, "ncclGetUniqueId");
    UGVC_SYM(CommInitRank, "ncclCommInitRank");
    UGVC_SYM(CommDestroy, "ncclCommDestroy");
    UGVC_SYM(CommCount, "ncclCommCount");
    UGVC_SYM(CommUserRank, "ncclCommUserRank");
    UGVC_SYM(CommCuDevice, "ncclCommCuDevice");
    UGVC_SYM(AllGather, "ncclAllGather");
    UGVC_SYM(GroupStart, "ncclGroupStart");
    UGVC_SYM(GroupEnd, "ncclGroupEnd");
    UGVC_SYM(GetErrorString, "ncclGetErrorString");
#undef UGVC_SYM
    g_rccl.h = h;
    return 0;
}

#define UGVC_NCCL(expr)                                                                  \
    do {                                                                                 \
        ncclResult_t _r = (expr);                                                        \
        if (_r != ncclSuccess)                                                           \
            return ugvc::fail(std::string(#expr) + ": " + g_rccl.GetErrorString(_r));    \
    } while (0)

}  // namespace ugvc

using namespace ugvc;

extern "C" {

int ugvc_comm_unique_id(uint8_t id[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (!id) return fail("id is NULL");
    if (load_rccl()) return -1;
    ncclUniqueId u;
    UGVC_NCCL(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, 128);
    return 0;
}

int ugvc_comm_init(ugvc_ctx* ctx, const uint8_t id[128], int rank, int world) {
    if (!ctx || !id) return fail("NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail("bad rank/world");
    if (load_rccl()) return -1;
    UGVC_HIP(hipSetDevice(ctx->device));
    if (ctx->comm) return fail("communicator already initialised");
    ncclUniqueId u;
    memcpy(&u, id, 128);
    ncclComm_t comm = nullptr;
    UGVC_NCCL(g_rccl.CommInitRank(&comm, world, u, rank));
    ctx->comm = comm;
    ctx->rank = rank;
    ctx->world = world;
    // the collective runs on its own stream so the gather of pass i overlaps the kernels of pass i+1
    UGVC_HIP(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
    UGVC_HIP(hipEventCreateWithFlags(&ctx->ev_res_ready, hipEventDisableTiming));
    for (int k = 0; k < 2; ++k) {
        UGVC_HIP(hipEventCreateWithFlags(&ctx->ev_gather_done[k], hipEventDisableTiming));
        ctx->gather_pending[k] = 0;
    }
    return 0;
}

int ugvc_comm_info(ugvc_ctx* ctx, int* nranks, int* rank, int* device) {
    if (!ctx) return fail("ctx is NULL");
    if (!ctx->comm) return fail("communicator not initialised (ugvc_comm_init)");
    ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
    int a = 0, b = 0, c = 0;
    UGVC_NCCL(g_rccl.CommCount(comm, &a));
    UGVC_NCCL(g_rccl.CommUserRank(comm, &b));
    UGVC_NCCL(g_rccl.CommCuDevice(comm, &c));
    if (nranks) *nranks = a;
    if (rank) *rank = b;
    if (device) *device = c;
    return 0;
}

int ugvc_comm_destroy(ugvc_ctx* ctx) {
    if (!ctx || !ctx->comm) return 0;
    (void)hipSetDevice(ctx->device);
    if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
    if (g_rccl.CommDestroy) g_rccl.CommDestroy(static_cast<ncclComm_t>(ctx->comm));
    if (ctx->comm_stream) { (void)hipStreamDestroy(ctx->comm_stream); ctx->comm_stream = nullptr; }
    if (ctx->ev_res_ready) { (void)hipEventDestroy(ctx->ev_res_ready); ctx->ev_res_ready = nullptr; }
    for (int k = 0; k < 2; ++k)
        if (ctx->ev_gather_done[k]) { (void)hipEventDestroy(ctx->ev_gather_done[k]); ctx->ev_gather_done[k] = nullptr; }
    ctx->comm = nullptr;
    ctx->world = 1;
    ctx->rank = 0;
    return 0;
}

// Pick the next gather buffer, make the context stream wait until its previous collective has
// drained, and hand back this rank's slot in it: the scoring pass can write its results there.
int ugvc_gather_target(ugvc_ctx* ctx, int64_t shard_cap, float** score, uint8_t** filter, uint8_t** flags) {
    if (!ctx) return fail("ctx is NULL");
    if (shard_cap < ctx->n) return fail("shard_cap smaller than this rank's variant count");
    UGVC_HIP(hipSetDevice(ctx->device));
    const int b = ctx->g_cur ^ 1;
    const size_t cap = (size_t)shard_cap, W = (size_t)ctx->world, r = (size_t)ctx->rank;
    if (ctx->g_score[b].cap < cap * W * 4 || ctx->g_filter[b].cap < cap * W || ctx->g_flags[b].cap < cap * W) {
        if (ctx->comm_stream) UGVC_HIP(hipStreamSynchronize(ctx->comm_stream));   // re-allocation: nothing may be in flight
        if (ensure(ctx->g_score[b], cap * W * 4) || ensure(ctx->g_filter[b], cap * W) || ensure(ctx->g_flags[b], cap * W)) return -1;
    }
    if (ctx->comm && ctx->gather_pending[b]) UGVC_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_gather_done[b], 0));
    ctx->g_cur = b;
    if (score) *score = ctx->g_score[b].as<float>() + r * cap;
    if (filter) *filter = ctx->g_filter[b].as<uint8_t>() + r * cap;
    if (flags) *flags = ctx->g_flags[b].as<uint8_t>() + r * cap;
    return 0;
}

// Issue the grouped in-place all-gathers of the current gather buffer on the communication stream,
// behind everything queued so far on the context stream.
int ugvc_gather_launch(ugvc_ctx* ctx, int64_t shard_cap) {
    if (!ctx) return fail("ctx is NULL");
    if (!ctx->comm) {
        if (ctx->world == 1) return 0;
        return fail("communicator not initialised (ugvc_comm_init)");
    }
    UGVC_HIP(hipSetDevice(ctx->device));
    const int b = ctx->g_cur;
    const size_t cap = (size_t)shard_cap, r = (size_t)ctx->rank;
    float* gs = ctx->g_score[b].as<float>();
    uint8_t* gf = ctx->g_filter[b].as<uint8_t>();
    uint8_t* gl = ctx->g_flags[b].as<uint8_t>();
    ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
    UGVC_HIP(hipEventRecord(ctx->ev_res_ready, ctx->stream));
    UGVC_HIP(hipStreamWaitEvent(ctx->comm_stream, ctx->ev_res_ready, 0));
    UGVC_NCCL(g_rccl.GroupStart());
    UGVC_NCCL(g_rccl.AllGather(gs + r * cap, gs, cap, ncclFloat32, comm, ctx->comm_stream));
    UGVC_NCCL(g_rccl.AllGather(gf + r * cap, gf, cap, ncclUint8, comm, ctx->comm_stream));
    UGVC_NCCL(g_rccl.AllGather(gl + r * cap, gl, cap, ncclUint8, comm, ctx->comm_stream));
    UGVC_NCCL(g_rccl.GroupEnd());
    UGVC_HIP(hipEventRecord(ctx->ev_gather_done[b], ctx->comm_stream));
    ctx->gather_pending[b] = 1;
    return 0;
}

int ugvc_allgather_resident(ugvc_ctx* ctx, int64_t shard_cap) {
    float* gs; uint8_t* gf; uint8_t* gl;
    if (ugvc_gather_target(ctx, shard_cap, &gs, &gf, &gl)) return -1;
    const size_t n = (size_t)ctx->n;
    if (n) {
        UGVC_HIP(hipMemcpyAsync(gs, ctx->r_score.p, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        UGVC_HIP(hipMemcpyAsync(gf, ctx->r_filter.p, n, hipMemcpyDeviceToDevice, ctx->stream));
        UGVC_HIP(hipMemcpyAsync(gl, ctx->r_flags.p, n, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return ugvc_gather_launch(ctx, shard_cap);
}

// Make the context stream wait for the outstanding collectives (end of a timed region, before a download).
int ugvc_gather_fence(ugvc_ctx* ctx) {
    if (!ctx) return fail("ctx is NULL");
    if (ctx->comm) {
        UGVC_HIP(hipSetDevice(ctx->device));
        for (int b = 0; b < 2; ++b)
            if (ctx->gather_pending[b]) UGVC_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_gather_done[b], 0));
    }
    return 0;
}

int ugvc_gather_fence(ugvc_ctx* ctx);

int ugvc_gathered_download(ugvc_ctx* ctx, int64_t shard_cap, int world, const ugvc_results* out) {
    if (!ctx || !out) return fail("NULL argument");
    if (world != ctx->world) return fail("world does not match the communicator");
    UGVC_HIP(hipSetDevice(ctx->device));
    const size_t tot = (size_t)shard_cap * (size_t)world;
    const int b = ctx->g_cur;
    if (ctx->g_score[b].cap < tot * 4) return fail("nothing gathered yet (ugvc_allgather_resident)");
    if (ugvc_gather_fence(ctx)) return -1;
    if (out->tree_score) UGVC_HIP(copy_out(ctx, out->tree_score, ctx->g_score[b].p, tot * 4));
    if (out->filter) UGVC_HIP(copy_out(ctx, out->filter, ctx->g_filter[b].p, tot));
    if (out->flags) UGVC_HIP(copy_out(ctx, out->flags, ctx->g_flags[b].p, tot));
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
