// dce_comm.hip -- the ONE exchange of the multi-GPU path: every rank's packed result rows travel to
// the root in a single RCCL gather over xGMI, issued by this library on a stream of its own behind the ctx stream.
//
// The windows of a sequence are independent (reference utils/data_handler.py:55-57; no BatchNorm, Dropout off), so
// a node's GPUs take contiguous window ranges and nothing is exchanged inside the model.  What is exchanged:
//   dce_gather_results   (n_r,68)-byte rows (16 fp32 logits + 4 contact bits, written in that form by the tail /
//                        combine kernels) -> root: ncclGather when every rank holds the same number of rows, else one
//                        ncclGroup of ncclSend / ncclRecv with the true row counts (shards differ by at most one row)
//   dce_allreduce_counts the 16x16 confusion counts of the accuracy epilogue (2 KB, ncclAllReduce int64 sum)
//
// RCCL is bound at run time (dlopen + dlsym), like the HIP runtime it has to share with the process: a PyTorch-ROCm
// process already holds its own librccl.so/libamdhip64.so pair and a second pair cannot open the device.  Order:
// $DCE_RCCL_LIB, a librccl already mapped into the process, then librccl.so.1 / librccl.so / $ROCM_PATH/lib.
// Bootstrap is ncclUniqueId (128 bytes) carried by the host however it likes (file, env, a TCP store).
#include "dce_ctx.h"

#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>

namespace {

struct Rccl {
    void* handle = nullptr;
    std::string where, error;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    const char*  (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void rccl_bind()
{
    Rccl& r = g_rccl;
    std::vector<std::pair<std::string, int>> cands;     // (name, dlopen flags)
    if (const char* e = getenv("DCE_RCCL_LIB")) {
        if (strcmp(e, "none") == 0) { r.error = "switched off by DCE_RCCL_LIB=none"; return; }      // tests of the callers' error paths
        cands.push_back({e, RTLD_NOW | RTLD_LOCAL});
    }
    for (const char* n : {"librccl.so", "librccl.so.1"}) cands.push_back({n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD});   // already mapped
    for (const char* n : {"librccl.so.1", "librccl.so"}) cands.push_back({n, RTLD_NOW | RTLD_LOCAL});
    const char* rocm = getenv("ROCM_PATH");
    cands.push_back({std::string(rocm ? rocm : "/opt/rocm") + "/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL});
    for (auto& cnd : cands) {
        r.handle = dlopen(cnd.first.c_str(), cnd.second);
        if (r.handle) { r.where = cnd.first + ((cnd.second & RTLD_NOLOAD) ? " (already in the process)" : ""); break; }
    }
    if (!r.handle) { r.error = "librccl could not be loaded (set DCE_RCCL_LIB to its path)"; return; }
#define BIND(field, sym)                                                                        \
    do { r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, sym));                   \
         if (!r.field) { r.error = std::string(sym) + " not found in " + r.where; r.handle = nullptr; return; } } while (0)
    BIND(GetVersion, "ncclGetVersion");        BIND(GetUniqueId, "ncclGetUniqueId");
    BIND(CommInitRank, "ncclCommInitRank");    BIND(CommDestroy, "ncclCommDestroy");
    BIND(CommCount, "ncclCommCount");          BIND(CommUserRank, "ncclCommUserRank");
    BIND(CommGetAsyncError, "ncclCommGetAsyncError");
    BIND(GetErrorString, "ncclGetErrorString");
    BIND(GroupStart, "ncclGroupStart");        BIND(GroupEnd, "ncclGroupEnd");
    BIND(Send, "ncclSend");                    BIND(Recv, "ncclRecv");
    BIND(Gather, "ncclGather");                BIND(AllReduce, "ncclAllReduce");
#undef BIND
}

Rccl* rccl(dce_ctx* c)
{
    std::call_once(g_rccl_once, rccl_bind);
    if (!g_rccl.handle) { fail(c, DCE_ERR_COMM, "RCCL unavailable: %s", g_rccl.error.c_str()); return nullptr; }
    return &g_rccl;
}

#define NCCL_TRY(c, r, expr)                                                                    \
    do { ncclResult_t n_ = (expr); if (n_ != ncclSuccess)                                       \
        return fail((c), DCE_ERR_COMM, "%s failed: %s", #expr, (r)->GetErrorString(n_)); } while (0)

int need_comm(dce_ctx* c, const char* who)
{
    if (!c) return DCE_ERR_ARG;
    if (!c->comm) return fail(c, DCE_ERR_STATE, "%s: no communicator (call dce_comm_init first)", who);
    return DCE_OK;
}

// comm_stream picks up behind everything queued on the ctx stream so far
int comm_follow_ctx(dce_ctx* c)
{
    HIP_TRY(c, hipEventRecord(c->comm_ready, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(c->comm_stream, c->comm_ready, 0));
    return DCE_OK;
}

}  // namespace

extern "C" {

int dce_comm_get_unique_id(uint8_t id[DCE_COMM_ID_BYTES])
{
    if (!id) return fail(nullptr, DCE_ERR_ARG, "dce_comm_get_unique_id: NULL id");
    static_assert(sizeof(ncclUniqueId) == DCE_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    Rccl* r = rccl(nullptr);
    if (!r) return DCE_ERR_COMM;
    ncclUniqueId u;
    NCCL_TRY(nullptr, r, r->GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return DCE_OK;
}

int dce_comm_init(dce_ctx* c, int rank, int world, const uint8_t id[DCE_COMM_ID_BYTES])
{
    if (!c) return DCE_ERR_ARG;
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(c, DCE_ERR_ARG, "dce_comm_init: bad rank %d / world %d / id", rank, world);
    if (c->comm) return fail(c, DCE_ERR_STATE, "dce_comm_init: this ctx already has a communicator");
    Rccl* r = rccl(c);
    if (!r) return DCE_ERR_COMM;
    DEVICE_GUARD(c);
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    NCCL_TRY(c, r, r->CommInitRank(&comm, world, u, rank));       // collective over the ranks: blocks until all have called
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_world = world;
    c->comm_issued = 0;
    if (!c->comm_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    if (!c->comm_ready) HIP_TRY(c, hipEventCreateWithFlags(&c->comm_ready, hipEventDisableTiming));
    for (auto& e : c->comm_done) if (!e) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return DCE_OK;
}

int dce_comm_info(dce_ctx* c, int* rank, int* world, int* rccl_version, char* library, int library_len)
{
    int rc = need_comm(c, "dce_comm_info");
    if (rc) return rc;
    Rccl* r = rccl(c);
    if (!r) return DCE_ERR_COMM;
    // what RCCL itself reports for this communicator, not what dce_comm_init was told
    if (rank) NCCL_TRY(c, r, r->CommUserRank((ncclComm_t)c->comm, rank));
    if (world) NCCL_TRY(c, r, r->CommCount((ncclComm_t)c->comm, world));
    if (rccl_version) NCCL_TRY(c, r, r->GetVersion(rccl_version));
    if (library && library_len > 0) snprintf(library, (size_t)library_len, "%s", r->where.c_str());
    return DCE_OK;
}

int dce_gather_results(dce_ctx* c, const uint8_t* packed_local, int64_t n_local, uint8_t* packed_all,
                       const int64_t* rows_per_rank, int root, int async)
{
    int rc = need_comm(c, "dce_gather_results");
    if (rc) return rc;
    if ((rc = dce_internal_quiesce(c))) return rc;
    Rccl* r = rccl(c);
    if (!r) return DCE_ERR_COMM;
    const int W = c->comm_world, me = c->comm_rank;
    if (root < 0 || root >= W || n_local < 0 || (n_local > 0 && !packed_local))
        return fail(c, DCE_ERR_ARG, "dce_gather_results: bad argument");
    if (((reinterpret_cast<uintptr_t>(packed_local) | reinterpret_cast<uintptr_t>(packed_all)) & 3u) != 0)
        return fail(c, DCE_ERR_ARG, "dce_gather_results: the packed row buffers must be 4-byte aligned");
    bool uniform = true;
    int64_t total = 0;
    if (rows_per_rank) {
        if (rows_per_rank[me] != n_local)
            return fail(c, DCE_ERR_ARG, "dce_gather_results: rank %d holds %lld rows, rows_per_rank says %lld",
                        me, (long long)n_local, (long long)rows_per_rank[me]);
        for (int g = 0; g < W; ++g) {
            if (rows_per_rank[g] < 0) return fail(c, DCE_ERR_ARG, "dce_gather_results: negative row count for rank %d", g);
            uniform = uniform && rows_per_rank[g] == n_local;
            total += rows_per_rank[g];
        }
    } else {
        total = n_local * W;
    }
    if (me == root && total > 0 && !packed_all) return fail(c, DCE_ERR_ARG, "dce_gather_results: the root needs packed_all");
    DEVICE_GUARD(c);
    if ((rc = comm_follow_ctx(c))) return rc;
    hipStream_t cs = c->comm_stream;
    ncclComm_t comm = (ncclComm_t)c->comm;
    if (uniform) {
        // (off the root the receive pointer is never written; a caller may pass NULL there, RCCL gets a valid pointer anyway)
        if (n_local > 0)
            NCCL_TRY(c, r, r->Gather(packed_local, (me == root || packed_all) ? (void*)packed_all : (void*)packed_local,
                                     (size_t)n_local * dce::PACKED_ROW, ncclUint8, root, comm, cs));
    } else {
        // ragged shards: ONE group of point-to-point transfers, each block straight into its place on the root
        NCCL_TRY(c, r, r->GroupStart());
        ncclResult_t e = ncclSuccess;
        if (n_local > 0) e = r->Send(packed_local, (size_t)n_local * dce::PACKED_ROW, ncclUint8, root, comm, cs);
        if (me == root) {
            int64_t off = 0;
            for (int g = 0; g < W && e == ncclSuccess; ++g) {
                if (rows_per_rank[g] > 0)
                    e = r->Recv(packed_all + off * dce::PACKED_ROW, (size_t)rows_per_rank[g] * dce::PACKED_ROW, ncclUint8, g, comm, cs);
                off += rows_per_rank[g];
            }
        }
        const ncclResult_t e2 = r->GroupEnd();
        if (e != ncclSuccess || e2 != ncclSuccess)
            return fail(c, DCE_ERR_COMM, "grouped ncclSend/ncclRecv failed: %s", r->GetErrorString(e != ncclSuccess ? e : e2));
    }
    if (async) {
        // Depth-two pipeline for a caller that alternates two send (and receive) buffers: this gather runs behind the
        // kernels queued after it; work queued on the ctx stream from now on waits only for the gather issued BEFORE
        // this one -- the one whose buffers the next step is about to overwrite.
        const int slot = (int)(c->comm_issued & 1);
        HIP_TRY(c, hipEventRecord(c->comm_done[slot], cs));
        if (c->comm_issued > 0) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->comm_done[slot ^ 1], 0));
        c->comm_issued += 1;
    } else {
        // stream-ordered: later work on the ctx stream (e.g. dce_unpack_results on the root) sees the gathered rows
        HIP_TRY(c, hipEventRecord(c->comm_done[0], cs));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->comm_done[0], 0));     // comm_stream is in order: earlier gathers are done too
        c->comm_issued = 0;
    }
    return DCE_OK;
}

int dce_allreduce_counts(dce_ctx* c, int64_t* counts, int on_device)
{
    int rc = need_comm(c, "dce_allreduce_counts");
    if (rc) return rc;
    if ((rc = dce_internal_quiesce(c))) return rc;
    if (!counts) return fail(c, DCE_ERR_ARG, "dce_allreduce_counts: NULL counts");
    Rccl* r = rccl(c);
    if (!r) return DCE_ERR_COMM;
    DEVICE_GUARD(c);
    int64_t* d = counts;
    if (!on_device) {
        if (c->d_in_bytes < 256 * sizeof(int64_t)) {
            if (c->d_in) { HIP_TRY(c, dce::dev_free(c->d_in)); c->d_in = nullptr; c->d_in_bytes = 0; }
            HIP_TRY(c, dce::dev_alloc(&c->d_in, c->tuning.guard_alloc, 4096));
            c->d_in_bytes = 4096;
        }
        d = reinterpret_cast<int64_t*>(c->d_in);
        HIP_TRY(c, hipMemcpyAsync(d, counts, 256 * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    }
    if ((rc = comm_follow_ctx(c))) return rc;
    NCCL_TRY(c, r, r->AllReduce(d, d, 256, ncclInt64, ncclSum, (ncclComm_t)c->comm, c->comm_stream));
    HIP_TRY(c, hipEventRecord(c->comm_done[0], c->comm_stream));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->comm_done[0], 0));
    if (!on_device) {
        HIP_TRY(c, hipMemcpyAsync(counts, d, 256 * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return DCE_OK;
}

int dce_comm_sync(dce_ctx* c)
{
    int rc = need_comm(c, "dce_comm_sync");
    if (rc) return rc;
    DEVICE_GUARD(c);
    HIP_TRY(c, hipStreamSynchronize(c->comm_stream));
    Rccl* r = rccl(c);
    ncclResult_t async_err = ncclSuccess;
    if (r && r->CommGetAsyncError((ncclComm_t)c->comm, &async_err) == ncclSuccess && async_err != ncclSuccess)
        return fail(c, DCE_ERR_COMM, "RCCL reported an asynchronous error: %s", r->GetErrorString(async_err));
    return DCE_OK;
}

int dce_comm_destroy(dce_ctx* c)
{
    if (!c) return DCE_ERR_ARG;
    if (!c->comm) return DCE_OK;
    Rccl* r = rccl(c);
    DEVICE_GUARD(c);
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    ncclComm_t comm = (ncclComm_t)c->comm;
    c->comm = nullptr;
    c->comm_world = 0;
    if (r) NCCL_TRY(c, r, r->CommDestroy(comm));
    return DCE_OK;
}

}  // extern "C"
