// rsik_comm.hpp — the RCCL all-gather entry points of the C ABI (rsik_comm_*, rsik_allgather)
// (included by rsik_lib.hip inside its extern "C" block, after the context and error helpers)
#pragma once

// ------------------------------------------------------------------------------------------
// Multi-GPU: the all-gather of the final arrays over RCCL, for hosts without torch.distributed.
// librccl is opened at run time (dlopen), so single-GPU users never need it installed.
// ------------------------------------------------------------------------------------------
namespace {
struct NcclUid { char internal[128]; };  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
struct Rccl {
    void* so = nullptr;
    int (*GetUniqueId)(NcclUid*) = nullptr;
    int (*CommInitRank)(void**, int, NcclUid, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};
Rccl* rccl() {
    static Rccl R;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* nm : names) {
            R.so = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (R.so) break;
        }
        if (!R.so) { R.err = "librccl.so not found (dlopen)"; return &R; }
        R.GetUniqueId = (int (*)(NcclUid*))dlsym(R.so, "ncclGetUniqueId");
        R.CommInitRank = (int (*)(void**, int, NcclUid, int))dlsym(R.so, "ncclCommInitRank");
        R.CommDestroy = (int (*)(void*))dlsym(R.so, "ncclCommDestroy");
        R.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(R.so, "ncclAllGather");
        R.GetErrorString = (const char* (*)(int))dlsym(R.so, "ncclGetErrorString");
        if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllGather) R.err = "librccl.so lacks an expected symbol";
    }
    return &R;
}
int rccl_fail(rsik_ctx* ctx, const char* what, int code) {
    Rccl* R = rccl();
    return fail(ctx, RSIK_E_HIP, std::string(what) + ": " + ((R->GetErrorString && code) ? R->GetErrorString(code) : R->err.c_str()));
}
}  // namespace

int rsik_comm_unique_id(void* id128) {
    Rccl* R = rccl();
    if (!id128 || !R->err.empty()) return fail(nullptr, RSIK_E_HIP, "rsik_comm_unique_id: " + (id128 ? R->err : std::string("NULL buffer")));
    NcclUid u;
    int rc = R->GetUniqueId(&u);
    if (rc != 0) return rccl_fail(nullptr, "ncclGetUniqueId", rc);
    std::memcpy(id128, u.internal, sizeof u.internal);
    return RSIK_OK;
}

int rsik_comm_init_rank(rsik_ctx* ctx, int nranks, int rank, const void* id128, void** comm) {
    if (!ctx) return RSIK_E_INVALID;
    if (!id128 || !comm || nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, RSIK_E_INVALID, "rsik_comm_init_rank: bad argument");
    Rccl* R = rccl();
    if (!R->err.empty()) return fail(ctx, RSIK_E_HIP, "rsik_comm_init_rank: " + R->err);
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    NcclUid u;
    std::memcpy(u.internal, id128, sizeof u.internal);
    *comm = nullptr;
    int rc = R->CommInitRank(comm, nranks, u, rank);
    if (rc != 0) return rccl_fail(ctx, "ncclCommInitRank", rc);
    return RSIK_OK;
}

int rsik_comm_destroy(rsik_ctx* ctx, void* comm) {
    if (!ctx) return RSIK_E_INVALID;
    if (!comm) return RSIK_OK;
    Rccl* R = rccl();
    if (!R->err.empty()) return fail(ctx, RSIK_E_HIP, "rsik_comm_destroy: " + R->err);
    int rc = R->CommDestroy(comm);
    if (rc != 0) return rccl_fail(ctx, "ncclCommDestroy", rc);
    return RSIK_OK;
}

int rsik_allgather(rsik_ctx* ctx, void* comm, const void* send, void* recv, size_t bytes_per_rank) {
    if (!ctx) return RSIK_E_INVALID;
    if (!comm || !recv || (!send && bytes_per_rank)) return fail(ctx, RSIK_E_INVALID, "rsik_allgather: NULL argument");
    if (bytes_per_rank == 0) return RSIK_OK;
    Rccl* R = rccl();
    if (!R->err.empty()) return fail(ctx, RSIK_E_HIP, "rsik_allgather: " + R->err);
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = R->AllGather(send, recv, bytes_per_rank, /*ncclInt8*/ 0, comm, ctx->stream);
    if (rc != 0) return rccl_fail(ctx, "ncclAllGather", rc);
    return RSIK_OK;
}
