// comm.cpp -- the data-parallel exchange of the sharded Lloyd iteration for hosts that do not go through
// torch.distributed: thin wrappers over RCCL (the ncclComm_t is the caller's), SURVEY.md section 8b item 7.
//
// The reference has no communication of its own (its only multi-GPU mode is faiss' internal replica sharding,
// processors/cluster_creator.py:42-56 with gpu=True).  The sharded k-means of this library exchanges ONE packed partial
// [k*(d+1) + 2] fp32 per iteration and adds the ranks' partials in ascending rank order, so that every rank -- and the
// CPU restatement's n_shards mode the tests compare with -- gets the same bits; an ncclAllReduce(sum) does not promise an order, hence all-gather +
// the fixed-order sum of at_sum_parts_f32 / at_centroid_finalize_f32.
//
// RCCL is looked up at run time among the libraries the process has ALREADY loaded (a host that created the
// communicator has one); nothing new is ever loaded -- a second copy of RCCL must not be handed another copy's
// communicator -- and the library itself keeps no link-time dependency on it.  No loaded copy: AT_E_COMM.
#include <dlfcn.h>

#include <mutex>

#include "at_internal.h"

namespace {

typedef int (*allgather_fn)(const void*, void*, size_t, int /*ncclDataType_t*/, void* /*ncclComm_t*/, hipStream_t);
typedef int (*count_fn)(void*, int*);
typedef const char* (*errstr_fn)(int);

struct Rccl {
    allgather_fn all_gather = nullptr;
    count_fn comm_count = nullptr;
    count_fn comm_rank = nullptr;
    errstr_fn error_string = nullptr;
    bool tried = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::mutex m;
    std::lock_guard<std::mutex> g(m);
    if (!r.tried) {
        r.tried = true;
        // Only a copy of RCCL that the process has loaded ALREADY is bound: the communicator the caller hands over was
        // created by some copy, and calling into another one with it is undefined behaviour.  The global scope first
        // (a host linked against RCCL), then -- without loading anything, RTLD_NOLOAD -- the sonames a copy loaded
        // with RTLD_LOCAL goes by (torch's bundled librccl.so is loaded that way, as a dependency of libtorch_hip.so).
        void* h = dlopen(nullptr, RTLD_NOW);
        if (h && !dlsym(h, "ncclAllGather")) h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "librccl.so.2"})
            if (!h) {
                h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
                if (h && !dlsym(h, "ncclAllGather")) h = nullptr;
            }
        if (h) {
            r.all_gather = reinterpret_cast<allgather_fn>(dlsym(h, "ncclAllGather"));
            r.comm_count = reinterpret_cast<count_fn>(dlsym(h, "ncclCommCount"));
            r.comm_rank = reinterpret_cast<count_fn>(dlsym(h, "ncclCommUserRank"));
            r.error_string = reinterpret_cast<errstr_fn>(dlsym(h, "ncclGetErrorString"));
        }
    }
    return r;
}

constexpr int NCCL_FLOAT32 = 7;   // ncclFloat32 (rccl.h: ncclInt8 0 ... ncclFloat16 6, ncclFloat32 7, ncclFloat64 8)

}  // namespace

extern "C" {

// parts [n_ranks][count] <- every rank's part [count] (rank r's at parts + r*count), over the caller's communicator.
int at_comm_allgather_f32(at_ctx* ctx, void* nccl_comm, const float* part, float* parts, int64_t count, void* stream_) {
    AT_REQUIRE(ctx && nccl_comm && part && parts && count > 0, "at_comm_allgather_f32: bad arguments");
    Rccl& r = rccl();
    if (!r.all_gather) return at_fail(AT_E_COMM, "at_comm_allgather_f32: no RCCL is loaded in this process (the communicator's own copy is looked up, nothing is loaded)");
    AT_HIP(hipSetDevice(ctx->device));
    const int rc = r.all_gather(part, parts, (size_t)count, NCCL_FLOAT32, nccl_comm, static_cast<hipStream_t>(stream_));
    if (rc != 0) return at_fail(AT_E_COMM, "at_comm_allgather_f32: ncclAllGather failed: %s", r.error_string ? r.error_string(rc) : "?");
    return AT_OK;
}

// out [count] <- part of rank 0 + part of rank 1 + ... in ascending rank order (fp32, the same bits on every rank):
// the all-reduce of the sharded Lloyd iteration.  parts_scratch: [n_ranks * count] floats of the caller's.
int at_comm_allreduce_ordered_f32(at_ctx* ctx, void* nccl_comm, const float* part, float* parts_scratch, float* out,
                                  int64_t count, void* stream_) {
    AT_REQUIRE(ctx && nccl_comm && part && parts_scratch && out && count > 0, "at_comm_allreduce_ordered_f32: bad arguments");
    Rccl& r = rccl();
    if (!r.all_gather || !r.comm_count) return at_fail(AT_E_COMM, "at_comm_allreduce_ordered_f32: no RCCL is loaded in this process (the communicator's own copy is looked up, nothing is loaded)");
    int n_ranks = 0;
    const int rcq = r.comm_count(nccl_comm, &n_ranks);
    if (rcq != 0 || n_ranks < 1) return at_fail(AT_E_COMM, "at_comm_allreduce_ordered_f32: ncclCommCount failed");
    int rc = at_comm_allgather_f32(ctx, nccl_comm, part, parts_scratch, count, stream_);
    if (rc) return rc;
    return at_sum_parts_f32(ctx, parts_scratch, count, n_ranks, count, out, stream_);
}

}  // extern "C"
