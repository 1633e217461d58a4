// bf_comm: the all-gather of the multi-GPU partition (include/bf_comm.h).  RCCL is loaded with dlopen at first use - libbf_hip.so does not link it,
// a single-GPU host never touches it - and called on the caller's HIP stream; a callback transport exists for tests and for hosts with their own
// fabric.  Only the handful of RCCL entry points used here are declared (signatures as in rccl/rccl.h of ROCm 7.2: ncclGetUniqueId :187,
// ncclCommInitRank :220, ncclCommDestroy :260, ncclGetErrorString :339, ncclAllGather :678).
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "bf_internal.h"
#include "../../include/bf_comm.h"

namespace {

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[BF_COMM_UNIQUE_ID_BYTES]; } ncclUniqueId;
const int NCCL_SUCCESS = 0;
const int NCCL_UINT8 = 1;          // ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    std::string error;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy the process already has (PyTorch ships its own librccl.so) is preferred: two RCCL instances in one process would each
        // open their own set of IPC handles
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) { r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL); if (r.handle) break; }
        if (const char* e = getenv("BF_RCCL_LIBRARY")) { if (!r.handle) r.handle = dlopen(e, RTLD_NOW | RTLD_LOCAL); }
        for (const char* n : names) { if (r.handle) break; r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL); }
        if (!r.handle) { const char* why = dlerror(); r.error = std::string("librccl not found: ") + (why ? why : "?"); return; }      // (dlerror clears itself: one call)
        r.GetUniqueId = (int (*)(ncclUniqueId*))dlsym(r.handle, "ncclGetUniqueId");
        r.CommInitRank = (int (*)(ncclComm_t*, int, ncclUniqueId, int))dlsym(r.handle, "ncclCommInitRank");
        r.CommDestroy = (int (*)(ncclComm_t))dlsym(r.handle, "ncclCommDestroy");
        r.GetErrorString = (const char* (*)(int))dlsym(r.handle, "ncclGetErrorString");
        r.AllGather = (int (*)(const void*, void*, size_t, int, ncclComm_t, hipStream_t))dlsym(r.handle, "ncclAllGather");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.GetErrorString || !r.AllGather) r.error = "librccl lacks an expected symbol";
    });
    return r;
}

int rcclReady() {
    Rccl& r = rccl();
    if (!r.error.empty()) { bf::set_error("%s", r.error.c_str()); return BF_ERR_INVALID_ARG; }
    return BF_OK;
}

#define BF_NCCL_TRY(expr) do { const int _e = (expr); if (_e != NCCL_SUCCESS) { bf::set_error("%s: %s", #expr, rccl().GetErrorString(_e)); return BF_ERR_HIP; } } while (0)

}  // namespace

struct bf_comm {
    uint32_t world = 1, rank = 0;
    ncclComm_t nccl = nullptr; bool ownsNccl = false;
    bf_all_gather_fn fn = nullptr; void* user = nullptr;
    // staging of bf_chunk_exchange
    uint8_t *d_send = nullptr, *d_recv = nullptr; uint64_t stageBytes = 0;
};

extern "C" {

int bf_comm_unique_id(uint8_t id[BF_COMM_UNIQUE_ID_BYTES]) {
    BF_REQUIRE(id, "null argument");
    const int rc = rcclReady();
    if (rc != BF_OK) return rc;
    ncclUniqueId u;
    BF_NCCL_TRY(rccl().GetUniqueId(&u));
    memcpy(id, u.internal, BF_COMM_UNIQUE_ID_BYTES);
    return BF_OK;
}

int bf_comm_create_rccl(const uint8_t id[BF_COMM_UNIQUE_ID_BYTES], uint32_t world, uint32_t rank, bf_comm** out) {
    BF_REQUIRE(id && out && world >= 1 && rank < world, "bad argument");
    const int rc = rcclReady();
    if (rc != BF_OK) return rc;
    ncclUniqueId u;
    memcpy(u.internal, id, BF_COMM_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    BF_NCCL_TRY(rccl().CommInitRank(&c, (int)world, u, (int)rank));
    bf_comm* m = new bf_comm;
    m->world = world; m->rank = rank; m->nccl = c; m->ownsNccl = true;
    *out = m;
    return BF_OK;
}

int bf_comm_from_rccl(void* nccl_comm, uint32_t world, uint32_t rank, bf_comm** out) {
    BF_REQUIRE(nccl_comm && out && world >= 1 && rank < world, "bad argument");
    const int rc = rcclReady();
    if (rc != BF_OK) return rc;
    bf_comm* m = new bf_comm;
    m->world = world; m->rank = rank; m->nccl = (ncclComm_t)nccl_comm; m->ownsNccl = false;
    *out = m;
    return BF_OK;
}

int bf_comm_create_callback(bf_all_gather_fn fn, void* user, uint32_t world, uint32_t rank, bf_comm** out) {
    BF_REQUIRE(fn && out && world >= 1 && rank < world, "bad argument");
    bf_comm* m = new bf_comm;
    m->world = world; m->rank = rank; m->fn = fn; m->user = user;
    *out = m;
    return BF_OK;
}

int bf_comm_destroy(bf_comm* c) {
    if (!c) return BF_OK;
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->nccl && c->ownsNccl) (void)rccl().CommDestroy(c->nccl);
    delete c;
    return BF_OK;
}

int bf_comm_world(bf_comm* c, uint32_t* world, uint32_t* rank) {
    BF_REQUIRE(c, "null comm");
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    return BF_OK;
}

int bf_comm_all_gather(bf_comm* c, const void* d_send, void* d_recv, uint64_t bytes, void* hip_stream) {
    BF_REQUIRE(c && d_send && d_recv, "null argument");
    if (bytes == 0) return BF_OK;
    if (c->fn) {
        const int rc = c->fn(c->user, d_send, d_recv, bytes, hip_stream);
        if (rc != 0) { bf::set_error("all-gather callback failed with %d", rc); return BF_ERR_HIP; }
        return BF_OK;
    }
    BF_REQUIRE(c->nccl, "communicator without a transport");
    BF_NCCL_TRY(rccl().AllGather(d_send, d_recv, (size_t)bytes, NCCL_UINT8, c->nccl, (hipStream_t)hip_stream));
    return BF_OK;
}

int bf_chunk_exchange(bf_comm* c, const void* h_mine, void* h_all, uint64_t package_bytes, void* hip_stream) {
    BF_REQUIRE(c && h_mine && h_all && package_bytes > 0, "bad argument");
    hipStream_t st = (hipStream_t)hip_stream;
    if (c->world == 1) { memcpy(h_all, h_mine, package_bytes); return BF_OK; }
    if (c->stageBytes < package_bytes) {
        if (c->d_send) (void)hipFree(c->d_send);
        if (c->d_recv) (void)hipFree(c->d_recv);
        c->d_send = c->d_recv = nullptr; c->stageBytes = 0;
        BF_HIP_TRY(BF_MALLOC((void**)&c->d_send, package_bytes));
        BF_HIP_TRY(BF_MALLOC((void**)&c->d_recv, package_bytes * c->world));
        c->stageBytes = package_bytes;
    }
    BF_HIP_TRY(hipMemcpyAsync(c->d_send, h_mine, package_bytes, hipMemcpyHostToDevice, st));
    const int rc = bf_comm_all_gather(c, c->d_send, c->d_recv, package_bytes, hip_stream);
    if (rc != BF_OK) return rc;
    BF_HIP_TRY(hipMemcpyAsync(h_all, c->d_recv, package_bytes * c->world, hipMemcpyDeviceToHost, st));
    BF_HIP_TRY(hipStreamSynchronize(st));
    return BF_OK;
}

}  // extern "C"
