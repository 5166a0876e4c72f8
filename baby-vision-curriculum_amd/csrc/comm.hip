// Communication group of the C ABI (include/bvc.h, "communication"): the gradient all-reduce of the data-parallel wrapper and the
// embedding all-gather of the global-batch SimCLR loss, on RCCL, with the communication stream and its event fences owned HERE
// (one process per GPU; the host thread that drives the step's context drives its communicator).
//
// Reference semantics it serves:
//   DDP's reducer:   pretraining/generative/pretrain_videomae.py:180-181,312 (gradients averaged over ranks during backward)
//   AllReduce:       pretraining/generative/ddputils.py:53-68
//   AllGather:       pretraining/predictive/distributed.py:49-76
//   process group:   pretraining/generative/pretrain_videomae.py:87-90 (dist.init_process_group("nccl", rank, world_size))
//
// RCCL is bound at run time (dlopen), never at link time: a process that has imported torch already holds a librccl.so (torch
// ships its own), and two RCCL instances in one process is what to avoid.  The library that is already mapped is preferred
// (found with dl_iterate_phdr), then librccl.so.1 / librccl.so from the loader path.  The ABI used - ncclGetUniqueId,
// ncclCommInitRank, ncclAllReduce, ncclAllGather, ncclBroadcast, ncclCommDestroy, ncclGetErrorString - is that of
// /opt/rocm/include/rccl/rccl.h and has been stable across the 2.x series.
#include <dlfcn.h>
#include <link.h>
#include <string.h>

#include <mutex>
#include <string>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "common.h"

namespace bvc {
namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string path, error;
};

int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* data) {
    if (info->dlpi_name && strstr(info->dlpi_name, "librccl.so")) {
        *static_cast<std::string*>(data) = info->dlpi_name;
        return 1;
    }
    return 0;
}

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        std::string loaded;
        dl_iterate_phdr(find_loaded_rccl, &loaded);
        const char* candidates[] = {loaded.empty() ? nullptr : loaded.c_str(), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* c : candidates) {
            if (!c) continue;
            r.handle = dlopen(c, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) { r.path = c; break; }
        }
        if (!r.handle) { r.error = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : "?"); return; }
        auto sym = [&](const char* name) -> void* {
            void* p = dlsym(r.handle, name);
            if (!p && r.error.empty()) r.error = std::string("RCCL symbol missing: ") + name;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
    });
    return r;
}

}  // namespace
}  // namespace bvc

// One communicator = one RCCL rank + the communication stream + the fences between it and the caller's streams.
struct bvc_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;          // the communication stream (non-blocking, so it never syncs with the null stream)
    static constexpr int kFences = 16;     // producer -> comm fences, reused round robin (a bucket's fence is consumed long before 16 more are recorded)
    hipEvent_t fence[kFences] = {};
    int next_fence = 0;
    hipEvent_t tail = nullptr;             // recorded on the communication stream after each collective
    bool pending = false;                  // something was enqueued on the communication stream since the last bvc_comm_wait
};

#define BVC_CHECK_NCCL(expr)                                                                                          \
    do {                                                                                                              \
        ncclResult_t _r = (expr);                                                                                     \
        if (_r != ncclSuccess) {                                                                                      \
            bvc::set_error("%s failed: %s (%s:%d)", #expr, bvc::rccl().GetErrorString ? bvc::rccl().GetErrorString(_r) : "?", __FILE__, __LINE__); \
            return BVC_ERR_HIP;                                                                                       \
        }                                                                                                             \
    } while (0)

static int need_rccl() {
    bvc::Rccl& r = bvc::rccl();
    if (!r.handle || !r.error.empty()) {
        bvc::set_error("RCCL unavailable: %s", r.error.c_str());
        return BVC_ERR_HIP;
    }
    return BVC_OK;
}

extern "C" {

int bvc_comm_unique_id(void* id_out) {
    BVC_REQUIRE(id_out != nullptr, "comm_unique_id: null buffer");
    static_assert(sizeof(ncclUniqueId) == BVC_COMM_ID_BYTES, "ncclUniqueId is 128 bytes in the RCCL ABI this header describes");
    if (int rc = need_rccl()) return rc;
    ncclUniqueId id;
    BVC_CHECK_NCCL(bvc::rccl().GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return BVC_OK;
}

int bvc_comm_init(int rank, int world, const void* id, bvc_comm** out) {
    BVC_REQUIRE(out != nullptr && id != nullptr, "comm_init: null argument");
    BVC_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d of %d", rank, world);
    if (int rc = need_rccl()) return rc;
    bvc_comm* c = new bvc_comm();
    c->rank = rank;
    c->world = world;
    auto fail = [&](int rc) { bvc_comm_destroy(c); return rc; };
    if (hipGetDevice(&c->device) != hipSuccess) { bvc::set_error("comm_init: hipGetDevice failed"); return fail(BVC_ERR_HIP); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { bvc::set_error("comm_init: stream creation failed"); return fail(BVC_ERR_HIP); }
    for (int i = 0; i < bvc_comm::kFences; ++i)
        if (hipEventCreateWithFlags(&c->fence[i], hipEventDisableTiming) != hipSuccess) { bvc::set_error("comm_init: event creation failed"); return fail(BVC_ERR_HIP); }
    if (hipEventCreateWithFlags(&c->tail, hipEventDisableTiming) != hipSuccess) { bvc::set_error("comm_init: event creation failed"); return fail(BVC_ERR_HIP); }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclResult_t r = bvc::rccl().CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        bvc::set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, bvc::rccl().GetErrorString(r));
        c->comm = nullptr;
        return fail(BVC_ERR_HIP);
    }
    *out = c;
    return BVC_OK;
}

int bvc_comm_destroy(bvc_comm* c) {
    if (!c) return BVC_OK;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)bvc::rccl().CommDestroy(c->comm);
    for (int i = 0; i < bvc_comm::kFences; ++i)
        if (c->fence[i]) (void)hipEventDestroy(c->fence[i]);
    if (c->tail) (void)hipEventDestroy(c->tail);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return BVC_OK;
}

void* bvc_comm_stream(const bvc_comm* c) { return c ? (void*)c->stream : nullptr; }
int bvc_comm_rank(const bvc_comm* c) { return c ? c->rank : -1; }
int bvc_comm_world(const bvc_comm* c) { return c ? c->world : -1; }

const char* bvc_comm_library(void) {
    bvc::Rccl& r = bvc::rccl();
    static thread_local std::string s;
    int v = 0;
    if (r.GetVersion) (void)r.GetVersion(&v);
    s = r.path + " (version " + std::to_string(v) + ")" + (r.error.empty() ? "" : " [" + r.error + "]");
    return s.c_str();
}

// In-place sum (average != 0: mean) over ranks of count f32 at buf, on the communication stream, ordered after everything
// enqueued so far on producer_stream (the stream whose kernels wrote buf).  Returns at once.
int bvc_allreduce_bucket(bvc_comm* c, float* buf, int64_t count, int average, void* producer_stream) {
    BVC_REQUIRE(c && c->comm, "allreduce_bucket: no communicator");
    BVC_REQUIRE(buf != nullptr && count > 0, "allreduce_bucket: empty bucket");
    hipEvent_t ev = c->fence[c->next_fence];
    c->next_fence = (c->next_fence + 1) % bvc_comm::kFences;
    BVC_CHECK_HIP(hipEventRecord(ev, (hipStream_t)producer_stream));
    BVC_CHECK_HIP(hipStreamWaitEvent(c->stream, ev, 0));
    BVC_CHECK_NCCL(bvc::rccl().AllReduce(buf, buf, (size_t)count, ncclFloat32, average ? ncclAvg : ncclSum, c->comm, c->stream));
    c->pending = true;
    return BVC_OK;
}

// `stream` waits for every collective enqueued on the communication stream so far (end of backward, before the optimizer).
int bvc_comm_wait(bvc_comm* c, void* stream) {
    BVC_REQUIRE(c != nullptr, "comm_wait: no communicator");
    if (!c->pending) return BVC_OK;
    BVC_CHECK_HIP(hipEventRecord(c->tail, c->stream));
    BVC_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, c->tail, 0));
    c->pending = false;
    return BVC_OK;
}

// The three collectives below are consumed by the caller's very next kernels, yet they run on the COMMUNICATION stream like the
// buckets do: one communicator is then driven from one stream only, in program order (RCCL executes a communicator's operations in
// issue order; issuing them from two streams adds nothing but the chance of a cross-stream wait cycle).  hop_in orders the
// communication stream behind everything enqueued on the caller's stream, hop_out orders the caller's stream behind the collective.
static int hop_in(bvc_comm* c, hipStream_t s) {
    if (s == c->stream) return BVC_OK;
    hipEvent_t ev = c->fence[c->next_fence];
    c->next_fence = (c->next_fence + 1) % bvc_comm::kFences;
    BVC_CHECK_HIP(hipEventRecord(ev, s));
    BVC_CHECK_HIP(hipStreamWaitEvent(c->stream, ev, 0));
    return BVC_OK;
}
static int hop_out(bvc_comm* c, hipStream_t s) {
    if (s == c->stream) return BVC_OK;
    BVC_CHECK_HIP(hipEventRecord(c->tail, c->stream));
    BVC_CHECK_HIP(hipStreamWaitEvent(s, c->tail, 0));
    return BVC_OK;
}

// recv[r * bytes_per_rank ...] = rank r's send buffer, for every r; ordered after `stream`, `stream` continues after it.
int bvc_allgather(bvc_comm* c, const void* send, void* recv, int64_t bytes_per_rank, void* stream) {
    BVC_REQUIRE(c && c->comm, "allgather: no communicator");
    BVC_REQUIRE(send && recv && bytes_per_rank > 0, "allgather: empty buffer");
    if (int rc = hop_in(c, (hipStream_t)stream)) return rc;
    BVC_CHECK_NCCL(bvc::rccl().AllGather(send, recv, (size_t)bytes_per_rank, ncclUint8, c->comm, c->stream));
    return hop_out(c, (hipStream_t)stream);
}

// In-place sum / mean over ranks of count f32 (the backward of the all-gather: sum, then the caller keeps its own rows; the loss
// scalar, which thereby follows the step's last gradient bucket on the communication stream).
int bvc_allreduce(bvc_comm* c, float* buf, int64_t count, int average, void* stream) {
    BVC_REQUIRE(c && c->comm, "allreduce: no communicator");
    BVC_REQUIRE(buf != nullptr && count > 0, "allreduce: empty buffer");
    if (int rc = hop_in(c, (hipStream_t)stream)) return rc;
    BVC_CHECK_NCCL(bvc::rccl().AllReduce(buf, buf, (size_t)count, ncclFloat32, average ? ncclAvg : ncclSum, c->comm, c->stream));
    return hop_out(c, (hipStream_t)stream);
}

// buf of root -> buf of every rank (the module-state sync at wrap time).
int bvc_broadcast(bvc_comm* c, void* buf, int64_t bytes, int root, void* stream) {
    BVC_REQUIRE(c && c->comm, "broadcast: no communicator");
    BVC_REQUIRE(buf != nullptr && bytes > 0 && root >= 0 && root < c->world, "broadcast: bad argument");
    if (int rc = hop_in(c, (hipStream_t)stream)) return rc;
    BVC_CHECK_NCCL(bvc::rccl().Broadcast(buf, buf, (size_t)bytes, ncclUint8, root, c->comm, c->stream));
    return hop_out(c, (hipStream_t)stream);
}

}  // extern "C"
