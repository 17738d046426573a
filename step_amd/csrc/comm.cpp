// Data-parallel collectives of the training step on RCCL's C API (SURVEY.md 8b "grad_allreduce(handle, stream, flat_grad*, n)", 8e: one
// gradient all-reduce per step, plus the small sums of the time-sliced graph learner).  The reference gets these from
// torch.nn.parallel.DistributedDataParallel (easytorch wraps the model when GPU_NUM > 1: step/STEP_PEMS04.py:30, STEP_PEMS07.py:29,83,117);
// here they are plain ncclAllReduce calls issued in stream order -- no work objects, no host synchronisation, no Python between the backward's
// kernels and the collective (one rank through torch.distributed: +0.45 ms of host time per step, profiles/r04_b_C2_rccl1_noshard_timeline.md).
//
// librccl is NOT linked: it is resolved at the first step_comm_* call with dlopen, preferring the copy already loaded into the process
// (torch ships its own librccl.so; two copies in one process would be two runtimes), so the library loads -- and everything else works --
// on a box without RCCL.  A communicator handle owns: the ncclComm_t, one side stream for the overlapped all-reduce of the fc weight
// gradient, and two events.  Buffers are the caller's; all reductions are in place.
#include "common.h"
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
char g_rccl_dlerror[512] = "";          // the loader's message of the failed dlopen, captured once where it happened

void keep_dlerror() {
    const char* e = dlerror();          // (one call: dlerror() clears the message it returns)
    snprintf(g_rccl_dlerror, sizeof g_rccl_dlerror, "%s", e ? e : "");
}

void load_rccl() {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    // $STEP_RCCL_LIB is an override: that file and no other (RTLD_LOCAL: its symbols must not interpose the copy torch has loaded; the
    // two-rank test on one device points it at tests/fake_rccl).  Otherwise a copy that is already in the process first (RTLD_NOLOAD), then
    // the usual search.
    const char* forced = getenv("STEP_RCCL_LIB");
    if (forced && forced[0]) {
        h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!h) { keep_dlerror(); return; }
    }
    for (int pass = 0; pass < 2 && !h; ++pass)
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) break;
        }
    if (!h) { keep_dlerror(); return; }
    g_rccl.handle = h;
#define STEP_RCCL_SYM(field, name) g_rccl.field = (decltype(g_rccl.field))dlsym(h, name)
    STEP_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    STEP_RCCL_SYM(CommInitRank, "ncclCommInitRank");
    STEP_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    STEP_RCCL_SYM(AllReduce, "ncclAllReduce");
    STEP_RCCL_SYM(Broadcast, "ncclBroadcast");
    STEP_RCCL_SYM(GetVersion, "ncclGetVersion");
    STEP_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef STEP_RCCL_SYM
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.Broadcast && g_rccl.GetErrorString;
}

int need_rccl(const char* what) {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.ok) {
        step_set_error("%s: librccl could not be loaded (dlopen of librccl.so / $STEP_RCCL_LIB failed or symbols are missing): %s", what,
                       g_rccl_dlerror[0] ? g_rccl_dlerror : "no dlerror");
        return STEP_ERR_HIP;
    }
    return STEP_OK;
}

#define STEP_NCCL(call, what)                                                          \
    do {                                                                               \
        ncclResult_t r_ = (call);                                                      \
        if (r_ != ncclSuccess) {                                                       \
            step_set_error("%s: RCCL error: %s", what, g_rccl.GetErrorString(r_));     \
            return STEP_ERR_HIP;                                                       \
        }                                                                              \
    } while (0)
#define STEP_HIPCK(call, what)                                                         \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            step_set_error("%s: %s", what, hipGetErrorString(e_));                     \
            return STEP_ERR_HIP;                                                       \
        }                                                                              \
    } while (0)

struct StepComm {
    uint32_t magic;
    ncclComm_t comm;
    int nranks, rank;
    hipStream_t side;            // the overlapped (asynchronous) all-reduce runs here
    bool own_side;               // created here (destroyed here), or handed over by step_comm_set_side_stream (the caller's)
    hipEvent_t ready, done;      // compute stream -> side stream, side stream -> compute stream
    bool pending;                // an asynchronous reduction has been queued and not joined yet
};
constexpr uint32_t COMM_MAGIC = 0x53434D31;      // "SCM1"

StepComm* as_comm(void* h) {
    StepComm* c = (StepComm*)h;
    return (c && c->magic == COMM_MAGIC) ? c : nullptr;
}

int dtype_of(int step_dtype, ncclDataType_t* out, const char* what) {
    switch (step_dtype) {
        case STEP_COMM_F32: *out = ncclFloat32; return STEP_OK;
        case STEP_COMM_F64: *out = ncclFloat64; return STEP_OK;
        case STEP_COMM_U8: *out = ncclUint8; return STEP_OK;
        default: step_set_error("%s: unknown element type %d", what, step_dtype); return STEP_ERR_ARG;
    }
}

}  // namespace

extern "C" int step_comm_available(void) {
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.ok ? 1 : 0;
}

extern "C" int step_comm_version(void) {
    std::call_once(g_rccl_once, load_rccl);
    int v = 0;
    if (g_rccl.ok && g_rccl.GetVersion) (void)g_rccl.GetVersion(&v);
    return v;
}

extern "C" int step_comm_unique_id(void* id128) {
    STEP_REQUIRE(id128, "comm_unique_id: null output");
    STEP_TRY(need_rccl("comm_unique_id"));
    static_assert(sizeof(ncclUniqueId) == STEP_COMM_ID_BYTES, "ncclUniqueId is not 128 bytes");
    STEP_NCCL(g_rccl.GetUniqueId((ncclUniqueId*)id128), "comm_unique_id");
    return STEP_OK;
}

extern "C" int step_comm_init_rank(const void* id128, int nranks, int rank, void** comm_out) {
    STEP_REQUIRE(id128 && comm_out, "comm_init_rank: null argument");
    STEP_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "comm_init_rank: rank %d of %d", rank, nranks);
    STEP_TRY(need_rccl("comm_init_rank"));
    StepComm* c = new StepComm();
    c->magic = COMM_MAGIC; c->nranks = nranks; c->rank = rank; c->pending = false; c->side = nullptr; c->own_side = true; c->ready = c->done = nullptr; c->comm = nullptr;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        step_set_error("comm_init_rank: RCCL error: %s", g_rccl.GetErrorString(r));
        delete c;
        return STEP_ERR_HIP;
    }
    hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
    if (e != hipSuccess) {
        step_set_error("comm_init_rank: %s", hipGetErrorString(e));
        (void)g_rccl.CommDestroy(c->comm);
        delete c;
        return STEP_ERR_HIP;
    }
    *comm_out = c;
    return STEP_OK;
}

extern "C" int step_comm_destroy(void* comm) {
    StepComm* c = as_comm(comm);
    STEP_REQUIRE(c, "comm_destroy: not a communicator handle");
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->side && c->own_side) (void)hipStreamDestroy(c->side);
    c->magic = 0;
    delete c;
    return STEP_OK;
}

// The overlapped all-reduce runs on `stream` from now on (the caller's: e.g. one it has probed with step_streams_concurrent to run
// concurrently with its compute stream -- a stream that shares the compute stream's hardware queue would put the 100 MB all-reduce in front
// of the rest of the backward).  The communicator's own stream is released.
extern "C" int step_comm_set_side_stream(void* comm, void* stream) {
    StepComm* c = as_comm(comm);
    STEP_REQUIRE(c, "comm_set_side_stream: not a communicator handle");
    STEP_REQUIRE(stream, "comm_set_side_stream: the null stream cannot be the side stream");
    STEP_REQUIRE(!c->pending, "comm_set_side_stream: a reduction is pending (join first)");
    if (c->side && c->own_side) {
        (void)hipStreamSynchronize(c->side);
        (void)hipStreamDestroy(c->side);
    }
    c->side = (hipStream_t)stream;
    c->own_side = false;
    return STEP_OK;
}

// in place, in stream order on `stream`: buf <- sum (average = 0) or mean (average = 1) over the ranks
extern "C" int step_comm_allreduce(void* comm, void* buf, long n, int dtype, int average, void* stream) {
    StepComm* c = as_comm(comm);
    STEP_REQUIRE(c, "comm_allreduce: not a communicator handle");
    STEP_REQUIRE(buf && n > 0, "comm_allreduce: null buffer or n = %ld", n);
    ncclDataType_t dt;
    STEP_TRY(dtype_of(dtype, &dt, "comm_allreduce"));
    STEP_REQUIRE(!(average && dtype == STEP_COMM_U8), "comm_allreduce: no mean of bytes");
    STEP_NCCL(g_rccl.AllReduce(buf, buf, (size_t)n, dt, average ? ncclAvg : ncclSum, c->comm, (hipStream_t)stream), "comm_allreduce");
    return STEP_OK;
}

// The gradient all-reduce of the step (mean over the ranks, in place), OVERLAPPED: it is ordered behind everything queued on `stream` so
// far, runs on the communicator's side stream, and `stream` goes on with the rest of the backward; step_grad_allreduce_join() makes
// `stream` wait for every reduction begun since the last join.  (C2 88.9 MB, C4 107.6 MB -- 98 % of it the fc weight
// gradient, finished early in the backward.)
extern "C" int step_grad_allreduce_begin(void* comm, float* flat_grad, long n, void* stream) {
    StepComm* c = as_comm(comm);
    STEP_REQUIRE(c, "grad_allreduce_begin: not a communicator handle");
    STEP_REQUIRE(flat_grad && n > 0, "grad_allreduce_begin: null buffer or n = %ld", n);
    STEP_HIPCK(hipEventRecord(c->ready, (hipStream_t)stream), "grad_allreduce_begin");
    STEP_HIPCK(hipStreamWaitEvent(c->side, c->ready, 0), "grad_allreduce_begin");
    STEP_NCCL(g_rccl.AllReduce(flat_grad, flat_grad, (size_t)n, ncclFloat32, ncclAvg, c->comm, c->side), "grad_allreduce_begin");
    c->pending = true;
    return STEP_OK;
}

extern "C" int step_grad_allreduce_join(void* comm, void* stream) {
    StepComm* c = as_comm(comm);
    STEP_REQUIRE(c, "grad_allreduce_join: not a communicator handle");
    if (!c->pending) return STEP_OK;
    STEP_HIPCK(hipEventRecord(c->done, c->side), "grad_allreduce_join");
    STEP_HIPCK(hipStreamWaitEvent((hipStream_t)stream, c->done, 0), "grad_allreduce_join");
    c->pending = false;
    return STEP_OK;
}

// the whole flat gradient buffer in one call on `stream` itself (no overlap; what SURVEY.md 8b sketches as grad_allreduce)
extern "C" int step_grad_allreduce(void* comm, float* flat_grad, long n, void* stream) {
    return step_comm_allreduce(comm, flat_grad, n, STEP_COMM_F32, 1, stream);
}

extern "C" int step_comm_broadcast(void* comm, void* buf, long n, int dtype, int root, void* stream) {
    StepComm* c = as_comm(comm);
    STEP_REQUIRE(c, "comm_broadcast: not a communicator handle");
    STEP_REQUIRE(buf && n > 0 && root >= 0 && root < c->nranks, "comm_broadcast: bad arguments (n = %ld, root = %d)", n, root);
    ncclDataType_t dt;
    STEP_TRY(dtype_of(dtype, &dt, "comm_broadcast"));
    STEP_NCCL(g_rccl.Broadcast(buf, buf, (size_t)n, dt, root, c->comm, (hipStream_t)stream), "comm_broadcast");
    return STEP_OK;
}
