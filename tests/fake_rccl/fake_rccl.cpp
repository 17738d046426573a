// TEST INFRASTRUCTURE -- a stand-in for librccl that moves data between PROCESSES THAT SHARE ONE GPU.
//
// RCCL refuses two ranks on the same device and a gpurun box has one GPU, so the N > 1 side of csrc/comm.cpp / step_amd/comm.py (unique id
// carried by the process group, ncclCommInitRank with nranks = 2, ncclAvg of the flat gradient on the side stream behind an event, the
// small f32 / f64 sums of the time-sliced graph learner in stream order, the broadcast) could only ever run with one rank.  This library
// exports the seven RCCL entry points comm.cpp resolves with dlsym and implements them through POSIX shared memory and host staging:
// stream-synchronise, device -> shared slot, barrier, every rank reduces all slots, barrier, host -> device.  It blocks the host (nothing
// here is a performance statement) but keeps RCCL's contract -- in place, in stream order, every rank gets the same result.  Loaded only
// when $STEP_RCCL_LIB names it (tests/test_gpu_comm_two_ranks.py); never shipped, never linked.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {
constexpr size_t CHUNK = 8u << 20;          // bytes staged per rank and round
constexpr size_t HEADER = 4096;
struct Shared {
    std::atomic<long> arrived;
};
}  // namespace

struct ncclComm {
    int nranks, rank;
    Shared* sh;
    char* slots;
    size_t map_bytes;
    long generation;
    char name[128];
    std::vector<char> tmp;
};

namespace {
bool barrier(ncclComm* c) {
    const long target = ++c->generation * c->nranks;
    c->sh->arrived.fetch_add(1, std::memory_order_acq_rel);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->sh->arrived.load(std::memory_order_acquire) < target) {
        std::this_thread::sleep_for(std::chrono::microseconds(20));
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;       // a peer died: fail, do not hang
    }
    return true;
}
size_t elem_size(ncclDataType_t t) {
    switch (t) {
        case ncclFloat32: return 4;
        case ncclFloat64: return 8;
        case ncclUint8: return 1;
        default: return 0;
    }
}
template <typename T>
void reduce(ncclComm* c, size_t n, bool avg) {
    T* out = (T*)c->tmp.data();
    for (size_t i = 0; i < n; ++i) {
        T s = ((const T*)c->slots)[i];
        for (int r = 1; r < c->nranks; ++r) s += ((const T*)(c->slots + (size_t)r * CHUNK))[i];
        out[i] = avg ? (T)(s / (T)c->nranks) : s;
    }
}
}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int* v) { *v = 29901; return ncclSuccess; }       // (no RCCL release carries this number)
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake RCCL: failure (peer lost, shared memory, or unsupported type)"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    static int counter = 0;
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/step_fake_rccl_%d_%ld_%d", (int)getpid(), (long)time(nullptr), counter++);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || rank < 0 || rank >= nranks || id.internal[0] != '/') return ncclInvalidArgument;
    ncclComm* c = new ncclComm();
    c->nranks = nranks; c->rank = rank; c->generation = 0;
    strncpy(c->name, id.internal, sizeof(c->name) - 1);
    c->map_bytes = HEADER + (size_t)nranks * CHUNK;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { delete c; return ncclSystemError; }
    void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->sh = (Shared*)p;                       // (a fresh object is zero-filled: the arrival counter starts at 0 for whoever comes first)
    c->slots = (char*)p + HEADER;
    c->tmp.resize(CHUNK);
    if (!barrier(c)) { munmap(p, c->map_bytes); delete c; return ncclSystemError; }
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclInvalidArgument;
    (void)barrier(c);
    munmap((void*)c->sh, c->map_bytes);
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, hipStream_t stream) {
    const size_t esz = elem_size(dt);
    if (!c || !esz || (op != ncclSum && op != ncclAvg) || (op == ncclAvg && dt == ncclUint8)) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    const size_t per = CHUNK / esz;
    for (size_t off = 0; off < count; off += per) {
        const size_t n = count - off < per ? count - off : per;
        if (hipMemcpy(c->slots + (size_t)c->rank * CHUNK, (const char*)send + off * esz, n * esz, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        if (!barrier(c)) return ncclSystemError;
        if (dt == ncclFloat32) reduce<float>(c, n, op == ncclAvg);
        else if (dt == ncclFloat64) reduce<double>(c, n, op == ncclAvg);
        else reduce<unsigned char>(c, n, false);
        if (!barrier(c)) return ncclSystemError;           // every rank has read every slot: they may be overwritten
        if (hipMemcpy((char*)recv + off * esz, c->tmp.data(), n * esz, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t dt, int root, ncclComm_t c, hipStream_t stream) {
    const size_t esz = elem_size(dt);
    if (!c || !esz || root < 0 || root >= c->nranks) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    const size_t per = CHUNK / esz;
    for (size_t off = 0; off < count; off += per) {
        const size_t n = count - off < per ? count - off : per;
        if (c->rank == root && hipMemcpy(c->slots, (const char*)send + off * esz, n * esz, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        if (!barrier(c)) return ncclSystemError;
        if (c->rank != root && hipMemcpy((char*)recv + off * esz, c->slots, n * esz, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        if (!barrier(c)) return ncclSystemError;
    }
    return ncclSuccess;
}

}  // extern "C"
