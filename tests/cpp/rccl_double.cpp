// TEST DOUBLE for librccl (tests only; never shipped, never loaded unless SKH_RCCL_LIB names it).
//
// skh_gather_tiles' N > 1 branch -- ncclGroupStart / N-1 ncclRecv on the root / one ncclSend per other rank / ncclGroupEnd, chunk offsets,
// zero padding of the short ranks -- needs N ranks with a communicator.  A 1-GPU box cannot form one with the real RCCL (two ranks, one
// device: ncclCommInitRank refuses), so the branch had never run before the driver's 8-GPU job.  This library implements the ten symbols
// the loader in strelka_hip.hip binds for N PROCESSES SHARING ONE GPU: the unique id names a POSIX shared-memory control block; a send
// stages its buffer in a shared-memory segment (after the stream has drained), a receive copies it from there onto the stream; grouped
// calls are queued and executed at ncclGroupEnd, as the real library does.  Same call sequence, same buffers, same sizes as on 8 GPUs --
// only the transport differs.
//
//   g++ -std=c++17 -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o librccl_double.so rccl_double.cpp -L/opt/rocm/lib -lamdhip64 -lrt
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct
{
    char internal[128];
} ncclUniqueId;
typedef enum
{
    ncclSuccess = 0,
    ncclSystemError = 2,
    ncclInvalidArgument = 4
} ncclResult_t;
typedef int ncclDataType_t; // 7 = ncclFloat32
}

namespace
{
constexpr int MAX_RANKS = 64;
struct Slot
{
    std::atomic<uint64_t> ready; // sequence number of the last message this rank has staged ...
    std::atomic<uint64_t> done; // ... and of the last one its receiver has consumed
    uint64_t bytes;
};
struct Control
{
    std::atomic<int> arrived;
    std::atomic<int> left;
    int world;
    Slot slot[MAX_RANKS];
};
struct Op
{
    bool send;
    void* buf;
    size_t bytes;
    int peer;
    hipStream_t stream;
};
} // namespace
struct ncclComm
{
    std::string name;
    Control* ctl = nullptr;
    int rank = 0, world = 1;
    uint64_t sent = 0; // messages this rank has sent
    std::vector<uint64_t> received; // per peer
};
namespace
{
thread_local int g_group = 0;
thread_local std::vector<std::pair<ncclComm*, Op>> g_ops;

size_t dtype_size(ncclDataType_t t)
{
    return t == 7 ? 4 : (t == 8 ? 8 : (t <= 1 ? 1 : 4));
}
bool wait_for(const std::atomic<uint64_t>& a, uint64_t v)
{
    const auto t0 = std::chrono::steady_clock::now();
    while (a.load(std::memory_order_acquire) < v)
    {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
            return false;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return true;
}
std::string seg_name(const ncclComm* c, int rank, uint64_t seq)
{
    return c->name + "_r" + std::to_string(rank) + "_" + std::to_string(seq);
}
ncclResult_t run(ncclComm* c, const Op& op)
{
    if (op.peer < 0 || op.peer >= c->world || op.peer == c->rank)
        return ncclInvalidArgument;
    if (op.send)
    {
        if (hipStreamSynchronize(op.stream) != hipSuccess) // what the stream wrote into the buffer is there now
            return ncclSystemError;
        const uint64_t seq = ++c->sent;
        const std::string nm = seg_name(c, c->rank, seq);
        const int fd = shm_open(nm.c_str(), O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)op.bytes) != 0)
            return ncclSystemError;
        void* p = mmap(nullptr, op.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED)
            return ncclSystemError;
        const hipError_t e = hipMemcpy(p, op.buf, op.bytes, hipMemcpyDeviceToHost);
        munmap(p, op.bytes);
        Slot& s = c->ctl->slot[c->rank];
        s.bytes = op.bytes;
        s.ready.store(seq, std::memory_order_release);
        const bool ok = e == hipSuccess && wait_for(s.done, seq); // a send completes when its receiver has the data
        shm_unlink(nm.c_str());
        return ok ? ncclSuccess : ncclSystemError;
    }
    const uint64_t seq = ++c->received[op.peer];
    Slot& s = c->ctl->slot[op.peer];
    if (!wait_for(s.ready, seq))
        return ncclSystemError;
    if (s.bytes != op.bytes) // (send and receive must agree on the size, as with the real library)
        return ncclInvalidArgument;
    const std::string nm = seg_name(c, op.peer, seq);
    const int fd = shm_open(nm.c_str(), O_RDONLY, 0600);
    if (fd < 0)
        return ncclSystemError;
    void* p = mmap(nullptr, op.bytes, PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED)
        return ncclSystemError;
    hipError_t e = hipMemcpyAsync(op.buf, p, op.bytes, hipMemcpyHostToDevice, op.stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(op.stream);
    munmap(p, op.bytes);
    s.done.store(seq, std::memory_order_release);
    return e == hipSuccess ? ncclSuccess : ncclSystemError;
}
ncclResult_t submit(ncclComm* c, const Op& op)
{
    if (g_group > 0)
    {
        g_ops.emplace_back(c, op);
        return ncclSuccess;
    }
    return run(c, op);
}
} // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    static std::atomic<int> counter{ 0 };
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/skh_rccl_double_%d_%d", (int)getpid(), counter.fetch_add(1));
    const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)sizeof(Control)) != 0)
        return ncclSystemError;
    close(fd); // (zero-filled by ftruncate: arrived = 0, every sequence number 0)
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks || id.internal[0] != '/')
        return ncclInvalidArgument;
    const int fd = shm_open(id.internal, O_RDWR, 0600);
    if (fd < 0)
        return ncclSystemError;
    void* p = mmap(nullptr, sizeof(Control), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED)
        return ncclSystemError;
    ncclComm* c = new ncclComm;
    c->name = id.internal;
    c->ctl = static_cast<Control*>(p);
    c->rank = rank, c->world = nranks;
    c->received.assign((size_t)nranks, 0);
    c->ctl->world = nranks;
    c->ctl->arrived.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->ctl->arrived.load() < nranks) // ncclCommInitRank is collective
    {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
        {
            munmap(p, sizeof(Control));
            delete c;
            return ncclSystemError;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    *comm = c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c)
        return ncclInvalidArgument;
    if (c->ctl->left.fetch_add(1) + 1 == c->world)
        shm_unlink(c->name.c_str()); // the last one out removes the control block
    munmap(c->ctl, sizeof(Control));
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st)
{
    return c ? submit(c, Op{ true, const_cast<void*>(buf), count * dtype_size(t), peer, st }) : ncclInvalidArgument;
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st)
{
    return c ? submit(c, Op{ false, buf, count * dtype_size(t), peer, st }) : ncclInvalidArgument;
}
ncclResult_t ncclGroupStart()
{
    ++g_group;
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd()
{
    if (g_group <= 0)
        return ncclInvalidArgument;
    if (--g_group > 0)
        return ncclSuccess;
    // a rank of the gather either sends (one message to the root) or receives (N - 1 of them): the queued calls run in posting order.  A
    // group that mixes the two could deadlock on this double's blocking sends and is refused -- the gather never posts one.
    bool anySend = false, anyRecv = false;
    for (auto& co : g_ops)
        (co.second.send ? anySend : anyRecv) = true;
    ncclResult_t r = (anySend && anyRecv) ? ncclInvalidArgument : ncclSuccess;
    for (auto& co : g_ops)
        if (r == ncclSuccess)
            r = run(co.first, co.second);
    g_ops.clear();
    return r;
}
const char* ncclGetErrorString(ncclResult_t r)
{
    return r == ncclSuccess ? "no error" : (r == ncclInvalidArgument ? "invalid argument (rccl test double)" : "system error (rccl test double)");
}
ncclResult_t ncclCommCount(const ncclComm_t c, int* n)
{
    if (!c || !n)
        return ncclInvalidArgument;
    *n = c->world;
    return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r)
{
    if (!c || !r)
        return ncclInvalidArgument;
    *r = c->rank;
    return ncclSuccess;
}
}
