// In-process stand-in for the RCCL entry points the library uses (tests only, see shim/rccl/rccl.h): communicators of
// one process, point-to-point sends and receives that are matched when the outermost group ends. Everything is
// synchronous host memory, so "matched" means copied. A send without its receive (or the reverse), or a pair whose
// sizes differ, is reported as an error instead of hanging as it would on the real thing.
#include <rccl/rccl.h>

#include <memory>
#include <mutex>
#include <vector>

struct CommWorld { int size; };
struct emu_comm { int rank, size, device; std::shared_ptr<CommWorld> world; };

namespace {
struct Op { bool send; const void* src; void* dst; size_t bytes; int from, to; CommWorld* world; bool matched; };
std::mutex gLock;
std::vector<Op> gOps;
int gDepth = 0;

size_t TypeSize(ncclDataType_t t) {
    switch (t) { case ncclInt8: case ncclUint8: return 1; case ncclFloat16: return 2; case ncclInt32: case ncclUint32: case ncclFloat32: return 4; default: return 8; }
}

ncclResult_t Flush() {
    ncclResult_t result = ncclSuccess;
    for (Op& s : gOps) {
        if (!s.send || s.matched) continue;
        for (Op& r : gOps) {
            if (r.send || r.matched || r.world != s.world || r.from != s.from || r.to != s.to) continue;
            if (r.bytes != s.bytes) { result = ncclInvalidArgument; break; }
            if (s.bytes) memmove(r.dst, s.src, s.bytes);
            r.matched = s.matched = true;
            break;
        }
    }
    for (const Op& o : gOps) if (!o.matched && result == ncclSuccess) result = ncclInvalidUsage;
    gOps.clear();
    return result;
}

ncclResult_t Post(Op op) {
    std::lock_guard<std::mutex> hold(gLock);
    gOps.push_back(op);
    return gDepth > 0 ? ncclSuccess : Flush();
}
}  // namespace

extern "C" {

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    if (!comms || ndev < 1) return ncclInvalidArgument;
    int have = 0;
    (void)hipGetDeviceCount(&have);
    auto world = std::make_shared<CommWorld>();
    world->size = ndev;
    for (int i = 0; i < ndev; ++i) {
        const int dev = devlist ? devlist[i] : i;
        if (dev < 0 || dev >= have) return ncclInvalidArgument;
        comms[i] = new emu_comm{i, ndev, dev, world};
    }
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return ncclSuccess; }
ncclResult_t ncclGroupStart() { std::lock_guard<std::mutex> hold(gLock); ++gDepth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    std::lock_guard<std::mutex> hold(gLock);
    if (gDepth <= 0) return ncclInvalidUsage;
    return --gDepth == 0 ? Flush() : ncclSuccess;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t) {
    if (!comm || peer < 0 || peer >= comm->size || (count && !buf)) return ncclInvalidArgument;
    return Post(Op{true, buf, nullptr, count * TypeSize(type), comm->rank, peer, comm->world.get(), false});
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t) {
    if (!comm || peer < 0 || peer >= comm->size || (count && !buf)) return ncclInvalidArgument;
    return Post(Op{false, nullptr, buf, count * TypeSize(type), peer, comm->rank, comm->world.get(), false});
}
const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclInvalidArgument: return "invalid argument (emulated RCCL: sizes of a send and its receive differ?)";
        case ncclInvalidUsage: return "invalid usage (emulated RCCL: a send or receive without its counterpart)";
        default: return "error (emulated RCCL)";
    }
}

}  // extern "C"

static int gHandle;
void* emu_dlopen(const char*, int) { return &gHandle; }
void* emu_dlsym(void*, const char* symbol) {
    const struct { const char* name; void* fn; } table[] = {
        {"ncclCommInitAll", (void*)&ncclCommInitAll}, {"ncclCommDestroy", (void*)&ncclCommDestroy}, {"ncclGroupStart", (void*)&ncclGroupStart},
        {"ncclGroupEnd", (void*)&ncclGroupEnd}, {"ncclSend", (void*)&ncclSend}, {"ncclRecv", (void*)&ncclRecv}, {"ncclGetErrorString", (void*)&ncclGetErrorString}};
    for (const auto& e : table) if (strcmp(e.name, symbol) == 0) return e.fn;
    return nullptr;
}
