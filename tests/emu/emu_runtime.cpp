// Host execution of HIP kernels for the tests (see shim/hip/hip_runtime.h).
//
// A launch runs its work-groups on a few host threads, one work-group at a time per thread. Inside a work-group
// every lane is a fiber (ucontext); a lane runs until it reaches a wave- or block-wide operation, where it parks
// until all lanes of its wave (or work-group) that are still running have reached the same operation. Lanes that
// reach DIFFERENT operations, or a wave that can never complete one, abort the process with a message that names
// the lanes: on the GPU the same mistake is a hang or garbage.
#include <hip/hip_runtime.h>

#include <sys/mman.h>
#include <ucontext.h>
#include <xmmintrin.h>

#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace {

constexpr size_t kFiberStack = 256 * 1024;

struct WaveState {
    uint32_t alive = 0, arrived = 0;
    uint64_t generation = 0;
    int kind = 0; unsigned offset = 0;
    uint64_t pendingMask = 0;
    uint64_t mask[2] = {0, 0};
    uint64_t value[2][64];
};

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
    int waitsOn = 0;              // 0 runnable, 1 wave, 2 work-group
    uint64_t waitGeneration = 0;
};

struct Group {
    std::vector<Fiber> fibers;
    std::vector<WaveState> waves;
    ucontext_t scheduler;
    unsigned current = 0, threads = 0;
    dim3 block;
    const std::function<void()>* body = nullptr;
    // work-group barrier
    uint32_t alive = 0, arrived = 0; uint64_t generation = 0;
    std::vector<char*> stacks;    // kept between work-groups of one host thread
};

thread_local Group* tlsGroup = nullptr;

[[noreturn]] void Die(const char* what, const Group& g) {
    fprintf(stderr, "[hip emulation] %s (work-group %u,%u,%u; lane %u of wave %u)\n", what, blockIdx.x, blockIdx.y, blockIdx.z, g.current & 63u, g.current >> 6);
    for (size_t w = 0; w < g.waves.size(); ++w) {
        const WaveState& s = g.waves[w];
        fprintf(stderr, "  wave %zu: %u running, %u arrived at operation %d, arrived mask %016llx\n", w, s.alive, s.arrived, s.kind, (unsigned long long)s.pendingMask);
    }
    abort();
}

void SetThreadIndex(const Group& g, unsigned t) {
    threadIdx.x = t % g.block.x;
    threadIdx.y = (t / g.block.x) % g.block.y;
    threadIdx.z = t / (g.block.x * g.block.y);
}

void FiberMain() {
    Group* g = tlsGroup;
    (*g->body)();
    g = tlsGroup;
    g->fibers[g->current].done = true;
    // returning resumes uc_link: the scheduler
}

void Park(Group& g, int what, uint64_t generation) {
    Fiber& f = g.fibers[g.current];
    f.waitsOn = what; f.waitGeneration = generation;
    swapcontext(&f.ctx, &g.scheduler);
}

void CompleteWave(WaveState& w) {
    w.mask[w.generation & 1u] = w.pendingMask;
    w.pendingMask = 0; w.arrived = 0; w.kind = 0;
    ++w.generation;
}

void RunGroup(Group& g, const std::function<void()>& body, dim3 block) {
    const unsigned n = block.x * block.y * block.z;
    g.block = block; g.threads = n; g.body = &body;
    g.fibers.assign(n, Fiber());
    g.waves.assign((n + 63u) / 64u, WaveState());
    g.alive = n; g.arrived = 0; g.generation = 0;
    while (g.stacks.size() < n) {
        void* p = mmap(nullptr, kFiberStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
        if (p == MAP_FAILED) { perror("[hip emulation] mmap"); abort(); }
        g.stacks.push_back((char*)p);
    }
    for (unsigned t = 0; t < n; ++t) {
        Fiber& f = g.fibers[t];
        f.stack = g.stacks[t];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = kFiberStack; f.ctx.uc_link = &g.scheduler;
        makecontext(&f.ctx, FiberMain, 0);
        g.waves[t >> 6].alive++;
    }
    tlsGroup = &g;
    unsigned left = n;
    while (left) {
        bool progress = false;
        for (unsigned t = 0; t < n; ++t) {
            Fiber& f = g.fibers[t];
            if (f.done) continue;
            if (f.waitsOn == 1 && g.waves[t >> 6].generation == f.waitGeneration) continue;
            if (f.waitsOn == 2 && g.generation == f.waitGeneration) continue;
            f.waitsOn = 0;
            g.current = t;
            SetThreadIndex(g, t);
            swapcontext(&g.scheduler, &f.ctx);
            progress = true;
            if (f.done) {
                --left;
                // a lane that has returned takes no part in what its wave or work-group still does
                WaveState& w = g.waves[t >> 6];
                --w.alive;
                if (w.arrived && w.arrived == w.alive) CompleteWave(w);
                --g.alive;
                if (g.arrived && g.arrived == g.alive) { g.arrived = 0; ++g.generation; }
            }
        }
        if (!progress) Die("deadlock: the lanes still running wait for operations that cannot complete", g);
    }
    tlsGroup = nullptr;
}

unsigned EnvNumber(const char* name, unsigned fallback) {
    const char* v = getenv(name);
    if (!v || !*v) return fallback;
    const long n = strtol(v, nullptr, 10);
    return n > 0 ? (unsigned)n : fallback;
}

}  // namespace

namespace emu {

uint64_t WaveExchange(int kind, uint64_t value, unsigned offset) {
    Group& g = *tlsGroup;
    const unsigned t = g.current, lane = t & 63u;
    WaveState& w = g.waves[t >> 6];
    const uint64_t generation = w.generation;
    const unsigned slot = (unsigned)(generation & 1u);
    if (w.arrived == 0) { w.kind = kind; w.offset = offset; }
    else if (w.kind != kind || w.offset != offset) Die("the lanes of a wave reached different wave-wide operations", g);
    w.value[slot][lane] = value;
    w.pendingMask |= 1ull << lane;
    if (++w.arrived == w.alive) CompleteWave(w);
    else Park(g, 1, generation);
    // (tlsGroup is re-read: the compiler must not assume anything across the switch)
    WaveState& r = tlsGroup->waves[t >> 6];
    const uint64_t mask = r.mask[slot];
    const uint64_t* v = r.value[slot];
    switch (kind) {
        case kBallot: {
            uint64_t out = 0;
            for (unsigned l = 0; l < 64; ++l) if ((mask >> l & 1u) && v[l]) out |= 1ull << l;
            return out;
        }
        case kFirstLane: return v[__builtin_ctzll(mask)];
        case kShuffleDown: {
            const unsigned from = lane + offset;
            return (from < 64u && (mask >> from & 1u)) ? v[from] : v[lane];
        }
    }
    Die("unknown wave-wide operation", g);
}

void BlockBarrier() {
    Group& g = *tlsGroup;
    const uint64_t generation = g.generation;
    if (++g.arrived == g.alive) { g.arrived = 0; ++g.generation; }
    else Park(g, 2, generation);
}

void Launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const uint64_t groups = (uint64_t)grid.x * grid.y * grid.z;
    if (groups == 0 || block.x * block.y * block.z == 0) return;
    const unsigned workers = (unsigned)std::min<uint64_t>(groups, EnvNumber("RT_EMU_THREADS", 4));
    std::atomic<uint64_t> next{0};
    auto work = [&]() {
        // denormals are flushed on the device (inputs and results)
        _mm_setcsr(_mm_getcsr() | 0x8040u);
        Group g;
        gridDim = grid; blockDim = block;
        for (;;) {
            const uint64_t i = next.fetch_add(1);
            if (i >= groups) break;
            blockIdx.x = (unsigned)(i % grid.x); blockIdx.y = (unsigned)(i / grid.x % grid.y); blockIdx.z = (unsigned)(i / ((uint64_t)grid.x * grid.y));
            RunGroup(g, body, block);
        }
        for (char* s : g.stacks) munmap(s, kFiberStack);
    };
    // every work-group runs on a thread of its own making, never on the caller's: the caller's floating-point mode
    // and thread-local state stay untouched
    std::vector<std::thread> pool;
    for (unsigned i = 0; i < workers; ++i) pool.emplace_back(work);
    for (auto& t : pool) t.join();
}

}  // namespace emu

// ---- runtime API ----
struct emu_stream { int device; };
struct emu_event { std::chrono::steady_clock::time_point at; bool recorded; };

namespace {
thread_local int tlsDevice = 0;
int DeviceCount() { return (int)EnvNumber("RT_EMU_DEVICES", 1); }
}  // namespace

const char* hipGetErrorString(hipError_t e) {
    switch (e) {
        case hipSuccess: return "no error";
        case hipErrorInvalidValue: return "invalid argument";
        case hipErrorOutOfMemory: return "out of memory";
        case hipErrorInvalidDevice: return "invalid device ordinal";
        default: return "error (emulated runtime)";
    }
}
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = DeviceCount(); return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= DeviceCount()) return hipErrorInvalidDevice; tlsDevice = d; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = tlsDevice; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* value, hipDeviceAttribute_t, int) { *value = (int)EnvNumber("RT_EMU_CUS", 2); return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
// ---- the table of "device" allocations: lets the assembly interpreter (tests/emu/isa) tell a valid access from a stray one ----
namespace {
std::mutex gRangesLock;
std::map<uintptr_t, size_t> gRanges;
bool Known(uintptr_t a, size_t bytes) {
    auto it = gRanges.upper_bound(a);
    if (it == gRanges.begin()) return false;
    --it;
    return a >= it->first && a + bytes <= it->first + it->second;
}
}  // namespace
hipError_t emu_malloc(void** p, size_t bytes) {
    *p = nullptr;
    if (bytes == 0) bytes = 1;
    void* q = nullptr;
    if (posix_memalign(&q, 256, bytes) != 0) return hipErrorOutOfMemory;
    memset(q, 0xA5, bytes);       // device memory arrives uninitialised: make reliance on zeros visible
    { std::lock_guard<std::mutex> g(gRangesLock); gRanges[(uintptr_t)q] = bytes; }
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void* p) { if (p) { std::lock_guard<std::mutex> g(gRangesLock); gRanges.erase((uintptr_t)p); } free(p); return hipSuccess; }
hipError_t hipHostFree(void* p) { return hipFree(p); }

// ---- for tests/emu/isa: the launch hook and lane-wide memory access with every address checked against the table ----
namespace { int (*gLaunchHook)(const char*, const unsigned*, const unsigned char*, size_t) = nullptr; }
namespace emu {
bool OfferLaunch(const char* kernel, dim3 grid, dim3 block, const unsigned char* args, size_t bytes) {
    if (!gLaunchHook) return false;
    const unsigned dims[6] = {grid.x, grid.y, grid.z, block.x, block.y, block.z};
    return gLaunchHook(kernel, dims, args, bytes) != 0;
}
}  // namespace emu
extern "C" {
void emu_set_launch_hook(int (*hook)(const char*, const unsigned*, const unsigned char*, size_t)) { gLaunchHook = hook; }
void emu_register_range(const void* p, size_t bytes) { std::lock_guard<std::mutex> g(gRangesLock); gRanges[(uintptr_t)p] = bytes; }
void emu_forget_range(const void* p) { std::lock_guard<std::mutex> g(gRangesLock); gRanges.erase((uintptr_t)p); }
// out[d * 64 + lane] = dword d at addr[lane], for the lanes in mask; returns the lanes whose address is no allocation's
uint64_t emu_gather(const uint64_t* addr, uint64_t mask, int ndw, uint32_t* out) {
    uint64_t bad = 0;
    std::lock_guard<std::mutex> g(gRangesLock);
    for (int l = 0; l < 64; ++l) {
        if (!(mask >> l & 1u)) continue;
        if ((addr[l] & 3u) || !Known((uintptr_t)addr[l], (size_t)ndw * 4)) { bad |= 1ull << l; continue; }
        for (int d = 0; d < ndw; ++d) out[d * 64 + l] = ((const uint32_t*)addr[l])[d];
    }
    return bad;
}
uint64_t emu_scatter(const uint64_t* addr, uint64_t mask, int ndw, const uint32_t* in) {
    uint64_t bad = 0;
    std::lock_guard<std::mutex> g(gRangesLock);
    for (int l = 0; l < 64; ++l) {
        if (!(mask >> l & 1u)) continue;
        if ((addr[l] & 3u) || !Known((uintptr_t)addr[l], (size_t)ndw * 4)) { bad |= 1ull << l; continue; }
        for (int d = 0; d < ndw; ++d) ((uint32_t*)addr[l])[d] = in[d * 64 + l];
    }
    return bad;
}
uint64_t emu_atomic_add_u32(const uint64_t* addr, uint64_t mask, const uint32_t* data, uint32_t* before) {
    uint64_t bad = 0;
    std::lock_guard<std::mutex> g(gRangesLock);
    for (int l = 0; l < 64; ++l) {
        if (!(mask >> l & 1u)) continue;
        if ((addr[l] & 3u) || !Known((uintptr_t)addr[l], 4)) { bad |= 1ull << l; continue; }
        before[l] = __atomic_fetch_add((uint32_t*)addr[l], data[l], __ATOMIC_RELAXED);
    }
    return bad;
}
int emu_known(const void* p, size_t bytes) { std::lock_guard<std::mutex> g(gRangesLock); return Known((uintptr_t)p, bytes) ? 1 : 0; }
// out = fma(a, b, c) in single precision, one rounding
void emu_vfma(const float* a, const float* b, const float* c, float* out, int n) { for (int i = 0; i < n; ++i) out[i] = fmaf(a[i], b[i], c[i]); }
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) { if (bytes) memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t) { if (bytes) memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemset(void* dst, int value, size_t bytes) { if (bytes) memset(dst, value, bytes); return hipSuccess; }
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t) { if (bytes) memset(dst, value, bytes); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { *s = new emu_stream{tlsDevice}; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event{{}, false}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->at = std::chrono::steady_clock::now(); e->recorded = true; return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    if (!a->recorded || !b->recorded) return hipErrorInvalidValue;
    *ms = std::chrono::duration<float, std::milli>(b->at - a->at).count();
    return hipSuccess;
}
