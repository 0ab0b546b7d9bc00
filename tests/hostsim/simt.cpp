// tests/hostsim/simt.cpp -- TEST INFRASTRUCTURE (see hip/hip_runtime.h): the fibre scheduler that executes a kernel launch
// on the host, and the synchronous stand-ins of the HIP runtime calls the engine makes.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <vector>

#if !defined(__x86_64__)
#error "tests/hostsim: the fibre switch is written for x86-64 (System V)"
#endif
// Fibre switch: callee-saved registers and the stack pointer, nothing else (glibc's swapcontext also saves the signal mask:
// one system call per switch, which was 90 % of the simulator's run time -- a lane switches twice per collective).
extern "C" void og_simt_switch(void** save_sp, void* load_sp);
asm(".text\n"
    ".p2align 4\n"
    ".globl og_simt_switch\n"
    ".hidden og_simt_switch\n"
    ".type og_simt_switch,@function\n"
    "og_simt_switch:\n"
    "    pushq %rbp\n"
    "    pushq %rbx\n"
    "    pushq %r12\n"
    "    pushq %r13\n"
    "    pushq %r14\n"
    "    pushq %r15\n"
    "    movq %rsp, (%rdi)\n"
    "    movq %rsi, %rsp\n"
    "    popq %r15\n"
    "    popq %r14\n"
    "    popq %r13\n"
    "    popq %r12\n"
    "    popq %rbx\n"
    "    popq %rbp\n"
    "    ret\n"
    ".size og_simt_switch, .-og_simt_switch\n");

namespace simt {
namespace {

constexpr unsigned WAVE = 64;
constexpr size_t STACK_BYTES = 256 * 1024;
constexpr unsigned MAX_LANES = 1024;

enum Op : int { OP_NONE = 0, OP_ALL, OP_ANY, OP_SHFL_XOR, OP_FIRST, OP_WAVE_SYNC, OP_BARRIER };
enum State : int { READY = 0, BLOCKED, DONE };

struct Lane {
    void* sp = nullptr; // saved stack pointer while the lane is switched out
    LaneCtx lc;
    State state = DONE;
    Op op = OP_NONE;
    uint32_t in = 0, arg = 0, out = 0;
};

struct Sched {
    std::vector<Lane> lanes;
    char* stacks = nullptr; // MAX_LANES x STACK_BYTES of address space, committed page by page as the fibres touch it
    void* main_sp = nullptr;
    Lane* cur = nullptr;
    const std::function<void()>* body = nullptr;
};

std::recursive_mutex g_launch_lock; // __shared__ variables are function statics: one workgroup at a time, process-wide
thread_local Sched* g_sched = nullptr;
thread_local LaneCtx g_host_lane = {{0, 0, 0}, {0, 0, 0}, {1, 1, 1}, {1, 1, 1}};

extern "C" void og_simt_lane_entry()
{
    Sched* s = g_sched;
    Lane* me = s->cur;
    (*s->body)();
    me->state = DONE;
    og_simt_switch(&me->sp, s->main_sp);
    abort(); // (a finished lane is never resumed)
}

uint32_t block_on(Op op, uint32_t in, uint32_t arg)
{
    Sched* s = g_sched;
    if (!s || !s->cur) { // host code calling a device helper outside a launch: a wave of one
        switch (op) {
        case OP_ALL:
        case OP_ANY: return in ? 1u : 0u;
        default: return in;
        }
    }
    Lane* me = s->cur;
    me->op = op;
    me->in = in;
    me->arg = arg;
    me->state = BLOCKED;
    og_simt_switch(&me->sp, s->main_sp);
    return me->out;
}

[[noreturn]] void deadlock(Sched& s, unsigned n)
{
    fprintf(stderr, "hostsim: deadlock in workgroup (%u,%u,%u): lanes wait at different collectives\n", s.lanes[0].lc.bid.x, s.lanes[0].lc.bid.y,
            s.lanes[0].lc.bid.z);
    for (unsigned w = 0; w < (n + WAVE - 1) / WAVE; ++w) {
        int cnt[8] = {0};
        int done = 0;
        for (unsigned l = w * WAVE; l < std::min(n, (w + 1) * WAVE); ++l) {
            if (s.lanes[l].state == DONE) ++done;
            else ++cnt[s.lanes[l].op];
        }
        fprintf(stderr, "  wave %u: done %d all %d any %d shfl %d first %d wave_sync %d barrier %d\n", w, done, cnt[OP_ALL], cnt[OP_ANY], cnt[OP_SHFL_XOR],
                cnt[OP_FIRST], cnt[OP_WAVE_SYNC], cnt[OP_BARRIER]);
    }
    abort();
}

void run_block(Sched& s, unsigned n)
{
    unsigned alive = n;
    for (unsigned l = 0; l < n; ++l) {
        Lane& L = s.lanes[l];
        // a fresh stack that og_simt_switch can "return" into: six zeroed callee-saved registers, then the entry point;
        // after the pops and the ret the stack pointer is 8 below a 16-byte boundary, as at any function entry
        uintptr_t top = (uintptr_t)(s.stacks + (size_t)(l + 1) * STACK_BYTES);
        top &= ~(uintptr_t)15;
        void** sp = (void**)(top - 8); // [sp] = return address of the entry function (never used)
        *sp = nullptr;
        *--sp = (void*)&og_simt_lane_entry;
        for (int r = 0; r < 6; ++r) *--sp = nullptr;
        L.sp = sp;
        L.state = READY;
        L.op = OP_NONE;
    }
    const unsigned n_waves = (n + WAVE - 1) / WAVE;
    // The order in which ready lanes run between rendezvous points is the simulator's choice; a kernel that is correct on
    // the hardware gives the same result under any of them.  OG_HOSTSIM_SCHEDULE=reverse runs the waves (and the lanes) last
    // first, =rotate starts one wave further on every pass: a hand-off that lacks a barrier shows up as a changed result.
    static const int sched = [] {
        const char* e = getenv("OG_HOSTSIM_SCHEDULE");
        return !e ? 0 : (!strcmp(e, "reverse") ? 1 : (!strcmp(e, "rotate") ? 2 : 0));
    }();
    unsigned pass = 0;
    while (alive) {
        bool progress = false;
        ++pass;
        for (unsigned k = 0; k < n; ++k) {
            const unsigned l = sched == 1 ? n - 1 - k : (sched == 2 ? (k + pass * WAVE) % n : k);
            Lane& L = s.lanes[l];
            if (L.state != READY) continue;
            s.cur = &L;
            og_simt_switch(&s.main_sp, L.sp);
            s.cur = nullptr;
            progress = true;
            if (L.state == DONE) --alive;
        }
        // wave-level collectives: every lane of the wave that has not returned must be at the same one
        for (unsigned w = 0; w < n_waves; ++w) {
            const unsigned lo = w * WAVE, hi = std::min(n, lo + WAVE);
            Op op = OP_NONE;
            bool ok = true, any_blocked = false;
            for (unsigned l = lo; l < hi && ok; ++l) {
                const Lane& L = s.lanes[l];
                if (L.state == DONE) continue;
                if (L.state != BLOCKED || L.op == OP_BARRIER) {
                    ok = false;
                    break;
                }
                if (op == OP_NONE) op = L.op;
                else if (op != L.op) ok = false;
                any_blocked = true;
            }
            if (!ok || !any_blocked) continue;
            uint32_t all = 1, any = 0, first = 0;
            bool have_first = false;
            for (unsigned l = lo; l < hi; ++l) {
                const Lane& L = s.lanes[l];
                if (L.state == DONE) continue;
                all &= L.in ? 1u : 0u;
                any |= L.in ? 1u : 0u;
                if (!have_first) {
                    first = L.in;
                    have_first = true;
                }
            }
            for (unsigned l = lo; l < hi; ++l) {
                Lane& L = s.lanes[l];
                if (L.state == DONE) continue;
                switch (op) {
                case OP_ALL: L.out = all; break;
                case OP_ANY: L.out = any; break;
                case OP_FIRST: L.out = first; break;
                case OP_SHFL_XOR: {
                    const unsigned src = lo + ((l - lo) ^ L.arg);
                    // (a lane outside the wave or one that has returned: the hardware hands back the caller's own value)
                    L.out = (src < hi && s.lanes[src].state != DONE) ? s.lanes[src].in : L.in;
                    break;
                }
                default: L.out = 0; break;
                }
            }
            for (unsigned l = lo; l < hi; ++l)
                if (s.lanes[l].state == BLOCKED) {
                    s.lanes[l].state = READY;
                    s.lanes[l].op = OP_NONE;
                }
            progress = true;
        }
        // the workgroup barrier
        {
            bool all_at_barrier = alive > 0;
            for (unsigned l = 0; l < n && all_at_barrier; ++l)
                if (s.lanes[l].state != DONE && !(s.lanes[l].state == BLOCKED && s.lanes[l].op == OP_BARRIER)) all_at_barrier = false;
            if (all_at_barrier) {
                for (unsigned l = 0; l < n; ++l)
                    if (s.lanes[l].state == BLOCKED) {
                        s.lanes[l].state = READY;
                        s.lanes[l].op = OP_NONE;
                    }
                progress = true;
            }
        }
        if (!progress) deadlock(s, n);
    }
}

void do_launch(dim3 grid, dim3 block, const std::function<void()>& body)
{
    std::lock_guard<std::recursive_mutex> lk(g_launch_lock);
    const unsigned n = block.x * block.y * block.z;
    if (!n || !grid.x || !grid.y || !grid.z) return;
    if (n > MAX_LANES) {
        fprintf(stderr, "hostsim: workgroup of %u lanes\n", n);
        abort();
    }
    // one scheduler (lane table + stack space) per host thread and nesting level, kept between launches
    thread_local std::vector<Sched*> pool;
    thread_local unsigned depth = 0;
    if (pool.size() <= depth) pool.push_back(nullptr);
    if (!pool[depth]) {
        pool[depth] = new Sched;
        void* m = mmap(nullptr, (size_t)MAX_LANES * STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (m == MAP_FAILED) {
            perror("hostsim: mmap of the fibre stacks");
            abort();
        }
        pool[depth]->stacks = (char*)m;
    }
    Sched& s = *pool[depth];
    if (s.lanes.size() < n) s.lanes.resize(n);
    s.body = &body;
    Sched* outer = g_sched;
    g_sched = &s;
    ++depth;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                for (unsigned l = 0; l < n; ++l) {
                    LaneCtx& c = s.lanes[l].lc;
                    c.tid = {l % block.x, (l / block.x) % block.y, l / (block.x * block.y)};
                    c.bid = {bx, by, bz};
                    c.bdim = {block.x, block.y, block.z};
                    c.gdim = {grid.x, grid.y, grid.z};
                }
                run_block(s, n);
            }
    --depth;
    g_sched = outer;
}

LaneCtx* api_cur() { return (g_sched && g_sched->cur) ? &g_sched->cur->lc : &g_host_lane; }
int api_all(int p) { return (int)block_on(OP_ALL, p ? 1u : 0u, 0); }
int api_any(int p) { return (int)block_on(OP_ANY, p ? 1u : 0u, 0); }
uint32_t api_shfl_xor(uint32_t v, int m) { return block_on(OP_SHFL_XOR, v, (uint32_t)m); }
uint32_t api_first(uint32_t v) { return block_on(OP_FIRST, v, 0); }
void api_wave_sync() { (void)block_on(OP_WAVE_SYNC, 0, 0); }
void api_syncthreads() { (void)block_on(OP_BARRIER, 0, 0); }
// a polling lane: stays READY, runs again on the scheduler's next pass (after every other ready lane has had its turn).  A poll
// whose condition can never come true spins for ever -- the tests run under a timeout -- instead of being reported as a deadlock.
void api_yield()
{
    Sched* s = g_sched;
    if (!s || !s->cur) return;
    Lane* me = s->cur;
    og_simt_switch(&me->sp, s->main_sp);
}

Api g_api = {api_cur, api_all, api_any, api_shfl_xor, api_first, api_wave_sync, api_syncthreads, do_launch, api_yield};

// x86 traps on an integer division by zero; the GPU does not (the quotient is garbage, typically in a lane whose result is
// never used).  Say where it happened -- object + offset, for llvm-symbolizer on a unit compiled with OG_HOSTSIM_DEBUG=1 --
// and which lane it was, then die: the simulator does not guess what the hardware would have produced.
void on_sigfpe(int, siginfo_t* si, void*)
{
    Dl_info info;
    char buf[512];
    int n;
    if (dladdr(si->si_addr, &info) && info.dli_fname)
        n = snprintf(buf, sizeof buf, "hostsim: SIGFPE (integer division by zero?) at %s+0x%zx (%s)\n", info.dli_fname,
                     (size_t)((char*)si->si_addr - (char*)info.dli_fbase), info.dli_sname ? info.dli_sname : "?");
    else
        n = snprintf(buf, sizeof buf, "hostsim: SIGFPE at %p\n", si->si_addr);
    if (n > 0) (void)!write(2, buf, (size_t)n);
    if (g_sched && g_sched->cur) {
        const LaneCtx& c = g_sched->cur->lc;
        n = snprintf(buf, sizeof buf, "hostsim:   lane %u of workgroup %u (block of %u lanes, grid of %u)\n", c.tid.x, c.bid.x, c.bdim.x, c.gdim.x);
        if (n > 0) (void)!write(2, buf, (size_t)n);
    }
    _exit(134);
}

struct Init {
    Init()
    {
        api_slot() = &g_api;
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_sigaction = on_sigfpe;
        sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
        sigaction(SIGFPE, &sa, nullptr);
    }
} g_init;

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

} // namespace

Api* default_api() { return &g_api; }

} // namespace simt

// ---- runtime API ----------------------------------------------------------------------------------------------------
namespace {
thread_local int t_device = 0;
int n_devices()
{
    const char* e = getenv("OG_HOSTSIM_DEVICES"); // (tests of the cluster path: several simulated devices)
    const int n = e ? atoi(e) : 1;
    return n < 1 ? 1 : (n > 8 ? 8 : n);
}
} // namespace

hipError_t hipGetDeviceCount(int* n)
{
    *n = n_devices();
    return hipSuccess;
}
hipError_t hipSetDevice(int d)
{
    if (d < 0 || d >= n_devices()) return hipErrorInvalidDevice;
    t_device = d;
    return hipSuccess;
}
hipError_t hipGetDevice(int* d)
{
    *d = t_device;
    return hipSuccess;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d)
{
    if (d < 0 || d >= n_devices()) return hipErrorInvalidDevice;
    memset(p, 0, sizeof *p);
    snprintf(p->name, sizeof p->name, "hostsim (x86 fibres, test infrastructure)");
    snprintf(p->gcnArchName, sizeof p->gcnArchName, "hostsim");
    p->totalGlobalMem = (size_t)8 << 30;
    p->multiProcessorCount = 256;
    p->clockRate = 2400000;
    p->warpSize = 64;
    p->sharedMemPerBlock = 160 * 1024;
    p->maxThreadsPerBlock = 1024;
    return hipSuccess;
}
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e)
{
    switch (e) {
    case hipSuccess: return "no error";
    case hipErrorInvalidValue: return "invalid value";
    case hipErrorOutOfMemory: return "out of memory";
    case hipErrorInvalidDevice: return "invalid device";
    case hipErrorNotReady: return "not ready";
    case hipErrorNotSupported: return "not supported by the host simulator";
    default: return "hostsim error";
    }
}
hipError_t hipMalloc(void** p, size_t n)
{
    *p = n ? aligned_alloc(256, (n + 255) / 256 * 256) : nullptr;
    if (n && !*p) return hipErrorOutOfMemory;
    if (*p) memset(*p, 0xA5, n); // (device memory is not zeroed: make reads of uninitialised memory visible)
    return hipSuccess;
}
hipError_t hipFree(void* p)
{
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned)
{
    *p = n ? aligned_alloc(256, (n + 255) / 256 * 256) : nullptr;
    return (n && !*p) ? hipErrorOutOfMemory : hipSuccess;
}
hipError_t hipHostFree(void* p)
{
    free(p);
    return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind)
{
    if (n) memmove(dst, src, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t)
{
    if (n) memmove(dst, src, n);
    return hipSuccess;
}
hipError_t hipMemset(void* dst, int v, size_t n)
{
    if (n) memset(dst, v, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t)
{
    if (n) memset(dst, v, n);
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags)
{
    *s = new ihipStream_t{t_device, flags};
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s)
{
    delete s;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e)
{
    *e = new ihipEvent_t{0.0, false};
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e)
{
    delete e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t)
{
    if (!e) return hipErrorInvalidValue;
    e->t_ms = simt::now_ms();
    e->recorded = true;
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b)
{
    if (!a || !b || !a->recorded || !b->recorded) return hipErrorInvalidValue;
    *ms = (float)(b->t_ms - a->t_ms);
    return hipSuccess;
}
hipError_t hipModuleLoadData(hipModule_t*, const void*) { return hipErrorNotSupported; }
hipError_t hipModuleUnload(hipModule_t) { return hipSuccess; }
hipError_t hipModuleGetFunction(hipFunction_t*, hipModule_t, const char*) { return hipErrorNotSupported; }
hipError_t hipModuleLaunchKernel(hipFunction_t, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, hipStream_t, void**, void**)
{
    return hipErrorNotSupported;
}
