// tests/emu/dev/emu_rt.cpp — TEST INFRASTRUCTURE: the scheduler of the emulated device (tests/emu/dev/hip/hip_runtime.h).
//
// One workgroup at a time; each work-item is a fiber with its own stack; a fiber runs until it finishes or blocks in __syncthreads / a
// wavefront operation.  When every live lane of a wavefront is blocked, the lanes waiting at the same call site are served together (the
// group is the operation's EXEC mask); when lanes of one wavefront wait at DIFFERENT call sites the wavefront has diverged around a
// wavefront operation: the group at the lowest code address goes first (right for if-without-else and for loops with a divergent trip
// count, which is all the kernels of this tree do) and the event is counted (EMU_STRICT=1: abort instead).  __syncthreads releases when
// every live work-item of the workgroup waits in it.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <dlfcn.h>
#include <unistd.h>
#include <cxxabi.h>
#include <map>
#include <mutex>
#include <string>
#include <cstdio>
#include <vector>

// EMU_TSAN (tests/emu/dev/build.sh with EMU_TSAN=1): the kernels' translation unit is compiled with -fsanitize=thread and every work-item is a
// ThreadSanitizer fiber.  Happens-before edges are exactly the device's: a wavefront operation orders the lanes that meet in it, __syncthreads
// orders the workgroup, a launch is ordered after the previous launch.  Nothing else is: two work-items that touch one address without such
// an edge between them (and without atomics) are reported as a data race — LDS exchanged on the strength of lockstep, two workgroups of a
// launch writing one word, a kernel reading what its own launch writes elsewhere.  This file itself is NOT instrumented.
#ifdef EMU_TSAN
extern "C" {
void* __tsan_get_current_fiber(void);
void* __tsan_create_fiber(unsigned flags);
void __tsan_switch_to_fiber(void* fiber, unsigned flags);
void __tsan_acquire(void* addr);
void __tsan_release(void* addr);
void AnnotateBenignRaceSized(const char* file, int line, const volatile void* mem, long size, const char* desc);
void AnnotateIgnoreReadsBegin(const char* file, int line);
void AnnotateIgnoreReadsEnd(const char* file, int line);
void AnnotateIgnoreWritesBegin(const char* file, int line);
void AnnotateIgnoreWritesEnd(const char* file, int line);
}
extern char __start_emu_lds[], __stop_emu_lds[];
#define TSAN(...) __VA_ARGS__
#else
#define TSAN(...)
#endif

// EMU_ASAN (build.sh with EMU_ASAN=1): the kernels' translation unit under -fsanitize=address — out-of-bounds and use-after-free accesses of
// device memory (every hipMalloc is its own heap block of the exact size), of LDS arrays and of kernel-local arrays.  The fiber switches are
// announced to the runtime (it tracks the bounds of the current stack).
#ifdef EMU_ASAN
extern "C" {
void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
}
#define ASAN(...) __VA_ARGS__
#else
#define ASAN(...)
#endif

namespace emu {

enum State : int { READY, WAIT_WAVE, WAIT_BLOCK, DONE };

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    int state = DONE;
    unsigned lane = 0, wave = 0;
    dim3 tidx;
    // pending wavefront operation
    const void* site = nullptr;
    int op = 0;
    uint64_t in = 0, aux = 0, aux2 = 0, out = 0;
    void* tsan = nullptr;        // the detector's identity of this work-item
    bool ignoring = false;
    void* asan_fake = nullptr;
};

thread_local Fiber* cur = nullptr;
thread_local dim3 t_idx, b_idx, b_dim, g_dim;
thread_local void* dyn_lds = nullptr;
thread_local unsigned lane_in_wave = 0;

static thread_local void* sched_sp = nullptr;
static thread_local const Body* body = nullptr;
static thread_local unsigned long long n_divergent_sites = 0;
static std::map<std::pair<const void*, const void*>, unsigned long long> divergent_pairs;   // (served first, left waiting) -> count
static unsigned long long n_launches = 0, n_blocks = 0, n_wave_ops = 0;   // EMU_STATS=1: printed at exit

constexpr size_t STACK_BYTES = 1u << 20;   // per work-item (the macro-op backends keep a few KB of locals; the host compile does not optimise them away)

extern "C" void emu_switch(void** save_sp, void* to_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch,.-emu_switch
)");

#ifdef EMU_TSAN
static void* sched_tsan = nullptr;
static char launch_sync, launch_done, block_sync, block_start, block_done, wave_sync[16];
// The workgroups of a launch run one after another on the same static LDS.  Two passes cover both kinds of race (EMU_TSAN_MODE):
//   "lds"  (default) consecutive workgroups are ordered (the next one's LDS is new memory): races inside a workgroup — LDS and global — are found;
//   "grid"           workgroups are NOT ordered and LDS is exempt: races between the workgroups of one launch on global memory are found.
static bool grid_mode = false;
static bool grid_mode_enabled() { static const bool g = [] { const char* m = getenv("EMU_TSAN_MODE"); return m && !strcmp(m, "grid"); }(); return g; }
#endif
ASAN(static const void* sched_stack_bottom = nullptr; static size_t sched_stack_size = 0;)
static void yield_to_scheduler() {
    TSAN(__tsan_switch_to_fiber(sched_tsan, 1);)   // 1 = no synchronisation by the switch itself
    Fiber* me = cur;
    ASAN(__sanitizer_start_switch_fiber(me->state == DONE ? nullptr : &me->asan_fake, sched_stack_bottom, sched_stack_size);)
    emu_switch(&me->sp, sched_sp);
    ASAN(__sanitizer_finish_switch_fiber(me->asan_fake, &sched_stack_bottom, &sched_stack_size);)
}

void duplicate_lane(bool dup) {
#ifdef EMU_TSAN
    if (dup && !cur->ignoring) { AnnotateIgnoreReadsBegin(__FILE__, __LINE__); AnnotateIgnoreWritesBegin(__FILE__, __LINE__); cur->ignoring = true; }
#else
    (void)dup;
#endif
}

static void fiber_entry() {
    ASAN(__sanitizer_finish_switch_fiber(nullptr, &sched_stack_bottom, &sched_stack_size);)
    TSAN(__tsan_acquire(&launch_sync); if (!grid_mode) __tsan_acquire(&block_start);)
    body->run();
    TSAN(if (!grid_mode) __tsan_release(&block_done);)
    TSAN(if (cur->ignoring) { AnnotateIgnoreReadsEnd(__FILE__, __LINE__); AnnotateIgnoreWritesEnd(__FILE__, __LINE__); cur->ignoring = false; })
    TSAN(__tsan_release(&launch_done);)
    cur->state = DONE;
    yield_to_scheduler();
    __builtin_trap();
}

static void prepare(Fiber& f) {
    if (!f.stack) {
        f.stack = (char*)mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (f.stack == (char*)MAP_FAILED) { fprintf(stderr, "[emu] cannot map a fiber stack\n"); abort(); }
    }
    uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
    void** s = (void**)top;
    *--s = nullptr;               // the return address fiber_entry would return to (it never does)
    *--s = (void*)&fiber_entry;   // popped by emu_switch's ret
    for (int i = 0; i < 6; ++i) *--s = nullptr;
    f.sp = s;
    f.state = READY;
    TSAN(if (!f.tsan) f.tsan = __tsan_create_fiber(0);)
}

uint64_t wave_op(int op, uint64_t in, uint64_t aux, uint64_t aux2) {
    Fiber* f = cur;
    f->site = __builtin_return_address(0);
    f->op = op; f->in = in; f->aux = aux; f->aux2 = aux2;
    f->state = WAIT_WAVE;
    TSAN(__tsan_release(&wave_sync[f->wave & 15]);)
    yield_to_scheduler();
    TSAN(__tsan_acquire(&wave_sync[f->wave & 15]);)
    return f->out;
}

void sync_threads() {
    cur->state = WAIT_BLOCK;
    TSAN(__tsan_release(&block_sync);)
    yield_to_scheduler();
    TSAN(__tsan_acquire(&block_sync);)
}

// DPP controls the kernels use (kernels_vm_seed.hpp mov32<CTRL>): row_shr:n = 0x110 + n, row_shl:n = 0x100 + n, row_ror:n = 0x120 + n,
// quad_perm = 0x00..0xff, row_bcast15 = 0x142, row_bcast31 = 0x143, wave_shr:1 = 0x138, row_mirror = 0x140, row_half_mirror = 0x141
static int dpp_source(unsigned ctrl, int lane, bool& valid) {
    valid = true;
    const int row = lane & ~15, i = lane & 15;
    if (ctrl <= 0xff) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
    if (ctrl >= 0x101 && ctrl <= 0x10f) { const int s = i + (int)(ctrl - 0x100); valid = s < 16; return row + (s & 15); }
    if (ctrl >= 0x111 && ctrl <= 0x11f) { const int s = i - (int)(ctrl - 0x110); valid = s >= 0; return row + (s & 15); }
    if (ctrl >= 0x121 && ctrl <= 0x12f) return row + ((i - (int)(ctrl - 0x120)) & 15);
    if (ctrl == 0x138) { valid = lane > 0; return lane - 1; }
    if (ctrl == 0x140) return row + (15 - i);
    if (ctrl == 0x141) return row + ((i & 8) | (7 - (i & 7)));
    if (ctrl == 0x142) { valid = lane >= 16; return row - 1; }                 // row_bcast15: lane 15 of the previous row
    if (ctrl == 0x143) { valid = lane >= 32; return (lane & 32) - 1; }          // row_bcast31: lane 31 to the upper half
    fprintf(stderr, "[emu] update_dpp control 0x%x is not emulated\n", ctrl); abort();
}

static void serve(Fiber** g, int n) {   // the lanes of one wavefront waiting at one call site, in lane order
    Fiber* by_lane[64] = {};
    for (int k = 0; k < n; ++k) by_lane[g[k]->lane] = g[k];
    const int op = g[0]->op;
    uint64_t ballot = 0;
    if (op == OP_BALLOT) for (int k = 0; k < n; ++k) ballot |= (uint64_t)(g[k]->in & 1) << g[k]->lane;
    for (int k = 0; k < n; ++k) {
        Fiber* f = g[k];
        if (f->op != op) { fprintf(stderr, "[emu] two different wavefront operations at one call site\n"); abort(); }
        switch (op) {
        case OP_BALLOT: f->out = ballot; break;
        case OP_READFIRST: f->out = g[0]->in; break;
        case OP_READLANE: {
            Fiber* s = by_lane[f->aux & 63];
            if (!s) { fprintf(stderr, "[emu] readlane from lane %u, which is not at this call site (site %p)\n", (unsigned)(f->aux & 63), f->site); abort(); }
            f->out = s->in;
        } break;
        case OP_SHFL: { Fiber* s = by_lane[f->aux & 63]; f->out = s ? s->in : f->in; } break;
        case OP_SHFL_UP: { const int src = (int)f->lane - (int)f->aux; Fiber* s = src >= 0 ? by_lane[src] : nullptr; f->out = s ? s->in : f->in; } break;
        case OP_SHFL_XOR: { Fiber* s = by_lane[(f->lane ^ (unsigned)f->aux) & 63]; f->out = s ? s->in : f->in; } break;
        case OP_DPP: {
            const unsigned ctrl = (unsigned)(f->aux & 0xffff), row_mask = (unsigned)(f->aux >> 16) & 15, bank_mask = (unsigned)(f->aux >> 20) & 15;
            const bool bound_ctrl = (f->aux >> 24) & 1;
            bool valid;
            const int src = dpp_source(ctrl, (int)f->lane, valid);
            const bool enabled = ((row_mask >> (f->lane >> 4)) & 1) && ((bank_mask >> ((f->lane >> 2) & 3)) & 1);
            Fiber* s = valid && src >= 0 && src < 64 ? by_lane[src] : nullptr;
            if (!enabled) f->out = f->aux2;
            else if (s) f->out = s->in;
            else f->out = bound_ctrl ? 0 : f->aux2;
        } break;
        case OP_WAVE_BARRIER: f->out = 0; break;
        default: abort();
        }
    }
    for (int k = 0; k < n; ++k) g[k]->state = READY;
}

static void run_block(Fiber* fb, unsigned n_threads) {
    unsigned live = n_threads;
    const unsigned n_waves = (n_threads + 63) / 64;
    for (unsigned t = 0; t < n_threads; ++t) prepare(fb[t]);
    static const bool strict = getenv("EMU_STRICT") != nullptr;
    while (live) {
        // a wavefront runs until each of its lanes has finished or waits in the workgroup barrier
        for (unsigned w = 0; w < n_waves; ++w) {
            const unsigned t0 = w * 64, t1 = std::min(n_threads, t0 + 64);
            while (true) {
                for (unsigned t = t0; t < t1; ++t) {
                    Fiber& f = fb[t];
                    if (f.state != READY) continue;
                    cur = &f; t_idx = f.tidx; lane_in_wave = f.lane;
                    TSAN(__tsan_switch_to_fiber(f.tsan, 1);)
                    ASAN(void* sched_fake = nullptr; __sanitizer_start_switch_fiber(&sched_fake, f.stack, STACK_BYTES);)
                    emu_switch(&sched_sp, f.sp);
                    ASAN(__sanitizer_finish_switch_fiber(sched_fake, nullptr, nullptr);)
                    if (f.state == DONE) --live;
                }
                // every live lane is blocked: serve the wavefront operation at the lowest call site
                const void* lowest = nullptr; bool diverged = false;
                for (unsigned t = t0; t < t1; ++t) {
                    const Fiber& f = fb[t];
                    if (f.state != WAIT_WAVE) continue;
                    if (lowest && f.site != lowest) diverged = true;
                    if (!lowest || (uintptr_t)f.site < (uintptr_t)lowest) lowest = f.site;
                }
                if (!lowest) break;
                if (diverged) {
                    ++n_divergent_sites;
                    for (unsigned t = t0; t < t1; ++t) if (fb[t].state == WAIT_WAVE && fb[t].site != lowest) { ++divergent_pairs[{lowest, fb[t].site}]; break; }
                    if (strict) { fprintf(stderr, "[emu] EMU_STRICT: a wavefront waits at two call sites of wavefront operations\n"); abort(); }
                }
                Fiber* g[64]; int n = 0;
                for (unsigned t = t0; t < t1; ++t) if (fb[t].state == WAIT_WAVE && fb[t].site == lowest) g[n++] = &fb[t];
                serve(g, n);
                ++n_wave_ops;
            }
        }
        // the workgroup barrier: every live work-item waits in it
        for (unsigned t = 0; t < n_threads; ++t) if (fb[t].state == WAIT_BLOCK) fb[t].state = READY;
    }
}

void launch_body(dim3 grid, dim3 block, size_t lds_bytes, const Body& b) {
    if (cur) { fprintf(stderr, "[emu] a kernel launched a kernel\n"); abort(); }
    static thread_local std::vector<Fiber> fb;
    static thread_local std::vector<char> lds;
    const unsigned n_threads = block.x * block.y * block.z;
    if (n_threads == 0 || (size_t)grid.x * grid.y * grid.z == 0) return;
    if (n_threads > 1024) { fprintf(stderr, "[emu] workgroup of %u work-items\n", n_threads); abort(); }
    // (race-detector build, grid mode: consecutive workgroups run on two disjoint sets of fibers — one identity and one stack would order them)
    unsigned sets = 1;
    TSAN(if (grid_mode_enabled()) sets = 2;)
    if (fb.size() < (size_t)sets * n_threads) fb.resize((size_t)sets * n_threads);
    lds.assign(lds_bytes + 16, 0);
    dyn_lds = lds.data();
    body = &b;
    ++n_launches;
#ifdef EMU_TSAN
    static bool once = false;
    if (!once) {
        once = true;
        grid_mode = grid_mode_enabled();
        if (grid_mode) AnnotateBenignRaceSized(__FILE__, __LINE__, __start_emu_lds, __stop_emu_lds - __start_emu_lds, "LDS of consecutive workgroups (grid mode)");
    }
    if (grid_mode && lds_bytes) {
        static void* annotated = nullptr; static size_t annotated_size = 0;
        if (annotated != lds.data() || annotated_size < lds.size()) { annotated = lds.data(); annotated_size = lds.size(); AnnotateBenignRaceSized(__FILE__, __LINE__, annotated, (long)annotated_size, "dynamic LDS (grid mode)"); }
    }
    sched_tsan = __tsan_get_current_fiber(); __tsan_release(&launch_sync);
#endif
    b_dim = block; g_dim = grid;
    for (unsigned t = 0; t < sets * n_threads; ++t) {
        const unsigned u = t % n_threads;
        fb[t].tidx = dim3(u % block.x, (u / block.x) % block.y, u / (block.x * block.y));
        fb[t].lane = u & 63; fb[t].wave = u >> 6;
    }
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                b_idx = dim3(x, y, z);
                ++n_blocks;
#ifdef EMU_TSAN   // a fresh workgroup's LDS: garbage as on the device
                if (!grid_mode) __tsan_acquire(&block_done);
                memset(__start_emu_lds, 0xCD, __stop_emu_lds - __start_emu_lds);
                if (!grid_mode) __tsan_release(&block_start);
#endif
                run_block(fb.data() + (sets == 2 && (n_blocks & 1) ? n_threads : 0), n_threads);
            }
    TSAN(__tsan_acquire(&launch_done);)
    cur = nullptr; body = nullptr;
}

// device memory: host memory filled with a pattern (hipMalloc does not zero; code that relies on zeros must show)
static unsigned long long n_allocs = 0, n_alloc_bytes = 0;
static std::string where(const void* p) {
    Dl_info di;
    if (!dladdr(p, &di) || !di.dli_sname) return "?";
    int st = 0;
    char* d = abi::__cxa_demangle(di.dli_sname, nullptr, nullptr, &st);
    std::string r = (st == 0 && d ? d : di.dli_sname);
    free(d);
    if (r.size() > 100) r = r.substr(0, 100) + "...";
    char off[32]; snprintf(off, sizeof off, " + 0x%lx", (unsigned long)((const char*)p - (const char*)di.dli_saddr));
    return r + off;
}
static void print_stats() {   // EMU_STATS=1: to stderr; EMU_STATS=<path prefix>: appended to <prefix>.<pid>
    const char* dest = getenv("EMU_STATS");
    FILE* o = stderr;
    if (dest && strchr(dest, '/')) { o = fopen((std::string(dest) + "." + std::to_string((long)getpid())).c_str(), "a"); if (!o) o = stderr; }
    for (auto& kv : divergent_pairs)
        fprintf(o, "[emu] divergent x%llu: served %s | left waiting %s\n", kv.second, where(kv.first.first).c_str(), where(kv.first.second).c_str());
    struct rusage ru; getrusage(RUSAGE_SELF, &ru);
    fprintf(o, "[emu] process: %ld minor page faults, %.1f s user, %.1f s system\n", ru.ru_minflt, ru.ru_utime.tv_sec + ru.ru_utime.tv_usec / 1e6, ru.ru_stime.tv_sec + ru.ru_stime.tv_usec / 1e6);
    fprintf(o, "[emu] %llu allocations (%.1f MB), %llu launches, %llu workgroups, %llu wavefront operations served, %llu divergent\n", n_allocs,
            n_alloc_bytes / 1e6, n_launches, n_blocks, n_wave_ops, n_divergent_sites);
    if (o != stderr) fclose(o);
}
// Freed blocks are kept and reused (a page fault is expensive in this container, and glibc returns large blocks to the system).
static std::multimap<size_t, void*> free_blocks;
static std::map<void*, size_t> block_size;
static std::mutex alloc_mutex;
void* alloc(size_t n) {
    if (!n) return nullptr;
    if (!n_allocs++ && getenv("EMU_STATS")) atexit(print_stats);
    n_alloc_bytes += n;
#ifdef EMU_ASAN   // every allocation its own heap block of the exact size: the runtime's red zones and quarantine do the checking
    { void* q = nullptr; if (posix_memalign(&q, 256, n)) return nullptr; memset(q, 0xCD, n); return q; }
#endif
    const size_t cap = (n + 4095) & ~(size_t)4095;
    void* p = nullptr;
    {
    std::lock_guard<std::mutex> lock(alloc_mutex);
    auto it = free_blocks.lower_bound(cap);
    if (it != free_blocks.end() && it->first <= cap + cap / 4) { p = it->second; free_blocks.erase(it); }
    else {
        if (posix_memalign(&p, 4096, cap)) return nullptr;
        block_size[p] = cap;
    }
    }
    static const bool zero = getenv("EMU_ZERO_ALLOC") != nullptr;
    memset(p, zero ? 0 : 0xCD, n);
    return p;
}
void release(void* p) {
    if (!p) return;
    ASAN(free(p); return;)
    std::lock_guard<std::mutex> lock(alloc_mutex);
    auto it = block_size.find(p);
    if (it == block_size.end()) { fprintf(stderr, "[emu] hipFree of an address hipMalloc did not return\n"); abort(); }
    free_blocks.emplace(it->second, p);
}

extern "C" unsigned long long zk_emu_divergent_wave_sites() { return n_divergent_sites; }
unsigned long long buffer_ops[4] = {0, 0, 0, 0};
extern "C" void zk_emu_buffer_ops(unsigned long long out[4]) { for (int i = 0; i < 4; ++i) out[i] = buffer_ops[i]; }

}  // namespace emu
