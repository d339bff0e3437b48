// TEST-ONLY: the fiber runtime behind tests/hostsim/fakehip/hip/hip_runtime.h — runs a "launch" one workgroup at a time, every device
// thread a cooperative fiber (ucontext), with workgroup barriers and wave64 rendezvous.  DEFINITIONS: include it in exactly one
// translation unit of a test library (devsim.cpp, enginesim.cpp).  Nothing in the product includes this file.
#pragma once
#include <hip/hip_runtime.h>
#include <functional>

// ---- the fiber runtime behind fakehip ------------------------------------------------------------------------------------
namespace fakehip {
State S;
static std::function<void()> g_body;
static constexpr size_t kStack = 256 * 1024;
static int g_readers[16][2];

static int live_count() { int n = 0; for (auto& f : S.fib) n += f.done ? 0 : 1; return n; }
static int live_in_wave(int w) {
    int n = 0;
    for (int l = 0; l < kWave; ++l) { const size_t i = (size_t)w * kWave + l; if (i < S.fib.size() && !S.fib[i].done) n++; }
    return n;
}
// the switch between the scheduler and a fiber: on x86-64 a dozen instructions (callee-saved registers + the stack pointer) instead of
// swapcontext, whose two signal-mask system calls per switch were a third of the suites' time; ucontext elsewhere
#if defined(__x86_64__)
extern "C" void fh_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl fh_switch
    .type fh_switch,@function
fh_switch:
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
    .size fh_switch,.-fh_switch
)");
static void* g_sched_sp;
static void to_scheduler() { fh_switch(&S.fib[S.cur].sp, g_sched_sp); }
static void to_fiber(uint32_t t) { fh_switch(&g_sched_sp, S.fib[t].sp); }
static void trampoline();
static void fiber_entry() { trampoline(); abort(); }              // (a finished fiber is never resumed)
static void fiber_init(Fiber& f, char* stack, size_t size) {
    void** top = (void**)(((uintptr_t)stack + size) & ~(uintptr_t)15);
    *--top = nullptr;                                             // where the entry function's caller would have left its return address
    *--top = (void*)fiber_entry;                                  // fh_switch's `ret` goes here
    for (int i = 0; i < 6; ++i) *--top = nullptr;                 // rbp rbx r12 r13 r14 r15
    f.sp = top;
}
#else
static void to_scheduler() { swapcontext(&S.fib[S.cur].ctx, &S.sched); }
static void to_fiber(uint32_t t) { swapcontext(&S.sched, &S.fib[t].ctx); }
static void trampoline();
static void fiber_init(Fiber& f, char* stack, size_t size) {
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = stack; f.ctx.uc_stack.ss_size = size; f.ctx.uc_link = &S.sched;
    makecontext(&f.ctx, (void (*)())trampoline, 0);
}
#endif
void yield() { to_scheduler(); }
void barrier() {
    const unsigned long long gen = S.bar_gen;
    S.bar_waiting++;
    for (;;) {
        if (S.bar_gen != gen) break;
        if (S.bar_waiting == live_count()) { S.bar_waiting = 0; S.bar_gen++; break; }
        S.fib[S.cur].waiting = 1;
        yield();
    }
    S.fib[S.cur].waiting = 0;
    S.progress++;
}
unsigned long long wave_exchange(unsigned long long v, int, unsigned long long* all, unsigned long long* live_mask) {
    const int tid = S.cur, w = tid / kWave, lane = tid % kWave;
    const int buf = (int)(S.lane_gen[tid]++ & 1);
    S.wx[w][buf][lane] = v;
    S.warrived[w][buf]++;
    while (S.warrived[w][buf] < live_in_wave(w)) { S.fib[tid].waiting = 2; yield(); }
    S.fib[tid].waiting = 0;
    S.progress++;
    unsigned long long lm = 0;
    for (int l = 0; l < kWave; ++l) {
        const size_t i = (size_t)w * kWave + l;
        if (i < S.fib.size() && !S.fib[i].done) lm |= 1ull << l;
        if (all) all[l] = S.wx[w][buf][l];
    }
    if (live_mask) *live_mask = lm;
    // the last reader re-arms the buffer (everybody has arrived, so nobody can be two operations ahead)
    if (++g_readers[w][buf] == live_in_wave(w)) { g_readers[w][buf] = 0; S.warrived[w][buf] = 0; }
    return 0;
}
static void trampoline() {
    g_body();
    S.fib[S.cur].done = true;
    S.progress++;
    to_scheduler();
}
static std::vector<char*> g_stacks;
template <class F> void launch(dim3 grid, dim3 block, const void* kernarg, F body) {
    g_body = body;
    S.gdim = grid; S.bdim = block; S.kernarg = kernarg;
    while (g_stacks.size() < block.x) g_stacks.push_back((char*)malloc(kStack));
    std::vector<uint32_t> order(grid.x);
    for (uint32_t b = 0; b < grid.x; ++b) order[b] = S.block_order == 1 ? grid.x - 1 - b : b;
    if (S.block_order == 2)
        for (uint32_t b = grid.x; b > 1; --b) {
            S.rng ^= S.rng << 13; S.rng ^= S.rng >> 7; S.rng ^= S.rng << 17;
            std::swap(order[b - 1], order[S.rng % b]);
        }
    for (uint32_t bi = 0; bi < grid.x; ++bi) {
        const uint32_t b = order[bi];
        S.bidx = dim3(b);
        S.fib.assign(block.x, Fiber{});
        S.bar_waiting = 0;
        memset(S.warrived, 0, sizeof(S.warrived)); memset(S.lane_gen, 0, sizeof(S.lane_gen)); memset(g_readers, 0, sizeof(g_readers));
        for (uint32_t t = 0; t < block.x; ++t) {
            Fiber& f = S.fib[t];
            f.done = false; f.waiting = 0;
            fiber_init(f, g_stacks[t], kStack);
        }
        // run-to-yield, round robin (S.chaos reverses the order of every other pass and yields inside atomics)
        unsigned pass = 0;
        for (;;) {
            const unsigned long long before = S.progress;
            bool any = false;
            for (uint32_t k = 0; k < block.x; ++k) {
                const uint32_t t = (S.chaos && (pass & 1)) ? block.x - 1 - k : k;
                if (S.fib[t].done) continue;
                any = true;
                S.cur = (int)t; S.tidx = dim3(t);
                to_fiber(t);
            }
            if (!any) break;
            if (S.progress == before) {
                fprintf(stderr, "[devsim] deadlock in workgroup %u: ", b);
                for (uint32_t t = 0; t < block.x; ++t) if (!S.fib[t].done) fprintf(stderr, "%u:%d ", t, S.fib[t].waiting);
                fprintf(stderr, "\n");
                abort();
            }
            pass++;
        }
    }
}
}  // namespace fakehip
