// tests/hipemu/hipemu.cpp -- TEST INFRASTRUCTURE: fiber scheduler behind tests/hipemu/hip/hip_runtime.h.
#include "hip/hip_runtime.h"
#include <stdexcept>
#include <mutex>

hipemu_idx threadIdx, blockIdx, blockDim, gridDim;

namespace {
constexpr size_t STACK = 512 * 1024;
// minimal x86-64 System V context switch (callee-saved registers + stack pointer); ucontext's swapcontext makes two
// sigprocmask system calls per switch, which dominated the run time of barrier-heavy kernels
struct Fiber { void* sp = nullptr; char* stack = nullptr; bool done = true; hipemu_idx tid; };
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
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
.size hipemu_switch,.-hipemu_switch
)");
struct Block {
    std::vector<Fiber> f; int cur = -1; int nlive = 0;
    int arrived = 0; unsigned gen = 0;
    // wave state
    int w_arrived[16]; unsigned w_gen[16]; unsigned w_val[2][16][64]; unsigned w_tog[16][64]; bool w_pred[16][64]; int w_live[16];
};
Block B;
void* sched_sp = nullptr;
const std::function<void()>* g_body = nullptr;

void fiber_main() {
    (*g_body)();
    Fiber& me = B.f[B.cur];
    me.done = true; B.nlive--; B.w_live[B.cur / 64]--;
    hipemu_switch(&me.sp, sched_sp);
    __builtin_unreachable();
}
void yield_() { Fiber& me = B.f[B.cur]; hipemu_switch(&me.sp, sched_sp); }
}

void hipemu_syncthreads() {
    unsigned gen = B.gen;
    if (++B.arrived >= B.nlive) { B.arrived = 0; B.gen++; }
    else while (B.gen == gen) yield_();
}

static void wave_barrier(int w) {
    unsigned gen = B.w_gen[w];
    if (++B.w_arrived[w] >= B.w_live[w]) { B.w_arrived[w] = 0; B.w_gen[w]++; }
    else while (B.w_gen[w] == gen) yield_();
}

unsigned hipemu_wave_exchange_impl(unsigned v, int src_lane, bool) {
    // double-buffered: a lane can only reach its second-next exchange after every lane has left this one
    int t = B.cur, w = t / 64, l = t & 63;
    const unsigned tog = (B.w_tog[w][l]++) & 1;
    B.w_val[tog][w][l] = v;
    wave_barrier(w);
    return B.w_val[tog][w][src_lane & 63];
}

unsigned long long hipemu_ballot_impl(bool p) {
    int t = B.cur, w = t / 64, l = t & 63;
    B.w_pred[w][l] = p;
    wave_barrier(w);
    unsigned long long m = 0;
    int n = (int)B.f.size() - w * 64; if (n > 64) n = 64;
    for (int i = 0; i < n; i++) if (B.w_pred[w][i] && !B.f[w * 64 + i].done) m |= 1ull << i;
    wave_barrier(w);
    return m;
}

// One kernel at a time: the model keeps its block state (and the statics that stand in for LDS) in globals.  Host code with several
// threads (welle.io_amd/host/gpu_node_receiver.cpp: one per shard) then simply takes turns, kernel by kernel.
static std::mutex g_launch_mutex;
void hipemu_launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    std::lock_guard<std::mutex> one_at_a_time(g_launch_mutex);
    const int nt = (int)(block.x * block.y * block.z);
    if (nt > 1024) throw std::runtime_error("hipemu: block too large");
    if ((int)B.f.size() < nt) {
        size_t old = B.f.size(); B.f.resize(nt);
        for (size_t i = old; i < B.f.size(); i++) B.f[i].stack = (char*)malloc(STACK);
    }
    g_body = &body;
    gridDim = {grid.x, grid.y, grid.z}; blockDim = {block.x, block.y, block.z};
    for (unsigned bz = 0; bz < grid.z; bz++) for (unsigned by = 0; by < grid.y; by++) for (unsigned bx = 0; bx < grid.x; bx++) {
        B.nlive = nt; B.arrived = 0; B.gen = 0;
        memset(B.w_tog, 0, sizeof B.w_tog);
        for (int w = 0; w < 16; w++) { B.w_arrived[w] = 0; B.w_gen[w] = 0; int n = nt - 64 * w; B.w_live[w] = n < 0 ? 0 : (n > 64 ? 64 : n); }
        for (int t = 0; t < nt; t++) {
            Fiber& f = B.f[t];
            // initial frame: six zeroed callee-saved registers, then the entry address; rsp after the `ret` into
            // fiber_main must be 8 mod 16 as after a call
            uintptr_t top = ((uintptr_t)f.stack + STACK) & ~(uintptr_t)15;
            void** sp = (void**)(top - 8);
            *--sp = (void*)fiber_main;
            for (int k = 0; k < 6; k++) *--sp = nullptr;
            f.sp = sp;
            f.done = false;
            f.tid = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
        }
        while (B.nlive > 0) {
            for (int t = 0; t < nt; t++) {
                if (B.f[t].done) continue;
                B.cur = t; threadIdx = B.f[t].tid; blockIdx = {bx, by, bz};
                hipemu_switch(&sched_sp, B.f[t].sp);
            }
        }
    }
    g_body = nullptr;
}
