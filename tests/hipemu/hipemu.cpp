// tests/hipemu/hipemu.cpp -- TEST INFRASTRUCTURE: fiber scheduler behind tests/hipemu/hip/hip_runtime.h.
#include "hip/hip_runtime.h"
#include <stdexcept>

hipemu_idx threadIdx, blockIdx, blockDim, gridDim;

namespace {
constexpr size_t STACK = 512 * 1024;
struct Fiber { ucontext_t ctx; char* stack = nullptr; bool done = true; hipemu_idx tid; };
struct Block {
    std::vector<Fiber> f; int cur = -1; int nlive = 0;
    int arrived = 0; unsigned gen = 0;
    // wave state
    int w_arrived[16]; unsigned w_gen[16]; unsigned w_val[16][64]; bool w_pred[16][64]; int w_live[16];
};
Block B;
ucontext_t sched_ctx;
const std::function<void()>* g_body = nullptr;

void fiber_main() {
    (*g_body)();
    Fiber& me = B.f[B.cur];
    me.done = true; B.nlive--; B.w_live[B.cur / 64]--;
    swapcontext(&me.ctx, &sched_ctx);
}
void yield_() { Fiber& me = B.f[B.cur]; swapcontext(&me.ctx, &sched_ctx); }
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

unsigned hipemu_wave_exchange(unsigned v, int src_lane, bool) {
    int t = B.cur, w = t / 64, l = t & 63;
    B.w_val[w][l] = v;
    wave_barrier(w);
    unsigned r = B.w_val[w][src_lane & 63];
    wave_barrier(w);
    return r;
}

unsigned long long hipemu_ballot(bool p) {
    int t = B.cur, w = t / 64, l = t & 63;
    B.w_pred[w][l] = p;
    wave_barrier(w);
    unsigned long long m = 0;
    int n = (int)B.f.size() - w * 64; if (n > 64) n = 64;
    for (int i = 0; i < n; i++) if (B.w_pred[w][i] && !B.f[w * 64 + i].done) m |= 1ull << i;
    wave_barrier(w);
    return m;
}

void hipemu_launch(dim3 grid, dim3 block, const std::function<void()>& body) {
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
        for (int w = 0; w < 16; w++) { B.w_arrived[w] = 0; B.w_gen[w] = 0; int n = nt - 64 * w; B.w_live[w] = n < 0 ? 0 : (n > 64 ? 64 : n); }
        for (int t = 0; t < nt; t++) {
            Fiber& f = B.f[t];
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = STACK; f.ctx.uc_link = &sched_ctx;
            makecontext(&f.ctx, fiber_main, 0);
            f.done = false;
            f.tid = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
        }
        while (B.nlive > 0) {
            for (int t = 0; t < nt; t++) {
                if (B.f[t].done) continue;
                B.cur = t; threadIdx = B.f[t].tid; blockIdx = {bx, by, bz};
                swapcontext(&sched_ctx, &B.f[t].ctx);
            }
        }
    }
    g_body = nullptr;
}
