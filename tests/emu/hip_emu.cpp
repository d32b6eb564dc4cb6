// Fiber scheduler behind hip_emu.h (test infrastructure only).
#include "hip_emu.h"

#include <ucontext.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

namespace emu {

struct Fiber {
    ucontext_t ctx;
    dim3 tid;
    int lane = 0, wave = 0;
    bool done = false;
    char* stack = nullptr;
};

dim3 g_blockIdx, g_blockDim, g_gridDim;
int g_last_error = 0;

static const size_t STACK = 256 * 1024;
static std::vector<Fiber> g_fibers;
static Fiber* g_cur = nullptr;
static ucontext_t g_main;
static const std::function<void()>* g_body = nullptr;
static int g_nthreads = 0;
static int g_bar_count = 0;
static unsigned g_bar_gen = 0;
static unsigned long g_events = 0;   // bumped on every release/finish; a round without events = deadlock

struct WaveState {
    int count = 0;
    unsigned gen = 0;
    int lanes = 64;
    alignas(16) unsigned char buf[64 * 64];
};
static std::vector<WaveState> g_waves;

const dim3& cur_tid() { return g_cur->tid; }
int cur_lane() { return g_cur->lane; }

static inline void yield() { swapcontext(&g_cur->ctx, &g_main); }

void syncthreads() {
    unsigned gen = g_bar_gen;
    if (++g_bar_count == g_nthreads) {
        g_bar_count = 0;
        ++g_bar_gen;
        ++g_events;
    } else {
        while (g_bar_gen == gen) yield();
    }
}

static void wave_sync(WaveState& w) {
    unsigned gen = w.gen;
    if (++w.count == w.lanes) {
        w.count = 0;
        ++w.gen;
        ++g_events;
    } else {
        while (w.gen == gen) yield();
    }
}

void wave_exchange(const void* mine, size_t bytes, void* all64) {
    if (bytes > 64) { fprintf(stderr, "emu: wave_exchange payload too large\n"); abort(); }
    WaveState& w = g_waves[g_cur->wave];
    memcpy(w.buf + (size_t)g_cur->lane * bytes, mine, bytes);
    wave_sync(w);
    memcpy(all64, w.buf, bytes * 64);
    wave_sync(w);
}

static void fiber_entry() {
    (*g_body)();
    g_cur->done = true;
    swapcontext(&g_cur->ctx, &g_main);
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    int nt = (int)(block.x * block.y * block.z);
    if (nt <= 0 || nt > 1024) { g_last_error = 1; return; }
    if ((int)g_fibers.size() < nt) {
        size_t old = g_fibers.size();
        g_fibers.resize(nt);
        for (size_t i = old; i < (size_t)nt; ++i) g_fibers[i].stack = (char*)malloc(STACK);
    }
    int nw = (nt + 63) / 64;
    g_waves.assign(nw, WaveState());
    g_body = &body;
    g_blockDim = block;
    g_gridDim = grid;
    g_nthreads = nt;
    int mode = 0;
    unsigned seed = 1;
    if (const char* e = getenv("SVCMI_EMU_ORDER")) {
        if (strncmp(e, "reverse", 7) == 0) mode = 1;
        else if (strncmp(e, "shuffle", 7) == 0) { mode = 2; if (e[7] == ':') seed = (unsigned)atoi(e + 8); }
    }
    std::mt19937 rng(seed);
    std::vector<int> order(nt);
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_blockIdx = dim3(bx, by, bz);
                g_bar_count = 0;
                for (int w = 0; w < nw; ++w) {
                    g_waves[w].count = 0;
                    g_waves[w].lanes = std::min(64, nt - 64 * w);
                }
                for (int t = 0; t < nt; ++t) {
                    Fiber& f = g_fibers[t];
                    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                    f.lane = t & 63;
                    f.wave = t >> 6;
                    f.done = false;
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = STACK;
                    f.ctx.uc_link = &g_main;
                    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
                }
                int remaining = nt;
                while (remaining > 0) {
                    unsigned long ev = g_events;
                    // SVCMI_EMU_ORDER = reverse | shuffle[:seed]: another order of the fibers inside every scheduling round.  A kernel whose
                    // result depends on it has a cross-thread hazard inside one barrier interval (a missing __syncthreads): the race check
                    // of tests/test_kernels_emu.py runs the half-step kernels under all three orders.
                    for (int t = 0; t < nt; ++t) order[t] = mode == 1 ? nt - 1 - t : t;
                    if (mode == 2) std::shuffle(order.begin(), order.end(), rng);
                    for (int tt = 0; tt < nt; ++tt) {
                        const int t = order[tt];
                        Fiber& f = g_fibers[t];
                        if (f.done) continue;
                        g_cur = &f;
                        swapcontext(&g_main, &f.ctx);
                        if (f.done) { --remaining; ++g_events; }
                    }
                    if (g_events == ev) {   // every live fiber is parked and nothing was released
                        fprintf(stderr, "emu: deadlock (divergent barrier / early return?) in block %u,%u,%u\n", bx, by, bz);
                        abort();
                    }
                }
            }
    g_cur = nullptr;
}

}  // namespace emu
