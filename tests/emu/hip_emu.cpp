// Fiber scheduler behind hip_emu.h (test infrastructure only).
//
// Two execution modes:
//   * default: the blocks of a launch run one after another on the calling thread, each as blockDim cooperative fibers.  With the
//     sanitizer build (build_emu(sanitize=True): -fsanitize=address,undefined, loaded under LD_PRELOAD=libasan.so) every `__shared__`
//     array is a static object of exactly its declared size with red zones around it, every device buffer (a torch CPU tensor =
//     an intercepted posix_memalign) has red zones too, and the LDS-DMA emulation's copies are checked like any other access: an LDS
//     overrun or a global access outside its tensor is reported at the faulting source line of the KERNEL.
//   * SVCMI_EMU_BLOCKS=K (K >= 2; needs a -DSVCMI_EMU_TLS build = build_emu(tls=True)): K blocks of a launch are resident at once,
//     each on its own OS thread with its own fibers and its own copy of every `__shared__` array (thread_local storage), and the
//     threads take turns round by round -- block b's fibers run one scheduling round, then block b+1's, ... -- so the blocks of a
//     launch interleave at barrier granularity the way co-resident workgroups of a CU do.  A kernel whose result depends on it
//     (a block that reads rows another block of the same launch writes, state left in "LDS" by the previous block, a ticket
//     protocol that assumes block order) differs from the sequential run.
#include "hip_emu.h"

#include <ucontext.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

namespace emu {

struct Fiber {
    ucontext_t ctx;
    dim3 tid;
    int lane = 0, wave = 0;
    bool done = false;
    char* stack = nullptr;
};

struct WaveState {
    int count = 0;
    unsigned gen = 0;
    int lanes = 64;
    alignas(16) unsigned char buf[64 * 64];
};

// per executing OS thread (one in the default mode, K in the concurrent-blocks mode)
thread_local dim3 g_blockIdx;
dim3 g_blockDim, g_gridDim;
int g_last_error = 0;

static const size_t STACK = 256 * 1024;
static thread_local std::vector<Fiber> g_fibers;
static thread_local Fiber* g_cur = nullptr;
static thread_local ucontext_t g_main;
static const std::function<void()>* g_body = nullptr;
static int g_nthreads = 0;
static thread_local int g_bar_count = 0;
static thread_local unsigned g_bar_gen = 0;
static thread_local unsigned long g_events = 0;   // bumped on every release/finish; a round without events = deadlock
static thread_local std::vector<WaveState> g_waves;

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

// lock-step baton of the concurrent-blocks mode: exactly one block thread runs at a time, turns pass round-robin after every
// scheduling round (deterministic: the interleaving is a function of the launch alone)
struct Baton {
    std::mutex m;
    std::condition_variable cv;
    int turn = 0, k = 1;
    std::vector<char> alive;
    void wait_turn(int me) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return turn == me; });
    }
    void pass(int me) {
        std::unique_lock<std::mutex> l(m);
        int t = me;
        for (int i = 0; i < k; ++i) {
            t = (t + 1) % k;
            if (alive[t]) break;
        }
        turn = t;
        cv.notify_all();
    }
};

struct Order {
    int mode = 0;
    std::mt19937 rng{1};
};

// all the blocks `first, first + step, ...` of the (flattened) grid on the calling thread; `bt` != null: pass the baton after every round
static void run_blocks(dim3 grid, dim3 block, long long first, long long step, Baton* bt, int me, int mode, unsigned seed) {
    const int nt = g_nthreads;
    if ((int)g_fibers.size() < nt) {
        size_t old = g_fibers.size();
        g_fibers.resize(nt);
        for (size_t i = old; i < (size_t)nt; ++i) g_fibers[i].stack = (char*)malloc(STACK);
    }
    const int nw = (nt + 63) / 64;
    g_waves.assign(nw, WaveState());
    std::mt19937 rng(seed);
    std::vector<int> order(nt);
    const long long total = (long long)grid.x * grid.y * grid.z;
    for (long long lin = first; lin < total; lin += step) {
        const unsigned bx = (unsigned)(lin % grid.x), by = (unsigned)((lin / grid.x) % grid.y), bz = (unsigned)(lin / ((long long)grid.x * grid.y));
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
            if (bt) bt->wait_turn(me);
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
            if (bt) bt->pass(me);
        }
    }
    g_cur = nullptr;
    if (bt) {      // out of blocks: leave the rotation (and hand the turn on if it came back to this thread meanwhile)
        std::unique_lock<std::mutex> l(bt->m);
        bt->alive[me] = 0;
        if (bt->turn == me) {
            int t = me;
            for (int i = 0; i < bt->k; ++i) {
                t = (t + 1) % bt->k;
                if (bt->alive[t]) break;
            }
            bt->turn = t;
            bt->cv.notify_all();
        }
    }
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    int nt = (int)(block.x * block.y * block.z);
    if (nt <= 0 || nt > 1024) { g_last_error = 1; return; }
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
    int k = 1;
    if (const char* e = getenv("SVCMI_EMU_BLOCKS")) k = atoi(e);
    const long long total = (long long)grid.x * grid.y * grid.z;
    if (k > total) k = (int)total;
#ifndef SVCMI_EMU_TLS
    if (k > 1) { fprintf(stderr, "emu: SVCMI_EMU_BLOCKS needs the -DSVCMI_EMU_TLS build (per-thread __shared__ storage)\n"); abort(); }
#endif
    if (k <= 1) {
        run_blocks(grid, block, 0, 1, nullptr, 0, mode, seed);
        return;
    }
    Baton bt;
    bt.k = k;
    bt.alive.assign(k, 1);
    std::vector<std::thread> th;
    for (int i = 0; i < k; ++i) th.emplace_back([&, i] { run_blocks(grid, block, i, k, &bt, i, mode, seed + (unsigned)i); });
    for (auto& t : th) t.join();
}

}  // namespace emu

// ---- self-tests of the harness (called by tests/test_emu_hardened.py through ctypes; they prove the checks can SEE what they claim to)
namespace {
void canary_lds_kernel(float* out, int idx) {
    __shared__ float s[64];
    const int t = (int)threadIdx.x;
    s[t] = (float)t;
    __syncthreads();
    float* volatile base = s;        // (through a pointer, as the kernels address their tiles: the red zone, not a static bounds check, must catch it)
    if (t == 0) base[idx] = 1.0f;    // idx == 64: one float past the block's LDS -- the sanitizer build must report it
    __syncthreads();
    out[blockIdx.x * 64 + t] = s[t];
}
// every block first publishes a word, then (after a barrier) reads its RIGHT neighbour's: run one after another, block b never sees block
// b + 1's word (0); with two resident blocks interleaved round by round it does
void canary_order_kernel(int* flags, int* seen, int nblocks) {
    const int b = (int)blockIdx.x;
    if (threadIdx.x == 0) flags[b] = b + 1;
    __syncthreads();
    __syncthreads();
    if (threadIdx.x == 0) seen[b] = b + 1 < nblocks ? flags[b + 1] : -1;
}
}  // namespace

extern "C" int emu_selftest_lds(float* out, int blocks, int idx) {
    emu::launch(dim3((unsigned)blocks), dim3(64), [=]() { canary_lds_kernel(out, idx); });
    return emu::g_last_error;
}
extern "C" int emu_selftest_order(int* flags, int* seen, int blocks) {
    emu::launch(dim3((unsigned)blocks), dim3(64), [=]() { canary_order_kernel(flags, seen, blocks); });
    return emu::g_last_error;
}
