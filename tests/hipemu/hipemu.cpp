// hipemu.cpp -- TEST INFRASTRUCTURE ONLY: fiber scheduler behind tests/hipemu/hipemu.h.
#include <vector>

#include "hipemu.h"

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
)");

namespace hipemu {

struct PendingDma {                 // one queued global_load_lds_dwordx4 wave-instruction
    char* base;
    const void* src[64];
    bool lane[64];
};

struct Wave {
    std::vector<PendingDma> pending; // issued, not yet landed (oldest first)
    int wait_n = 0;
    const void* gsrc[64];
    void* ldst[64];
    int live = 0, count = 0;
    uint64_t gen = 0;
    uint32_t in32[64], out32[64];
    int arg[64], mode[64];
    hipemu_half8 a[64], b[64];
    hipemu_floatx4 c[64], d[64];
    uint64_t tr[64];
    bool present[64];
};

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = false;
    dim3 tid;
    int lane = 0;
    Wave* wave = nullptr;
};

static constexpr size_t kStack = 128 * 1024;
Fiber* cur = nullptr;
dim3 g_blockIdx, g_blockDim, g_gridDim;
static void* sched_sp = nullptr;
static const std::function<void()>* g_body = nullptr;
static int blk_live = 0, blk_count = 0;
static uint64_t blk_gen = 0;
static std::vector<char*> stack_pool;

dim3& cur_tid() { return cur->tid; }

static void yield() { hipemu_switch(&cur->sp, sched_sp); }

// hipemu_set_dma_eager(1): copies land at issue (the other extreme of the latency range; a DMA into a ring stage that a slower
// wave still reads corrupts that wave's operands).  Default: copies land as late as the waits allow.
static int g_dma_eager = 0;
static bool dma_eager() { return g_dma_eager != 0; }

static void land(Wave* w, size_t keep) {          // retire all but the `keep` newest queued copies, oldest first
    while (w->pending.size() > keep) {
        const PendingDma& d = w->pending.front();
        for (int i = 0; i < 64; ++i)
            if (d.lane[i]) memcpy(d.base + 16 * i, d.src[i], 16);
        w->pending.erase(w->pending.begin());
    }
}

static void release_checks_on_exit(Fiber* f) {
    blk_live--;
    if (blk_live > 0 && blk_count == blk_live) { blk_count = 0; blk_gen++; }
    Wave* w = f->wave;
    w->live--;
    w->present[f->lane] = false;
    if (w->live == 0) land(w, 0);
    if (w->live > 0 && w->count == w->live) {
        fprintf(stderr, "hipemu: a lane exited while its wave waits in a collective\n");
        abort();
    }
}

static void fiber_entry() {
    (*g_body)();
    cur->done = true;
    release_checks_on_exit(cur);
    yield();
    abort();
}

void sync_threads() {
    uint64_t my = blk_gen;
    blk_count++;
    if (blk_count == blk_live) { blk_count = 0; blk_gen++; return; }
    while (blk_gen == my) yield();
}

template <typename F> static void collective(F&& compute_all) {
    Wave* w = cur->wave;
    uint64_t my = w->gen;
    w->count++;
    if (w->count == w->live) {
        compute_all(w);
        w->count = 0;
        w->gen++;
        return;
    }
    while (w->gen == my) yield();
}

uint32_t wave_shfl(uint32_t v, int a, int mode) {
    Wave* w = cur->wave;
    int l = cur->lane;
    w->in32[l] = v; w->arg[l] = a; w->mode[l] = mode;
    collective([](Wave* w) {
        for (int i = 0; i < 64; ++i) {
            if (!w->present[i]) continue;
            int src = w->mode[i] == 0 ? (w->arg[i] & 63) : w->mode[i] == 1 ? (i ^ w->arg[i]) : (i + w->arg[i]);
            if (src < 0 || src > 63 || !w->present[src]) src = i;
            w->out32[i] = w->in32[src];
        }
    });
    return w->out32[l];
}

void glds16(const void* gptr, void* lptr) {
    Wave* w = cur->wave;
    int l = cur->lane;
    w->gsrc[l] = gptr; w->ldst[l] = lptr;
    collective([](Wave* w) {
        int first = 0;
        while (first < 64 && !w->present[first]) ++first;
        PendingDma d;
        d.base = (char*)w->ldst[first];
        for (int i = 0; i < 64; ++i) { d.lane[i] = w->present[i]; d.src[i] = w->gsrc[i]; }
        w->pending.push_back(d);
        if (dma_eager()) land(w, 0);                  // zero-latency mode: exposes write-after-read on a stage still being read
    });
}

void wait_vmcnt(int n) {
    Wave* w = cur->wave;
    w->wait_n = n;                                 // wave-uniform by construction (an immediate operand on the hardware)
    collective([](Wave* w) { land(w, (size_t)w->wait_n); });
}

// ds_read_b64_tr_b16 as measured on gfx950 (tools/probes/tr16_probe.hip): every lane supplies the LDS address of a
// word of 4 contiguous 16-bit elements; within each group of 16 lanes, lane i receives element (i & 3) of the words
// supplied by lanes (i >> 2) + 4*j, j = 0..3.  With lane l pointing at row (l & 15) >> 2, columns (l & 3)*4.. of a
// [4][16] block, lane i therefore gets column (i & 15), rows 0..3.
uint64_t ds_read_tr16_b64(const void* lptr) {
    Wave* w = cur->wave;
    int l = cur->lane;
    w->gsrc[l] = lptr;
    collective([](Wave* w) {
        for (int i = 0; i < 64; ++i) {
            if (!w->present[i]) continue;
            uint16_t out[4];
            for (int b = 0; b < 4; ++b) {
                const int src = (i & ~15) + ((i & 15) >> 2) + 4 * b;
                if (!w->present[src]) { fprintf(stderr, "hipemu: transpose read with a missing lane\n"); abort(); }
                out[b] = ((const uint16_t*)w->gsrc[src])[i & 3];
            }
            memcpy(&w->tr[i], out, 8);
        }
    });
    return w->tr[l];
}

hipemu_floatx4 mfma_16x16x32_f16(hipemu_half8 a, hipemu_half8 b, hipemu_floatx4 c) {
    Wave* w = cur->wave;
    int l = cur->lane;
    w->a[l] = a; w->b[l] = b; w->c[l] = c;
    collective([](Wave* w) {
        if (w->live != 64) { fprintf(stderr, "hipemu: MFMA with %d live lanes\n", w->live); abort(); }
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                float acc = w->c[j + 16 * (i >> 2)][i & 3];
                for (int g = 0; g < 4; ++g)
                    for (int e = 0; e < 8; ++e)
                        acc += (float)w->a[i + 16 * g][e] * (float)w->b[j + 16 * g][e];
                w->d[j + 16 * (i >> 2)][i & 3] = acc;
            }
    });
    return w->d[l];
}

void run_grid(dim3 grid, dim3 block, const std::function<void()>& body) {
    const int nthreads = block.x * block.y * block.z;
    const int nwaves = (nthreads + 63) / 64;
    while ((int)stack_pool.size() < nthreads) {
        void* p = nullptr;
        if (posix_memalign(&p, 64, kStack)) abort();
        stack_pool.push_back((char*)p);
    }
    std::vector<Fiber> fibers(nthreads);
    std::vector<Wave> waves(nwaves);
    g_gridDim = grid; g_blockDim = block; g_body = &body;
    Fiber* saved_cur = cur;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_blockIdx = dim3(bx, by, bz);
                blk_live = nthreads; blk_count = 0; blk_gen = 0;
                for (auto& w : waves) { w = Wave(); }
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = fibers[t];
                    f.done = false;
                    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                    f.lane = t & 63;
                    f.wave = &waves[t >> 6];
                    f.wave->live++;
                    f.wave->present[f.lane] = true;
                    f.stack = stack_pool[t];
                    uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
                    void** s = (void**)(top - 64);
                    for (int i = 0; i < 6; ++i) s[i] = nullptr;
                    s[6] = (void*)&fiber_entry;
                    s[7] = nullptr;
                    f.sp = s;
                }
                int remaining = nthreads;
                uint64_t spins = 0;
                while (remaining > 0) {
                    remaining = 0;
                    for (int t = 0; t < nthreads; ++t) {
                        Fiber& f = fibers[t];
                        if (f.done) continue;
                        cur = &f;
                        hipemu_switch(&sched_sp, f.sp);
                        if (!f.done) remaining++;
                    }
                    if (++spins > (1ull << 34)) { fprintf(stderr, "hipemu: deadlock?\n"); abort(); }
                }
            }
    cur = saved_cur;
}

}  // namespace hipemu

extern "C" void hipemu_set_dma_eager(int on) { hipemu::g_dma_eager = on; }
