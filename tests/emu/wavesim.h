// wavesim.h -- TEST INFRASTRUCTURE ONLY (never part of the product library): a lock-step emulator of one gfx950 workgroup on the
// host, used to run the device code of particlemdi.jl_amd/csrc/pmdi_sweep2_body.h on the CPU against the oracle before (and beside)
// the GPU parity tests.  It is not a CPU path of the product: nothing under particlemdi.jl_amd/ includes it, the C-ABI library is
// built without it, and what it checks is kernel LOGIC (indexing, bookkeeping, arithmetic order), not memory ordering.
//
// Model: every lane of the workgroup is a fiber (ucontext) with its own stack, i.e. its own "registers".  A lane runs until it
// reaches a collective -- a wave-wide one (ballot, shuffle, readlane, wave barrier) or the workgroup barrier -- and parks there.
// When every unfinished lane of a wave is parked, the lanes waiting at the same call site are resolved together (lanes at another
// site are in another branch: the hardware would run them under another exec mask), exactly like a wave executing that instruction
// with those lanes active.  LDS is one host buffer per workgroup; LDS atomics are plain read-modify-writes (fibers never preempt).
// WAVESIM_SHUFFLE=<seed> runs the lanes of a wave in a random order between collectives: code that depends on the order of
// unsynchronised LDS accesses inside a wave then fails loudly instead of passing by accident.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <vector>

namespace wavesim {

enum Kind { K_NONE = 0, K_BALLOT, K_SHFL, K_WAVEBAR, K_BLOCKBAR };

struct Lane {
    ucontext_t ctx;
    char *stack = nullptr;
    bool done = false;
    int kind = K_NONE, site = 0;
    uint64_t val = 0;        // operand: predicate / value
    int src = 0;             // shuffle source lane
    uint64_t res = 0;        // result
};

struct Block {
    int T = 0;
    int bid = 0;
    std::vector<Lane> lanes;
    unsigned char *lds = nullptr;
    size_t lds_bytes = 0;
    ucontext_t sched;
    int cur = -1;
    void (*entry)(void *) = nullptr;
    void *arg = nullptr;
    unsigned rng = 0;
    long long collectives = 0;
};

inline Block *&current() { static thread_local Block *b = nullptr; return b; }
inline int tid() { return current()->cur; }
inline int bid() { return current()->bid; }
inline unsigned char *lds_base() { return current()->lds; }

inline void park(int kind, int site, uint64_t val, int src)
{
    Block *b = current();
    Lane &l = b->lanes[b->cur];
    l.kind = kind; l.site = site; l.val = val; l.src = src;
    swapcontext(&l.ctx, &b->sched);
}

inline uint64_t ballot(bool p, int site) { park(K_BALLOT, site, p ? 1 : 0, 0); return current()->lanes[current()->cur].res; }
inline uint64_t shfl64(uint64_t v, int src, int site) { park(K_SHFL, site, v, src); return current()->lanes[current()->cur].res; }
inline void wave_barrier(int site) { park(K_WAVEBAR, site, 0, 0); }
inline void block_barrier(int site) { park(K_BLOCKBAR, site, 0, 0); }

inline void trampoline()
{
    Block *b = current();
    b->entry(b->arg);
    b->lanes[b->cur].done = true;
    b->lanes[b->cur].kind = K_NONE;
    swapcontext(&b->lanes[b->cur].ctx, &b->sched);
}

// resolve the wave-wide collectives of wave w: every unfinished lane is parked; group by (kind, site)
inline void resolve_wave(Block *b, int w)
{
    const int lo = w * 64, hi = (lo + 64 < b->T) ? lo + 64 : b->T;
    bool doneflag[64] = {false};
    for (int i = lo; i < hi; ++i) {
        Lane &li = b->lanes[i];
        if (li.done || li.kind == K_NONE || li.kind == K_BLOCKBAR || doneflag[i - lo]) continue;
        // the group of lanes at the same collective
        uint64_t mask = 0, bal = 0;
        for (int j = i; j < hi; ++j) {
            Lane &lj = b->lanes[j];
            if (!lj.done && lj.kind == li.kind && lj.site == li.site) {
                mask |= 1ull << (j - lo);
                if (lj.val & 1) bal |= 1ull << (j - lo);
            }
        }
        for (int j = i; j < hi; ++j) {
            if (!((mask >> (j - lo)) & 1)) continue;
            Lane &lj = b->lanes[j];
            if (li.kind == K_BALLOT) lj.res = bal;
            else if (li.kind == K_SHFL) {
                const int s = lj.src & 63;
                // an inactive source lane returns the lane's own value (the hardware returns garbage: code must not depend on it)
                lj.res = ((mask >> s) & 1) ? b->lanes[lo + s].val : lj.val;
            }
            doneflag[j - lo] = true;
        }
        for (int j = i; j < hi; ++j)
            if ((mask >> (j - lo)) & 1) b->lanes[j].kind = K_NONE;      // runnable again
        b->collectives += 1;
    }
}

// Run one workgroup of T lanes to completion.  entry(arg) is the kernel body, called once per lane.
inline void run_block(int T, int bid, size_t lds_bytes, void (*entry)(void *), void *arg, size_t stack_bytes = 256 * 1024)
{
    Block blk;
    blk.T = T; blk.bid = bid; blk.entry = entry; blk.arg = arg;
    blk.lanes.resize(T);
    blk.lds_bytes = lds_bytes;
    blk.lds = (unsigned char *)aligned_alloc(64, (lds_bytes + 63) & ~(size_t)63);
    memset(blk.lds, 0xA5, lds_bytes);              // LDS is not zero at kernel start
    const char *sh = getenv("WAVESIM_SHUFFLE");
    blk.rng = sh ? (unsigned)atoi(sh) * 2654435761u + 12345u : 0;
    Block *saved = current();
    current() = &blk;
    for (int i = 0; i < T; ++i) {
        Lane &l = blk.lanes[i];
        l.stack = (char *)malloc(stack_bytes);
        getcontext(&l.ctx);
        l.ctx.uc_stack.ss_sp = l.stack;
        l.ctx.uc_stack.ss_size = stack_bytes;
        l.ctx.uc_link = nullptr;
        makecontext(&l.ctx, (void (*)())trampoline, 0);
    }
    const int nw = (T + 63) / 64;
    for (;;) {
        bool progressed = false, all_done = true;
        int worder[16];
        for (int w = 0; w < nw; ++w) worder[w] = w;
        if (blk.rng)                                 // ... and the waves of the workgroup in a random order, too
            for (int i = nw - 1; i > 0; --i) {
                blk.rng = blk.rng * 1664525u + 1013904223u;
                const int j = (int)((blk.rng >> 8) % (unsigned)(i + 1));
                const int t = worder[i]; worder[i] = worder[j]; worder[j] = t;
            }
        for (int wi = 0; wi < nw; ++wi) {
            const int w = worder[wi];
            const int lo = w * 64, hi = (lo + 64 < T) ? lo + 64 : T;
            // run every runnable lane of the wave up to its next collective
            int order[64];
            const int cnt = hi - lo;
            for (int i = 0; i < cnt; ++i) order[i] = lo + i;
            if (blk.rng)
                for (int i = cnt - 1; i > 0; --i) {
                    blk.rng = blk.rng * 1664525u + 1013904223u;
                    const int j = (int)((blk.rng >> 8) % (unsigned)(i + 1));
                    const int t = order[i]; order[i] = order[j]; order[j] = t;
                }
            for (int oi = 0; oi < cnt; ++oi) {
                Lane &l = blk.lanes[order[oi]];
                if (l.done || l.kind != K_NONE) continue;
                blk.cur = order[oi];
                swapcontext(&blk.sched, &l.ctx);
                progressed = true;
            }
            bool any_live = false;
            for (int i = lo; i < hi; ++i) if (!blk.lanes[i].done) any_live = true;
            if (any_live) { all_done = false; resolve_wave(&blk, w); }
        }
        if (all_done) break;
        // workgroup barrier: every unfinished lane parked at one
        bool at_bar = true, any_bar = false, any_runnable = false;
        for (int i = 0; i < T; ++i) {
            const Lane &l = blk.lanes[i];
            if (l.done) continue;
            if (l.kind == K_BLOCKBAR) any_bar = true; else at_bar = false;
            if (l.kind == K_NONE) any_runnable = true;
        }
        if (any_bar && at_bar) {
            const int site = [&] { for (int i = 0; i < T; ++i) if (!blk.lanes[i].done) return blk.lanes[i].site; return 0; }();
            for (int i = 0; i < T; ++i)
                if (!blk.lanes[i].done) {
                    if (blk.lanes[i].site != site) { fprintf(stderr, "wavesim: lanes wait at different workgroup barriers (lines %d and %d)\n", site, blk.lanes[i].site); abort(); }
                    blk.lanes[i].kind = K_NONE;
                }
            progressed = true;
        } else if (!progressed && !any_runnable) {
            fprintf(stderr, "wavesim: deadlock (some lanes wait at a workgroup barrier while others wait elsewhere or have returned)\n");
            for (int i = 0; i < T; i += 1)
                if (!blk.lanes[i].done && (i % 64 == 0 || blk.lanes[i].site != blk.lanes[i - 1].site))
                    fprintf(stderr, "  lane %d: kind %d at line %d\n", i, blk.lanes[i].kind, blk.lanes[i].site);
            abort();
        }
    }
    for (int i = 0; i < T; ++i) free(blk.lanes[i].stack);
    free(blk.lds);
    current() = saved;
}

}  // namespace wavesim
