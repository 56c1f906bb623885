// tests/emu/emu_runtime.cpp -- TEST INFRASTRUCTURE ONLY: fiber scheduler behind
// tests/emu/hip/hip_runtime.h (see that file's header).
#include <ucontext.h>

#include <stdio.h>
#include <vector>

#include "hip/hip_runtime.h"

emu_idx threadIdx, blockIdx, blockDim, gridDim;
namespace tfft { alignas(16) unsigned char tfft_smem[160 * 1024]; }

namespace {
constexpr size_t STACK = 256 * 1024;
struct Fiber { ucontext_t ctx; emu_idx tid; bool done; int waiting; };   // waiting: 0 running, 1 at a block barrier, 2 at a wave barrier
ucontext_t sched_ctx;
std::vector<Fiber> fibers;
std::vector<char*> stacks;
Fiber* current = nullptr;
const std::function<void()>* body_fn = nullptr;

void trampoline() {
    (*body_fn)();
    current->done = true;
    swapcontext(&current->ctx, &sched_ctx);
}
}  // namespace

void emu_syncthreads() { current->waiting = 1; swapcontext(&current->ctx, &sched_ctx); }
void emu_wave_sync() { current->waiting = 2; swapcontext(&current->ctx, &sched_ctx); }

// ballot among the 64 fibers of the wave: deposit, barrier, read, barrier, clear, barrier (the third
// barrier keeps a fast lane's next deposit from being wiped by a slow lane's clear)
namespace { unsigned long long ballot_acc[16]; }
static unsigned linear_tid() { return current->tid.x + blockDim.x * (current->tid.y + blockDim.y * current->tid.z); }
unsigned emu_lane() { return linear_tid() & 63u; }
unsigned long long emu_ballot(int pred) {
    const unsigned w = linear_tid() >> 6, l = linear_tid() & 63u;
    if (pred) ballot_acc[w] |= 1ull << l;
    emu_wave_sync();
    const unsigned long long r = ballot_acc[w];
    emu_wave_sync();
    ballot_acc[w] = 0;
    emu_wave_sync();
    return r;
}

// the value lane 0 of the wave holds (every lane of the wave calls)
namespace { unsigned rfl_acc[16]; }
unsigned emu_readfirstlane(unsigned v) {
    const unsigned w = linear_tid() >> 6, l = linear_tid() & 63u;
    if (l == 0) rfl_acc[w] = v;
    emu_wave_sync();
    const unsigned r = rfl_acc[w];
    emu_wave_sync();
    return r;
}

void emu_launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    if (shmem > sizeof(tfft::tfft_smem)) { fprintf(stderr, "emu: %zu bytes of LDS requested\n", shmem); abort(); }
    const size_t nt = (size_t)block.x * block.y * block.z;
    if (nt > 1024) { fprintf(stderr, "emu: %zu threads per block\n", nt); abort(); }
    while (stacks.size() < nt) stacks.push_back((char*)malloc(STACK));
    fibers.resize(nt);
    body_fn = &body;
    blockDim = {block.x, block.y, block.z};
    gridDim = {grid.x, grid.y, grid.z};
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                blockIdx = {bx, by, bz};
                size_t i = 0;
                for (unsigned tz = 0; tz < block.z; tz++)
                    for (unsigned ty = 0; ty < block.y; ty++)
                        for (unsigned tx = 0; tx < block.x; tx++, i++) {
                            Fiber& f = fibers[i];
                            f.tid = {tx, ty, tz}; f.done = false; f.waiting = 0;
                            getcontext(&f.ctx);
                            f.ctx.uc_stack.ss_sp = stacks[i]; f.ctx.uc_stack.ss_size = STACK; f.ctx.uc_link = nullptr;
                            makecontext(&f.ctx, trampoline, 0);
                        }
                // Scheduler with real barrier semantics: a fiber parked at a block barrier resumes only when
                // every live fiber of the block is parked there; one parked at a wave barrier resumes when every
                // live fiber of its wave (64 consecutive linear thread ids) is parked at a wave barrier.
                size_t live = nt;
                auto run = [&](Fiber& f) {
                    f.waiting = 0; current = &f; threadIdx = f.tid;
                    swapcontext(&sched_ctx, &f.ctx);
                    if (f.done) live--;
                };
                while (live) {
                    bool progressed = false;
                    for (size_t k = 0; k < nt; k++) if (!fibers[k].done && fibers[k].waiting == 0) { run(fibers[k]); progressed = true; }
                    for (size_t w0 = 0; w0 < nt; w0 += 64) {          // wave barriers
                        const size_t w1 = w0 + 64 < nt ? w0 + 64 : nt;
                        bool any = false, all = true;
                        for (size_t k = w0; k < w1; k++) if (!fibers[k].done) { any = true; if (fibers[k].waiting != 2) all = false; }
                        if (any && all) { for (size_t k = w0; k < w1; k++) if (!fibers[k].done) run(fibers[k]); progressed = true; }
                    }
                    bool any = false, all = true;                      // block barrier
                    for (size_t k = 0; k < nt; k++) if (!fibers[k].done) { any = true; if (fibers[k].waiting != 1) all = false; }
                    if (any && all) { for (size_t k = 0; k < nt; k++) if (!fibers[k].done) run(fibers[k]); progressed = true; }
                    if (!progressed && live) { fprintf(stderr, "emu: barrier deadlock (divergent barriers) in block %u,%u,%u\n", bx, by, bz); abort(); }
                }
            }
    body_fn = nullptr;
}
