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
struct Fiber { ucontext_t ctx; emu_idx tid; bool done; };
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

void emu_syncthreads() { swapcontext(&current->ctx, &sched_ctx); }

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
                            f.tid = {tx, ty, tz}; f.done = false;
                            getcontext(&f.ctx);
                            f.ctx.uc_stack.ss_sp = stacks[i]; f.ctx.uc_stack.ss_size = STACK; f.ctx.uc_link = nullptr;
                            makecontext(&f.ctx, trampoline, 0);
                        }
                size_t live = nt;
                while (live) {           // one round = every live fiber runs up to its next barrier
                    for (size_t k = 0; k < nt; k++) {
                        Fiber& f = fibers[k];
                        if (f.done) continue;
                        current = &f; threadIdx = f.tid;
                        swapcontext(&sched_ctx, &f.ctx);
                        if (f.done) live--;
                    }
                }
            }
    body_fn = nullptr;
}
