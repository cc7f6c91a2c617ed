// cuda_emu.h -- CUDA execution-model emulation for the cross-check harness (build container only).
//
// This is OUR code: it gives meaning to __global__/__shared__/cooperative_groups/atomicAdd so that
// the reference's kernel TEXT (included by line range from /root/reference at build time, into /tmp,
// never into this repository) can execute on host threads.  Because the execution model is ours,
// the result is EVIDENCE about the restatement in ../tgs_oracle.c, not a build of the reference:
// DESIGN.md and ../tgs_oracle.c keep the status "parity unpinned".
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <barrier>
#include <functional>
#include <thread>
#include <vector>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(...)
#define __shared__ static

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint3 { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

inline float min(float a, float b) { return a < b ? a : b; }
inline float max(float a, float b) { return a > b ? a : b; }
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
inline unsigned min(unsigned a, int b) { return min(a, (unsigned)b); }   // CUDA: int converts to unsigned
inline unsigned min(int a, unsigned b) { return min((unsigned)a, b); }
inline unsigned max(unsigned a, int b) { return max(a, (unsigned)b); }
inline unsigned max(int a, unsigned b) { return max((unsigned)a, b); }

inline void __trap() { abort(); }

struct EmuCtx {
    dim3 gridDim, blockDim, blockIdx, threadIdx;
    std::barrier<>* bar = nullptr;
    std::atomic<int>* counters = nullptr;   // two alternating vote counters
    int phase = 0;
};
extern thread_local EmuCtx emu_ctx;

inline float atomicAdd(float* addr, float v)
{
    std::atomic_ref<float> r(*addr);
    return r.fetch_add(v, std::memory_order_relaxed);
}

inline int __syncthreads_count(int pred)
{
    EmuCtx& c = emu_ctx;
    std::atomic<int>& cnt = c.counters[c.phase & 1];
    if (pred) cnt.fetch_add(1);
    c.bar->arrive_and_wait();
    int v = cnt.load();
    c.bar->arrive_and_wait();
    // the counter of this phase is reset by thread 0 only after everyone has read it; the next call
    // uses the other counter, so a fast thread cannot disturb a slow reader
    if (c.threadIdx.x == 0 && c.threadIdx.y == 0) cnt.store(0);
    c.phase++;
    return v;
}

namespace cooperative_groups {
struct grid_group {
    unsigned long long thread_rank() const
    {
        const EmuCtx& c = emu_ctx;
        unsigned long long b = (unsigned long long)c.blockIdx.y * c.gridDim.x + c.blockIdx.x;
        unsigned long long t = (unsigned long long)c.threadIdx.y * c.blockDim.x + c.threadIdx.x;
        return b * ((unsigned long long)c.blockDim.x * c.blockDim.y) + t;
    }
};
struct thread_block {
    dim3 group_index() const { return emu_ctx.blockIdx; }
    dim3 thread_index() const { return emu_ctx.threadIdx; }
    unsigned thread_rank() const { return emu_ctx.threadIdx.y * emu_ctx.blockDim.x + emu_ctx.threadIdx.x; }
    void sync() const { emu_ctx.bar->arrive_and_wait(); }
};
inline grid_group this_grid() { return grid_group(); }
inline thread_block this_thread_block() { return thread_block(); }
}  // namespace cooperative_groups
namespace cg = cooperative_groups;

// Kernels without block-level synchronisation: every CUDA thread is run in turn on the caller.
template <class F> void emu_launch_serial(dim3 grid, dim3 block, F&& fn)
{
    EmuCtx& c = emu_ctx;
    c.gridDim = grid; c.blockDim = block; c.bar = nullptr;
    for (unsigned by = 0; by < grid.y; by++)
        for (unsigned bx = 0; bx < grid.x; bx++)
            for (unsigned ty = 0; ty < block.y; ty++)
                for (unsigned tx = 0; tx < block.x; tx++) {
                    c.blockIdx = dim3(bx, by, 0); c.threadIdx = dim3(tx, ty, 0);
                    fn();
                }
}

// Kernels with barriers: one host thread per CUDA thread of a block, blocks one after the other
// (which is what makes `__shared__` -> `static` sound).
template <class F> void emu_launch_blocks(dim3 grid, dim3 block, F&& fn)
{
    const int nt = (int)(block.x * block.y);
    std::barrier<> bar(nt);
    std::atomic<int> counters[2];
    counters[0] = 0; counters[1] = 0;
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; t++) {
        pool.emplace_back([&, t]() {
            EmuCtx& c = emu_ctx;
            c.gridDim = grid; c.blockDim = block; c.bar = &bar; c.counters = counters; c.phase = 0;
            c.threadIdx = dim3(t % block.x, t / block.x, 0);
            for (unsigned by = 0; by < grid.y; by++)
                for (unsigned bx = 0; bx < grid.x; bx++) {
                    c.blockIdx = dim3(bx, by, 0);
                    fn();
                    bar.arrive_and_wait();   // block boundary: statics are reused by the next block
                }
        });
    }
    for (auto& th : pool) th.join();
}
