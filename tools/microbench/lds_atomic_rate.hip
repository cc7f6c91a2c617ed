// Throughput of LDS float atomics (ds_add_f32) on gfx950 by lane pattern: cycles of LDS time per instruction per CU.
// Build: hipcc -O3 --offload-arch=gfx950 lds_atomic_rate.hip -o lds_atomic_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int ITERS = 2000;
constexpr int WAVES = 16;

template <int MODE>
__global__ __launch_bounds__(1024) void k_rate(const uint32_t* __restrict__ addr_tab, const uint32_t* __restrict__ active_tab, float* out)
{
    __shared__ double acc8[6400];
    float* acc = reinterpret_cast<float*>(acc8);
    for (int i = threadIdx.x; i < 2 * 6400; i += 1024) acc[i] = 0.f;
    __syncthreads();
    const uint32_t a = addr_tab[threadIdx.x];
    const bool on = active_tab[threadIdx.x] != 0;
    float v = (float)threadIdx.x;
    if (on) {
#pragma unroll 4
        for (int it = 0; it < ITERS; it++) {
            if (MODE == 1) atomicAdd(&acc[a], v);
            else if (MODE == 0) acc[a] = v;
            else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned int*>(acc) + a, (unsigned int)it);
            else if (MODE == 3) atomicAdd(reinterpret_cast<unsigned long long*>(acc8) + a, (unsigned long long)it);
            else if (MODE == 4) atomicAdd(acc8 + a, (double)v);
            else if (MODE == 5) atomicMax(reinterpret_cast<unsigned int*>(acc) + a, (unsigned int)it);
            else if (MODE == 6) { const float o = acc[a]; acc[a] = o + v; }
            asm volatile("" : "+v"(v));
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[a];
}

int main()
{
    const int nblk = 256;
    uint32_t *d_addr, *d_act; float* d_out;
    hipMalloc(&d_addr, 1024 * 4); hipMalloc(&d_act, 1024 * 4); hipMalloc(&d_out, nblk * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    srand(1);
    struct Pat { const char* name; int atomic; };
    const char* names[] = {"64 lanes, consecutive addresses", "16 lanes (lane < 16), consecutive", "64 lanes, random addresses", "64 lanes, (pixel*513 + j) with 16 random j",
                           "36 lanes of the block-list kernel (16+16+4 averaged as 12)", "64 lanes, one address", "4 lanes", "1 lane", "ds_write_b32, 64 lanes consecutive", "64 lanes, stride 2", "32 lanes consecutive",
                           "ds_add_u32, 64 lanes consecutive", "ds_add_u64, 64 lanes consecutive", "ds_add_f64, 64 lanes consecutive", "ds_max_u32, 64 lanes consecutive", "ds_read + v_add + ds_write (not atomic), 64 lanes", "ds_add_u32, 64 lanes random",
                           "ds_add_f64, k_render_bwd's pattern: acc[pq][j], 16 random j per wave (row stride 385)", "ds_add_f64, entry-major: acc[j][pq], 12 doubles per j", "ds_add_f64, entry-major, 16 doubles per j",
                           "ds_add_f64, 64 lanes random addresses", "ds_add_f64, acc[pq][j] with row stride 392 (= 8 mod 32 double-banks)"};
    for (int p = 0; p < 22; p++) {
        std::vector<uint32_t> addr(1024), act(1024);
        for (int t = 0; t < 1024; t++) {
            const int lane = t & 63, wv = t >> 6;
            uint32_t a = 0, on = 1;
            switch (p) {
            case 0: case 8: case 11: case 12: case 13: case 14: case 15: a = wv * 64 + lane; break;
            case 16: a = rand() % (9 * 513); break;
            case 1: a = wv * 64 + lane; on = lane < 16; break;
            case 2: a = rand() % (9 * 513); break;
            case 3: { static uint32_t js[16][16]; if (lane == 0 && wv == 0) for (auto& r : js) for (auto& x : r) x = rand() % 512;
                      a = ((lane >> 2) & 3) * 513 + js[wv][(lane >> 4) * 4 + (lane & 3)]; } break;
            case 4: a = (lane >> 4) * 513 + (rand() % 512); on = (lane & 15) < 3; break;
            case 5: a = 7; break;
            case 6: a = wv * 64 + lane; on = lane < 4; break;
            case 7: a = wv * 64; on = lane == 0; break;
            case 9: a = (wv * 64 + lane) * 2; break;
            case 10: a = wv * 64 + lane; on = lane < 32; break;
            case 17: case 18: case 19: case 21: {             // lane = (row r = lane >> 4, pixel pq = (lane >> 2) & 3, entry slot e = lane & 3): 16 random entries j per wave
                static uint32_t js[16][16]; if (lane == 0 && wv == 0) for (auto& r : js) for (auto& x : r) x = rand() % 384;
                const uint32_t j = js[wv][(lane >> 4) * 4 + (lane & 3)], pq = (lane >> 2) & 3;
                a = p == 17 ? pq * 385 + j : p == 18 ? j * 12 + pq : p == 19 ? j * 16 + pq : pq * 392 + j; } break;
            case 20: a = rand() % (9 * 385); break;
            }
            addr[t] = a; act[t] = on;
        }
        hipMemcpy(d_addr, addr.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(d_act, act.data(), 4096, hipMemcpyHostToDevice);
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
#define L(M) hipLaunchKernelGGL(k_rate<M>, dim3(nblk), dim3(1024), 0, 0, d_addr, d_act, d_out)
            if (p == 8) L(0); else if (p == 11 || p == 16) L(2); else if (p == 12) L(3); else if (p == 13 || p >= 17) L(4); else if (p == 14) L(5); else if (p == 15) L(6); else L(1);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        // one workgroup per CU (256 CUs): LDS time per instruction = ms * clock / (16 waves * ITERS)
        const double cyc = ms * 1e-3 * 2.4e9 / (WAVES * (double)ITERS);
        printf("%-62s %8.3f ms  %7.1f clk / instruction (one CU, 16 waves issuing)\n", names[p], ms, cyc);
    }
    return 0;
}
