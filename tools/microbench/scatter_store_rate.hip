// Scattered-store rate of an MI355X: every thread stores B bytes to a pseudo-random row of a buffer far larger than the L2s.
//   hipcc -O3 --offload-arch=gfx950 scatter_store_rate.hip -o scatter_store_rate && ./scatter_store_rate
// Why: three stages of the rasterizer write one small piece per (tile, Gaussian) instance to an address nothing orders -- k_scatter's 8-byte keys,
// k_render_bwd's 48-byte slab rows (flush, and the zero rows behind a tile's deepest contributor) -- and sit at the same ~50-75 G pieces/s whatever
// the piece's size.  This probe pins that number outside the library: rows per second by bytes per row and by how the row is stored.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <chrono>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// MODE 0: 1 byte per row   1: 8 bytes   2: 16 bytes   3: 48 bytes as three 16-byte stores of one lane   4: 48 bytes by three neighbouring lanes (one instruction)
// 5: 8 bytes to CONSECUTIVE rows (the coalesced reference)   6: 48 bytes to rows on a 64-byte pitch (one cache line per row)
template <int MODE>
__global__ __launch_bounds__(256) void k_store(unsigned char* buf, uint32_t rows, uint32_t n, uint32_t salt)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (MODE == 4) {
        const uint32_t e = t >> 2, part = t & 3u;
        if (e >= n || part == 3u) return;
        const uint32_t r = mix(e ^ salt) % rows;
        reinterpret_cast<float4*>(buf + (size_t)r * 48)[part] = make_float4(1.f, 2.f, 3.f, 4.f);
        return;
    }
    if (t >= n) return;
    const uint32_t r = MODE == 5 ? t % rows : mix(t ^ salt) % rows;
    if (MODE == 0) buf[(size_t)r * 48] = 1;
    else if (MODE == 1 || MODE == 5) *reinterpret_cast<uint2*>(buf + (size_t)r * (MODE == 5 ? 8 : 48)) = make_uint2(t, salt);
    else if (MODE == 2) *reinterpret_cast<float4*>(buf + (size_t)r * 48) = make_float4(1.f, 2.f, 3.f, 4.f);
    else if (MODE == 7) {                                 // 48 bytes, one lane, three NON-TEMPORAL stores (global_store_dwordx4 ... nt)
        typedef float v4 __attribute__((ext_vector_type(4)));
        v4* p = reinterpret_cast<v4*>(buf + (size_t)r * 48);
        const v4 a = {1.f, 2.f, 3.f, 4.f};
        __builtin_nontemporal_store(a, p); __builtin_nontemporal_store(a, p + 1); __builtin_nontemporal_store(a, p + 2);
    } else if (MODE == 8) {                               // 8 bytes, non-temporal
        typedef unsigned v2 __attribute__((ext_vector_type(2)));
        const v2 a = {t, salt};
        __builtin_nontemporal_store(a, reinterpret_cast<v2*>(buf + (size_t)r * 48));
    } else {
        float4* p = reinterpret_cast<float4*>(buf + (size_t)r * (MODE == 6 ? 64 : 48));
        p[0] = make_float4(1.f, 2.f, 3.f, 4.f); p[1] = make_float4(5.f, 6.f, 7.f, 8.f); p[2] = make_float4(9.f, 10.f, 11.f, 12.f);
    }
}

// a vector-bound kernel for the overlap test: 256 threads x iters dependent-free v_fma_f32 (8 accumulators), workgroups of 1024 threads like k_render_bwd's
__global__ __launch_bounds__(1024) void k_fma(float* out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float m = 0.999f, c = 1e-3f;
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c); a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
    }
    out[blockIdx.x * 1024 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main()
{
    const uint32_t rows = 3u << 20;                      // 3.1 M rows x 48 B = 151 MB (x 64 B = 201 MB for mode 6): the slab of a frame with every splat x 8
    unsigned char* buf = nullptr;
    if (hipMalloc(&buf, (size_t)rows * 64) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(buf, 0, (size_t)rows * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"1 byte per row", "8 bytes per row", "16 bytes per row", "48 bytes per row, one lane, three stores", "48 bytes per row, three lanes, one instruction",
                           "8 bytes, consecutive addresses (coalesced)", "48 bytes per row on a 64-byte pitch (one line per row)", "48 bytes per row, one lane, three non-temporal stores", "8 bytes per row, non-temporal"};
    printf("# tools/microbench/scatter_store_rate.hip on MI355X: n threads each store one piece to a pseudo-random row of a %u-row buffer (48-byte pitch unless said otherwise)\n", rows);
    for (int mode = 0; mode < 9; mode++) {
        for (uint32_t n : {750000u, 2300000u, 9200000u}) {
            const uint32_t threads = mode == 4 ? n * 4u : n;
            const dim3 grid((threads + 255) / 256);
            float best = 1e30f;
            for (int rep = 0; rep < 5; rep++) {
                hipEventRecord(e0);
                switch (mode) {
                case 0: hipLaunchKernelGGL(k_store<0>, grid, dim3(256), 0, 0, buf, rows, n, 77u + rep); break;
                case 1: hipLaunchKernelGGL(k_store<1>, grid, dim3(256), 0, 0, buf, rows, n, 77u + rep); break;
                case 2: hipLaunchKernelGGL(k_store<2>, grid, dim3(256), 0, 0, buf, rows, n, 77u + rep); break;
                case 3: hipLaunchKernelGGL(k_store<3>, grid, dim3(256), 0, 0, buf, rows, n, 77u + rep); break;
                case 4: hipLaunchKernelGGL(k_store<4>, grid, dim3(256), 0, 0, buf, rows, n, 77u + rep); break;
                case 5: hipLaunchKernelGGL(k_store<5>, grid, dim3(256), 0, 0, buf, rows, n, 77u + rep); break;
                case 6: hipLaunchKernelGGL(k_store<6>, grid, dim3(256), 0, 0, buf, rows, n, 77u + rep); break;
                case 7: hipLaunchKernelGGL(k_store<7>, grid, dim3(256), 0, 0, buf, rows, n, 77u + rep); break;
                default: hipLaunchKernelGGL(k_store<8>, grid, dim3(256), 0, 0, buf, rows, n, 77u + rep); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;     // (first launch: code load)
            }
            const int bytes = mode == 0 ? 1 : (mode == 1 || mode == 5 || mode == 8) ? 8 : mode == 2 ? 16 : 48;
            printf("%-58s n %8u  %8.1f us  %7.1f G rows/s  %8.1f GB/s of payload\n", names[mode], n, best * 1e3, n / (best * 1e-3) / 1e9, (double)n * bytes / (best * 1e-3) / 1e9);
        }
    }
    // Do scattered stores ride beside vector work?  k_fma (2 workgroups of 1024 threads per CU, all SIMDs busy) alone, the 2.3 M 48-byte rows alone, both at once on two streams.
    {
        float* out = nullptr; hipMalloc(&out, (size_t)512 * 8 * 1024 * 4);
        hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
        hipEvent_t a0, a1, b0, b1; hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1);
        const uint32_t n = 2300000u; const dim3 gs((n + 255) / 256);
        for (int iters : {350, 700}) {
            for (int what = 0; what < 3; what++) {           // 0: fma alone, 1: stores alone, 2: both
                float best = 1e30f, bfma = 0, bst = 0;
                for (int rep = 0; rep < 4; rep++) {
                    hipDeviceSynchronize();
                    const auto t0 = std::chrono::steady_clock::now();
                    if (what != 1) { hipEventRecord(a0, sa); hipLaunchKernelGGL(k_fma, dim3(512 * 4), dim3(1024), 0, sa, out, iters); hipEventRecord(a1, sa); }
                    if (what != 0) { hipEventRecord(b0, sb); hipLaunchKernelGGL(k_store<3>, gs, dim3(256), 0, sb, buf, rows, n, 99u + rep); hipEventRecord(b1, sb); }
                    hipDeviceSynchronize();
                    const float wall = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
                    float f = 0, st = 0;
                    if (what != 1) hipEventElapsedTime(&f, a0, a1);
                    if (what != 0) hipEventElapsedTime(&st, b0, b1);
                    if (rep > 0 && wall < best) { best = wall; bfma = f; bst = st; }
                }
                printf("overlap test, k_fma %5d iterations: %-12s wall %7.1f us   k_fma %7.1f us   2.3 M scattered 48-byte rows %7.1f us\n", iters,
                       what == 0 ? "fma alone" : what == 1 ? "stores alone" : "both at once", best * 1e3, bfma * 1e3, bst * 1e3);
            }
        }
    }
    return 0;
}
