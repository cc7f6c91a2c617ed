// FETCH_SIZE / WRITE_SIZE calibration on gfx950 for the access patterns of this library's kernels (MI355X_MICROARCH.md, HBM section:
// "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths are uncalibrated: calibrate on
// a known byte count in your own access pattern before trusting an absolute").  Each kernel below touches a KNOWN number of bytes of a
// buffer that is larger than the Infinity Cache and is read / written once, cold:
//   k_stream16   16 B per lane, consecutive lanes consecutive addresses        (SH rows, records recA / recB, slab rows: 3 x 16 B)
//   k_stream8     8 B per lane                                                  (keys, recC, quadrant masks)
//   k_stream4     4 B per lane                                                  (depths, slots, n_contrib, images)
//   k_gather64   one 64-B line per lane at a random line index (4 x 16 B)      (k_finalize: the pack line of a sorted instance)
//   k_gather48   three 16-B pieces of a 48-B row at a random row               (k_preprocess_bwd: slab rows of a Gaussian, Gaussian-major)
//   k_store16 / k_store8 / k_scatter8: streaming 16-B / 8-B stores and 8-B stores to random slots (k_scatter's keys)
// Run under  rocprofv3 --kernel-trace --pmc FETCH_SIZE  and  --pmc WRITE_SIZE  (separate passes); tools/fetch_calibration.py divides the
// counter by the known bytes -> profiles/<round>_fetch_calibration.json, which tools/pmc_to_json.py applies per kernel.
// Build: hipcc -O3 --offload-arch=gfx950 fetch_calibration.hip -o fetch_calibration
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

constexpr size_t BYTES = 1ull << 30;            // 1 GiB per buffer: 4 x the Infinity Cache

__global__ __launch_bounds__(256) void k_stream16(const float4* __restrict__ p, size_t n, float* out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_stream8(const float2* __restrict__ p, size_t n, float* out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float2 v = p[i]; acc += v.x + v.y; }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_stream4(const float* __restrict__ p, size_t n, float* out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_gather64(const float4* __restrict__ p, const uint32_t* __restrict__ idx, size_t n_lines, float* out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * 256) {
        const float4* l = p + 4 * (size_t)idx[i];
        const float4 a = l[0], b = l[1], c = l[2], d = l[3];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_gather48(const float4* __restrict__ p, const uint32_t* __restrict__ idx, size_t n_rows, float* out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_rows; i += (size_t)gridDim.x * 256) {
        const float4* l = p + 3 * (size_t)idx[i];
        const float4 a = l[0], b = l[1], c = l[2];
        acc += a.x + b.y + c.z;
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_store16(float4* __restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ __launch_bounds__(256) void k_store8(float2* __restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_float2(1.f, (float)i);
}
__global__ __launch_bounds__(256) void k_scatter8(float2* __restrict__ p, const uint32_t* __restrict__ idx, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[idx[i]] = make_float2(1.f, (float)i);
}
// evict: write another GiB so that nothing of the measured buffer is left in L2 / Infinity Cache
__global__ __launch_bounds__(256) void k_evict(float4* __restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

int main()
{
    float4 *buf, *evict; uint32_t* idx; float* out;
    hipMalloc(&buf, BYTES); hipMalloc(&evict, BYTES); hipMalloc(&out, 256);
    const size_t n_lines = BYTES / 64, n_rows48 = BYTES / 48, n_slots8 = BYTES / 8 / 16;      // (scatter: 8 M random slots of the 128 M)
    hipMalloc(&idx, n_lines * 4);
    std::vector<uint32_t> h(n_lines);
    std::iota(h.begin(), h.end(), 0u);
    std::mt19937 rng(1234);
    std::shuffle(h.begin(), h.end(), rng);
    hipMemcpy(idx, h.data(), n_lines * 4, hipMemcpyHostToDevice);
    hipMemset(buf, 0, BYTES);
    const dim3 grid(256 * 16), blk(256);
    auto cold = [&] { hipLaunchKernelGGL(k_evict, grid, blk, 0, 0, evict, BYTES / 16); hipDeviceSynchronize(); };
    printf("# known bytes per launch (read or written once, cold):\n");
    for (int rep = 0; rep < 3; rep++) {
        cold(); hipLaunchKernelGGL(k_stream16, grid, blk, 0, 0, buf, BYTES / 16, out);
        cold(); hipLaunchKernelGGL(k_stream8, grid, blk, 0, 0, (const float2*)buf, BYTES / 8, out);
        cold(); hipLaunchKernelGGL(k_stream4, grid, blk, 0, 0, (const float*)buf, BYTES / 4, out);
        cold(); hipLaunchKernelGGL(k_gather64, grid, blk, 0, 0, buf, idx, n_lines, out);
        cold(); hipLaunchKernelGGL(k_gather48, grid, blk, 0, 0, buf, idx, n_rows48 < n_lines ? n_rows48 : n_lines, out);    // (indices < n_lines <= rows that fit: 48 B rows, 64 B lines)
        cold(); hipLaunchKernelGGL(k_store16, grid, blk, 0, 0, buf, BYTES / 16);
        cold(); hipLaunchKernelGGL(k_store8, grid, blk, 0, 0, (float2*)buf, BYTES / 8);
        cold(); hipLaunchKernelGGL(k_scatter8, grid, blk, 0, 0, (float2*)buf, idx, n_slots8);
        hipDeviceSynchronize();
    }
    const size_t rows48 = n_rows48 < n_lines ? n_rows48 : n_lines;
    printf("k_stream16 read %zu\nk_stream8 read %zu\nk_stream4 read %zu\nk_gather64 read %zu index %zu\nk_gather48 read %zu index %zu\n", BYTES, BYTES, BYTES,
           n_lines * 64, n_lines * 4, rows48 * 48, rows48 * 4);
    printf("k_store16 write %zu\nk_store8 write %zu\nk_scatter8 write %zu index %zu\n", BYTES, BYTES, n_slots8 * 8, n_slots8 * 4);
    return 0;
}
