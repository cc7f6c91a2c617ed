// VALU issue rate of gfx950 by resident waves per SIMD: how many wave64 vector instructions per second the chip retires
// when every SIMD holds 1, 2, 4 or 8 waves that issue independent instructions back to back.  Settles the ceiling the
// render kernels are priced against (MI355X_MICROARCH.md: one wave alone issues a v_fma_f32 every 4 cycles, the SIMD-32
// executes one in 2 -- so 2+ waves per SIMD should reach 2 cycles per instruction; DESIGN.md of round 1 assumed 4).
// Build: hipcc -O3 --offload-arch=gfx950 valu_issue_rate.hip -o valu_issue_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int ITERS = 4000;       // loop trips; each trip = 32 instructions of the stream under test
constexpr int PER_TRIP = 32;

// MODE 0: v_fma_f32 (8 independent accumulators)      MODE 1: v_mul_f32_dpp quad_perm (8 independent)
// MODE 2: v_exp_f32 (8 independent)                    MODE 3: the render-loop mix: 6 fma/mul + 1 dpp + 1 exp per 8
// MODE 4: v_cndmask_b32_dpp + v_fma (the chain step)   MODE 5: dependent v_fma_f32 chain (1 accumulator)
// MODE 6 / 7 (round 4): the render mix of MODE 3 with the SCALAR density of the real loops beside it -- k_render_fwd carries 0.71 scalar
// instructions per vector instruction (s_and / s_or / s_andn2_b64 on the lane masks done / live / fail / upd / stop, s_mov_b64 vcc in
// front of every v_cndmask_b32_dpp), k_render_bwd 0.37 (profiles/r03_f_sq_counters.json): 6 resp. 3 scalar instructions per 8 vector ones.
// One scalar unit serves a CU's four SIMDs; if it co-limits the loops, the VECTOR rate of these modes falls below MODE 3's.
template <int MODE>
__global__ __launch_bounds__(256) void k_valu(float* out, float seed, unsigned long long* clocks)
{
    extern __shared__ float lds_pad[];       // dynamic LDS only caps the number of resident workgroups per CU
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float m = 0.999f, c = 1e-3f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
    const v2f pm = {m, m}, pc = {c, c};
    const unsigned lds_addr = (threadIdx.x & 255u) * 4u;
    if (MODE == 11) lds_pad[threadIdx.x] = a0;
    unsigned long long m0 = __builtin_amdgcn_ballot_w64(a0 > 3.f), m1 = __builtin_amdgcn_ballot_w64(a0 > 5.f), m2 = ~m0;     // lane masks in SGPR pairs
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
#define R8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define FMA(x) "v_fma_f32 %[" #x "], %[" #x "], %[m], %[c]\n\t"
#define DPP(x) "v_mul_f32_dpp %[" #x "], %[" #x "], %[m] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define EXP(x) "v_exp_f32 %[" #x "], %[" #x "]\n\t"
#define CND(x) "v_cndmask_b32_dpp %[" #x "], %[" #x "], %[m], vcc quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
#define OPS : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6), [a7] "+v"(a7) : [m] "v"(m), [c] "v"(c) : "vcc"
        if (MODE == 0) asm volatile(R8(FMA) R8(FMA) R8(FMA) R8(FMA) OPS);
        else if (MODE == 1) asm volatile(R8(DPP) R8(DPP) R8(DPP) R8(DPP) OPS);
        else if (MODE == 2) asm volatile(R8(EXP) R8(EXP) R8(EXP) R8(EXP) OPS);
        else if (MODE == 3) {
#define MIX FMA(a0) FMA(a1) FMA(a2) DPP(a3) FMA(a4) FMA(a5) EXP(a6) FMA(a7)
            asm volatile(MIX MIX MIX MIX OPS);
        } else if (MODE == 4) {
#define CH CND(a0) FMA(a1) CND(a2) FMA(a3) CND(a4) FMA(a5) CND(a6) FMA(a7)
            asm volatile(CH CH CH CH OPS);
        } else if (MODE == 5) {
#define D8 FMA(a0) FMA(a0) FMA(a0) FMA(a0) FMA(a0) FMA(a0) FMA(a0) FMA(a0)
            asm volatile(D8 D8 D8 D8 OPS);
        } else if (MODE == 8 || MODE == 9) {
            // round 5: packed fp32 (two fp32 operations per lane and instruction on a 64-bit register pair) -- does it issue at the rate of v_fma_f32?
#define PKF(x) "v_pk_fma_f32 %[" #x "], %[" #x "], %[m], %[c]\n\t"
#define PKM(x) "v_pk_mul_f32 %[" #x "], %[" #x "], %[m]\n\t"
#define OPSP : [a0] "+v"(p0), [a1] "+v"(p1), [a2] "+v"(p2), [a3] "+v"(p3), [a4] "+v"(p4), [a5] "+v"(p5), [a6] "+v"(p6), [a7] "+v"(p7) : [m] "v"(pm), [c] "v"(pc)
            if (MODE == 8) asm volatile(R8(PKF) R8(PKF) R8(PKF) R8(PKF) OPSP);
            else asm volatile(R8(PKM) R8(PKM) R8(PKM) R8(PKM) OPSP);
        } else if (MODE == 10) {
#define WSHR(x) "v_mov_b32_dpp %[" #x "], %[" #x "] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            asm volatile(R8(WSHR) R8(WSHR) R8(WSHR) R8(WSHR) OPS);
        } else if (MODE == 12 || MODE == 13) {
            // gfx950's row-granular swaps: v_permlane32_swap (a.hi <-> b.lo), v_permlane16_swap (odd rows of a <-> even rows of b)
#define SW32(x, y) "v_permlane32_swap_b32 %[" #x "], %[" #y "]\n\t"
#define SW16(x, y) "v_permlane16_swap_b32 %[" #x "], %[" #y "]\n\t"
#define S8_32 SW32(a0, a1) SW32(a2, a3) SW32(a4, a5) SW32(a6, a7) SW32(a0, a2) SW32(a1, a3) SW32(a4, a6) SW32(a5, a7)
#define S8_16 SW16(a0, a1) SW16(a2, a3) SW16(a4, a5) SW16(a6, a7) SW16(a0, a2) SW16(a1, a3) SW16(a4, a6) SW16(a5, a7)
            if (MODE == 12) asm volatile(S8_32 S8_32 S8_32 S8_32 OPS);
            else asm volatile(S8_16 S8_16 S8_16 S8_16 OPS);
        } else if (MODE == 14) {
            // the reduction step built on them: 1 swap + 1 v_add_f32 per pair of values
#define RS(x, y) SW32(x, y) "v_add_f32 %[" #x "], %[" #x "], %[" #y "]\n\t"
#define RS8 RS(a0, a1) RS(a2, a3) RS(a4, a5) RS(a6, a7)
            asm volatile(RS8 RS8 RS8 RS8 OPS);
        } else if (MODE == 11) {
            // 7 v_fma_f32 + 1 ds_read_b32 per 8: does an LDS read ride beside the vector stream (the SSIM window from LDS instead of DPP shifts)?
#define LDSR(x) "ds_read_b32 %[" #x "], %[ad]\n\t"
#define MIXL FMA(a0) FMA(a1) FMA(a2) FMA(a3) FMA(a4) FMA(a5) FMA(a6) "s_waitcnt lgkmcnt(1)\n\t" LDSR(a7)
            asm volatile(MIXL MIXL MIXL MIXL "s_waitcnt lgkmcnt(0)\n\t" : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6), [a7] "+v"(a7) : [m] "v"(m), [c] "v"(c), [ad] "v"(lds_addr) : "memory");
        } else {
#define SAND "s_and_b64 %[s0], %[s0], %[s1]\n\t"
#define SOR "s_or_b64 %[s1], %[s1], %[s2]\n\t"
#define SAN2 "s_andn2_b64 %[s2], %[s2], %[s0]\n\t"
#define SMOV "s_mov_b64 vcc, %[s0]\n\t"
#define OPSS : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6), [a7] "+v"(a7), [s0] "+s"(m0), [s1] "+s"(m1), [s2] "+s"(m2) : [m] "v"(m), [c] "v"(c) : "vcc", "scc"
#define MIXS6 FMA(a0) SAND FMA(a1) SOR FMA(a2) SMOV DPP(a3) SAN2 FMA(a4) SAND FMA(a5) EXP(a6) SOR FMA(a7)
#define MIXS3 FMA(a0) SAND FMA(a1) FMA(a2) SMOV DPP(a3) FMA(a4) FMA(a5) SOR EXP(a6) FMA(a7)
            if (MODE == 6) asm volatile(MIXS6 MIXS6 MIXS6 MIXS6 OPSS);
            else asm volatile(MIXS3 MIXS3 MIXS3 MIXS3 OPSS);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)__builtin_popcountll(m0 ^ m1 ^ m2)
                                          + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
    if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

int main()
{
    const int CUS = 256, ROUNDS = 4;
    float* d_out; unsigned long long* d_clk;
    hipMalloc(&d_out, (size_t)CUS * 8 * ROUNDS * 256 * 4);
    hipMalloc(&d_clk, (size_t)CUS * 8 * ROUNDS * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"v_fma_f32 x8 independent", "v_mul_f32_dpp quad_perm x8 independent", "v_exp_f32 x8 independent",
                           "render mix: 6 fma + 1 mul_dpp + 1 exp", "chain step: v_cndmask_b32_dpp + v_fma_f32 alternating", "v_fma_f32 dependent chain",
                           "render mix + 6 SALU per 8 VALU (k_render_fwd: 0.71)", "render mix + 3 SALU per 8 VALU (k_render_bwd: 0.37)",
                           "v_pk_fma_f32 x8 independent (2 fma per lane each)", "v_pk_mul_f32 x8 independent", "v_mov_b32_dpp wave_shr:1 x8 independent",
                           "7 v_fma_f32 + 1 ds_read_b32 per 8 (reads counted)", "v_permlane32_swap_b32 x8", "v_permlane16_swap_b32 x8",
                           "v_permlane32_swap_b32 + v_add_f32 alternating"};
    printf("# tools/microbench/valu_issue_rate.hip on MI355X (gfx950): 256-thread workgroups (one wave per SIMD each), W workgroups resident per CU\n"
           "# (capped through dynamic LDS), %d x %d wave64 vector instructions per wave; G winstr/s = VECTOR wave-instructions retired chip-wide per second (the scalar ones of modes 6 / 7 ride beside them, uncounted);\n"
           "# cyc/instr/SIMD = in-kernel cycles (s_memtime) / (W * instructions per wave): 2.0 = the SIMD-32 execute rate, 4.0 = one wave alone.\n", ITERS, PER_TRIP);
    for (int mode = 0; mode < 15; mode++) {
        for (int W : {1, 2, 4, 8}) {
            const size_t lds = (size_t)(160 * 1024 / W) - 1024;          // W workgroups fit one CU's 160 KiB, W + 1 do not
            const int grid = CUS * W * ROUNDS;
            hipFuncSetAttribute((const void*)k_valu<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
            hipFuncSetAttribute((const void*)k_valu<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
            hipFuncSetAttribute((const void*)k_valu<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
            hipFuncSetAttribute((const void*)k_valu<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
            hipFuncSetAttribute((const void*)k_valu<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
            hipFuncSetAttribute((const void*)k_valu<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
            hipFuncSetAttribute((const void*)k_valu<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
            hipFuncSetAttribute((const void*)k_valu<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
            hipFuncSetAttribute((const void*)k_valu<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
            hipFuncSetAttribute((const void*)k_valu<9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
            hipFuncSetAttribute((const void*)k_valu<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
            hipFuncSetAttribute((const void*)k_valu<11>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
            hipFuncSetAttribute((const void*)k_valu<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
            hipFuncSetAttribute((const void*)k_valu<13>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
            hipFuncSetAttribute((const void*)k_valu<14>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
            float ms = 0;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
#define L(M) hipLaunchKernelGGL(k_valu<M>, dim3(grid), dim3(256), lds, 0, d_out, 1.0f, d_clk)
                switch (mode) { case 0: L(0); break; case 1: L(1); break; case 2: L(2); break; case 3: L(3); break; case 4: L(4); break; case 5: L(5); break; case 6: L(6); break; case 7: L(7); break; case 8: L(8); break; case 9: L(9); break; case 10: L(10); break; case 11: L(11); break; case 12: L(12); break; case 13: L(13); break; default: L(14); }
                if (hipError_t er = hipGetLastError()) { printf("launch of mode %d failed: %s\n", mode, hipGetErrorString(er)); break; }
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            std::vector<unsigned long long> clk((size_t)grid * 2);
            hipMemcpy(clk.data(), d_clk, clk.size() * 8, hipMemcpyDeviceToHost);
            double cyc = 0, real = 0;
            for (int b = 0; b < grid; b++) { cyc += (double)clk[2 * b]; real += (double)clk[2 * b + 1]; }
            cyc /= grid; real /= grid;
            const double instr_per_wave = (double)ITERS * PER_TRIP;
            const double total = instr_per_wave * 4.0 * grid;                     // 4 waves per workgroup
            const double ghz = cyc / (real * 10.0);                               // s_memrealtime ticks at 100 MHz
            printf("%-56s W=%d  %8.3f ms  %8.1f G winstr/s  %5.2f cyc/instr/SIMD  in-kernel clock %.2f GHz\n", names[mode], W, ms,
                   total / (ms * 1e-3) / 1e9, cyc / (W * instr_per_wave), ghz);
        }
    }
    return 0;
}
