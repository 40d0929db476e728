// valu_peak_bench.hip — THROUGHPUT of the gfx950 vector pipe per instruction class (dev tool, round 3).
//
// Question settled here: how many cycles does a SIMD need per wave64 VALU instruction when it
// has enough independent work — 4 (SIMD-16 arithmetic, what bench.py assumed in rounds 1-2) or 2 (the hardware
// guide's SIMD-32, MI355X_MICROARCH.md:52-53,430)?  Every kernel below runs 8 INDEPENDENT chains per lane (no
// instruction reads a result younger than 8 instructions), at 1 / 2 / 4 / 8 waves per SIMD on all 256 CUs, and is
// timed with HIP events; cycles per wave-instruction per SIMD = time · f_clk · 1024 SIMDs ÷ (waves · instructions).
// f_clk is measured (s_memtime ticks at 100 MHz vs. the shader's cycle counter is not portable, so: the clock that
// the 1-wave dependent-chain leg implies is printed beside the nominal 2.4 GHz).
//
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_peak_bench valu_peak_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define ITERS 32768
#define CHAINS 8

enum { FMA = 0, MUL_ADD, EXP, RCP, DPP_ADD, PERM32, PERM16, PK_FMA, FMA_DEP, MIX_BLEND, CMP_VCC, CNDMASK_VCC, CMP_CND, CMP_SGPR_CND, MIN_LIT, MOV, MUL_LO, MUL_U24, MAD_U64, LSHL_ADD_U64, MED3, NMODES };
static const char* kNames[NMODES] = {"v_fma_f32 (8 indep chains)", "v_mul_f32 + v_add_f32", "v_exp_f32",
                                     "v_rcp_f32", "v_add_f32_dpp quad_perm", "v_permlane32_swap",
                                     "v_permlane16_swap", "v_pk_fma_f32 (2 flop-pairs/inst)",
                                     "v_fma_f32 (ONE dependent chain)", "blend-like mix: 6 fma + exp + rcp",
                                     "v_cmp_lt_f32 -> vcc", "v_cndmask_b32 (vcc)", "v_cmp (vcc) + v_cndmask (vcc)",
                                     "v_cmp_e64 -> sgpr pair + v_cndmask_e64", "v_min_f32 with a literal", "v_mov_b32",
                                     "v_mul_lo_u32", "v_mul_u32_u24", "v_mad_u64_u32", "v_lshl_add_u64", "v_med3_f32"};
// VALU instructions issued per chain per iteration
static const int kInstPerChainIter[NMODES] = {1, 2, 1, 1, 1, 1, 1, 1, 1, 8, 1, 1, 2, 2, 1, 1, 1, 1, 1, 1, 1};

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    float v[CHAINS], w[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) { v[c] = threadIdx.x * 0.001f + c; w[c] = 0.5f + 0.01f * c; }
    typedef float float2_ __attribute__((ext_vector_type(2)));
    float2_ pv[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) pv[c] = float2_{v[c], w[c]};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (MODE == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[c]) : "v"(w[c]));
            if (MODE == MUL_ADD) {
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
            }
            if (MODE == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[c]));
            if (MODE == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[c]));
            if (MODE == DPP_ADD) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[c]) : "v"(w[c]));
            if (MODE == PERM32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[c]), "+v"(w[c]));
            if (MODE == PERM16) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(v[c]), "+v"(w[c]));
            if (MODE == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(pv[c]));
            if (MODE == CMP_VCC) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(v[c]), "v"(w[c]) : "vcc");
            if (MODE == CNDMASK_VCC) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[c]) : "v"(w[c]) : );
            if (MODE == CMP_CND) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(v[c]), "v"(w[c]) : "vcc");
                asm volatile("s_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[c]) : "v"(w[c]) : "vcc");
            }
            if (MODE == CMP_SGPR_CND) {
                unsigned long long m;
                asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(v[c]), "v"(w[c]));
                asm volatile("s_nop 1\n\tv_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[c]) : "v"(w[c]), "s"(m));
            }
            if (MODE == MIN_LIT) asm volatile("v_min_f32 %0, 0x3f7d70a4, %0" : "+v"(v[c]));
            if (MODE == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(v[c]) : "v"(w[c]));
            if (MODE == MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
            if (MODE == MUL_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
            if (MODE == MAD_U64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(pv[c]) : "v"(v[c]), "v"(w[c]) : "vcc");
            if (MODE == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 2, %0" : "+v"(pv[c]));
            if (MODE == MED3) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(v[c]) : "v"(w[c]));
            if (MODE == MIX_BLEND) {  // the shape of one (entry, pixel) evaluation: geometry fmas, exp, rcp, recurrences
                float t;
                asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(t) : "v"(v[c]), "v"(w[c]));
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(t) : "v"(w[c]));
                asm volatile("v_exp_f32 %0, %0" : "+v"(t));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(t) : "v"(w[c]));
                asm volatile("v_rcp_f32 %0, %0" : "+v"(t));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[c]) : "v"(t));
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[c]) : "v"(t));
                asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(v[c]) : "v"(t));
            }
        }
        if (MODE == FMA_DEP) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[0]) : "v"(w[0]));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += v[c] + w[c] + pv[c].x + pv[c].y;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(float* out, int waves_per_simd, double clk_ghz, int cus) {
    // 256-thread workgroups = one wave per SIMD of a CU; `waves_per_simd` workgroups per CU
    const int grid = cus * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(out, ITERS);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        k<MODE><<<grid, 256>>>(out, ITERS);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double inst_per_wave = (double)ITERS * CHAINS * kInstPerChainIter[MODE];
    const double cyc = best * 1e-3 * clk_ghz * 1e9;        // cycles the launch lasted
    const double per_inst = cyc / (inst_per_wave * waves_per_simd);  // SIMD cycles per wave-instruction
    const double ginst = inst_per_wave * waves_per_simd * cus * 4 / (best * 1e-3) * 1e-9;
    printf("%-36s waves/SIMD %d  %8.4f ms  %6.2f cyc/wave-inst/SIMD  %8.1f G wave-inst/s (chip)\n", kNames[MODE],
           waves_per_simd, best, per_inst, ginst);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

// shader clock actually sustained under a VALU load: s_memtime cycles ÷ wall time of one long launch
__global__ void clock_probe(unsigned long long* out, float* sink) {
    float a = threadIdx.x, b = 1.0001f;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < (1 << 20); i++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

int main(int argc, char** argv) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double clk = p.clockRate * 1e-6;  // kHz → GHz
    const int cus = p.multiProcessorCount;
    printf("# %s: %d CUs, clockRate %.3f GHz (cycles below assume this clock), ITERS %d, %d independent chains per lane\n",
           p.name, cus, clk, ITERS, CHAINS);
    printf("# peak if 2 cyc/inst: %.0f G wave-inst/s; if 4 cyc/inst: %.0f G wave-inst/s\n", cus * 4 * clk / 2, cus * 4 * clk / 4);
    float* out; hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float));
    {
        unsigned long long* cw; hipMallocManaged(&cw, 16);
        clock_probe<<<cus * 8, 256>>>(cw, out);
        hipDeviceSynchronize();
        int wall_khz = 0; hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
        printf("# clock probe (all SIMDs busy with v_fma_f32): %llu s_memtime ticks in %llu wall-clock ticks at %d kHz -> "
               "s_memtime runs at %.1f MHz (a constant counter, NOT the shader clock on this part: cycles below use clockRate)\n",
               cw[0], cw[1], wall_khz, (double)cw[0] / ((double)cw[1] / (wall_khz * 1e3)) * 1e-6);
    }
    const int ws[] = {1, 2, 4, 8};
#define SWEEP(M) for (int w : ws) run<M>(out, w, clk, cus);
    SWEEP(FMA) SWEEP(MUL_ADD) SWEEP(EXP) SWEEP(RCP) SWEEP(DPP_ADD) SWEEP(PERM32) SWEEP(PERM16) SWEEP(PK_FMA)
    SWEEP(FMA_DEP) SWEEP(MIX_BLEND) SWEEP(CMP_VCC) SWEEP(CNDMASK_VCC) SWEEP(CMP_CND) SWEEP(CMP_SGPR_CND) SWEEP(MIN_LIT) SWEEP(MOV) SWEEP(MUL_LO) SWEEP(MUL_U24) SWEEP(MAD_U64) SWEEP(LSHL_ADD_U64) SWEEP(MED3)
    return 0;
}
