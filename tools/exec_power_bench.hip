// exec_power_bench.hip — does the sustained vector-issue rate of the chip depend on how many lanes are enabled?
// (dev tool, round 4.)  tools/valu_peak_bench.hip shows the vector pipe of a warm MI355X issuing at 0.55-0.76 of its
// nominal rate depending on the instruction mix — a power limit, not a pipeline one.  If disabled lanes save power, a
// kernel whose lanes are only partly live (the blend kernels: 58 %) would run faster with EXEC narrowed to the live lanes
// than with all lanes computing on masked values.  Same FMA loop, 8 independent chains, 8 waves per SIMD, all CUs,
// EXEC = all 64 lanes / 48 / 32 / 16; also with all lanes enabled but half of them multiplying zeros.
// Build: hipcc --offload-arch=gfx950 -O3 -o exec_power_bench exec_power_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITERS 32768
#define CHAINS 8

template <int LANES, bool ZEROS, int PAT = 0>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    float v[CHAINS], w[CHAINS];
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
        v[c] = threadIdx.x * 0.001f + c; w[c] = 0.5f + 0.01f * c;
        if (ZEROS && lane >= 32) { v[c] = 0.f; w[c] = 0.f; }
    }
    const bool on = PAT == 0 ? lane < LANES : PAT == 1 ? (lane & 1) == 0 : PAT == 2 ? lane != 5 : (lane & 31) < 16;
    if (on) {   // (the loop runs with EXEC = the selected lanes)
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[c]) : "v"(w[c]));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += v[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int LANES, bool ZEROS, int PAT = 0>
static void run(float* out, int cus, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = cus * 8;
    for (int i = 0; i < 12; i++) k<LANES, ZEROS, PAT><<<grid, 256>>>(out, ITERS);   // ≈ 40 ms: past the power-state ramp
    float best = 1e30f, sum = 0.f;
    for (int rep = 0; rep < 8; rep++) {
        hipEventRecord(e0); k<LANES, ZEROS, PAT><<<grid, 256>>>(out, ITERS); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
    }
    const double inst = (double)ITERS * CHAINS * 8 * cus * 4;
    printf("%-44s best %7.4f ms  mean %7.4f ms  %7.1f G wave-inst/s (chip, best)\n", name, best, sum / 8, inst / (best * 1e-3) * 1e-9);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* out; hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    printf("# %s, %d CUs; v_fma_f32, 8 chains, 8 waves/SIMD; nominal 2-cycle peak %.0f G wave-inst/s\n", p.name, cus, cus * 4 * p.clockRate * 1e-6 / 2);
    for (int round = 0; round < 2; round++) {
        run<64, false>(out, cus, "EXEC = 64 lanes");
        run<48, false>(out, cus, "EXEC = 48 lanes");
        run<32, false>(out, cus, "EXEC = 32 lanes");
        run<16, false>(out, cus, "EXEC = 16 lanes");
        run<64, true>(out, cus, "EXEC = 64 lanes, lanes 32-63 compute on zeros");
        run<64, false, 2>(out, cus, "EXEC = 63 lanes (lane 5 off)");
        run<64, false, 1>(out, cus, "EXEC = even lanes");
        run<64, false, 3>(out, cus, "EXEC = lanes 0-15 and 32-47");
    }
    return 0;
}
