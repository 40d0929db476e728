// atomic_line_bench.hip — what a wave pays for committing per-record float atomics as 1, 2 or 3 vector
// instructions (dev tool behind NOTES.md "Atomic line transactions").
//
// Layout as in blend_bwd: records of 16 floats (one 64-B line); lane l of a wave serves record slot l >> 3 with
// value index l & 7; a batch = 8 random records.  Modes:
//   0  one instruction: values 0..7 of the 8 records                      ( 8 line transactions per batch)
//   1  + value 8 from lanes vi == 0 in a second instruction               (16)
//   2  + value 9 from lanes vi == 1 in a third instruction                (24)
//   3  two instructions, every record's 10 values in ONE of them: even slots' values 0..7 from their own lanes
//      plus their values 8, 9 from lanes 0, 1 of the odd neighbour group; then the odd slots'   (8)
// Prints ns per batch per wave-slot and the line-transaction rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ void __launch_bounds__(256) k(float* rec, uint32_t nrec, int batches) {
    const int lane = threadIdx.x & 63, slot = lane >> 3, vi = lane & 7;
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int b = 0; b < batches; b++) {
        // neighbouring waves / batches touch nearby records, like neighbouring tiles share Gaussians
        const uint32_t base = hash32(wave / 8 * 131u + b) % (nrec - 64);
        const uint32_t g = base + hash32(slot * 977u + b + wave) % 64u;
        const uint32_t g_nb = base + hash32((slot ^ 1) * 977u + b + wave) % 64u;
        float* r = rec + 16 * (size_t)g;
        float* r_nb = rec + 16 * (size_t)g_nb;
        const float v = 1.0f + vi;
        if (MODE <= 2) {
            atomicAdd(r + vi, v);
            if (MODE >= 1 && vi == 0) atomicAdd(r + 8, v);
            if (MODE >= 2 && vi == 1) atomicAdd(r + 9, v);
        } else {
            const bool odd = slot & 1;
            {
                float* p = odd ? r_nb + 8 + vi : r + vi;
                if (!odd || vi < 2) atomicAdd(p, v);
            }
            {
                float* p = odd ? r + vi : r_nb + 8 + vi;
                if (odd || vi < 2) atomicAdd(p, v);
            }
        }
    }
}

template <int MODE>
static void run(float* rec, uint32_t nrec, int wgs, int batches, int lines_per_batch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, rec, nrec, batches);
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, rec, nrec, batches);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double nb = (double)wgs * 4 * batches;
    printf("mode %d: %.3f ms for %.2f M batches  -> %.1f G line transactions/s, %.2f G batches/s\n", MODE, ms, nb / 1e6,
           nb * lines_per_batch / ms / 1e6, nb / ms / 1e6);
}

int main(int argc, char** argv) {
    const uint32_t nrec = 1u << 20;
    const int wgs = argc > 1 ? atoi(argv[1]) : 8160, batches = argc > 2 ? atoi(argv[2]) : 56;
    float* rec;
    hipMalloc(&rec, (size_t)nrec * 64);
    hipMemset(rec, 0, (size_t)nrec * 64);
    run<0>(rec, nrec, wgs, batches, 8);
    run<1>(rec, nrec, wgs, batches, 16);
    run<2>(rec, nrec, wgs, batches, 24);
    run<3>(rec, nrec, wgs, batches, 8);
    return 0;
}
