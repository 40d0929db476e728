// sh_stage_bench.hip — what bounds a preprocess-shaped streaming kernel whose per-Gaussian SH rows go through LDS
// (dev tool behind NOTES.md, old §4 preprocess rows).
//
// Shape of preprocess_fwd: 256 threads per block, one Gaussian per thread; 40 B of own inputs per thread, the block's
// ROWF-float SH rows copied flat (float4, coalesced) into LDS, a barrier, ROWF LDS reads + FMAs per thread, 68 B
// written per thread.  Modes:
//   0  one group of 256 Gaussians per block (the product's structure)
//   1  G consecutive groups per block, the NEXT group's rows in flight into registers while the current one is
//      consumed (software pipeline; same LDS footprint)
//   2  as 0 with 64-thread blocks (same waves per CU, finer phases)
// Prints GB/s of algorithmic traffic (inputs + rows + outputs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int ROWF, int T>
__device__ __forceinline__ void consume(const float* lds, const float* own, float* out, size_t g, int tid, size_t P) {
    const float* row = lds + tid * ROWF;
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < ROWF / 3; k++) {
        acc[0] += own[k % 10] * row[k]; acc[1] += own[(k + 1) % 10] * row[ROWF / 3 + k]; acc[2] += own[(k + 2) % 10] * row[2 * (ROWF / 3) + k];
    }
    if (g < P) {
#pragma unroll
        for (int k = 0; k < 17; k++) out[(size_t)k * P + g] = acc[k % 3] + own[k % 10];
    }
}

template <int ROWF, int T>
__global__ void __launch_bounds__(T) k_single(const float* __restrict__ own_in, const float* __restrict__ rows, float* __restrict__ out, size_t P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const size_t g0 = (size_t)blockIdx.x * T, g = g0 + tid, gl = g < P ? g : P - 1;
    float own[10];
#pragma unroll
    for (int k = 0; k < 10; k++) own[k] = own_in[(size_t)k * P + gl];
    const int nG = (int)((P - g0) < (size_t)T ? (P - g0) : (size_t)T);
    const int n4 = nG * ROWF / 4;
    constexpr int U = (ROWF + 3) / 4;  // float4 per thread
    const float4* src = reinterpret_cast<const float4*>(rows + g0 * ROWF);
    float4 v[U];
#pragma unroll
    for (int it = 0; it < U; it++) { const int j = it * T + tid; v[it] = src[j < n4 ? j : n4 - 1]; }
#pragma unroll
    for (int it = 0; it < U; it++) { const int j = it * T + tid; if (j < n4) reinterpret_cast<float4*>(lds)[j] = v[it]; }
    __syncthreads();
    consume<ROWF, T>(lds, own, out, g, tid, P);
}

template <int ROWF, int G>
__global__ void __launch_bounds__(256) k_pipe(const float* __restrict__ own_in, const float* __restrict__ rows, float* __restrict__ out, size_t P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int T = 256;
    const int tid = threadIdx.x;
    constexpr int U = (ROWF + 3) / 4;
    const size_t ngroups = (P + T - 1) / T;
    size_t grp = (size_t)blockIdx.x * G;
    float4 v[U];
    float own_n[10];
    auto issue = [&](size_t gi) {
        const size_t g0 = gi * T, gl = g0 + tid < P ? g0 + tid : P - 1;
#pragma unroll
        for (int k = 0; k < 10; k++) own_n[k] = own_in[(size_t)k * P + gl];
        const int nG = (int)((P - g0) < (size_t)T ? (P - g0) : (size_t)T);
        const int n4 = nG * ROWF / 4;
        const float4* src = reinterpret_cast<const float4*>(rows + g0 * ROWF);
#pragma unroll
        for (int it = 0; it < U; it++) { const int j = it * T + tid; v[it] = src[j < n4 ? j : n4 - 1]; }
    };
    if (grp < ngroups) issue(grp);
    for (int i = 0; i < G && grp < ngroups; i++, grp++) {
        const size_t g0 = grp * T;
        const int nG = (int)((P - g0) < (size_t)T ? (P - g0) : (size_t)T);
        const int n4 = nG * ROWF / 4;
        float own[10];
#pragma unroll
        for (int k = 0; k < 10; k++) own[k] = own_n[k];
        if (i) __syncthreads();  // previous group consumed
#pragma unroll
        for (int it = 0; it < U; it++) { const int j = it * T + tid; if (j < n4) reinterpret_cast<float4*>(lds)[j] = v[it]; }
        if (i + 1 < G && grp + 1 < ngroups) issue(grp + 1);
        __syncthreads();
        consume<ROWF, T>(lds, own, out, g0 + tid, tid, P);
    }
}


// mode 3: the row is staged one third at a time (GGRt's channel-major rows: one colour channel), LDS = 256 × ROWF/3
// floats; the next third's loads are in flight while the current one is consumed.  The rows' 128-B lines are
// fetched three times (L2 hits after the first).
template <int ROWF>
__global__ void __launch_bounds__(256) k_chunk(const float* __restrict__ own_in, const float* __restrict__ rows, float* __restrict__ out, size_t P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int T = 256, C = ROWF / 3;
    const int tid = threadIdx.x;
    const size_t g0 = (size_t)blockIdx.x * T, g = g0 + tid, gl = g < P ? g : P - 1;
    float own[10];
#pragma unroll
    for (int k = 0; k < 10; k++) own[k] = own_in[(size_t)k * P + gl];
    const int nG = (int)((P - g0) < (size_t)T ? (P - g0) : (size_t)T);
    const float* src = rows + g0 * ROWF;
    int off[C];
#pragma unroll
    for (int it = 0; it < C; it++) {
        const int j = it * T + tid, gg = j / C, kk = j - gg * C;
        off[it] = (gg < nG ? gg : nG - 1) * ROWF + kk;
    }
    float v[C];
#pragma unroll
    for (int it = 0; it < C; it++) v[it] = src[off[it]];
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        if (c) __syncthreads();
#pragma unroll
        for (int it = 0; it < C; it++) lds[it * T + tid] = v[it];
        if (c < 2) {
#pragma unroll
            for (int it = 0; it < C; it++) v[it] = src[off[it] + (c + 1) * C];
        }
        __syncthreads();
        const float* row = lds + tid * C;
#pragma unroll
        for (int k = 0; k < C; k++) acc[c] += own[(k + c) % 10] * row[k];
    }
    if (g < P) {
#pragma unroll
        for (int k = 0; k < 17; k++) out[(size_t)k * P + g] = acc[k % 3] + own[k % 10];
    }
}


// backward shape: the rows are read AND a gradient row of the same length is written.  k_single_rw: whole rows through
// LDS, flat float4 in and out (the product's structure); k_chunk_rw: one third at a time, 4-B loads and stores.
template <int ROWF>
__global__ void __launch_bounds__(256) k_single_rw(const float* __restrict__ own_in, const float* __restrict__ rows, float* __restrict__ out, float* __restrict__ drows, size_t P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int T = 256;
    const int tid = threadIdx.x;
    const size_t g0 = (size_t)blockIdx.x * T, g = g0 + tid, gl = g < P ? g : P - 1;
    float own[10];
#pragma unroll
    for (int k = 0; k < 10; k++) own[k] = own_in[(size_t)k * P + gl];
    const int nG = (int)((P - g0) < (size_t)T ? (P - g0) : (size_t)T);
    const int n4 = nG * ROWF / 4;
    constexpr int U = (ROWF + 3) / 4;
    const float4* src = reinterpret_cast<const float4*>(rows + g0 * ROWF);
    float4 v[U];
#pragma unroll
    for (int it = 0; it < U; it++) { const int j = it * T + tid; v[it] = src[j < n4 ? j : n4 - 1]; }
#pragma unroll
    for (int it = 0; it < U; it++) { const int j = it * T + tid; if (j < n4) reinterpret_cast<float4*>(lds)[j] = v[it]; }
    __syncthreads();
    float* row = lds + tid * ROWF;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < ROWF; k++) { const float s = row[k]; acc += own[k % 10] * s; row[k] = own[(k + 1) % 10] * own[k % 7]; }
    __syncthreads();
    float4* dst = reinterpret_cast<float4*>(drows + g0 * ROWF);
    for (int j = tid; j < n4; j += T) dst[j] = reinterpret_cast<const float4*>(lds)[j];
    if (g < P) {
#pragma unroll
        for (int k = 0; k < 13; k++) out[(size_t)k * P + g] = acc + own[k % 10];
    }
}

template <int ROWF>
__global__ void __launch_bounds__(256) k_chunk_rw(const float* __restrict__ own_in, const float* __restrict__ rows, float* __restrict__ out, float* __restrict__ drows, size_t P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int T = 256, C = ROWF / 3;
    const int tid = threadIdx.x;
    const size_t g0 = (size_t)blockIdx.x * T, g = g0 + tid, gl = g < P ? g : P - 1;
    float own[10];
#pragma unroll
    for (int k = 0; k < 10; k++) own[k] = own_in[(size_t)k * P + gl];
    const int nG = (int)((P - g0) < (size_t)T ? (P - g0) : (size_t)T);
    const float* src = rows + g0 * ROWF;
    float* dst = drows + g0 * ROWF;
    int off[C];
#pragma unroll
    for (int it = 0; it < C; it++) {
        const int j = it * T + tid, gg = j / C, kk = j - gg * C;
        off[it] = (gg < nG ? gg : nG - 1) * ROWF + kk;
    }
    float v[C];
#pragma unroll
    for (int it = 0; it < C; it++) v[it] = src[off[it]];
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        if (c) __syncthreads();
#pragma unroll
        for (int it = 0; it < C; it++) lds[it * T + tid] = v[it];
        if (c < 2) {
#pragma unroll
            for (int it = 0; it < C; it++) v[it] = src[off[it] + (c + 1) * C];
        }
        __syncthreads();
        float* row = lds + tid * C;
#pragma unroll
        for (int k = 0; k < C; k++) { const float s = row[k]; acc += own[(k + c) % 10] * s; row[k] = own[(k + 1) % 10] * own[k % 7]; }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < C; it++) dst[off[it] + c * C] = lds[it * T + tid];
    }
    if (g < P) {
#pragma unroll
        for (int k = 0; k < 13; k++) out[(size_t)k * P + g] = acc + own[k % 10];
    }
}


// backward, hybrid: rows READ one third (by columns) at a time as in k_chunk, gradient rows WRITTEN whole, in three
// row ranges [0,84) [84,168) [168,256) — each range built in LDS by its owner threads and copied out flat (float4,
// full lines).  LDS = max(256·C, 88·ROWF) floats.
template <int ROWF>
__global__ void __launch_bounds__(256) k_hybrid_rw(const float* __restrict__ own_in, const float* __restrict__ rows, float* __restrict__ out, float* __restrict__ drows, size_t P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int T = 256, C = ROWF / 3;
    const int tid = threadIdx.x;
    const size_t g0 = (size_t)blockIdx.x * T, g = g0 + tid, gl = g < P ? g : P - 1;
    float own[10];
#pragma unroll
    for (int k = 0; k < 10; k++) own[k] = own_in[(size_t)k * P + gl];
    const int nG = (int)((P - g0) < (size_t)T ? (P - g0) : (size_t)T);
    const float* src = rows + g0 * ROWF;
    int off[C];
#pragma unroll
    for (int it = 0; it < C; it++) {
        const int j = it * T + tid, gg = j / C, kk = j - gg * C;
        off[it] = (gg < nG ? gg : nG - 1) * ROWF + kk;
    }
    float v[C];
#pragma unroll
    for (int it = 0; it < C; it++) v[it] = src[off[it]];
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        if (c) __syncthreads();
#pragma unroll
        for (int it = 0; it < C; it++) lds[it * T + tid] = v[it];
        if (c < 2) {
#pragma unroll
            for (int it = 0; it < C; it++) v[it] = src[off[it] + (c + 1) * C];
        }
        __syncthreads();
        const float* row = lds + tid * C;
#pragma unroll
        for (int k = 0; k < C; k++) acc += own[(k + c) % 10] * row[k];
    }
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const int r0 = r * 84, r1 = r == 2 ? 256 : r0 + 84;
        __syncthreads();
        if (tid >= r0 && tid < r1) {
            float* row = lds + (tid - r0) * ROWF;
#pragma unroll
            for (int k = 0; k < ROWF; k++) row[k] = own[(k + 1) % 10] * own[k % 7];
        }
        __syncthreads();
        const int n4 = ((r1 < nG ? r1 : nG) - r0) * ROWF / 4;
        float4* dst = reinterpret_cast<float4*>(drows + (g0 + r0) * ROWF);
        for (int j = tid; j < n4; j += T) dst[j] = reinterpret_cast<const float4*>(lds)[j];
    }
    if (g < P) {
#pragma unroll
        for (int k = 0; k < 13; k++) out[(size_t)k * P + g] = acc + own[k % 10];
    }
}

template <int ROWF>
void run(size_t P, const float* own, const float* rows, float* out, float* drows) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double bytes = (double)P * (40 + 4 * ROWF + 68);
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < 10; i++) launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms / 10 < best ? ms / 10 : best;
        }
        printf("ROWF %d %-28s %.4f ms  %.0f GB/s\n", ROWF, name, best, bytes / best * 1e-6);
    };
    const size_t lds256 = 256 * ROWF * 4, lds64 = 64 * ROWF * 4;
    CHECK(hipFuncSetAttribute((const void*)k_single<ROWF, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256));
    time("single 256", [&] { hipLaunchKernelGGL((k_single<ROWF, 256>), dim3((P + 255) / 256), dim3(256), lds256, 0, own, rows, out, P); });
    time("single 64", [&] { hipLaunchKernelGGL((k_single<ROWF, 64>), dim3((P + 63) / 64), dim3(64), lds64, 0, own, rows, out, P); });
    CHECK(hipFuncSetAttribute((const void*)k_pipe<ROWF, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256));
    CHECK(hipFuncSetAttribute((const void*)k_pipe<ROWF, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256));
    CHECK(hipFuncSetAttribute((const void*)k_pipe<ROWF, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256));
    time("pipe G=2", [&] { hipLaunchKernelGGL((k_pipe<ROWF, 2>), dim3((P + 511) / 512), dim3(256), lds256, 0, own, rows, out, P); });
    time("pipe G=4", [&] { hipLaunchKernelGGL((k_pipe<ROWF, 4>), dim3((P + 1023) / 1024), dim3(256), lds256, 0, own, rows, out, P); });
    time("pipe G=8", [&] { hipLaunchKernelGGL((k_pipe<ROWF, 8>), dim3((P + 2047) / 2048), dim3(256), lds256, 0, own, rows, out, P); });
    time("chunk3", [&] { hipLaunchKernelGGL((k_chunk<ROWF>), dim3((P + 255) / 256), dim3(256), 256 * (ROWF / 3) * 4, 0, own, rows, out, P); });
    CHECK(hipFuncSetAttribute((const void*)k_single_rw<ROWF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256));
    printf("(rw: bytes below exclude the %d-B gradient row; same formula)\n", 4 * ROWF);
    time("rw single", [&] { hipLaunchKernelGGL((k_single_rw<ROWF>), dim3((P + 255) / 256), dim3(256), lds256, 0, own, rows, out, drows, P); });
    time("rw hybrid", [&] { hipLaunchKernelGGL((k_hybrid_rw<ROWF>), dim3((P + 255) / 256), dim3(256), (256 * (ROWF / 3) > 88 * ROWF ? 256 * (ROWF / 3) : 88 * ROWF) * 4, 0, own, rows, out, drows, P); });
    time("rw chunk3", [&] { hipLaunchKernelGGL((k_chunk_rw<ROWF>), dim3((P + 255) / 256), dim3(256), 256 * (ROWF / 3) * 4, 0, own, rows, out, drows, P); });
    CHECK(hipGetLastError());
}

int main(int argc, char** argv) {
    const size_t P = argc > 1 ? (size_t)atol(argv[1]) : 1013760;
    float *own, *rows, *out, *drows;
    CHECK(hipMalloc(&own, P * 40)); CHECK(hipMalloc(&rows, P * 4 * 76)); CHECK(hipMalloc(&out, P * 68)); CHECK(hipMalloc(&drows, P * 4 * 76));
    CHECK(hipMemset(own, 0, P * 40)); CHECK(hipMemset(rows, 0, P * 4 * 76));
    run<48>(P, own, rows, out, drows);
    run<75>(P, own, rows, out, drows);
    CHECK(hipDeviceSynchronize());
    return 0;
}
