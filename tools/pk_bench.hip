// pk_bench.hip — does v_pk_fma_f32 double fp32 throughput per instruction on gfx950? (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define N 8192
template <int MODE>
__global__ void k(float* out) {
    float s[8]; f2 p[4];
    for (int i = 0; i < 8; i++) s[i] = threadIdx.x + i;
    for (int i = 0; i < 4; i++) p[i] = f2{(float)threadIdx.x + i, (float)threadIdx.x - i};
    const float c = 1.0001f, d = 0.5f; const f2 c2 = {1.0001f, 0.9999f}, d2 = {0.5f, 0.25f};
#pragma unroll 8
    for (int i = 0; i < N; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) s[j] = __builtin_fmaf(s[j], c, d);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(c2), "v"(d2));
        }
    }
    float r = 0; for (int i = 0; i < 8; i++) r += s[i]; for (int i = 0; i < 4; i++) r += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
    float* out; hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (mode == 0) k<0><<<256 * 8, 256>>>(out); else k<1><<<256 * 8, 256>>>(out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flops = 2.0 * 8 * N * 256.0 * 8 * 256;
            if (rep == 2) printf("%s: %.3f ms  %.1f TFLOP/s  (%d VALU instr per iter)\n", mode == 0 ? "8x v_fma_f32   " : "4x v_pk_fma_f32", ms, flops / ms / 1e9, mode == 0 ? 8 : 4);
        }
    }
    return 0;
}
