// xlane_bench.hip — cost of gfx950 cross-lane primitives (dev tool): cycles per wave-instruction,
// measured with s_memtime over a long dependent/independent chain, 1 wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int MODE>
__global__ void k(float* out, long long* cyc) {
    float a = threadIdx.x * 1.0f, b = a + 1.f, c = a + 2.f, d = a + 3.f;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; i++) {
        if (MODE == 0) { a = a * 1.0001f + b; b = b * 1.0001f + c; c = c * 1.0001f + d; d = d * 1.0001f + a; }
        if (MODE == 1) {  // permlane32_swap ×2 (independent pairs)
            auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
            auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(c), __float_as_uint(d), false, false);
            a = __uint_as_float(r[0]); b = __uint_as_float(r[1]); c = __uint_as_float(q[0]); d = __uint_as_float(q[1]);
        }
        if (MODE == 2) {
            auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
            auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(c), __float_as_uint(d), false, false);
            a = __uint_as_float(r[0]); b = __uint_as_float(r[1]); c = __uint_as_float(q[0]); d = __uint_as_float(q[1]);
        }
        if (MODE == 3) {  // 4 dpp adds (quad_perm), independent
            a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xB1, 0xf, 0xf, false));
            b += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b), 0xB1, 0xf, 0xf, false));
            c += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), 0x4E, 0xf, 0xf, false));
            d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x141, 0xf, 0xf, false));
        }
        if (MODE == 4) {  // row_ror:8 and row_bcast
            a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x128, 0xf, 0xf, false));
            b += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b), 0x128, 0xf, 0xf, false));
            c += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), 0x142, 0xa, 0xf, false));
            d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x143, 0xc, 0xf, false));
        }
        if (MODE == 5) {  // ds_swizzle / bpermute based shuffles
            a += __shfl_xor(a, 32); b += __shfl_xor(b, 16); c += __shfl_xor(c, 32); d += __shfl_xor(d, 16);
        }
        if (MODE == 6) { a = __expf(a) ; b = __expf(b); c = __expf(c); d = __expf(d); }
        if (MODE == 7) { a = __builtin_amdgcn_rcpf(a + 2.f); b = __builtin_amdgcn_rcpf(b + 2.f); c = __builtin_amdgcn_rcpf(c + 2.f); d = __builtin_amdgcn_rcpf(d + 2.f); }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMallocManaged(&cyc, 64);
    const char* names[] = {"4x v_fma (indep)", "2x permlane32_swap", "2x permlane16_swap", "4x dpp add quad/half_mirror", "4x dpp add ror8/bcast", "4x shfl_xor(bpermute)+add", "4x v_exp", "4x v_rcp(+add)"};
#define RUN(M) k<M><<<1, 64>>>(out, cyc); hipDeviceSynchronize(); printf("%-32s %8.2f cycles/iter (1 wave)\n", names[M], (double)cyc[M] / N);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    // throughput with 4 waves per SIMD (16 waves per CU)
#define RUNT(M) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); k<M><<<256 * 4, 256>>>(out, cyc); hipDeviceSynchronize(); hipEventRecord(e0); k<M><<<256 * 4, 256>>>(out, cyc); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-32s %8.3f ms for 4 waves/SIMD x %d iters -> %.2f cycles/iter/wave-slot @2.4GHz\n", names[M], ms, N, ms * 1e-3 * 2.4e9 / N / 4); }
    RUNT(0) RUNT(1) RUNT(2) RUNT(3) RUNT(4) RUNT(5) RUNT(6) RUNT(7)
    return 0;
}
