// util.hip — the streaming-copy yardstick (gfx950): what a pure HBM streaming kernel reaches on THIS part.
//
// bench.py prices the streaming kernels (preprocess_fwd / preprocess_bwd) against 8 TB/s (the spec) AND against a
// measured ceiling.  Until round 3 that ceiling was torch's `copy_` (≈ 5.0 TB/s); the hardware guide measures ≈ 6.3 TB/s
// for a float4 copy — this is that kernel, in the library so that the figure comes from the same binary and the same
// run as the kernels it is compared with (ggr_debug_copy: a debug entry point, no counterpart in the reference).
#include "ggr_common.h"

namespace ggr {

// one float4 (16 B) per lane and trip, UNROLL trips in flight per thread, grid-stride; plain (cached) loads and
// non-temporal stores (the destination is not read again)
typedef float v4f __attribute__((ext_vector_type(4)));   // (the non-temporal builtin wants a native vector type)

template <int UNROLL>
__global__ void __launch_bounds__(256)
copy_f4_kernel(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        v4f v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) __builtin_nontemporal_store(v[u], &dst[i + u * stride]);
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}

void launch_copy_f4(const void* src, void* dst, size_t bytes, int blocks, hipStream_t s) {
    const size_t n4 = bytes / 16;
    if (n4 == 0) return;
    if (blocks <= 0) blocks = 256 * 16;   // 16 workgroups per CU
    hipLaunchKernelGGL(copy_f4_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, (const v4f*)src, (v4f*)dst, n4);
}

}  // namespace ggr
