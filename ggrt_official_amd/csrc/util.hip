// util.hip — the streaming-copy yardstick (gfx950): what a pure HBM streaming kernel reaches on THIS part.
//
// bench.py prices the streaming kernels (preprocess_fwd / preprocess_bwd) against 8 TB/s (the spec) AND against a
// measured ceiling.  Until round 3 that ceiling was torch's `copy_` (≈ 5.0 TB/s); the hardware guide measures ≈ 6.3 TB/s
// for a float4 copy — this is that kernel, in the library so that the figure comes from the same binary and the same
// run as the kernels it is compared with (ggr_debug_copy: a debug entry point, no counterpart in the reference).
#include "ggr_common.h"

namespace ggr {

// ONE float4 (16 B) per thread, one 256-thread workgroup per 4 KB, plain load and store — the plainest shape is the
// fastest on this part (tools/copy_bench.hip, round 4, 512 MB … 1 GB per buffer, read + write counted):
//     this shape 6.14-6.24 TB/s (the guide's 6.29) | contiguous chunk per workgroup, 4 loads in flight 5.3-5.9 |
//     grid-stride with 4 far-apart streams per thread 4.2-5.1 | hipMemcpyDtoD 4.9-5.2 | read-only 5.0-5.5, write-only
//     4.3-4.5 TB/s: a kernel that only reads or only writes does NOT reach the copy's rate.
// `blocks` > 0 selects the grid-stride form instead (kept for the comparison).
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
copy_f4_kernel(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) dst[i] = src[i];
}

__global__ void __launch_bounds__(256)
copy_f4_stride_kernel(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}

void launch_copy_f4(const void* src, void* dst, size_t bytes, int blocks, hipStream_t s) {
    const size_t n4 = bytes / 16;
    if (n4 == 0) return;
    if (blocks > 0)
        hipLaunchKernelGGL(copy_f4_stride_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const v4f*)src, (v4f*)dst, n4);
    else
        hipLaunchKernelGGL(copy_f4_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (const v4f*)src, (v4f*)dst, n4);
}

// ---- the side-stream probe (api.hip side_stream) --------------------------------------------------------------------
// one wave that stays on the device for `ticks` of the 100 MHz wall clock, and one that does nothing
__global__ void spin_kernel(unsigned long long ticks, uint32_t* started /*host-visible word: set when the wave runs*/) {
    if (started && threadIdx.x == 0) __hip_atomic_store(started, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
__global__ void noop_kernel() {}
void launch_spin(unsigned long long ticks, uint32_t* started, hipStream_t s) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ticks, started);
}
void launch_noop(hipStream_t s) { hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, s); }

}  // namespace ggr
