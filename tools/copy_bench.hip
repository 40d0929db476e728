// copy_bench.hip — which float4 copy shape reaches the HBM streaming ceiling on this part (round 4).
// hipcc --offload-arch=gfx950 -O3 -o tools/copy_bench tools/copy_bench.hip ; ./tools/copy_bench
// Variants: grid-stride (UNROLL far-apart streams per thread) vs one contiguous chunk per workgroup; plain vs
// non-temporal stores / loads; workgroups per CU; buffer size (the 256 MiB Infinity Cache flatters small buffers).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT_ST, bool NT_LD>
__global__ void __launch_bounds__(256) copy_stride(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        v4f v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = NT_LD ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { if (NT_ST) __builtin_nontemporal_store(v[u], &dst[i + u * stride]); else dst[i + u * stride] = v[u]; }
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}
// workgroup b copies the contiguous range [b·per, (b+1)·per): UNROLL consecutive 4 KB segments in flight
template <int UNROLL, bool NT_ST, bool NT_LD>
__global__ void __launch_bounds__(256) copy_chunk(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n4, size_t per) {
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
    size_t i = lo + threadIdx.x;
    for (; i + (UNROLL - 1) * 256 < hi; i += UNROLL * 256) {
        v4f v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = NT_LD ? __builtin_nontemporal_load(&src[i + u * 256]) : src[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { if (NT_ST) __builtin_nontemporal_store(v[u], &dst[i + u * 256]); else dst[i + u * 256] = v[u]; }
    }
    for (; i < hi; i += 256) dst[i] = src[i];
}
template <int UNROLL>
__global__ void __launch_bounds__(256) read_only(const v4f* __restrict__ src, float* __restrict__ out, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        v4f v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}
template <int UNROLL>
__global__ void __launch_bounds__(256) write_only(v4f* __restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    v4f z = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) dst[i] = z;
}

// K arrays in, K arrays out, one float4 per thread and array (a structure-of-arrays streaming kernel like preprocess_fwd /
// preprocess_bwd, which read ≈ 6 and write ≈ 7 separate arrays): does the copy rate survive K concurrent streams?
// `pad4`: extra float4s between consecutive arrays (0: array k starts at k·n — with K a power of two every array starts on
// the same HBM channel / bank phase; a few KB of padding staggers them)
template <int K>
__global__ void __launch_bounds__(256) copy_soa(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n4_per_array, size_t pad4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4_per_array) return;
    v4f v[K];
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = src[(size_t)k * (n4_per_array + pad4) + i];
#pragma unroll
    for (int k = 0; k < K; k++) dst[(size_t)k * (n4_per_array + pad4) + i] = v[k];
}
// the same bytes with 4-byte accesses at a 12-byte stride (three scalar loads per "xyz" record, as means3D is read)
__global__ void __launch_bounds__(256) copy_xyz(const float* __restrict__ src, float* __restrict__ dst, size_t n3) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n3) return;
    const float a = src[3 * i], b = src[3 * i + 1], c = src[3 * i + 2];
    dst[3 * i] = a; dst[3 * i + 1] = b; dst[3 * i + 2] = c;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <class F> float best_ms(F f, int reps = 6) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int r = 0; r < reps + 1; r++) {
        CK(hipEventRecord(a, 0)); f(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r > 0 && ms < best) best = ms;
    }
    return best;
}
int main() {
    for (size_t mb : {256, 1024}) {
        const size_t bytes = mb << 20, n4 = bytes / 16;
        v4f *a, *b; float* o;
        CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 64));
        CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
        printf("== %zu MB per buffer\n", mb);
        auto rep = [&](const char* name, float ms, double moved) { printf("  %-52s %8.3f ms  %7.1f GB/s\n", name, ms, moved / (ms * 1e-3) / 1e9); };
        rep("hipMemcpyDtoD", best_ms([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }), 2.0 * bytes);
        for (int wg : {4, 8, 16, 32, 64}) {
            const int g = 256 * wg; char nm[96];
            snprintf(nm, sizeof nm, "stride u4 plain        %2d WG/CU", wg);
            rep(nm, best_ms([&] { hipLaunchKernelGGL((copy_stride<4, false, false>), dim3(g), dim3(256), 0, 0, a, b, n4); }), 2.0 * bytes);
            snprintf(nm, sizeof nm, "stride u4 nt-store     %2d WG/CU", wg);
            rep(nm, best_ms([&] { hipLaunchKernelGGL((copy_stride<4, true, false>), dim3(g), dim3(256), 0, 0, a, b, n4); }), 2.0 * bytes);
            snprintf(nm, sizeof nm, "stride u4 nt-both      %2d WG/CU", wg);
            rep(nm, best_ms([&] { hipLaunchKernelGGL((copy_stride<4, true, true>), dim3(g), dim3(256), 0, 0, a, b, n4); }), 2.0 * bytes);
            snprintf(nm, sizeof nm, "stride u1 plain        %2d WG/CU", wg);
            rep(nm, best_ms([&] { hipLaunchKernelGGL((copy_stride<1, false, false>), dim3(g), dim3(256), 0, 0, a, b, n4); }), 2.0 * bytes);
            snprintf(nm, sizeof nm, "stride u8 plain        %2d WG/CU", wg);
            rep(nm, best_ms([&] { hipLaunchKernelGGL((copy_stride<8, false, false>), dim3(g), dim3(256), 0, 0, a, b, n4); }), 2.0 * bytes);
            const size_t per = ((n4 + g - 1) / g + 255) / 256 * 256;
            snprintf(nm, sizeof nm, "chunk  u4 plain        %2d WG/CU", wg);
            rep(nm, best_ms([&] { hipLaunchKernelGGL((copy_chunk<4, false, false>), dim3(g), dim3(256), 0, 0, a, b, n4, per); }), 2.0 * bytes);
            snprintf(nm, sizeof nm, "chunk  u4 nt-store     %2d WG/CU", wg);
            rep(nm, best_ms([&] { hipLaunchKernelGGL((copy_chunk<4, true, false>), dim3(g), dim3(256), 0, 0, a, b, n4, per); }), 2.0 * bytes);
        }
        // one workgroup per 4·256 float4 (no loop): the shape of a "one thread per item" streaming kernel
        {
            const int g = (int)((n4 + 1023) / 1024);
            rep("chunk u4 plain, 1 trip per WG", best_ms([&] { hipLaunchKernelGGL((copy_chunk<4, false, false>), dim3(g), dim3(256), 0, 0, a, b, n4, (size_t)1024); }), 2.0 * bytes);
            const int g1 = (int)((n4 + 255) / 256);
            rep("chunk u1 plain, 1 float4 per thread", best_ms([&] { hipLaunchKernelGGL((copy_chunk<1, false, false>), dim3(g1), dim3(256), 0, 0, a, b, n4, (size_t)256); }), 2.0 * bytes);
        }
        {
            auto soa = [&](auto kern, int K, const char* nm) {
                for (size_t pad4 : {(size_t)0, (size_t)(4096 + 64) / 16 * 13, (size_t)1234567 / 16 * 16}) {
                    const size_t per = ((n4 - pad4 * K) / K) & ~(size_t)255;
                    const int g = (int)(per / 256);
                    char nm2[128]; snprintf(nm2, sizeof nm2, "%s pad %zu B", nm, pad4 * 16);
                    rep(nm2, best_ms([&] { hipLaunchKernelGGL(kern, dim3(g), dim3(256), 0, 0, a, b, per, pad4); }), 2.0 * per * K * 16);
                }
            };
            soa(copy_soa<1>, 1, "SoA  1+1");
            soa(copy_soa<2>, 2, "SoA  2+2");
            soa(copy_soa<4>, 4, "SoA  4+4");
            soa(copy_soa<8>, 8, "SoA  8+8");
            soa(copy_soa<12>, 12, "SoA 12+12");
            const size_t n3 = bytes / 12;
            rep("xyz: 3 scalar loads/stores at 12-B stride", best_ms([&] { hipLaunchKernelGGL(copy_xyz, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, 0, (const float*)a, (float*)b, n3); }), 2.0 * n3 * 12);
        }
        rep("read only  u4 16 WG/CU", best_ms([&] { hipLaunchKernelGGL((read_only<4>), dim3(256 * 16), dim3(256), 0, 0, a, o, n4); }), 1.0 * bytes);
        rep("read only  u8 16 WG/CU", best_ms([&] { hipLaunchKernelGGL((read_only<8>), dim3(256 * 16), dim3(256), 0, 0, a, o, n4); }), 1.0 * bytes);
        rep("write only    16 WG/CU", best_ms([&] { hipLaunchKernelGGL((write_only<1>), dim3(256 * 16), dim3(256), 0, 0, b, n4); }), 1.0 * bytes);
        CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(o));
    }
    return 0;
}
