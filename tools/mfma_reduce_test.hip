// mfma_reduce_test.hip — checks the operand layout assumptions behind the MFMA-based wave reductions of
// blend_bwd (v_mfma_f32_16x16x4_f32 used as "sum over the 64 lanes with polynomial pixel weights"):
//   stage 1:  D[i][j]  = Σ_k  V[i+16k] · ψ_j(k)         (A = the per-lane values, B = constants)
//   stage 2:  D2[a][j] = Σ_i  φ_a(i)   · D[i][j]        (A = constants, B = D's registers, 4 MFMAs)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_reduce_test tools/mfma_reduce_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k(const float* __restrict__ in, float* __restrict__ out) {
    const int lane = threadIdx.x;
    const int j = lane & 15, g = lane >> 4;
    const float v0 = in[lane], v1 = in[64 + lane];
    // B constants: vector 0 → columns 0..2 with ψ = 1, (2k-3), (2k-3)²; vector 1 → column 3 with ψ = 1
    const float ps = 2.f * g - 3.f;
    const float b0 = j == 0 ? 1.f : j == 1 ? ps : j == 2 ? ps * ps : 0.f;
    const float b1 = j == 3 ? 1.f : 0.f;
    v4f d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(v0, b0, d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(v1, b1, d, 0, 0, 0);
    // stage 2: A2_r at lane (a = lane & 15, k' = lane >> 4) = φ_a(i = 4k' + r)
    v4f d2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int a = lane & 15, i = 4 * g + r;
        const float fx = (float)(i & 7) - 3.5f, fy = (float)(i >> 3) - 0.5f;
        const float phi = a == 0 ? 1.f : a == 1 ? fx : a == 2 ? fy : a == 3 ? fx * fx : a == 4 ? fx * fy : a == 5 ? fy * fy : 0.f;
        d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(phi, d[r], d2, 0, 0, 0);
    }
    // D2[a][j] sits in lane (j, a/4), register a%4
#pragma unroll
    for (int r = 0; r < 4; r++) out[(4 * g + r) * 16 + j] = d2[r];
}

int main() {
    std::vector<float> h(128);
    for (int i = 0; i < 128; i++) h[i] = sinf(0.37f * i) + 0.01f * i;
    float *din, *dout;
    hipMalloc(&din, 128 * 4); hipMalloc(&dout, 256 * 4);
    hipMemcpy(din, h.data(), 128 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
    std::vector<float> o(256);
    hipMemcpy(o.data(), dout, 256 * 4, hipMemcpyDeviceToHost);
    // expected: pixel p: x' = (p&7)-3.5, y' = (p>>3)-3.5
    double M1 = 0, Mx = 0, My = 0, Mxx = 0, Mxy = 0, Myy = 0, S1 = 0;
    for (int p = 0; p < 64; p++) {
        const double x = (p & 7) - 3.5, y = (p >> 3) - 3.5, v = h[p];
        M1 += v; Mx += v * x; My += v * y; Mxx += v * x * x; Mxy += v * x * y; Myy += v * y * y; S1 += h[64 + p];
    }
    auto D2 = [&](int a, int j) { return (double)o[a * 16 + j]; };
    const double g1 = D2(0, 0), gx = D2(1, 0), gy = D2(2, 0) + D2(0, 1), gxx = D2(3, 0), gxy = D2(4, 0) + D2(1, 1),
                 gyy = D2(5, 0) + 2 * D2(2, 1) + D2(0, 2), gs = D2(0, 3);
    printf("M1  %.6f %.6f\nMx  %.6f %.6f\nMy  %.6f %.6f\nMxx %.6f %.6f\nMxy %.6f %.6f\nMyy %.6f %.6f\nS1  %.6f %.6f\n", M1, g1, Mx,
           gx, My, gy, Mxx, gxx, Mxy, gxy, Myy, gyy, S1, gs);
    const double err = fabs(M1 - g1) + fabs(Mx - gx) + fabs(My - gy) + fabs(Mxx - gxx) + fabs(Mxy - gxy) + fabs(Myy - gyy) + fabs(S1 - gs);
    printf("%s (abs err sum %.3g)\n", err < 1e-3 ? "LAYOUT OK" : "LAYOUT MISMATCH", err);
    return err < 1e-3 ? 0 : 1;
}
