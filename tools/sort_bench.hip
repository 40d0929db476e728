// sort_bench.hip — standalone micro-benchmark + self-check of ggr::radix_sort_pairs (dev tool).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ggrt_official_amd/csrc tools/sort_bench.hip \
//        ggrt_official_amd/csrc/binning.hip -o /tmp/sort_bench
#include "ggr_common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
    size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 10763014;
    int nbits = argc > 2 ? atoi(argv[2]) : 13;
    int mode = argc > 3 ? atoi(argv[3]) : 0;  // 0: structured tile ids, 1: random, 2: float depth bits
    std::vector<uint32_t> hk(n), hv(n);
    uint32_t seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    const uint32_t gx = 120, gy = 68;
    size_t i = 0;
    while (i < n) {
        if (mode == 0) {  // a Gaussian's rect: w×h tiles row-major
            uint32_t w = 1 + rnd() % 6, h = 1 + rnd() % 6, x0 = rnd() % (gx - w + 1), y0 = rnd() % (gy - h + 1);
            for (uint32_t y = 0; y < h && i < n; y++)
                for (uint32_t x = 0; x < w && i < n; x++) { hk[i] = (y0 + y) * gx + x0 + x; hv[i] = (uint32_t)i; i++; }
        } else if (mode == 1) { hk[i] = rnd() & ((nbits >= 32 ? 0 : (1u << nbits)) - 1u); hv[i] = (uint32_t)i; i++; }
        else { float z = 1.5f + (rnd() % 100000) * 0.000485f; uint32_t b; memcpy(&b, &z, 4); hk[i] = b; hv[i] = (uint32_t)i; i++; }
    }
    uint32_t *ka, *kb, *va, *vb, *hist;
    CK(hipMalloc(&ka, n * 4)); CK(hipMalloc(&kb, n * 4)); CK(hipMalloc(&va, n * 4)); CK(hipMalloc(&vb, n * 4));
    CK(hipMalloc(&hist, ggr_sort_hist_words(n) * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint32_t *ko, *vo;
    float best = 1e9f;
    for (int it = 0; it < 8; it++) {
        CK(hipMemcpy(ka, hk.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(va, hv.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipEventRecord(e0, s));
        ggr::radix_sort_pairs(ka, kb, va, vb, hist, n, nbits, &ko, &vo, s);
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    std::vector<uint32_t> rk(n), rv(n);
    CK(hipMemcpy(rk.data(), ko, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(rv.data(), vo, n * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> idx(n); std::iota(idx.begin(), idx.end(), 0u);
    const uint32_t mask = nbits >= 32 ? 0xFFFFFFFFu : ((1u << (8 * ((nbits + 7) / 8))) - 1u);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return (hk[a] & mask) < (hk[b] & mask); });
    size_t bad = 0;
    for (size_t j = 0; j < n; j++) if (rk[j] != hk[idx[j]] || rv[j] != hv[idx[j]]) bad++;
    printf("n=%zu nbits=%d mode=%d  best %.3f ms  (%.1f GB/s at 16 B/key/pass x %d passes)  mismatches=%zu\n", n, nbits, mode, best,
           n * 16.0 * ((nbits + 7) / 8) / best / 1e6, (nbits + 7) / 8, bad);
    return bad != 0;
}
