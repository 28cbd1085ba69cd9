// Experiment: where does the fixed ~20 us of a 64x64-tile fp32 MFMA GEMM launch go?
// Pure MFMA blocks (no loads), same grid/occupancy as the real kernel; records per-block wall clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <map>
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 2) void k(int iters, float* out, unsigned long long* t0, unsigned long long* t1,
                                              unsigned* hw, unsigned long long* cyc) {
    extern __shared__ float lds[];
    unsigned long long s = wall_clock64();
    unsigned long long c0 = clock64();
    f16v acc = {0};
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        a += 1e-6f;
    }
    float r = 0; for (int j = 0; j < 16; ++j) r += acc[j];
    if (threadIdx.x == 0) lds[0] = r;
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
    unsigned long long e = wall_clock64();
    if (threadIdx.x == 0) {
        t0[blockIdx.x] = s; t1[blockIdx.x] = e; cyc[blockIdx.x] = clock64() - c0;
        unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        hw[blockIdx.x] = (id & 0xffffff) | (xcc << 24);
    }
}
int main(int argc, char** argv) {
    int nblk = argc > 1 ? atoi(argv[1]) : 1264; int lds = argc > 2 ? atoi(argv[2]) : 32768;
    float* out; unsigned long long *t0, *t1, *cyc; unsigned* hw;
    hipMalloc(&out, (size_t)nblk * 256 * 4); hipMalloc(&t0, nblk * 8); hipMalloc(&t1, nblk * 8); hipMalloc(&cyc, nblk * 8); hipMalloc(&hw, nblk * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int iters : {1, 4, 8, 16, 23, 46, 92}) {
        for (int w = 0; w < 3; ++w) k<<<nblk, 256, lds>>>(iters, out, t0, t1, hw, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0); for (int w = 0; w < 20; ++w) k<<<nblk, 256, lds>>>(iters, out, t0, t1, hw, cyc);
        hipEventRecord(e1); hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> a(nblk), b(nblk), c(nblk); std::vector<unsigned> h(nblk);
        hipMemcpy(a.data(), t0, nblk * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), t1, nblk * 8, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), cyc, nblk * 8, hipMemcpyDeviceToHost); hipMemcpy(h.data(), hw, nblk * 4, hipMemcpyDeviceToHost);
        unsigned long long mn = *std::min_element(a.begin(), a.end()), mx = *std::max_element(b.begin(), b.end());
        std::vector<double> st, en, du, cy;
        for (int i = 0; i < nblk; ++i) { st.push_back((a[i] - mn) * 0.01); en.push_back((b[i] - mn) * 0.01); du.push_back((b[i] - a[i]) * 0.01); cy.push_back((double)c[i]); }
        auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
        std::map<unsigned, int> percu; for (auto x : h) percu[x & 0xff000fff & ~0xfu /*drop wave id*/]++;  // coarse
        int mxcu = 0, mncu = 1 << 30; for (auto& kv : percu) { mxcu = std::max(mxcu, kv.second); mncu = std::min(mncu, kv.second); }
        printf("iters %3d: event %.1f us/launch | span %.1f us | start p50 %.1f p90 %.1f max %.1f | end p10 %.1f p50 %.1f max %.1f | dur p10 %.1f p50 %.1f p90 %.1f | clk(p50) %.0f cyc -> %.2f GHz | ids %zu blk/id %d..%d | ideal %.1f us\n",
               iters, ms / 20 * 1e3, (mx - mn) * 0.01, pct(st, .5), pct(st, .9), pct(st, 1), pct(en, .1), pct(en, .5), pct(en, 1),
               pct(du, .1), pct(du, .5), pct(du, .9), pct(cy, .5), pct(cy, .5) / (pct(du, .5) * 1e3), percu.size(), mncu, mxcu,
               (double)nblk * 4 * iters * 16 * 64 / (256.0 * 4) / 2.4e3);
    }
    return 0;
}
