// Standalone microbenchmark: v_mfma_f64_16x16x4_f64 issue rate vs independent accumulators and
// waves per SIMD, plus v_fma_f64 VALU rate.  hipcc --offload-arch=gfx950 -O3 mfma_f64_rate.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(unsigned long long *st, int iters) {
    f64x4 c[NACC];
    double a[NACC], b[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) { c[i] = (f64x4){0, 0, 0, 0}; a[i] = 1.0 + 1e-9 * (threadIdx.x + i); b[i] = 1.0 - 1e-9 * (threadIdx.x + 3 * i); }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += NACC) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[i], c[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("" ::"v"(c[i][0]), "v"(c[i][1]), "v"(c[i][2]), "v"(c[i][3]));
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) { int w = blockIdx.x * 4 + (threadIdx.x >> 6); st[2 * w] = t1 - t0; st[2 * w + 1] = r1 - r0; }
}

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma4(unsigned long long *st, int iters) {
    double c[NACC], a[NACC], b[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) { c[i] = 0; a[i] = 1.0 + 1e-9 * (threadIdx.x + i); b[i] = 1.0 - 1e-9 * (threadIdx.x + 3 * i); }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += NACC) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[i], c[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("" ::"v"(c[i]));
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) { int w = blockIdx.x * 4 + (threadIdx.x >> 6); st[2 * w] = t1 - t0; st[2 * w + 1] = r1 - r0; }
}

template <int NACC>
__global__ __launch_bounds__(256) void k_fma(unsigned long long *st, int iters) {
    double c[NACC];
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-9 * threadIdx.x;
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += NACC) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_fma(c[i], a, b);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("" ::"v"(c[i]));
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) { int w = blockIdx.x * 4 + (threadIdx.x >> 6); st[2 * w] = t1 - t0; st[2 * w + 1] = r1 - r0; }
}

template <class K>
void run(const char *name, K kern, int nacc, int bpc, int iters, double flops_per_inst) {
    int blocks = 256 * bpc, waves = blocks * 4;
    unsigned long long *d;
    hipMalloc(&d, sizeof(unsigned long long) * 2 * waves);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * waves);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    std::vector<double> cyc(waves), ghz(waves);
    for (int w = 0; w < waves; ++w) { cyc[w] = (double)h[2 * w] / iters; ghz[w] = (double)h[2 * w] / (h[2 * w + 1] * 10.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    printf("%-8s nacc=%2d waves/SIMD=%d: %7.1f cyc/inst/wave (=> %6.1f per SIMD slot), clock %.2f GHz, wall %.1f TFLOP/s\n", name, nacc, bpc,
           cyc[waves / 2], cyc[waves / 2] / bpc, ghz[waves / 2], (double)waves * iters * flops_per_inst / (ms * 1e-3) * 1e-12);
    hipFree(d);
}

int main() {
    const int it = 1 << 15;
    for (int bpc : {1, 2, 4, 8}) {
        run("mfma", k_mfma<1>, 1, bpc, it, 2048);
        run("mfma", k_mfma<2>, 2, bpc, it, 2048);
        run("mfma", k_mfma<4>, 4, bpc, it, 2048);
        run("mfma", k_mfma<8>, 8, bpc, it, 2048);
        run("mfma", k_mfma<16>, 16, bpc, it, 2048);
    }
    for (int bpc : {1, 2, 4, 8}) {
        run("mfma4x4", k_mfma4<4>, 4, bpc, it * 4, 512);
        run("mfma4x4", k_mfma4<16>, 16, bpc, it * 4, 512);
    }
    for (int bpc : {1, 2, 4, 8}) {
        run("fma64", k_fma<8>, 8, bpc, it * 16, 128);
        run("fma64", k_fma<16>, 16, bpc, it * 16, 128);
    }
    return 0;
}
