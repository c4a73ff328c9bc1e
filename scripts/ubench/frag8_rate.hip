// Inner-loop rate of the 4x4x4 composite (MFMA + DPP rotations) from registers, no memory.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
template <int CTRL, int V> __device__ __forceinline__ double rot(double s) {
    int lo, hi;
    if (V == 0) {
        lo = __builtin_amdgcn_update_dpp(__double2loint(s), __double2loint(s), CTRL, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(__double2hiint(s), __double2hiint(s), CTRL, 0xF, 0xF, false);
    } else {
        lo = __builtin_amdgcn_mov_dpp(__double2loint(s), CTRL, 0xF, 0xF, true);
        hi = __builtin_amdgcn_mov_dpp(__double2hiint(s), CTRL, 0xF, 0xF, true);
    }
    return __hiloint2double(hi, lo);
}
// V: 0 update_dpp, 1 mov_dpp, 2 no rotation at all (MFMA only, wrong math, rate reference)
template <int V>
__global__ __launch_bounds__(256, 2) void k(const double *in, double *out, int iters) {
    double acc[4][4][4];
    f64x2 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = *(const f64x2 *)(in + threadIdx.x * 2 + i * 512); b[i] = *(const f64x2 *)(in + 2048 + threadIdx.x * 2 + i * 512); }
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[x][y][r] = 0;
    for (int t = 0; t < iters; ++t) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                double bv = h ? b[it].y : b[it].x;
                asm volatile("" : "+v"(bv));
                double r0 = bv, r1, r2, r3;
                if (V == 2) { r1 = bv; r2 = bv; r3 = bv; }
                else { r1 = rot<0x12C, V>(bv); r2 = rot<0x128, V>(bv); r3 = rot<0x124, V>(bv); }
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {
                    double av = h ? a[jt].y : a[jt].x;
                    acc[jt][it][0] = mfma4(av, r0, acc[jt][it][0]);
                    acc[jt][it][1] = mfma4(av, r1, acc[jt][it][1]);
                    acc[jt][it][2] = mfma4(av, r2, acc[jt][it][2]);
                    acc[jt][it][3] = mfma4(av, r3, acc[jt][it][3]);
                }
            }
    }
    double s = 0;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[x][y][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int V> void run(const char *name, int bpc) {
    double *in, *out;
    hipMalloc(&in, 8 * 8192); hipMemset(in, 0, 8 * 8192);
    int blocks = 256 * bpc, iters = 4000;
    hipMalloc(&out, 8 * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-12s waves/SIMD=%d: %.1f TFLOP/s\n", name, bpc, (double)blocks * 4 * iters * 128.0 * 512 / (ms * 1e-3) * 1e-12);
    hipFree(in); hipFree(out);
}
int main() {
    for (int bpc : {1, 2}) { run<0>("update_dpp", bpc); run<1>("mov_dpp", bpc); run<2>("mfma_only", bpc); }
}
