// Probe the lane maps of v_mfma_f64_4x4x4_4b_f64 with one-hot operands.
// For every (la, lb): A = onehot(lane == la), B = onehot(lane == lb) -> D (64 lanes).
// Output: [cfg][la][lb][lane] doubles; cfg 0: cbsz=0; cfg 1..4: cbsz=2, abid=0..3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CBSZ, int ABID>
__global__ void probe(double *out) {
    const int l = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
            out[((long)la * 64 + lb) * 64 + l] = d;
        }
}
int main(int argc, char **argv) {
    const size_t N = 64 * 64 * 64;
    double *d;
    hipMalloc(&d, sizeof(double) * N * 5);
    hipLaunchKernelGGL((probe<0, 0>), dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL((probe<2, 0>), dim3(1), dim3(64), 0, 0, d + N);
    hipLaunchKernelGGL((probe<2, 1>), dim3(1), dim3(64), 0, 0, d + 2 * N);
    hipLaunchKernelGGL((probe<2, 2>), dim3(1), dim3(64), 0, 0, d + 3 * N);
    hipLaunchKernelGGL((probe<2, 3>), dim3(1), dim3(64), 0, 0, d + 4 * N);
    std::vector<double> h(N * 5);
    hipMemcpy(h.data(), d, sizeof(double) * N * 5, hipMemcpyDeviceToHost);
    FILE *f = fopen(argc > 1 ? argv[1] : "mfma4_probe.bin", "wb");
    fwrite(h.data(), sizeof(double), h.size(), f);
    fclose(f);
    printf("wrote %zu doubles\n", h.size());
    return 0;
}
