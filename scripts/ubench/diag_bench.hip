// chol_diag_kernel in isolation: includes the production kernel source, factors block j of B
// synthetic items and reports the per-launch time in the latency regime (B = 64) and the
// throughput regime (B = 4096), checking L L' against the input tile.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../nowcastautogp_amd/csrc -I../../include \
//         diag_bench.hip -o diag_bench
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "../../nowcastautogp_amd/csrc/ngp_kernels.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
    using namespace ngp;
    const int j = 2, n0 = 256;            // block column 2 of a 4-block matrix: k-loop of 128
    for (int B : {64, 4096}) {
        JobGeom g{};
        g.B = B; g.n0 = n0; g.nb0 = n0 / NB; g.ld = n0; g.naux = 1; g.naux_pad = NB;
        g.item_stride = (int64_t)(n0 + NB) * n0; g.n_real = n0; g.D = 1;
        // rows of block j: L_j,k (k < 128) random small, diagonal tile K_jj = L L' + S S' with S
        // lower triangular and a strong diagonal, so the result of the factorisation is S
        std::vector<double> h((size_t)g.item_stride, 0.0), S((size_t)NB * NB, 0.0);
        unsigned long long sd = 12345;
        auto rnd = [&] { sd = sd * 6364136223846793005ULL + 1442695040888963407ULL; return ((sd >> 33) & 0xFFFFFF) / double(0x1000000) - 0.5; };
        for (int r = 0; r < NB; ++r)
            for (int k = 0; k < j * NB; ++k) h[(size_t)(j * NB + r) * n0 + k] = 0.3 * rnd();
        for (int r = 0; r < NB; ++r)
            for (int c = 0; c <= r; ++c) S[(size_t)r * NB + c] = (r == c) ? 2.0 + rnd() : 0.2 * rnd();
        for (int r = 0; r < NB; ++r)
            for (int c = 0; c < NB; ++c) {
                double v = 0.0;
                for (int k = 0; k < j * NB; ++k) v += h[(size_t)(j * NB + r) * n0 + k] * h[(size_t)(j * NB + c) * n0 + k];
                for (int k = 0; k < NB; ++k) v += S[(size_t)r * NB + k] * S[(size_t)c * NB + k];
                h[(size_t)(j * NB + r) * n0 + j * NB + c] = v;
            }
        double *dL, *dsrc, *ddinv, *dlogdet; int *dinfo;
        const size_t bytes = (size_t)g.item_stride * 8;
        CK(hipMalloc(&dL, bytes * B)); CK(hipMalloc(&dsrc, bytes));
        CK(hipMalloc(&ddinv, (size_t)B * NB * NB * 8)); CK(hipMalloc(&dlogdet, 8 * B)); CK(hipMalloc(&dinfo, 4 * B));
        CK(hipMemcpy(dsrc, h.data(), bytes, hipMemcpyHostToDevice));
        ChunkPtrs p{};
        p.L = dL; p.dinv = ddinv; p.logdet = dlogdet; p.info = dinfo;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        double tot = 0.0; const int reps = 6;
        for (int rep = 0; rep < reps; ++rep) {
            for (int b = 0; b < B; ++b) CK(hipMemcpyAsync(dL + (size_t)b * g.item_stride, dsrc, bytes, hipMemcpyDeviceToDevice, 0));
            CK(hipMemsetAsync(dlogdet, 0, 8 * B, 0)); CK(hipMemsetAsync(dinfo, 0, 4 * B, 0));
            CK(hipEventRecord(e0, 0));
            launch_chol_diag(g, p, B, j, 0, 0);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) tot += ms;
        }
        std::vector<double> out((size_t)g.item_stride); std::vector<double> dv((size_t)NB * NB); double ld; int info;
        CK(hipMemcpy(out.data(), dL + (size_t)(B - 1) * g.item_stride, bytes, hipMemcpyDeviceToHost));
        CK(hipMemcpy(dv.data(), ddinv + (size_t)(B - 1) * NB * NB, 8 * NB * NB, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&ld, dlogdet + B - 1, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&info, dinfo + B - 1, 4, hipMemcpyDeviceToHost));
        double errL = 0.0, ldref = 0.0, errM = 0.0;
        for (int r = 0; r < NB; ++r) {
            ldref += std::log(S[(size_t)r * NB + r]);
            for (int c = 0; c < NB; ++c) errL = std::fmax(errL, std::fabs(out[(size_t)(j * NB + r) * n0 + j * NB + c] - S[(size_t)r * NB + c]));
        }
        // strip order -> M, check M S = I
        std::vector<double> M((size_t)NB * NB, 0.0);
        for (int e = 0; e < NB * NB; ++e) {
            const int strip = e >> 6, l = e & 63, cb4 = strip >> 2, jt = strip & 3;
            M[(size_t)(4 * cb4 + (l & 3)) * NB + 16 * jt + 4 * ((l >> 2) & 3) + (l >> 4)] = dv[e];
        }
        for (int r = 0; r < NB; ++r)
            for (int c = 0; c < NB; ++c) {
                double v = 0.0;
                for (int k = 0; k < NB; ++k) v += M[(size_t)r * NB + k] * S[(size_t)k * NB + c];
                errM = std::fmax(errM, std::fabs(v - (r == c ? 1.0 : 0.0)));
            }
        printf("B=%5d  chol_diag %.1f us/launch  (%.3f us/item)  errL=%.2e errM=%.2e logdet err=%.2e info=%d\n", B,
               1e3 * tot / (reps - 1), 1e3 * tot / (reps - 1) / B, errL, errM, std::fabs(ld - ldref), info);
        CK(hipFree(dL)); CK(hipFree(dsrc)); CK(hipFree(ddinv)); CK(hipFree(dlogdet)); CK(hipFree(dinfo));
    }
    return 0;
}
