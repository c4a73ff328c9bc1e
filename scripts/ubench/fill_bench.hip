// fill_lattice_kernel in isolation (includes the production kernel source): n0 = 2048 lattice
// times, B items with a 7-node tree (Plus(Times(Linear, Periodic), ChangePoint(GammaExp, SqExp))),
// reports GB/s written.  -DFILL_MODE=1: stores only (no evaluation), 2: evaluation only.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../nowcastautogp_amd/csrc/ngp_kernels.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    using namespace ngp;
    const int n0 = 2048, B = 512, naux = 12;
    JobGeom g{};
    g.B = B; g.n0 = n0; g.nb0 = n0 / NB; g.ld = n0; g.da = 1; g.m = 10; g.naux = naux; g.naux_pad = NB;
    g.item_stride = (int64_t)(n0 + NB) * n0; g.n_real = n0; g.D = 1; g.lattice = 1;
    g.npts = n0 + 11; g.R = g.npts; g.maxstat = 3; g.maxcp = 1; g.h = 1.0 / (n0 - 1);
    DevProgram P{};
    // postfix: LIN PER TIMES GE SE CP PLUS
    const uint8_t ops[7] = {2, 5, 7, 4, 3, 8, 6};
    const double par[] = {0.3, 0.1, 0.5, 0.8, 0.25, 1.1, 0.4, 1.3, 0.9, 0.2, 0.7, 0.5, 0.1};
    const int npar[9] = {0, 1, 3, 2, 3, 3, 0, 0, 2};
    P.n_ops = 7; P.n_params = 13; P.noise = 0.1;
    int po = 0, ns = 0, nc = 0;
    for (int i = 0; i < 7; ++i) {
        P.ops[i] = ops[i]; P.poff[i] = (uint8_t)po; po += npar[ops[i]];
        if (ops[i] >= 3 && ops[i] <= 5) P.slot[i] = (uint8_t)ns++;
        if (ops[i] == 8) P.slot[i] = (uint8_t)nc++;
    }
    P.first[2] = 0; P.first[5] = 3; P.first[6] = 2;
    std::memcpy(P.params, par, sizeof(par));
    std::vector<DevProgram> hp(B, P);
    std::vector<double> t(g.npts), y(n0, 0.5);
    std::vector<int> q(g.npts);
    for (int i = 0; i < g.npts; ++i) { t[i] = i * g.h; q[i] = i; }
    DevProgram *dp; double *dt, *dy, *dL, *dtab, *dsig; int *dq;
    CK(hipMalloc(&dp, sizeof(DevProgram) * B)); CK(hipMalloc(&dt, 8 * g.npts)); CK(hipMalloc(&dy, 8 * n0));
    CK(hipMalloc(&dq, 4 * g.npts)); CK(hipMalloc(&dL, (size_t)g.item_stride * 8 * B));
    CK(hipMalloc(&dtab, 8ull * B * g.maxstat * g.R)); CK(hipMalloc(&dsig, 8ull * B * g.maxcp * g.npts));
    CK(hipMemcpy(dp, hp.data(), sizeof(DevProgram) * B, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, t.data(), 8 * g.npts, hipMemcpyHostToDevice));
    CK(hipMemcpy(dy, y.data(), 8 * n0, hipMemcpyHostToDevice));
    CK(hipMemcpy(dq, q.data(), 4 * g.npts, hipMemcpyHostToDevice));
    ChunkPtrs p{};
    p.L = dL; p.progs = dp; p.t0 = dt; p.taux = dt + n0; p.y0 = dy; p.tab = dtab; p.sig = dsig; p.qpts = dq;
    g.y_shared = 1;
    DevSpec sp{0, 0, 0, 0, 1e-5};
    launch_tables(g, p, B, sp, 0);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double tot = 0; const int reps = 5;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        launch_fill(g, p, B, sp, 0);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) tot += ms;
    }
    const double bytes = 8.0 * B * ((double)n0 * (n0 + NB) / 2.0 + (double)NB * n0);
    std::vector<double> h(4);
    CK(hipMemcpy(h.data(), dL + (size_t)(B - 1) * g.item_stride + 700 * (size_t)n0 + 300, 32, hipMemcpyDeviceToHost));
    printf("fill: %.3f ms per launch, %.0f GB/s written  (sample %.6f %.6f)\n", tot / (reps - 1), bytes / (tot / (reps - 1) * 1e-3) * 1e-9, h[0], h[1]);
    return 0;
}
