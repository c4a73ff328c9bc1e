// chol_small_kernel in isolation: includes the production kernel source, factors B synthetic
// items (value geometry: n0 main points + a few aux rows; gradient geometry: aux rows [I ; y']),
// checks L and W against a host factorisation, reports the per-launch time, and — with the kernel
// instantiated on a stamping probe — where every wave's time goes, step by step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 small_bench.hip -o small_bench
//   ./small_bench [n_real] [gradient 0/1] [B]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../nowcastautogp_amd/csrc/ngp_kernels.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ unsigned long long *g_stamps;   // [sweep 0..3][wave 0..7][160]

struct SmallStamp {
    unsigned long long *base = nullptr;
    int on = 0, wv = 0;
    __device__ __forceinline__ void begin(int item, int wave, int lane) {
        on = (item == 0 && lane == 0);
        wv = wave;
        base = g_stamps + wave * 160;
    }
    __device__ __forceinline__ void sweep(int si) { base = g_stamps + (si * 8 + wv) * 160; }
    __device__ __forceinline__ void mark(int slot) {
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        if (on && slot < 160) base[slot] = t;
    }
    template <class T> __device__ __forceinline__ void mark_after(int, T) {}
    __device__ __forceinline__ void drain() {}
    __device__ __forceinline__ void emit_diag(int, int, int) {}
    __device__ __forceinline__ void emit_col(int, int, int, int, int, int, bool) {}
};

int main(int argc, char **argv) {
    using namespace ngp;
    const int n_real = argc > 1 ? atoi(argv[1]) : 208;
    const int grad = argc > 2 ? atoi(argv[2]) : 0;
    const int B = argc > 3 ? atoi(argv[3]) : 24;
    JobGeom g{};
    g.B = B;
    g.n0 = grad ? (n_real + NB - 1) / NB * NB : n_real / NB * NB;
    const int tail = grad ? 0 : n_real - g.n0;
    g.nb0 = g.n0 / NB;
    g.ld = g.n0;
    g.n_real = grad ? n_real : g.n0;
    g.aux_identity = grad;
    g.naux = grad ? g.n0 + 1 : tail + 1;
    g.naux_pad = grad ? g.n0 + NB : (g.naux + NB - 1) / NB * NB;
    g.item_stride = (int64_t)(g.n0 + g.naux_pad) * g.n0;
    g.D = 1;
    g.short_series = 1;
    SmallPlan pl;
    if (!small_plan(g, &pl)) { printf("geometry not eligible\n"); return 1; }
    printf("n_real %d  n0 %d  gradient %d  B %d: nbe %d, %d sweep(s), %d panel blocks, %d B of LDS\n", n_real, g.n0,
           grad, B, pl.nbe, pl.nsweeps, pl.npanel, small_lds_bytes(pl));
    const int n0 = g.n0, nr = g.n_real;
    // K = S S' (S lower, strong diagonal) on the data rows, identity on the padding; aux rows random
    std::vector<double> h((size_t)g.item_stride, 0.0), Sx((size_t)n0 * n0, 0.0);
    unsigned long long sd = 12345;
    auto rnd = [&] { sd = sd * 6364136223846793005ULL + 1442695040888963407ULL; return ((sd >> 33) & 0xFFFFFF) / double(0x1000000) - 0.5; };
    for (int r = 0; r < n0; ++r)
        for (int c = 0; c <= r; ++c)
            Sx[(size_t)r * n0 + c] = (r >= nr || c >= nr) ? (r == c ? 1.0 : 0.0) : (r == c ? 2.0 + rnd() : 0.2 * rnd());
    for (int r = 0; r < n0; ++r)
        for (int c = 0; c < n0; ++c) {
            double v = 0.0;
            for (int k = 0; k <= std::min(r, c); ++k) v += Sx[(size_t)r * n0 + k] * Sx[(size_t)c * n0 + k];
            if (c / NB <= r / NB) h[(size_t)r * n0 + c] = v;     // lower 64-tiles (full diagonal tiles)
        }
    std::vector<double> X((size_t)g.naux * n0, 0.0);
    for (int a = 0; a < g.naux; ++a)
        for (int c = 0; c < n0; ++c) {
            double v;
            if (grad) v = a < n0 ? (a == c ? 1.0 : 0.0) : (c < nr ? rnd() : 0.0);
            else v = rnd();
            X[(size_t)a * n0 + c] = v;
            // gradient jobs: the identity rows are NOT in the slab (the kernel synthesises them);
            // poison them so that a block the kernel forgets to write shows
            h[(size_t)(n0 + a) * n0 + c] = (grad && a < n0) ? ((a / NB == c / NB + 1) ? 0.0 : 777.0) : v;
        }
    double *dL, *dsrc, *dlogdet, *dy0; int *dinfo; unsigned long long *dst;
    const size_t bytes = (size_t)g.item_stride * 8;
    CK(hipMalloc(&dL, bytes * B)); CK(hipMalloc(&dsrc, bytes)); CK(hipMalloc(&dlogdet, 8 * B)); CK(hipMalloc(&dinfo, 4 * B));
    CK(hipMalloc(&dst, 8 * 4 * 8 * 160)); CK(hipMemset(dst, 0, 8 * 4 * 8 * 160));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &dst, sizeof(dst)));
    CK(hipMemcpy(dsrc, h.data(), bytes, hipMemcpyHostToDevice));
    // gradient jobs: the kernel takes y' from the observations, not from the slab
    std::vector<double> hy((size_t)n0, 0.0);
    if (grad) for (int c = 0; c < n0; ++c) hy[(size_t)c] = X[(size_t)n0 * n0 + c];
    CK(hipMalloc(&dy0, 8 * (size_t)n0)); CK(hipMemcpy(dy0, hy.data(), 8 * (size_t)n0, hipMemcpyHostToDevice));
    g.y_shared = 1;
    ChunkPtrs p{};
    p.L = dL; p.logdet = dlogdet; p.info = dinfo; p.y0 = dy0;
    CK(hipFuncSetAttribute((const void *)chol_small_kernel<SmallStamp>, hipFuncAttributeMaxDynamicSharedMemorySize,
                           SM_LDS_FIXED + SM_MAX_PANEL * 2048));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int stamped = 0; stamped < 2; ++stamped) {
        double tot = 0.0; const int reps = 8;
        for (int rep = 0; rep < reps; ++rep) {
            for (int b = 0; b < B; ++b) CK(hipMemcpyAsync(dL + (size_t)b * g.item_stride, dsrc, bytes, hipMemcpyDeviceToDevice, 0));
            CK(hipMemsetAsync(dlogdet, 0, 8 * B, 0)); CK(hipMemsetAsync(dinfo, 0, 4 * B, 0));
            CK(hipEventRecord(e0, 0));
            if (stamped) hipLaunchKernelGGL(chol_small_kernel<SmallStamp>, dim3(B), dim3(SM_THREADS), small_lds_bytes(pl), 0, g, p, pl);
            else launch_chol_small(g, p, B, pl, 0);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 1) tot += ms;
        }
        printf("%s: %.1f us per launch\n", stamped ? "stamped" : "product", tot / 6 * 1e3);
        if (stamped) continue;
        std::vector<double> out((size_t)g.item_stride); double ldv; int info;
        CK(hipMemcpy(out.data(), dL + (size_t)(B - 1) * g.item_stride, bytes, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&ldv, dlogdet + B - 1, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&info, dinfo + B - 1, 4, hipMemcpyDeviceToHost));
        double eL = 0.0, eW = 0.0, ld_ref = 0.0;
        for (int r = 0; r < n0; ++r) {
            if (r < nr) ld_ref += log(Sx[(size_t)r * n0 + r]);
            for (int c = 0; c < n0; ++c)
                if (c / 16 < r / 16 && r < 16 * pl.nbe)      // the 16-blocks below the diagonal, data rows (the
                                                              // diagonal blocks' factors stay in the pivot wave)
                    eL = std::max(eL, fabs(out[(size_t)r * n0 + c] - (c <= r ? Sx[(size_t)r * n0 + c] : 0.0)));
        }
        // W = X S^-T by forward substitution on the host
        for (int a = 0; a < g.naux; ++a) {
            std::vector<double> w((size_t)n0);
            for (int c = 0; c < n0; ++c) {
                double v = X[(size_t)a * n0 + c];
                for (int k = 0; k < c; ++k) v -= w[(size_t)k] * Sx[(size_t)c * n0 + k];
                w[(size_t)c] = v / Sx[(size_t)c * n0 + c];
            }
            // W_I: what grad_kinv_small_kernel reads — data rows, from the row's own 16-block to the
            // end of the data columns; other aux rows: the data columns
            if (grad && a < n0 && a >= 16 * pl.nbe) continue;
            const int c0 = (grad && a < n0) ? a / 16 * 16 : 0;
            for (int c = c0; c < 16 * pl.nbe; ++c) eW = std::max(eW, fabs(out[(size_t)(n0 + a) * n0 + c] - w[(size_t)c]));
        }
        if (pl.nsweeps == 1) eL = 0.0;   // (L itself is not written by a single-sweep launch)
        printf("max |L - S| %.2e   max |W - X S^-T| %.2e   logdet %.12f (host %.12f)   info %d\n", eL, eW, ldv, ld_ref, info);
    }
    std::vector<unsigned long long> st(4 * 8 * 160);
    CK(hipMemcpy(st.data(), dst, 8 * st.size(), hipMemcpyDeviceToHost));
    // 100 MHz stamps: slot 8 (j + 1) + phase.  Pivot wave (wave 0 of a main sweep): +0 solves of j
    // done, block j + 1: +1 updated, +2 factored, +3 inverted, +4 step over.  Workers: +0 step
    // begins, +1 solves done, +2 barrier passed, +3 updates done.
    for (int si = 0; si < pl.nsweeps; ++si) {
        printf("sweep %d (%s)\n", si, pl.sw[si].main ? "main" : "aux only");
        for (int w : {0, 1, 7}) {
            const unsigned long long *t = st.data() + (size_t)(si * 8 + w) * 160;
            printf("  wave %d, us per step (", w);
            const bool piv = pl.sw[si].main && w == 0;
            printf(piv ? "update | factor | inverse | wait" : "solves | barrier | updates | barrier");
            printf("):\n");
            for (int j = 0; j < pl.nbe; ++j) {
                const unsigned long long *s = t + 8 * (j + 1);
                const unsigned long long nxt = t[8 * (j + 2)];
                if (!s[0]) continue;
                if (piv) printf("    j %2d: %5.2f %5.2f %5.2f %5.2f\n", j, (s[1] - s[0]) * 0.01, (s[2] - s[1]) * 0.01,
                                (s[3] - s[2]) * 0.01, (s[4] - s[3]) * 0.01);
                else printf("    j %2d: %5.2f %5.2f %5.2f %5.2f\n", j, (s[1] - s[0]) * 0.01, (s[2] - s[1]) * 0.01,
                            (s[3] - s[2]) * 0.01, nxt > s[3] ? (nxt - s[3]) * 0.01 : 0.0);
            }
        }
    }
    for (int w : {0, 1, 7}) {
        const unsigned long long *t = st.data() + (size_t)w * 160;
        const unsigned long long *tl = st.data() + (size_t)((pl.nsweeps - 1) * 8 + w) * 160;
        printf("wave %d: entry -> sweeps %.2f us, first sweep's prologue %.2f, last sweep's loop end -> kernel end %.2f\n", w,
               (t[151] - t[150]) * 0.01, t[7] > t[151] ? (t[7] - t[151]) * 0.01 : 0.0,
               (t[152] - tl[8 * (pl.nbe + 1)]) * 0.01);
    }
    const unsigned long long *t0 = st.data();
    printf("wave 0: entry -> sweeps %.2f us; whole kernel %.2f us\n", (t0[151] - t0[150]) * 0.01, (t0[152] - t0[150]) * 0.01);
    return 0;
}
