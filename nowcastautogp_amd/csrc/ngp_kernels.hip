// ngp_kernels.hip — hand-written gfx950 (CDNA4) kernels of the GP hot path.
//
// Data layout in HBM (per item = one particle's covariance kernel), row-major fp64:
//   factor storage  [(n0 + naux_pad) x n0]:
//       rows [0, n0)            K(t0,t0)+(noise+jitter)I, lower 64x64 blocks only; overwritten in
//                               place by its Cholesky factor L, one 64-wide block column per step
//       rows [n0, n0+naux)      "aux rows" X = [k(t_add,t0); k(t_new,t0); y0'] that ride along
//                               and become W = X L^-T (appended points, forecast points, data)
//   Everything downstream (log-marginal likelihoods for every scenario, predictive mean and
//   covariance) is Schur-complement algebra on the small Gram matrix G = W W'.
//
// Kernels (roofline class):
//   fill_kernel       RPN kernel-tree interpreter, one 64x64 tile per workgroup, 512-B row
//                     stores                                   (HBM-write + fp64 transcendental VALU)
//   chol_diag_kernel  C_jj -= L_j L_j' (MFMA), 64x64 Cholesky in LDS, 16x16 diagonal-block
//                     inverses                                              (latency-bound)
//   chol_col_kernel   C_rj -= L_r L_j' over k = 64 j (v_mfma_f64_16x16x4_f64, 64x64 tile per
//                     wave), then the 64-wide triangular solve as MFMA block substitution on the
//                     accumulator tiles without leaving registers    (fp64-MFMA-bound; dominant)
//   gram_kernel       G = W W'                                              (HBM-read-bound, small)
//   epilogue_kernel   dense Schur algebra per item + per-scenario solves     (latency-bound, tiny)
//
// MFMA operand maps used throughout (v_mfma_f64_16x16x4_f64, guide cdna_hip_programming.md §3):
//   A: lane l holds A[m = l&15][k = l>>4]     B: lane l holds B[k = l>>4][n = l&15]
//   D: lane l, register r holds D[m = (l>>4) + 4 r][n = l&15]
// so a D-layout tile is directly the B operand of a following product that sums over its row
// index (register r <-> k-slot), which is what keeps the triangular solve in registers.
#include "ngp_internal.h"

namespace ngp {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f64x4 mfma64(double a, double b, f64x4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------
// kernel-tree interpreter
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void load_program(DevProgram *dst, const DevProgram *src) {
    const unsigned long long *s = reinterpret_cast<const unsigned long long *>(src);
    unsigned long long *d = reinterpret_cast<unsigned long long *>(dst);
    for (unsigned i = threadIdx.x; i < sizeof(DevProgram) / 8; i += blockDim.x) d[i] = s[i];
}

__device__ __forceinline__ double cp_sigma(int form, double x, double loc, double scale) {
    const double u = form ? (x - loc) / scale : (loc - x) / scale;
    return 0.5 * (1.0 + tanh(u));
}

// Evaluate k(t1, t2) for the program held in LDS.  The evaluation stack is a register shift
// file (no runtime-indexed arrays, which would go to scratch); ops are workgroup-uniform.
__device__ double keval(const DevProgram &P, const DevSpec &sp, double t1, double t2) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
    int pi = 0;
    const int nops = P.n_ops;
    for (int i = 0; i < nops; ++i) {
        const int op = __builtin_amdgcn_readfirstlane((int)P.ops[i]);
        if (op < NGP_OP_PLUS) {
            double v;
            if (op == NGP_OP_CONSTANT) {
                v = P.params[pi];
                pi += 1;
            } else if (op == NGP_OP_LINEAR) {
                const double c = P.params[pi];
                v = P.params[pi + 1] + P.params[pi + 2] * (t1 - c) * (t2 - c);
                pi += 3;
            } else if (op == NGP_OP_SQEXP) {
                const double d = t1 - t2, l = P.params[pi];
                const double den = sp.se_form ? l : l * l;
                v = P.params[pi + 1] * exp(-0.5 * d * d / den);
                pi += 2;
            } else if (op == NGP_OP_GAMMAEXP) {
                const double d = fabs(t1 - t2);
                v = P.params[pi + 2] * exp(-pow(d / P.params[pi], P.params[pi + 1]));
                pi += 3;
            } else {  // NGP_OP_PERIODIC
                const double d = fabs(t1 - t2), l = P.params[pi];
                const double sn = sin(M_PI * d / P.params[pi + 1]);
                const double c = sp.periodic_form ? 2.0 / l : 2.0 / (l * l);
                v = P.params[pi + 2] * exp(-c * sn * sn);
                pi += 3;
            }
            s7 = s6; s6 = s5; s5 = s4; s4 = s3; s3 = s2; s2 = s1; s1 = s0; s0 = v;
        } else {
            double v;
            if (op == NGP_OP_PLUS) {
                v = s1 + s0;
            } else if (op == NGP_OP_TIMES) {
                v = s1 * s0;
            } else {
                const double kl = (op == NGP_OP_CHANGEPOINT) ? s1 : s0;
                const double kr = (op == NGP_OP_CHANGEPOINT) ? s0 : s1;
                const double loc = P.params[pi], sc = P.params[pi + 1];
                const double g1 = cp_sigma(sp.cp_form, t1, loc, sc);
                const double g2 = cp_sigma(sp.cp_form, t2, loc, sc);
                v = g1 * kl * g2 + (1.0 - g1) * kr * (1.0 - g2);
                pi += 2;
            }
            s0 = v; s1 = s2; s2 = s3; s3 = s4; s4 = s5; s5 = s6; s6 = s7;
        }
    }
    return s0;
}

// ---------------------------------------------------------------------------------------
// standalone covariance assembly (ngp_cov_batch; also the K22-style small blocks in tests)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cov_kernel(const DevProgram *progs, const double *t1,
                                                  int n1, const double *t2, int n2, int add_diag,
                                                  double *out, DevSpec sp) {
    __shared__ DevProgram P;
    const int b = blockIdx.y;
    load_program(&P, progs + b);
    __syncthreads();
    const long total = (long)n1 * n2;
    double *o = out + (long)b * total;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int i = (int)(e / n2), j = (int)(e % n2);
        double v = keval(P, sp, t1[i], t2[j]);
        if (add_diag && i == j) v += P.noise + sp.jitter;
        o[e] = v;
    }
}

// ---------------------------------------------------------------------------------------
// fill: K lower blocks + aux rows into the factor storage, one 64x64 tile per workgroup
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fill_kernel(JobGeom g, ChunkPtrs p, int ntri, DevSpec sp) {
    __shared__ DevProgram P;
    const int item = blockIdx.y;
    load_program(&P, p.progs + item);
    __syncthreads();
    const int tile = blockIdx.x;
    int r, c;            // block row / block column
    bool aux = false;
    if (tile < ntri) {   // lower-triangular block (r >= c): tile = r(r+1)/2 + c
        r = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
        while ((r + 1) * (r + 2) / 2 <= tile) ++r;
        while (r * (r + 1) / 2 > tile) --r;
        c = tile - r * (r + 1) / 2;
    } else {
        const int a = tile - ntri;
        r = a / g.nb0;   // aux tile row
        c = a % g.nb0;
        aux = true;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int col = c * NB + tx;
    const double t2 = p.t0[col];
    const double diag = P.noise + sp.jitter;
    double *Lit = p.L + (long)item * g.item_stride;
    const int naux_t = g.da + g.m;
    const double *y0 = p.y0 + (g.y_shared ? 0 : (long)item * g.n0);
    for (int rr = 0; rr < 16; ++rr) {
        const int lr = ty * 16 + rr;
        double v;
        long row;
        if (!aux) {
            row = (long)r * NB + lr;
            v = keval(P, sp, p.t0[row], t2);
            if (row == col) v += diag;
        } else {
            const int ar = r * NB + lr;
            row = (long)g.n0 + ar;
            if (ar < naux_t) v = keval(P, sp, p.taux[ar], t2);
            else if (ar == naux_t) v = y0[col];
            else v = 0.0;
        }
        Lit[row * g.ld + col] = v;
    }
}

// ---------------------------------------------------------------------------------------
// chol_diag: factor the 64x64 diagonal block of block column j
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chol_diag_kernel(JobGeom g, ChunkPtrs p, int j) {
    __shared__ double At[NB][NB + 1];
    __shared__ double Lt[NB][NB + 1];
    __shared__ double logs[NB];
    __shared__ int bad;
    const int item = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long ld = g.ld;
    double *Lit = p.L + (long)item * g.item_stride;
    double *Lj = Lit + (long)j * NB * ld;  // rows of block j
    const int kmax = j * NB;
    const int r16 = lane & 15, q = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;

    for (int e = tid; e < NB * (NB + 1); e += 256) (&Lt[0][0])[e] = 0.0;
    if (tid == 0) bad = 0;

    // ---- C_jj = K_jj - L_j L_j'   (each wave one 32x32 quadrant; the strictly upper one is
    //      never read by the factorisation and is skipped)
    if (!(wr == 0 && wc == 1)) {
        f64x4 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = (f64x4){0, 0, 0, 0};
        const double *pa = Lj + (long)(32 * wr + r16) * ld + 2 * q;
        const double *pb = Lj + (long)(32 * wc + r16) * ld + 2 * q;
        for (int kc = 0; kc < kmax; kc += 16) {
            double a[2][4], b[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f64x2 alo = *reinterpret_cast<const f64x2 *>(pa + (long)u * 16 * ld + kc);
                const f64x2 ahi = *reinterpret_cast<const f64x2 *>(pa + (long)u * 16 * ld + kc + 8);
                const f64x2 blo = *reinterpret_cast<const f64x2 *>(pb + (long)u * 16 * ld + kc);
                const f64x2 bhi = *reinterpret_cast<const f64x2 *>(pb + (long)u * 16 * ld + kc + 8);
                a[u][0] = alo.x; a[u][1] = alo.y; a[u][2] = ahi.x; a[u][3] = ahi.y;
                b[u][0] = blo.x; b[u][1] = blo.y; b[u][2] = bhi.x; b[u][3] = bhi.y;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = mfma64(a[mt][s], b[nt][s], acc[mt][nt]);
        }
        // D layout: register s of acc[mt][nt] is S[M = 32wr+16mt+q+4s][N = 32wc+16nt+r16]
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int M = 32 * wr + 16 * mt + q + 4 * s, N = 32 * wc + 16 * nt + r16;
                    At[M][N] = Lj[(long)M * ld + kmax + N] - acc[mt][nt][s];
                }
    }
    __syncthreads();

    // ---- right-looking Cholesky of the 64x64 tile in LDS; one barrier per column: column k is
    //      read-only during step k (scaled copies go to Lt), the trailing update writes j > k
    const int tx = tid & 15, ty = tid >> 4;
    for (int k = 0; k < NB; ++k) {
        __syncthreads();
        const double akk = At[k][k];
        const double dk = sqrt(akk);
        const double inv = 1.0 / dk;
        if (tid == 0) {
            if (!(akk > 0.0) && bad == 0) bad = k + 1;
            Lt[k][k] = dk;
            logs[k] = log(dk);
        }
        if (tid > k && tid < NB) Lt[tid][k] = At[tid][k] * inv;
        for (int i = k + 1 + ty; i < NB; i += 16) {
            const double lik = At[i][k] * inv;
            for (int jj = k + 1 + tx; jj <= i; jj += 16) At[i][jj] -= lik * (At[jj][k] * inv);
        }
    }
    __syncthreads();

    // ---- inverses of the four 16x16 diagonal blocks (column c of block b per thread)
    if (tid < NB) {
        const int b = tid >> 4, c = tid & 15;
        double x[TB];
#pragma unroll
        for (int i = 0; i < TB; ++i) {
            double sum = (i == c) ? 1.0 : 0.0;
#pragma unroll
            for (int pp = 0; pp < i; ++pp) sum -= Lt[TB * b + i][TB * b + pp] * x[pp];
            x[i] = sum / Lt[TB * b + i][TB * b + i];
        }
        double *dv = p.dinv + ((long)item * (NB / TB) + b) * (TB * TB);
#pragma unroll
        for (int i = 0; i < TB; ++i) dv[i * TB + c] = x[i];
    }
    // ---- write L_jj back (strict upper part zero)
    for (int e = tid; e < NB * NB; e += 256) {
        const int M = e >> 6, N = e & 63;
        Lj[(long)M * ld + kmax + N] = Lt[M][N];
    }
    if (tid == 0) {
        double s = 0.0;
        for (int k = 0; k < NB; ++k) s += logs[k];
        p.logdet[item] += s;
        if (bad && p.info[item] == 0) p.info[item] = kmax + bad;
    }
}

// ---------------------------------------------------------------------------------------
// chol_col: every row tile below the diagonal of block column j (and every aux tile):
//           C_rj -= L_r,0:k L_j,0:k'   then   L_rj = C_rj L_jj^-T
// One 64x64 tile per wave, 4 tiles per workgroup; the transposed tile C' is accumulated so the
// solve can consume the accumulators as MFMA B operands in place.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chol_col_kernel(JobGeom g, ChunkPtrs p, int Bc, int j,
                                                       int groups, int nmain, int ntiles) {
    const int wg = blockIdx.x;
    const int xcd = wg & 7, idx = wg >> 3;   // blocks b and b+8 share an XCD (speed only)
    const int item = (idx / groups) * 8 + xcd;
    const int grp = idx % groups;
    if (item >= Bc) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = grp * 4 + wave;
    if (tile >= ntiles) return;

    const long ld = g.ld;
    double *Lit = p.L + (long)item * g.item_stride;
    const double *Lj = Lit + (long)j * NB * ld;
    const long rowbase = (tile < nmain) ? (long)(j + 1 + tile) * NB
                                        : (long)g.n0 + (long)(tile - nmain) * NB;
    double *Lr = Lit + rowbase * ld;
    const int kmax = j * NB;
    const int r16 = lane & 15, q = lane >> 4;

    f64x4 acc[4][4];  // acc[jt][it]: S'[jj = 16jt + q + 4s][i = 16it + r16]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f64x4){0, 0, 0, 0};

    const double *pa = Lj + (long)r16 * ld + 2 * q;  // A operand: rows of block j (M = jj)
    const double *pb = Lr + (long)r16 * ld + 2 * q;  // B operand: rows of this tile (N = i)
    for (int kc = 0; kc < kmax; kc += 16) {
        double a[4][4], b[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f64x2 alo = *reinterpret_cast<const f64x2 *>(pa + (long)u * 16 * ld + kc);
            const f64x2 ahi = *reinterpret_cast<const f64x2 *>(pa + (long)u * 16 * ld + kc + 8);
            const f64x2 blo = *reinterpret_cast<const f64x2 *>(pb + (long)u * 16 * ld + kc);
            const f64x2 bhi = *reinterpret_cast<const f64x2 *>(pb + (long)u * 16 * ld + kc + 8);
            a[u][0] = alo.x; a[u][1] = alo.y; a[u][2] = ahi.x; a[u][3] = ahi.y;
            b[u][0] = blo.x; b[u][1] = blo.y; b[u][2] = bhi.x; b[u][3] = bhi.y;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    acc[jt][it] = mfma64(a[jt][s], b[it][s], acc[jt][it]);
    }

    // C' = K' - S'   (K_rj was put in place by fill_kernel)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int jj = 16 * jt + q + 4 * s, i = 16 * it + r16;
                acc[jt][it][s] = Lr[(long)i * ld + kmax + jj] - acc[jt][it][s];
            }

    // X' = L_jj^-1 C' by 16-row block substitution:
    //   X'_ct = Dinv_ct (C'_ct - sum_{jt<ct} L_jj[ct][jt] X'_jt),  X'_ct overwrites acc[ct]
    const double *dinv = p.dinv + (long)item * (NB / TB) * (TB * TB);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        f64x4 tmp[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) tmp[it] = acc[ct][it];
#pragma unroll
        for (int jt = 0; jt < ct; ++jt) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double a =
                    -Lj[(long)(16 * ct + r16) * ld + kmax + 16 * jt + q + 4 * s];
#pragma unroll
                for (int it = 0; it < 4; ++it) tmp[it] = mfma64(a, acc[jt][it][s], tmp[it]);
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) acc[ct][it] = (f64x4){0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const double a = dinv[ct * (TB * TB) + r16 * TB + q + 4 * s];
#pragma unroll
            for (int it = 0; it < 4; ++it) acc[ct][it] = mfma64(a, tmp[it][s], acc[ct][it]);
        }
    }
    // store L_rj: X'[c = 16ct + q + 4s][i = 16it + r16] -> L_r[i][kmax + c]
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                Lr[(long)(16 * it + r16) * ld + kmax + 16 * ct + q + 4 * s] = acc[ct][it][s];
}

// ---------------------------------------------------------------------------------------
// gram: G = W W' over the aux rows (lower triangle computed, mirrored on store)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gram_kernel(JobGeom g, const double *L, double *G) {
    const int item = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *W = L + (long)item * g.item_stride + (long)g.n0 * g.ld;
    double *Go = G + (long)item * g.naux * g.naux;
    const int npairs = g.naux * (g.naux + 1) / 2;
    for (int pr = wave; pr < npairs; pr += 4) {
        int a = (int)((sqrt(8.0 * pr + 1.0) - 1.0) * 0.5);
        while ((a + 1) * (a + 2) / 2 <= pr) ++a;
        while (a * (a + 1) / 2 > pr) --a;
        const int b = pr - a * (a + 1) / 2;
        const double *wa = W + (long)a * g.ld, *wb = W + (long)b * g.ld;
        double s = 0.0;
        for (int k = lane; k < g.n0; k += 64) s += wa[k] * wb[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) {
            Go[a * g.naux + b] = s;
            Go[b * g.naux + a] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------
// epilogue: Schur-complement algebra on G (one single-wave workgroup per item)
//   A = appended rows (da), T = forecast rows (m), Y = data row
//   S_AA = K_AA + nz I - G_AA = L_A L_A'        V_A = (K_TA - G_TA) L_A^-T
//   Sigma = K_TT - G_TT - V_A V_A' (+ nz I)     per scenario: z_A = L_A^-1 (y_A - G_AY)
//   logml_full = -1/2 (G_YY + |z_A|^2) - (logdet0 + sum log diag L_A) - (n+d)/2 log 2pi
//   mu = G_TY + V_A z_A ;  logml_base = same with the first `tail` appended rows only
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void epilogue_kernel(JobGeom g, EpiPtrs p, DevSpec sp) {
    __shared__ DevProgram P;
    __shared__ int bad;
    const int item = blockIdx.x, tid = threadIdx.x;
    load_program(&P, p.progs + item);
    if (tid == 0) bad = 0;
    __syncthreads();
    const int da = g.da, m = g.m, na = g.naux, Y = da + m;
    const double nz = P.noise + sp.jitter;
    const double *G = p.G + (long)item * na * na;
    const double *ta = p.taux, *tt = p.taux + da;
    double *work = p.work + (long)item * p.work_stride;
    double *LA = work;                  // [da x da]
    double *VA = LA + (long)da * da;    // [m x da]
    double *ldA = VA + (long)m * da;    // [da] log diag L_A

    for (int e = tid; e < da * da; e += 64) {
        const int a = e / da, b = e % da;
        double v = 0.0;
        if (b <= a) {
            v = keval(P, sp, ta[a], ta[b]) - (g.n0 ? G[a * na + b] : 0.0);
            if (a == b) v += nz;
        }
        LA[e] = v;
    }
    __syncthreads();
    for (int k = 0; k < da; ++k) {  // in-place right-looking Cholesky of S_AA
        const double akk = LA[k * da + k];
        const double dk = sqrt(akk);
        __syncthreads();
        if (tid == 0) {
            if (!(akk > 0.0) && bad == 0) bad = k + 1;
            LA[k * da + k] = dk;
            ldA[k] = log(dk);
        }
        for (int i = k + 1 + tid; i < da; i += 64) LA[i * da + k] /= dk;
        __syncthreads();
        for (int e = tid; e < (da - k - 1) * (da - k - 1); e += 64) {
            const int i = k + 1 + e / (da - k - 1), jj = k + 1 + e % (da - k - 1);
            if (jj <= i) LA[i * da + jj] -= LA[i * da + k] * LA[jj * da + k];
        }
        __syncthreads();
    }
    // V_A: one forecast row per thread, forward substitution along the appended points
    for (int i = tid; i < m; i += 64) {
        for (int a = 0; a < da; ++a) {
            double s = keval(P, sp, tt[i], ta[a]) - (g.n0 ? G[(da + i) * na + a] : 0.0);
            for (int pp = 0; pp < a; ++pp) s -= VA[i * da + pp] * LA[a * da + pp];
            VA[i * da + a] = s / LA[a * da + a];
        }
    }
    __syncthreads();
    if (p.sigma) {
        double *Sg = p.sigma + (long)item * m * m;
        for (int e = tid; e < m * m; e += 64) {
            const int i = e / m, jj = e % m;
            if (jj > i) continue;
            double s = keval(P, sp, tt[i], tt[jj]) - (g.n0 ? G[(da + i) * na + da + jj] : 0.0);
            for (int a = 0; a < da; ++a) s -= VA[i * da + a] * VA[jj * da + a];
            if (i == jj && g.noise_on_new) s += nz;
            Sg[i * m + jj] = s;
            Sg[jj * m + i] = s;
        }
    }
    const double q0 = g.n0 ? G[Y * na + Y] : 0.0;
    const double ld0 = p.logdet[item];
    const double LOG2PI = 1.8378770664093454836;
    const double *ya_base = p.ya + (g.y_shared ? 0 : (long)item * g.D * da);
    for (int s = tid; s < g.D; s += 64) {
        const double *ya = ya_base + (long)s * da;
        double *z = p.zbuf + ((long)item * g.D + s) * da;
        double quad = 0.0, quad_tail = 0.0, ldsum = 0.0, ld_tail = 0.0;
        for (int a = 0; a < da; ++a) {
            double e = ya[a] - (g.n0 ? G[a * na + Y] : 0.0);
            for (int pp = 0; pp < a; ++pp) e -= LA[a * da + pp] * z[pp];
            e /= LA[a * da + a];
            z[a] = e;
            quad += e * e;
            ldsum += ldA[a];
            if (a < g.tail) { quad_tail += e * e; ld_tail += ldA[a]; }
        }
        const int nfull = g.n0 + da;
        p.logml_full[(long)item * g.D + s] = -0.5 * (q0 + quad) - (ld0 + ldsum) - 0.5 * nfull * LOG2PI;
        if (s == 0)
            p.logml_base[item] =
                -0.5 * (q0 + quad_tail) - (ld0 + ld_tail) - 0.5 * (g.n0 + g.tail) * LOG2PI;
        if (p.mu) {
            double *mu = p.mu + ((long)item * g.D + s) * m;
            for (int i = 0; i < m; ++i) {
                double v = g.n0 ? G[(da + i) * na + Y] : 0.0;
                for (int a = 0; a < da; ++a) v += VA[i * da + a] * z[a];
                mu[i] = v;
            }
        }
    }
    if (tid == 0 && bad && p.info[item] == 0) p.info[item] = g.n0 + bad;
}

// ---------------------------------------------------------------------------------------
// microbenchmarks / self tests
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mfma_bench_kernel(double *out, int iters) {
    f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    for (int i = 0; i < iters; i += 4) {
        c0 = mfma64(a, b, c0);
        c1 = mfma64(a, b, c1);
        c2 = mfma64(a, b, c2);
        c3 = mfma64(a, b, c3);
    }
    const f64x4 r = c0 + c1 + c2 + c3;
    if (r[0] + r[1] + r[2] + r[3] == -1.0) out[blockIdx.x * 256 + threadIdx.x] = r[0];
}

// per-wave shader-clock cycles (s_memtime) and 100 MHz wall ticks (s_memrealtime) around the loop
__global__ __launch_bounds__(256) void mfma_bench_detail_kernel(unsigned long long *stamps,
                                                                int iters) {
    f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i += 4) {
        c0 = mfma64(a, b, c0);
        c1 = mfma64(a, b, c1);
        c2 = mfma64(a, b, c2);
        c3 = mfma64(a, b, c3);
    }
    const f64x4 r = c0 + c1 + c2 + c3;
    asm volatile("" ::"v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

__global__ void mfma_layout_probe_kernel(const double *A, const double *Bm, double *Dout) {
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];    // A[m][k], 16x4 row-major
    const double b = Bm[(l >> 4) * 16 + (l & 15)];  // B[k][n], 4x16 row-major
    const f64x4 d = mfma64(a, b, (f64x4){0, 0, 0, 0});
#pragma unroll
    for (int r = 0; r < 4; ++r) Dout[((l >> 4) + 4 * r) * 16 + (l & 15)] = d[r];
}

__global__ __launch_bounds__(256) void stream_write_kernel(f64x2 *dst, long n2) {
    const f64x2 v = {1.0, 2.0};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256)
        dst[i] = v;
}
__global__ __launch_bounds__(256) void stream_copy_kernel(f64x2 *dst, const f64x2 *src, long n2) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256)
        dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
void launch_fill(const JobGeom &g, const ChunkPtrs &p, int Bc, const DevSpec &sp, hipStream_t s) {
    if (g.n0 == 0) return;
    const int ntri = g.nb0 * (g.nb0 + 1) / 2;
    const int ntiles = ntri + (g.naux_pad / NB) * g.nb0;
    hipLaunchKernelGGL(fill_kernel, dim3(ntiles, Bc), dim3(256), 0, s, g, p, ntri, sp);
}

void launch_chol_diag(const JobGeom &g, const ChunkPtrs &p, int Bc, int j, const DevSpec &,
                      hipStream_t s) {
    hipLaunchKernelGGL(chol_diag_kernel, dim3(Bc), dim3(256), 0, s, g, p, j);
}

void launch_chol_col(const JobGeom &g, const ChunkPtrs &p, int Bc, int j, const DevSpec &,
                     hipStream_t s) {
    const int nmain = g.nb0 - 1 - j;
    const int ntiles = nmain + g.naux_pad / NB;
    if (ntiles <= 0) return;
    const int groups = (ntiles + 3) / 4;
    const int bpad = (Bc + 7) / 8 * 8;
    hipLaunchKernelGGL(chol_col_kernel, dim3(groups * bpad), dim3(256), 0, s, g, p, Bc, j, groups,
                       nmain, ntiles);
}

void launch_gram(const JobGeom &g, const double *L, double *G, int Bc, hipStream_t s) {
    if (g.n0 == 0) return;
    hipLaunchKernelGGL(gram_kernel, dim3(Bc), dim3(256), 0, s, g, L, G);
}

void launch_epilogue(const JobGeom &g, const EpiPtrs &p, const DevSpec &sp, hipStream_t s) {
    hipLaunchKernelGGL(epilogue_kernel, dim3(g.B), dim3(64), 0, s, g, p, sp);
}

void launch_cov(const DevProgram *progs, int B, const double *t1, int n1, const double *t2, int n2,
                int add_diag, double *out, const DevSpec &sp, hipStream_t s) {
    long total = (long)n1 * n2;
    int gx = (int)((total + 255) / 256);
    if (gx > 2048) gx = 2048;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(cov_kernel, dim3(gx, B), dim3(256), 0, s, progs, t1, n1, t2, n2, add_diag,
                       out, sp);
}

void launch_mfma_bench(double *out, int iters, int blocks, hipStream_t s) {
    hipLaunchKernelGGL(mfma_bench_kernel, dim3(blocks), dim3(256), 0, s, out, iters);
}
void launch_mfma_bench_detail(unsigned long long *stamps, int iters, int blocks, hipStream_t s) {
    hipLaunchKernelGGL(mfma_bench_detail_kernel, dim3(blocks), dim3(256), 0, s, stamps, iters);
}
void launch_mfma_layout_probe(const double *A, const double *Bm, double *Dout, hipStream_t s) {
    hipLaunchKernelGGL(mfma_layout_probe_kernel, dim3(1), dim3(64), 0, s, A, Bm, Dout);
}
void launch_stream_write(double *dst, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(stream_write_kernel, dim3(2048), dim3(256), 0, s,
                       reinterpret_cast<f64x2 *>(dst), (long)(n / 2));
}
void launch_stream_copy(double *dst, const double *src, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(stream_copy_kernel, dim3(2048), dim3(256), 0, s,
                       reinterpret_cast<f64x2 *>(dst), reinterpret_cast<const f64x2 *>(src),
                       (long)(n / 2));
}

}  // namespace ngp
