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
//   tables_kernel / fill_lattice_kernel   table-driven covariance fill on lattice times
//   fill_kernel       direct RPN kernel-tree interpreter, one 64x64 tile per workgroup, 512-B row
//                     stores                                   (HBM-write + fp64 transcendental VALU)
//   chol_diag_kernel  C_jj -= L_j L_j' (MFMA), 64x64 Cholesky four pivots per barrier round on
//                     packed lower-triangular LDS tiles, full inverse M = L_jj^-1 in MFMA strip
//                     order                                                 (latency-bound)
//   chol_col_glds_kernel / chol_col_kernel   C_rj -= L_r L_j' (v_mfma_f64_4x4x4_4b_f64 composite,
//                     64x64 tile per wave, LDS-DMA staged operands on the long k-loops), then the
//                     64-wide solve L_rj = C_rj M' on the accumulators (fp64-MFMA-bound; dominant)
//   diag_ahead_kernel pre-accumulation of the next-but-one diagonal tile on a side stream
//   gram_kernel       G = W W'                                              (HBM-read-bound, small)
//   epilogue_kernel   dense Schur algebra per item + per-scenario solves     (latency-bound, tiny)
//   grad_*            K^-1 = W_I W_I' (MFMA), reverse-mode contraction with aa' - K^-1
//   aux_update_kernel right-looking sweep of the aux rows through a resident factor
//
// MFMA operand maps used throughout (v_mfma_f64_16x16x4_f64, guide cdna_hip_programming.md §3):
//   A: lane l holds A[m = l&15][k = l>>4]     B: lane l holds B[k = l>>4][n = l&15]
//   D: lane l, register r holds D[m = (l>>4) + 4 r][n = l&15]
// so a D-layout tile is directly the B operand of a following product that sums over its row
// index (register r <-> k-slot), which is what keeps the triangular solve in registers.
#include <type_traits>

#include "ngp_internal.h"
#include "ngp_mfma.h"
#include "ngp_col_kernels.h"
#include "ngp_small_kernels.h"

namespace ngp {

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
__global__ __launch_bounds__(256) void fill_kernel(JobGeom g, ChunkPtrs p, int ntri, int tile_off,
                                                   DevSpec sp) {
    __shared__ DevProgram P;
    const int item = blockIdx.y;
    load_program(&P, p.progs + item);
    __syncthreads();
    const int tile = blockIdx.x + tile_off;   // tile_off = ntri: aux rows only (cached factor)
    int r, c;            // block row / block column
    bool aux = false;
    if (tile < ntri) {   // lower-triangular block (r >= c): tile = r(r+1)/2 + c
        r = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
        while ((r + 1) * (r + 2) / 2 <= tile) ++r;
        while (r * (r + 1) / 2 > tile) --r;
        c = tile - r * (r + 1) / 2;
    } else {
        const int a = tile - ntri;
        if (g.aux_identity) {   // only the y' tile row and the zero blocks (a, a-1), see launch_fill
            r = a < g.nb0 ? g.nb0 : a - g.nb0 + 1;
            c = a < g.nb0 ? a : a - g.nb0;
        } else {
            r = a / g.nb0;   // aux tile row
            c = a % g.nb0;
        }
        aux = true;
    }
    // thread = (column pair tx, 8-row group ty): two adjacent columns per thread -> 16-byte stores
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int col = c * NB + 2 * tx;
    const double t2a = p.t0[col], t2b = p.t0[col + 1];
    const double diag = P.noise + sp.jitter;
    double *Lit = p.L + (long)item * g.item_stride;
    const int naux_t = g.da + g.m;
    const double *y0 = p.y0 + (g.y_shared ? 0 : (long)item * g.n0);
    for (int rr = 0; rr < 8; ++rr) {
        const int lr = ty * 8 + rr;
        f64x2 v;
        long row;
        if (!aux) {
            row = (long)r * NB + lr;
            const double t1 = p.t0[row];
            v.x = keval(P, sp, t1, t2a);
            v.y = keval(P, sp, t1, t2b);
            if (row == col) v.x += diag;
            if (row == col + 1) v.y += diag;
            if (row >= g.n_real || col >= g.n_real) v.x = (row == col) ? 1.0 : 0.0;
            if (row >= g.n_real || col + 1 >= g.n_real) v.y = (row == col + 1) ? 1.0 : 0.0;
        } else if (g.aux_identity) {
            const int ar = r * NB + lr;
            row = (long)g.n0 + ar;
            v.x = ar < g.n0 ? (ar == col ? 1.0 : 0.0) : (ar == g.n0 ? y0[col] : 0.0);
            v.y = ar < g.n0 ? (ar == col + 1 ? 1.0 : 0.0) : (ar == g.n0 ? y0[col + 1] : 0.0);
        } else {
            const int ar = r * NB + lr;
            row = (long)g.n0 + ar;
            if (ar < naux_t) {
                v.x = keval(P, sp, p.taux[ar], t2a);
                v.y = keval(P, sp, p.taux[ar], t2b);
            } else if (ar == naux_t) {
                v.x = y0[col];
                v.y = y0[col + 1];
            } else {
                v.x = 0.0;
                v.y = 0.0;
            }
        }
        *reinterpret_cast<f64x2 *>(Lit + row * g.ld + col) = v;
        if (aux && p.auxX)   // mixed-precision jobs keep the untouched aux rows X for the refinement
            *reinterpret_cast<f64x2 *>(p.auxX + ((long)item * g.naux_pad + (row - g.n0)) * g.ld + col) = v;
    }
}

// ---------------------------------------------------------------------------------------
// table-driven fill.  Dates are integer days, so after AutoGP's [0,1] rescale every time sits on
// a lattice t = tmin + q h.  Every transcendental of the kernel grammar is then a function of
// either the integer distance |q_i - q_j| (SquaredExponential / GammaExponential / Periodic
// leaves) or of a single point (ChangePoint sigmoids): O(n) evaluations per leaf instead of
// O(n^2).  tables_kernel evaluates them once per item; fill_lattice_kernel is then pure
// lookups + FMAs and runs at the HBM-write rate.
// ---------------------------------------------------------------------------------------
__device__ double keval_stat(const DevProgram &P, const DevSpec &sp, int first, int last, double d);

__global__ __launch_bounds__(256) void tables_kernel(JobGeom g, ChunkPtrs p, DevSpec sp) {
    __shared__ DevProgram P;
    const int item = blockIdx.x;
    load_program(&P, (p.progs_src ? p.progs_src : p.progs) + item);
    __syncthreads();
    if (p.progs_src) load_program(const_cast<DevProgram *>(p.progs) + item, &P);   // see ChunkPtrs::progs_src
    double *tab = p.tab + (long)item * g.maxstat * g.R;
    double *sig = p.sig + (long)item * g.maxcp * g.npts;
    // gradient jobs: dt = [slot][3][R]: e (the leaf value without its amplitude) and the two
    // factors its lengthscale-type derivatives need, so the O(n^2) contraction is lookups + FMAs
    double *dt = p.dtab ? p.dtab + (long)item * g.maxstat * 3 * g.R : nullptr;
    if (!dt || g.tab_sub > 0) {
        // one table per maximal stationary subtree of the tree (reduced program): all a value job
        // needs; a gradient job keeps them BEHIND its per-leaf tables (slot g.tab_sub on) — its fill
        // then runs on the reduced-program kernels like a value job's, the contraction on the leaves
        // (subtree by subtree: keval_stat takes its opcodes wave-uniformly, so the lanes of a wave
        // must be in the same subtree)
        for (int k = 0; k < P.n_tab; ++k) {
            const int first = P.tb_first[k], last = P.tb_last[k];
            for (int idx = threadIdx.x; idx < g.R; idx += 256)
                tab[(long)(g.tab_sub + k) * g.R + idx] = keval_stat(P, sp, first, last, idx * g.h);
        }
    }
    // One pass over (node, lattice distance) pairs and one over (ChangePoint, point) pairs: a short
    // series (R of a few dozen — the early annealing steps of a fit) fills every leaf's table in ONE
    // round of the workgroup instead of a round per leaf, each a chain of fp64 transcendentals
    // (15 us of a 24-item call at n = 21).  Per entry the arithmetic is what it was.
    for (int e = threadIdx.x; e < P.n_ops * g.R; e += 256) {
        const int i = e / g.R, k = e - i * g.R;
        const int op = P.ops[i], slot = P.slot[i], pi = P.poff[i];
        double *d0 = dt ? dt + (long)slot * 3 * g.R : nullptr;
        if (!d0) break;   // leaf tables: gradient jobs only (value jobs tabulate whole subtrees, above)
        if (op == NGP_OP_SQEXP) {
            const double l = P.params[pi], a = P.params[pi + 1];
            const double den = sp.se_form ? l : l * l;
            const double d = k * g.h;
            const double ev = exp(-0.5 * d * d / den);
            tab[(long)slot * g.R + k] = a * ev;
            d0[k] = ev;
        } else if (op == NGP_OP_GAMMAEXP) {
            const double l = P.params[pi], gam = P.params[pi + 1], a = P.params[pi + 2];
            const double rr = k * g.h / l, u = pow(rr, gam), ev = exp(-u);
            tab[(long)slot * g.R + k] = a * ev;
            d0[k] = ev;
            d0[g.R + k] = ev * u;                                  // -> d / d lengthscale
            d0[2 * g.R + k] = (k > 0) ? ev * u * log(rr) : 0.0;    // -> d / d gamma
        } else if (op == NGP_OP_PERIODIC) {
            const double l = P.params[pi], per = P.params[pi + 1], a = P.params[pi + 2];
            const double c = sp.periodic_form ? 2.0 / l : 2.0 / (l * l);
            const double d = k * g.h, ang = M_PI * d / per;
            const double sn = sin(ang), ev = exp(-c * sn * sn);
            tab[(long)slot * g.R + k] = a * ev;
            d0[k] = ev;
            d0[g.R + k] = ev * sn * sn;                 // -> d / d lengthscale
            d0[2 * g.R + k] = ev * sn * cos(ang) * d;   // -> d / d period
        }
    }
    for (int e = threadIdx.x; e < P.n_ops * g.npts; e += 256) {
        const int i = e / g.npts, pt = e - i * g.npts;
        const int op = P.ops[i];
        if (op == NGP_OP_CHANGEPOINT || op == OP_CP_SWAPPED) {
            const int pi = P.poff[i];
            const double loc = P.params[pi], sc = P.params[pi + 1];
            const double t = pt < g.n0 ? p.t0[pt] : p.taux[pt - g.n0];
            sig[(long)P.slot[i] * g.npts + pt] = cp_sigma(sp.cp_form, t, loc, sc);
        }
    }
}

__device__ __forceinline__ double keval_lattice(const DevProgram &P, const double *tab,
                                                const double *sig, int R, int npts, double t1,
                                                double t2, int dq, int pt1, int pt2) {
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
            } else {
                const int slot = __builtin_amdgcn_readfirstlane((int)P.slot[i]);
                v = tab[(long)slot * R + dq];
                pi += (op == NGP_OP_SQEXP) ? 2 : 3;
            }
            s7 = s6; s6 = s5; s5 = s4; s4 = s3; s3 = s2; s2 = s1; s1 = s0; s0 = v;
        } else {
            double v;
            if (op == NGP_OP_PLUS) {
                v = s1 + s0;
            } else if (op == NGP_OP_TIMES) {
                v = s1 * s0;
            } else {
                const int slot = __builtin_amdgcn_readfirstlane((int)P.slot[i]);
                const double kl = (op == NGP_OP_CHANGEPOINT) ? s1 : s0;
                const double kr = (op == NGP_OP_CHANGEPOINT) ? s0 : s1;
                const double g1 = sig[(long)slot * npts + pt1];
                const double g2 = sig[(long)slot * npts + pt2];
                v = g1 * kl * g2 + (1.0 - g1) * kr * (1.0 - g2);
                pi += 2;
            }
            s0 = v; s1 = s2; s2 = s3; s3 = s4; s4 = s5; s5 = s6; s6 = s7;
        }
    }
    return s0;
}

// Value of the stationary subtree ops[first..last] (a postfix range of the full program) at
// distance d: the leaf formulas of tables_kernel / keval, operation for operation.
__device__ double keval_stat(const DevProgram &P, const DevSpec &sp, int first, int last, double d) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
    for (int i = first; i <= last; ++i) {
        const int op = __builtin_amdgcn_readfirstlane((int)P.ops[i]);
        const int pi = __builtin_amdgcn_readfirstlane((int)P.poff[i]);
        if (op < NGP_OP_PLUS) {
            double v;
            if (op == NGP_OP_CONSTANT) {
                v = P.params[pi];
            } else if (op == NGP_OP_SQEXP) {
                const double l = P.params[pi];
                const double den = sp.se_form ? l : l * l;
                v = P.params[pi + 1] * exp(-0.5 * d * d / den);
            } else if (op == NGP_OP_GAMMAEXP) {
                const double rr = d / P.params[pi], u = pow(rr, P.params[pi + 1]);
                v = P.params[pi + 2] * exp(-u);
            } else {  // NGP_OP_PERIODIC
                const double l = P.params[pi];
                const double c = sp.periodic_form ? 2.0 / l : 2.0 / (l * l);
                const double sn = sin(M_PI * d / P.params[pi + 1]);
                v = P.params[pi + 2] * exp(-c * sn * sn);
            }
            s7 = s6; s6 = s5; s5 = s4; s4 = s3; s3 = s2; s2 = s1; s1 = s0; s0 = v;
        } else {
            const double v = (op == NGP_OP_PLUS) ? s1 + s0 : s1 * s0;
            s0 = v; s1 = s2; s2 = s3; s3 = s4; s4 = s5; s5 = s6; s6 = s7;
        }
    }
    return s0;
}

// k(t1, t2) on lattice times through the REDUCED program (DevProgram::rops): table leaves by
// lattice distance dq, Linear in closed form, ChangePoint sigmoids by point
__device__ __forceinline__ double keval_reduced(const DevProgram &P, const double *tab,
                                                const double *sig, int R, int npts, double t1,
                                                double t2, int dq, int pt1, int pt2) {
    const int nops = P.n_rops;
    if (nops == 1 && P.rops[0] == OP_TABLE) return tab[dq];   // the whole tree is stationary
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
    auto linear = [&](int pi) {
        const double c = P.params[pi];
        return P.params[pi + 1] + P.params[pi + 2] * (t1 - c) * (t2 - c);
    };
    for (int i = 0; i < nops; ++i) {
        const int code = __builtin_amdgcn_readfirstlane((int)P.rops[i]);
        const int op = code & 15, lk = code >> 4;
        if (op == OP_TABLE || op == NGP_OP_LINEAR) {
            double v;
            if (op == OP_TABLE) {
                const int slot = __builtin_amdgcn_readfirstlane((int)P.rslot[i]);
                v = tab[(long)slot * R + dq];
            } else {
                v = linear(__builtin_amdgcn_readfirstlane((int)P.rpoff[i]));
            }
            s7 = s6; s6 = s5; s5 = s4; s4 = s3; s3 = s2; s2 = s1; s1 = s0; s0 = v;
        } else {
            // operands in evaluation order: a (first), b (second — the fused leaf if there is one)
            double a, b;
            if (lk) {
                const int lf = __builtin_amdgcn_readfirstlane((int)P.rleaf[i]);
                a = s0;
                b = (lk == RLEAF_TABLE) ? tab[(long)lf * R + dq] : linear(lf);
            } else {
                a = s1;
                b = s0;
            }
            double v;
            if (op == NGP_OP_PLUS) {
                v = a + b;
            } else if (op == NGP_OP_TIMES) {
                v = a * b;
            } else {
                const int slot = __builtin_amdgcn_readfirstlane((int)P.rslot[i]);
                const double kl = (op == NGP_OP_CHANGEPOINT) ? a : b;
                const double kr = (op == NGP_OP_CHANGEPOINT) ? b : a;
                const double g1 = sig[(long)slot * npts + pt1];
                const double g2 = sig[(long)slot * npts + pt2];
                v = g1 * kl * g2 + (1.0 - g1) * kr * (1.0 - g2);
            }
            if (lk) {
                s0 = v;
            } else {
                s0 = v; s1 = s2; s2 = s3; s3 = s4; s4 = s5; s5 = s6; s6 = s7;
            }
        }
    }
    return s0;
}

// split: workgroups per tile (1, 2, 4) — small launches are latency-bound on the 8 rows a thread
// walks, so they are cut into more, shorter workgroups (as in the gradient contraction)
// GRADJOB: the tables are per leaf (the contraction needs them that way) -> full program;
// otherwise per maximal stationary subtree -> reduced program
template <bool GRADJOB>
__global__ __launch_bounds__(256) void fill_lattice_kernel(JobGeom g, ChunkPtrs p, int ntri,
                                                           int tile_off, int split, DevSpec sp) {
    __shared__ DevProgram P;
    // p.fill_other (staged value jobs): the chunk's items that are not chain programs
    const int item = p.fill_other ? p.fill_other[blockIdx.y] - p.fill_base : (int)blockIdx.y;
    load_program(&P, p.progs + item);
    __syncthreads();
    const int tile = blockIdx.x / split + tile_off, sub = blockIdx.x % split;
    const int nrows = 8 / split;
    int r, c;
    bool aux = false;
    if (tile < ntri) {
        r = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
        while ((r + 1) * (r + 2) / 2 <= tile) ++r;
        while (r * (r + 1) / 2 > tile) --r;
        c = tile - r * (r + 1) / 2;
    } else {
        const int a = tile - ntri;
        if (g.aux_identity) {   // only the y' tile row and the zero blocks (a, a-1), see launch_fill
            r = a < g.nb0 ? g.nb0 : a - g.nb0 + 1;
            c = a < g.nb0 ? a : a - g.nb0;
        } else {
            r = a / g.nb0;   // aux tile row
            c = a % g.nb0;
        }
        aux = true;
    }
    // thread = (column pair tx, 8-row group ty): two adjacent columns per thread -> 16-byte stores
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int col = c * NB + 2 * tx;
    const double t2a = p.t0[col], t2b = p.t0[col + 1];
    const int q2a = p.qpts[col], q2b = p.qpts[col + 1];
    const double diag = P.noise + sp.jitter;
    double *Lit = p.L + (long)item * g.item_stride;
    const double *tab = p.tab + (long)item * g.maxstat * g.R;
    const double *sig = p.sig + (long)item * g.maxcp * g.npts;
    const int naux_t = g.da + g.m;
    const double *y0 = p.y0 + (g.y_shared ? 0 : (long)item * g.n0);
    auto kev = [&](const DevProgram &Pp, const double *tb, const double *sg, int R, int npts,
                   double t1, double t2, int dq, int pt1, int pt2) -> double {
        if constexpr (GRADJOB) return keval_lattice(Pp, tb, sg, R, npts, t1, t2, dq, pt1, pt2);
        else return keval_reduced(Pp, tb, sg, R, npts, t1, t2, dq, pt1, pt2);
    };
    for (int rr = 0; rr < nrows; ++rr) {
        const int lr = ty * 8 + sub * nrows + rr;
        f64x2 v;
        long row;
        if (!aux) {
            row = (long)r * NB + lr;
            const int q1 = p.qpts[row];
            const double t1 = p.t0[row];
            v.x = kev(P, tab, sig, g.R, g.npts, t1, t2a, abs(q1 - q2a), (int)row, col);
            v.y = kev(P, tab, sig, g.R, g.npts, t1, t2b, abs(q1 - q2b), (int)row, col + 1);
            if (row == col) v.x += diag;
            if (row == col + 1) v.y += diag;
            if (row >= g.n_real || col >= g.n_real) v.x = (row == col) ? 1.0 : 0.0;
            if (row >= g.n_real || col + 1 >= g.n_real) v.y = (row == col + 1) ? 1.0 : 0.0;
        } else if (g.aux_identity) {
            const int ar = r * NB + lr;
            row = (long)g.n0 + ar;
            v.x = ar < g.n0 ? (ar == col ? 1.0 : 0.0) : (ar == g.n0 ? y0[col] : 0.0);
            v.y = ar < g.n0 ? (ar == col + 1 ? 1.0 : 0.0) : (ar == g.n0 ? y0[col + 1] : 0.0);
        } else {
            const int ar = r * NB + lr;
            row = (long)g.n0 + ar;
            if (ar < naux_t) {
                const int q1 = p.qpts[g.n0 + ar];
                const double t1 = p.taux[ar];
                v.x = kev(P, tab, sig, g.R, g.npts, t1, t2a, abs(q1 - q2a), g.n0 + ar, col);
                v.y = kev(P, tab, sig, g.R, g.npts, t1, t2b, abs(q1 - q2b), g.n0 + ar, col + 1);
            } else if (ar == naux_t) {
                v.x = y0[col];
                v.y = y0[col + 1];
            } else if (g.aux_e1 && ar == naux_t + 1) {   // the Toeplitz gradient path: e_1' beside y'
                v.x = (col == 0) ? 1.0 : 0.0;
                v.y = 0.0;
            } else {
                v.x = 0.0;
                v.y = 0.0;
            }
        }
        *reinterpret_cast<f64x2 *>(Lit + row * g.ld + col) = v;
        if (aux && p.auxX)   // mixed-precision jobs keep the untouched aux rows X for the refinement
            *reinterpret_cast<f64x2 *>(p.auxX + ((long)item * g.naux_pad + (row - g.n0)) * g.ld + col) = v;
    }
}

// Stationary trees (the whole reduced program is ONE table: 45 of the 64 base kernels of the bench
// ensemble): K[i][j] = tab[|q_i - q_j|].  No program in LDS, no interpreter: sixteen lookups in
// flight per thread at full occupancy.  `p.fill_single` lists the chunk's such items.
__global__ __launch_bounds__(256) void fill_single_kernel(JobGeom g, ChunkPtrs p, int ntri,
                                                          DevSpec sp) {
    const int item = p.fill_single[blockIdx.y] - p.fill_base;
    const int tile = blockIdx.x;
    int r, c;
    bool aux = false;
    if (g.toep) {
        // Toeplitz jobs: only the diagonal tiles (they carry the noise) and the aux rows are
        // stored; the column kernels take every other tile from the table (struct_slice)
        if (tile < g.nb0) {
            r = c = tile;
        } else {
            const int a = tile - g.nb0;
            r = a / g.nb0;
            c = a % g.nb0;
            aux = true;
        }
    } else if (tile < ntri) {
        r = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
        while ((r + 1) * (r + 2) / 2 <= tile) ++r;
        while (r * (r + 1) / 2 > tile) --r;
        c = tile - r * (r + 1) / 2;
    } else {
        const int a = tile - ntri;
        r = a / g.nb0;
        c = a % g.nb0;
        aux = true;
    }
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int col = c * NB + 2 * tx;
    const int q2a = p.qpts[col], q2b = p.qpts[col + 1];
    const DevProgram *P = p.progs + item;
    const double diag = P->noise + sp.jitter;
    double *Lit = p.L + (long)item * g.item_stride;
    const double *tab = p.tab + (long)item * g.maxstat * g.R;   // slot 0: the tree's only table
    const int naux_t = g.da + g.m;
    const double *y0 = p.y0 + (g.y_shared ? 0 : (long)item * g.n0);
    f64x2 v[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        int pt = r * NB + ty * 8 + rr;
        if (aux) pt = (pt < naux_t) ? g.n0 + pt : 0;
        const int q1 = p.qpts[pt];
        v[rr].x = tab[abs(q1 - q2a)];
        v[rr].y = tab[abs(q1 - q2b)];
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int lr = ty * 8 + rr;
        f64x2 o = v[rr];
        long row;
        if (!aux) {
            row = (long)r * NB + lr;
            if (row == col) o.x += diag;
            if (row == col + 1) o.y += diag;
            if (row >= g.n_real || col >= g.n_real) o.x = (row == col) ? 1.0 : 0.0;
            if (row >= g.n_real || col + 1 >= g.n_real) o.y = (row == col + 1) ? 1.0 : 0.0;
        } else {
            const int ar = r * NB + lr;
            row = (long)g.n0 + ar;
            if (ar == naux_t) {
                o.x = y0[col];
                o.y = y0[col + 1];
            } else if (ar > naux_t) {
                o.x = 0.0;
                o.y = 0.0;
            }
        }
        *reinterpret_cast<f64x2 *>(Lit + row * g.ld + col) = o;
        if (aux && p.auxX)   // mixed-precision jobs keep the untouched aux rows X for the refinement
            *reinterpret_cast<f64x2 *>(p.auxX + ((long)item * g.naux_pad + (row - g.n0)) * g.ld + col) = o;
    }
}

// Chain programs (DevProgram::rchain with more than one instruction: one push, then only operations
// that carry their leaf — 17 of the 19 non-stationary base kernels of the bench ensemble): every
// instruction is decoded once per thread and applied to its 16 elements.  Same formulas and the
// same order of operations per element as keval_reduced: bit-identical values.  A kernel of its own
// (238 VGPRs would cost the single-lookup fill of stationary kernels its occupancy); `items` lists
// the chunk's chain items (ChunkPtrs::fill_chain).
// tpw: tiles a workgroup fills one after the other (large launches: the program is loaded and the
// barrier paid once for `tpw` tiles; the values do not depend on it)
__global__ __launch_bounds__(256) void fill_chain_kernel(JobGeom g, ChunkPtrs p, int ntri,
                                                         DevSpec sp, int ntiles, int tpw) {
    __shared__ DevProgram P;
    const int item = p.fill_chain[blockIdx.y] - p.fill_base;
    load_program(&P, p.progs + item);
    __syncthreads();
    for (int tile = blockIdx.x * tpw; tile < min((int)blockIdx.x * tpw + tpw, ntiles); ++tile) {
    int r, c;
    bool aux = false;
    if (tile < ntri) {
        r = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
        while ((r + 1) * (r + 2) / 2 <= tile) ++r;
        while (r * (r + 1) / 2 > tile) --r;
        c = tile - r * (r + 1) / 2;
    } else {
        const int a = tile - ntri;
        r = a / g.nb0;
        c = a % g.nb0;
        aux = true;
    }
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int col = c * NB + 2 * tx;
    const double t2a = p.t0[col], t2b = p.t0[col + 1];
    const int q2a = p.qpts[col], q2b = p.qpts[col + 1];
    const double diag = P.noise + sp.jitter;
    double *Lit = p.L + (long)item * g.item_stride;
    const double *tab = p.tab + (long)item * g.maxstat * g.R;
    const double *sig = p.sig + (long)item * g.maxcp * g.npts;
    const int naux_t = g.da + g.m;
    const double *y0 = p.y0 + (g.y_shared ? 0 : (long)item * g.n0);
    // Two passes of four rows: half the registers of one pass of eight (124 instead of 206 VGPRs:
    // four waves per SIMD instead of two) for one more decode of the program per thread.  Lattice
    // data of the rows first (aux rows past the last time point: any valid point, value unused).
    for (int half = 0; half < 2; ++half) {
        int q1[4], pt1[4];
        double t1[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int lr = ty * 8 + 4 * half + rr;
            int pt = r * NB + lr;
            if (aux) pt = (pt < naux_t) ? g.n0 + pt : 0;
            pt1[rr] = pt;
            q1[rr] = p.qpts[pt];
            t1[rr] = pt < g.n0 ? p.t0[pt] : p.taux[pt - g.n0];
        }
        double kv[4][2];
        const int nops = P.n_rops;
        for (int i = 0; i < nops; ++i) {
            const int code = __builtin_amdgcn_readfirstlane((int)P.rops[i]);
            const int op = code & 15;
            int lk = code >> 4, lf;
            if (i == 0) {
                lk = (op == OP_TABLE) ? RLEAF_TABLE : RLEAF_LINEAR;
                lf = __builtin_amdgcn_readfirstlane((int)(op == OP_TABLE ? P.rslot[0] : P.rpoff[0]));
            } else {
                lf = __builtin_amdgcn_readfirstlane((int)P.rleaf[i]);
            }
            double b[4][2];
            if (lk == RLEAF_TABLE) {
                const double *tb = tab + (long)lf * g.R;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    b[rr][0] = tb[abs(q1[rr] - q2a)];
                    b[rr][1] = tb[abs(q1[rr] - q2b)];
                }
            } else {
                const double cc = P.params[lf], b0 = P.params[lf + 1], b1 = P.params[lf + 2];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    b[rr][0] = b0 + b1 * (t1[rr] - cc) * (t2a - cc);
                    b[rr][1] = b0 + b1 * (t1[rr] - cc) * (t2b - cc);
                }
            }
            if (i == 0) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    kv[rr][0] = b[rr][0];
                    kv[rr][1] = b[rr][1];
                }
            } else if (op == NGP_OP_PLUS) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    kv[rr][0] = kv[rr][0] + b[rr][0];
                    kv[rr][1] = kv[rr][1] + b[rr][1];
                }
            } else if (op == NGP_OP_TIMES) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    kv[rr][0] = kv[rr][0] * b[rr][0];
                    kv[rr][1] = kv[rr][1] * b[rr][1];
                }
            } else {
                const int slot = __builtin_amdgcn_readfirstlane((int)P.rslot[i]);
                const double *sg = sig + (long)slot * g.npts;
                const double g2a = sg[col], g2b = sg[col + 1];
                const bool fwd = op == NGP_OP_CHANGEPOINT;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const double g1 = sg[pt1[rr]];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const double g2 = u ? g2b : g2a;
                        const double kl = fwd ? kv[rr][u] : b[rr][u];
                        const double kr = fwd ? b[rr][u] : kv[rr][u];
                        kv[rr][u] = g1 * kl * g2 + (1.0 - g1) * kr * (1.0 - g2);
                    }
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int lr = ty * 8 + 4 * half + rr;
            f64x2 v;
            v.x = kv[rr][0];
            v.y = kv[rr][1];
            long row;
            if (!aux) {
                row = (long)r * NB + lr;
                if (row == col) v.x += diag;
                if (row == col + 1) v.y += diag;
                if (row >= g.n_real || col >= g.n_real) v.x = (row == col) ? 1.0 : 0.0;
                if (row >= g.n_real || col + 1 >= g.n_real) v.y = (row == col + 1) ? 1.0 : 0.0;
            } else {
                const int ar = r * NB + lr;
                row = (long)g.n0 + ar;
                if (ar == naux_t) {
                    v.x = y0[col];
                    v.y = y0[col + 1];
                } else if (ar > naux_t) {
                    v.x = 0.0;
                    v.y = 0.0;
                }
            }
            *reinterpret_cast<f64x2 *>(Lit + row * g.ld + col) = v;
            if (aux && p.auxX)   // mixed-precision jobs keep the untouched aux rows X for the refinement
                *reinterpret_cast<f64x2 *>(p.auxX + ((long)item * g.naux_pad + (row - g.n0)) * g.ld + col) = v;
        }
    }
    }   // tiles of this workgroup
}

// order[slot] = the item with the slot-th largest number of fp64 tile products since the previous
// call (mixcnt[2 i + 1] counts them; prev keeps the snapshot), ties by index.  One workgroup.
__global__ __launch_bounds__(256) void mixed_order_kernel(const unsigned *mixcnt, unsigned *prev,
                                                          int32_t *order, int Bc) {
    extern __shared__ unsigned keys[];
    for (int i = threadIdx.x; i < Bc; i += 256) {
        const unsigned now = mixcnt[2 * i + 1];
        keys[i] = now - prev[i];
        prev[i] = now;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Bc; i += 256) {
        const unsigned k = keys[i];
        int rank = 0;
        for (int o = 0; o < Bc; ++o) rank += (keys[o] > k) || (keys[o] == k && o < i);
        order[rank] = i;
    }
}

// diag-ahead: K_(j+2,j+2) -= L_(j+2),[0,kmax) L_(j+2),[0,kmax)' — one workgroup per item, next to
// the fat launch (pre-accumulates the next-but-one diagonal tile so chol_diag's own k-loop is
// <= 128).  NW waves split the k-range and add their partial tiles in wave order through LDS: with
// few items and a long history (64 particles at n = 8192) a single wave per item took 0.8 ms per
// launch and the main stream waited for it at every second block column.
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void diag_ahead_kernel(JobGeom g, ChunkPtrs p, int j) {
    __shared__ double part[NW > 1 ? 64 * 64 : 1];   // [register][lane]: conflict-free
    const int item = blockIdx.x, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long ld = g.ld;
    double *Ld = p.L + (long)item * g.item_stride + (long)(j + 2) * NB * ld;
    const int r16 = lane & 15, q = lane >> 4;
    double acc4[4][4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc4[a][b][r] = 0.0;
    const double *pd = Ld + (long)r16 * ld + 2 * q;
    // k-range in NW pieces, each a multiple of 16 (gemm_rows' stage depth)
    const int kmax = j * NB, per = ((kmax / 16 + NW - 1) / NW) * 16;
    const int k0 = min(wave * per, kmax), k1 = min(k0 + per, kmax);
    // Self-product of the 64 rows: both 64-byte halves of a row's 128-byte line are requested
    // together, one 16-deep stage ahead.  (gemm_rows asks for the halves one MFMA stage apart, and
    // the PMC counters showed every line of these rows fetched from HBM twice: 1.85 x the
    // algorithmic bytes, profiles/r02/README.md.)  Same MFMA order as gemm_rows: bit-identical.
    if (k1 > k0) {
        Frag8<4> f[2][2];
        load_frag8(f[0][0], pd + k0, ld);
        load_frag8(f[0][1], pd + k0 + 8, ld);
        for (int kc = k0; kc < k1; kc += 32) {
            if (kc + 16 < k1) {
                load_frag8(f[1][0], pd + kc + 16, ld);
                load_frag8(f[1][1], pd + kc + 24, ld);
            }
            mfma_frag8(acc4, f[0][0], f[0][0]);
            mfma_frag8(acc4, f[0][1], f[0][1]);
            if (kc + 16 < k1) {
                if (kc + 32 < k1) {
                    load_frag8(f[0][0], pd + kc + 32, ld);
                    load_frag8(f[0][1], pd + kc + 40, ld);
                }
                mfma_frag8(acc4, f[1][0], f[1][0]);
                mfma_frag8(acc4, f[1][1], f[1][1]);
            }
        }
    }
    if constexpr (NW > 1) {
        for (int w = 1; w < NW; ++w) {          // wave w adds its tile; wave 0 collects last
            if (wave == w) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            double *e = &part[((a * 4 + b) * 4 + r) * 64 + lane];
                            *e = (w == 1) ? acc4[a][b][r] : *e + acc4[a][b][r];
                        }
            }
            __syncthreads();
        }
        if (wave != 0) return;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc4[a][b][r] += part[((a * 4 + b) * 4 + r) * 64 + lane];
    }
    subtract_in_place_perm(Ld, ld, (j + 2) * NB, acc4, lane);
}

// cached factor (ngp_factor_*): L and every M_j stay on the device; a query only needs its aux
// rows W = X L^-T.  Right-looking sweep over block columns, so each step is wide instead of a
// long k-loop on a single wave per aux tile:  W_j = C_j M_j' (chol_col_kernel, empty k-range),
// then this kernel:  C_c -= W_j L_(c,j)'  for every block column c > j, one wave per (aux tile, c).
__global__ __launch_bounds__(256, 2) void aux_update_kernel(JobGeom g, ChunkPtrs p, int j) {
    const int item = p.items ? p.items[blockIdx.y] : blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntl = g.naux_pad / NB;
    const int idx = blockIdx.x * 4 + wave;
    if (idx >= ntl * (g.nb0 - 1 - j)) return;
    const int a = idx % ntl, c = j + 1 + idx / ntl;
    const long ld = g.ld;
    double *Lit = p.L + (long)item * g.item_stride;
    double *Wa = Lit + ((long)g.n0 + (long)a * NB) * ld;      // aux tile rows (B operand)
    const double *Lc = Lit + (long)c * NB * ld;               // rows of block c (A operand)
    const int r16 = lane & 15, q = lane >> 4;
    double acc4[4][4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc4[x][y][r] = 0.0;
    gemm_rows<4>(acc4, Lc + (long)r16 * ld + 2 * q, Wa + (long)r16 * ld + 2 * q, ld, j * NB,
                 (j + 1) * NB);
    subtract_in_place_perm(Wa, ld, c * NB, acc4, lane);
}

// ---------------------------------------------------------------------------------------
// Gram refinement of NGP_PREC_MIXED jobs.  X = the aux rows as filled ([k(t_aux, t0) ; y']), P = L L'
// the mixed-precision factor, K the exact fp64 covariance.  With A ~ X K^-1:
//     R = X - A K                      (kapply_kernel: K re-evaluated tile by tile, never stored)
//     G = A X' + R A'                  (refine_gram_*: second-order accurate in the error of A)
//     A += (R L^-T) L^-1               (forward sweep = the resident-factor kernels; backward sweep
//                                       = aux_back_kernel)
// A_0 = W L^-1 from the W = X L^-T the factorisation leaves in the aux rows.
// ---------------------------------------------------------------------------------------
// position of M[R][C] inside the strip-ordered block inverse chol_diag writes
__device__ __forceinline__ int mstrip_index(int R, int C) {
    return (((R >> 2) * 4 + (C >> 4)) << 6) + (R & 3) + 4 * ((C >> 2) & 3) + 16 * (C & 3);
}

// items of the chunk a refinement launch works on: all of them, or the compacted list of those
// that have not converged yet
__device__ __forceinline__ int map_item(const ChunkPtrs &p, int i) {
    return p.items ? p.items[i] : i;
}

// Backward sweep, block column c (c descending across launches), two launches per block column:
//   aux_back_solve_kernel   A_c = C_c L_cc^-1 = C_c M_c, in place in the aux rows and out to Aout
//                           ([Bc][naux_pad][n0]; accumulate: +=); one workgroup per (aux tile, item)
//   aux_back_update_kernel  C_b -= A_c L_(c,b) for every b < c; one wave per (aux tile, b), the same
//                           4x4x4 MFMA tile product as the forward sweep with the L tile read
//                           transposed (k runs down its rows)
__global__ __launch_bounds__(256) void aux_back_solve_kernel(JobGeom g, ChunkPtrs p,
                                                             const double *Mc, double *Aout,
                                                             int accumulate, int c) {
    __shared__ double Ms[NB][NB + 1];   // M = L_cc^-1, natural layout
    __shared__ double Cs[NB][NB + 1];   // the aux tile's block column c
    const int item = map_item(p, blockIdx.y), at = blockIdx.x, tid = threadIdx.x;
    const long ld = g.ld;
    double *Caux = p.L + (long)item * g.item_stride + ((long)g.n0 + (long)at * NB) * ld + c * NB;
    const double *M = Mc + (long)item * (NB * NB);
    const int rows = min(NB, g.naux - at * NB);
    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e >> 6, jj = e & 63;
        Ms[i][jj] = (i >= jj) ? M[mstrip_index(i, jj)] : 0.0;
        Cs[i][jj] = (i < rows) ? Caux[(long)i * ld + jj] : 0.0;
    }
    __syncthreads();
    const int jc = tid & 63, w = tid >> 6;
    for (int a = w; a < rows; a += 4) {
        double sum = 0.0;
        for (int i = jc; i < NB; ++i) sum += Cs[a][i] * Ms[i][jc];
        Caux[(long)a * ld + jc] = sum;
        double *dst = Aout + ((long)item * g.naux_pad + at * NB + a) * ld + c * NB + jc;
        *dst = accumulate ? *dst + sum : sum;
    }
}

// 8 k-values of four 16-row fragments of a TRANSPOSED operand: element (m, k) lives at p[k * ld + m]
// (lane (r16, q): rows m = 16 u + r16, k = 2 q, 2 q + 1)
__device__ __forceinline__ void load_frag8_t(Frag8<4> &f, const double *p, long ld) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        f.v[u].x = p[16 * u];
        f.v[u].y = p[ld + 16 * u];
    }
}

// NIT: 16-row groups of the aux tile that hold real rows (1 for the usual d + m + 1 <= 16, else 4).
// The four waves of a workgroup take four different b and share A_c through LDS.
template <int NIT>
__global__ __launch_bounds__(256, 2) void aux_back_update_kernel(JobGeom g, ChunkPtrs p, int c) {
    constexpr int PITCH = NB + 2;
    __shared__ __attribute__((aligned(16))) double Acs[16 * NIT][PITCH];
    const int item = map_item(p, blockIdx.y);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ngrp = (c + 3) / 4;                    // workgroups per aux tile
    const int at = blockIdx.x / ngrp, b = (blockIdx.x % ngrp) * 4 + wave;
    const long ld = g.ld;
    double *Lit = p.L + (long)item * g.item_stride;
    double *Wa = Lit + ((long)g.n0 + (long)at * NB) * ld;          // aux tile rows
    for (int e = threadIdx.x; e < 16 * NIT * NB; e += 256) {
        const int r = e >> 6, k = e & 63;
        Acs[r][k] = Wa[(long)r * ld + c * NB + k];                 // A_c (aux_back_solve_kernel)
    }
    __syncthreads();
    if (b >= c) return;
    const double *Lcb = Lit + (long)c * NB * ld + (long)b * NB;    // L tile (c, b), read transposed
    const int r16 = lane & 15, q = lane >> 4;
    double acc4[4][NIT][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < NIT; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc4[x][y][r] = 0.0;
    // S'[jj][a] = sum_k L[64 c + k][64 b + jj] A_c[a][k]
    const double *pa = Lcb + (long)(2 * q) * ld + r16;
    Frag8<4> a[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) load_frag8_t(a[s], pa + (long)(8 * s) * ld, ld);   // whole tile in flight
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const f64x2 bv = *reinterpret_cast<const f64x2 *>(&Acs[16 * it + r16][8 * s + 2 * q]);
            const Rot4 bx = rot4(bv.x);
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) mfma16_as_4(acc4[jt][it], a[s].v[jt].x, bx);
            const Rot4 by = rot4(bv.y);
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) mfma16_as_4(acc4[jt][it], a[s].v[jt].y, by);
        }
    }
    // Wa[a][64 b + jj] -= S'[jj][a]
    const int jj0 = 4 * (r16 >> 2) + q;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double *e = Wa + (long)(16 * it + ((r16 + 4 * r) & 15)) * ld + b * NB + 16 * jt + jj0;
                *e -= acc4[jt][it][r];
            }
}

// R[:, col] = X[:, col] - sum_row A[:, row] K[row][col]: one thread per column (or two), the rows of
// A it multiplies are wave-uniform (LDS broadcast), K comes from the lattice tables (or the direct
// interpreter) element by element and is never stored.  One workgroup per (64 CPT columns, item,
// NACC aux rows); rows are walked in slabs of 64 whose A block, times and lattice coordinates sit
// in LDS.
// Three instantiations; on a lattice each item is taken by exactly one of the first two (the other
// returns at once):
//   KA_SINGLE   the item's tree is stationary as a whole = ONE table (DevProgram::rops): no
//               interpreter in the loop, two columns per thread, gathers issued eight rows ahead
//   KA_REDUCED  reduced-program interpreter, one column per thread (kept apart from KA_DIRECT: the
//               transcendental code of the direct interpreter cost it half its occupancy)
//   KA_DIRECT   irregular times: direct evaluation of the full program
enum { KA_SINGLE = 0, KA_REDUCED = 1, KA_DIRECT = 2, KA_CHAIN = 3 };
// Workgroup = 64 CPT columns x 4 row quarters: wave w walks the w-th quarter of the rows for the
// same columns and the four partial sums are added in wave order through LDS.  (One wave walking
// all n0 rows was the critical path: a launch took as long as the item with the longest program.)
template <int NACC, int CPT, int MODE>
__global__ __launch_bounds__(256) void kapply_kernel(JobGeom g, ChunkPtrs p, const double *A,
                                                     const double *X, double *Rout, DevSpec sp) {
    __shared__ DevProgram P;
    __shared__ double As[4][NACC][NB];
    __shared__ double t1s[4][NB];
    __shared__ int q1s[4][NB];
    __shared__ double red[3][CPT][NACC][64];
    const int item = map_item(p, blockIdx.y), tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int a0 = blockIdx.z * NACC;
    const int rows = min(NACC, g.naux - a0);
    if (rows <= 0) return;
    if (blockIdx.x * CPT * 64 >= g.n0) return;    // whole workgroup: the grid is sized for CPT = 1
    load_program(&P, p.progs + item);
    __syncthreads();
    constexpr bool SINGLE = MODE == KA_SINGLE;
    if constexpr (MODE != KA_DIRECT) {
        // workgroup-uniform: which of the three lattice instantiations owns this item
        const bool single = P.n_rops == 1 && P.rops[0] == OP_TABLE;
        const bool chain = !single && P.rchain;
        const int mine = single ? KA_SINGLE : (chain ? KA_CHAIN : KA_REDUCED);
        if (mine != MODE) return;
    }
    const long ld = g.ld;
    const double *Ai = A + ((long)item * g.naux_pad + a0) * ld;
    const double *Xi = X + ((long)item * g.naux_pad + a0) * ld;
    double *Ri = Rout + ((long)item * g.naux_pad + a0) * ld;
    const double *tab = MODE != KA_DIRECT ? p.tab + (long)item * g.maxstat * g.R : nullptr;
    const double *sig = MODE != KA_DIRECT ? p.sig + (long)item * g.maxcp * g.npts : nullptr;
    int col[CPT], q2[CPT];
    double t2[CPT];
    bool live[CPT];
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        const int cidx = (blockIdx.x * CPT + u) * 64 + lane;
        live[u] = cidx < g.n0;
        col[u] = live[u] ? cidx : g.n0 - 1;
        t2[u] = p.t0[col[u]];
        q2[u] = MODE != KA_DIRECT ? p.qpts[col[u]] : 0;
    }
    const double diag = P.noise + sp.jitter;
    double acc[CPT][NACC];
#pragma unroll
    for (int u = 0; u < CPT; ++u)
#pragma unroll
        for (int s = 0; s < NACC; ++s) acc[u][s] = 0.0;
    const int per = (g.nb0 + 3) / 4;              // 64-row slabs per wave
    for (int sl = 0; sl < per; ++sl) {
        const int slab = w * per + sl;
        const bool on = slab < g.nb0;             // wave-uniform
        const int r0 = slab * NB;
        __syncthreads();
        if (on) {
            for (int e = lane; e < NACC * NB; e += 64) {
                const int a = e >> 6, rr = e & 63;
                As[w][a][rr] = (a < rows) ? Ai[(long)a * ld + r0 + rr] : 0.0;
            }
            t1s[w][lane] = p.t0[r0 + lane];
            q1s[w][lane] = MODE != KA_DIRECT ? p.qpts[r0 + lane] : 0;
        }
        __syncthreads();
        if (!on) continue;
        if constexpr (MODE == KA_CHAIN) {
            // chain programs (see fill_chain_kernel): 8 rows per decode, same formulas and order of
            // operations per element as keval_reduced
            static_assert(MODE != KA_CHAIN || CPT == 1, "the chain instantiation is one column per thread");
            const int nops = P.n_rops;
            for (int r8 = 0; r8 < NB; r8 += 8) {
                double v[8];
                for (int i = 0; i < nops; ++i) {
                    const int code = __builtin_amdgcn_readfirstlane((int)P.rops[i]);
                    const int op = code & 15;
                    int lk = code >> 4, lf;
                    if (i == 0) {
                        lk = (op == OP_TABLE) ? RLEAF_TABLE : RLEAF_LINEAR;
                        lf = __builtin_amdgcn_readfirstlane(
                            (int)(op == OP_TABLE ? P.rslot[0] : P.rpoff[0]));
                    } else {
                        lf = __builtin_amdgcn_readfirstlane((int)P.rleaf[i]);
                    }
                    double b[8];
                    if (lk == RLEAF_TABLE) {
                        const double *tb = tab + (long)lf * g.R;
#pragma unroll
                        for (int k = 0; k < 8; ++k) b[k] = tb[abs(q1s[w][r8 + k] - q2[0])];
                    } else {
                        const double cc = P.params[lf], b0 = P.params[lf + 1], b1 = P.params[lf + 2];
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            b[k] = b0 + b1 * (t1s[w][r8 + k] - cc) * (t2[0] - cc);
                    }
                    if (i == 0) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = b[k];
                    } else if (op == NGP_OP_PLUS) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = v[k] + b[k];
                    } else if (op == NGP_OP_TIMES) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = v[k] * b[k];
                    } else {
                        const int slot = __builtin_amdgcn_readfirstlane((int)P.rslot[i]);
                        const double *sg = sig + (long)slot * g.npts;
                        const double g2 = sg[col[0]];
                        const bool fwd = op == NGP_OP_CHANGEPOINT;
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const double g1 = sg[r0 + r8 + k];
                            const double kl = fwd ? v[k] : b[k], kr = fwd ? b[k] : v[k];
                            v[k] = g1 * kl * g2 + (1.0 - g1) * kr * (1.0 - g2);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    double vv = v[k];
                    if (r0 + r8 + k == col[0]) vv += diag;
#pragma unroll
                    for (int s = 0; s < NACC; ++s) acc[0][s] += As[w][s][r8 + k] * vv;
                }
            }
            continue;
        }
#pragma unroll SINGLE ? 8 : 1
        for (int rr = 0; rr < NB; ++rr) {
            const int row = r0 + rr;
#pragma unroll
            for (int u = 0; u < CPT; ++u) {
                double v;
                if constexpr (SINGLE)
                    v = tab[abs(q1s[w][rr] - q2[u])];
                else if constexpr (MODE == KA_REDUCED)
                    v = keval_reduced(P, tab, sig, g.R, g.npts, t1s[w][rr], t2[u],
                                      abs(q1s[w][rr] - q2[u]), row, col[u]);
                else
                    v = keval(P, sp, t1s[w][rr], t2[u]);
                if (row == col[u]) v += diag;
#pragma unroll
                for (int s = 0; s < NACC; ++s) acc[u][s] += As[w][s][rr] * v;
            }
        }
    }
    __syncthreads();
    if (w > 0) {
#pragma unroll
        for (int u = 0; u < CPT; ++u)
#pragma unroll
            for (int s = 0; s < NACC; ++s) red[w - 1][u][s][lane] = acc[u][s];
    }
    __syncthreads();
    if (w > 0) return;
#pragma unroll
    for (int u = 0; u < CPT; ++u)
        if (live[u]) {
#pragma unroll
            for (int s = 0; s < NACC; ++s)   // static index: a runtime bound sends acc[] to scratch
                if (s < rows) {
                    const double sum = ((acc[u][s] + red[0][u][s][lane]) + red[1][u][s][lane]) +
                                       red[2][u][s][lane];
                    Ri[(long)s * ld + col[u]] = Xi[(long)s * ld + col[u]] - sum;
                }
        }
}

// S[a][b] = A_a . X_b + R_a . A_b and T[a][b] = R_a . A_b for all naux^2 pairs, one wave per pair
__global__ __launch_bounds__(256) void refine_gram_pairs_kernel(JobGeom g, const double *A,
                                                                const double *X, const double *R,
                                                                double *S, double *T, double *U,
                                                                const int32_t *items) {
    const int item = items ? items[blockIdx.y] : blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pr = blockIdx.x * 4 + wave;
    if (pr >= g.naux * g.naux) return;
    const int a = pr / g.naux, b = pr % g.naux;
    const long base = (long)item * g.naux_pad * g.ld;
    const double *Aa = A + base + (long)a * g.ld, *Ab = A + base + (long)b * g.ld;
    const double *Xb = X + base + (long)b * g.ld, *Ra = R + base + (long)a * g.ld;
    double s0 = 0.0, t0 = 0.0, rr = 0.0, xx = 0.0;
    for (int k = lane; k < g.n0; k += 64) {
        s0 += Aa[k] * Xb[k];
        t0 += Ra[k] * Ab[k];
        if (a == b) {   // wave-uniform: |R_a|^2 and |X_a|^2 for the contraction estimate
            rr += Ra[k] * Ra[k];
            xx += Xb[k] * Xb[k];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_down(s0, off, 64);
        t0 += __shfl_down(t0, off, 64);
        rr += __shfl_down(rr, off, 64);
        xx += __shfl_down(xx, off, 64);
    }
    if (lane == 0) {
        S[(long)item * g.naux * g.naux + pr] = s0 + t0;
        T[(long)item * g.naux * g.naux + pr] = t0;
        if (a == b) U[(long)item * g.naux + a] = xx > 0.0 ? rr / xx : 0.0;
    }
}

// G = (S + S') / 2;  delta[2 item] = max |T + T'| / 2 relative to sqrt(G_aa G_bb), the size of the
// correction this step applied;  delta[2 item + 1] = max_a |R_a| / |X_a|, how far (L L')^-1 K is
// from the identity along the rows of X (the factor by which the next correction is smaller)
__global__ __launch_bounds__(256) void refine_gram_final_kernel(JobGeom g, const double *S,
                                                                const double *T, const double *U,
                                                                double *G, double *delta,
                                                                const int32_t *items) {
    __shared__ double red[256];
    const int item = items ? items[blockIdx.x] : blockIdx.x;
    const int tid = threadIdx.x, na = g.naux;
    const double *Si = S + (long)item * na * na, *Ti = T + (long)item * na * na;
    double *Gi = G + (long)item * na * na;
    double dm = 0.0;
    for (int e = tid; e < na * na; e += 256) {
        const int a = e / na, b = e % na;
        Gi[e] = 0.5 * (Si[a * na + b] + Si[b * na + a]);
        const double gaa = Si[a * na + a], gbb = Si[b * na + b];
        const double den = sqrt(fabs(gaa * gbb));
        const double tv = 0.5 * fabs(Ti[a * na + b] + Ti[b * na + a]);
        // fmax drops NaN: a non-finite correction must come out as NaN (the item is then
        // never marked refined), so it is carried explicitly
        if (!(tv == tv) || !(den == den)) dm = NAN;
        else if (dm == dm) {
            if (den > 0.0) dm = fmax(dm, tv / den);
            else if (tv > 0.0) dm = INFINITY;
        }
    }
    red[tid] = dm;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            const double a = red[tid], b = red[tid + off];
            red[tid] = (a == a && b == b) ? fmax(a, b) : NAN;
        }
        __syncthreads();
    }
    if (tid == 0) {
        delta[2 * item] = red[0];
        double rho2 = 0.0;
        for (int a = 0; a < na; ++a) {
            const double u = U[(long)item * na + a];
            rho2 = (u == u && rho2 == rho2) ? fmax(rho2, u) : NAN;
        }
        delta[2 * item + 1] = sqrt(rho2);
    }
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
    if (g.n0 <= 512) {
        // short rows: one pair per thread, a serial dot product over at most 512 elements that sit
        // in L1 (the wave-per-pair form below spends its time on index arithmetic and shuffles: 46 us
        // for 276 pairs at n0 = 128)
        for (int e = threadIdx.x; e < g.naux * g.naux; e += 256) {
            const int a = e / g.naux, b = e % g.naux;
            if (b > a) continue;
            const double *wa = W + (long)a * g.ld, *wb = W + (long)b * g.ld;
            double s0 = 0.0, s1 = 0.0;
            for (int k = 0; k < g.n0; k += 2) {
                s0 += wa[k] * wb[k];
                s1 += wa[k + 1] * wb[k + 1];
            }
            const double sv = s0 + s1;
            Go[a * g.naux + b] = sv;
            Go[b * g.naux + a] = sv;
        }
        return;
    }
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
// The working set (L_A, V_A, log diag) lives in dynamic LDS when it fits (lds_work != 0; the usual
// case: a ragged tail of < 64 points plus a few appended ones) — the pivot loop of the da x da
// factorisation is a chain of dependent accesses, 114 us from global memory against ~20 from LDS
// at da = 22 — and in the per-item global work buffer otherwise.
__global__ __launch_bounds__(64) void epilogue_kernel(JobGeom g, EpiPtrs p, DevSpec sp, int lds_work) {
    __shared__ DevProgram P;
    __shared__ int bad;
    extern __shared__ double epi_dyn[];
    const int item = blockIdx.x, tid = threadIdx.x;
    load_program(&P, p.progs + item);
    if (tid == 0) bad = 0;
    __syncthreads();
    const int da = g.da, m = g.m, na = g.naux, Y = da + m;
    const double nz = P.noise + sp.jitter;
    const double *G = p.G + (long)item * na * na;
    double *work = lds_work ? epi_dyn : p.work + (long)item * p.work_stride;
    // k(aux point u, aux point v): table lookups when the item's lattice tables are at hand
    const double *tab = p.tab ? p.tab + (long)item * g.maxstat * g.R : nullptr;
    const double *sig = p.sig ? p.sig + (long)item * g.maxcp * g.npts : nullptr;
    auto kaux = [&](int u, int v) -> double {   // u, v index taux: appended points then forecast points
        if (tab)
            return keval_reduced(P, tab, sig, g.R, g.npts, p.taux[u], p.taux[v],
                                 abs(p.qpts[g.n0 + u] - p.qpts[g.n0 + v]), g.n0 + u, g.n0 + v);
        return keval(P, sp, p.taux[u], p.taux[v]);
    };
    double *LA = work;                  // [da x da]
    double *VA = LA + (long)da * da;    // [m x da]
    double *ldA = VA + (long)m * da;    // [da] log diag L_A

    for (int e = tid; e < da * da; e += 64) {
        const int a = e / da, b = e % da;
        double v = 0.0;
        if (b <= a) {
            v = kaux(a, b) - (g.n0 ? G[a * na + b] : 0.0);
            if (a == b) v += nz;
        }
        LA[e] = v;
    }
    __syncthreads();
    for (int k = 0; k < da; ++k) {  // in-place right-looking Cholesky of S_AA (one wave: row per thread)
        const double akk = LA[k * da + k];
        const double dk = sqrt(akk);
        __syncthreads();
        if (tid == 0) {
            if (!(akk > 0.0) && bad == 0) bad = k + 1;
            LA[k * da + k] = dk;
        }
        for (int i = k + 1 + tid; i < da; i += 64) LA[i * da + k] /= dk;
        __syncthreads();
        for (int i = k + 1 + tid; i < da; i += 64) {
            const double lik = LA[i * da + k];
            for (int jj = k + 1; jj <= i; ++jj) LA[i * da + jj] -= lik * LA[jj * da + k];
        }
        __syncthreads();
    }
    for (int a = tid; a < da; a += 64) ldA[a] = log(LA[a * da + a]);
    __syncthreads();
    // V_A: one forecast row per thread, forward substitution along the appended points
    for (int i = tid; i < m; i += 64) {
        for (int a = 0; a < da; ++a) {
            double s = kaux(da + i, a) - (g.n0 ? G[(da + i) * na + a] : 0.0);
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
            double s = kaux(da + i, da + jj) - (g.n0 ? G[(da + i) * na + da + jj] : 0.0);
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
    if (g.D <= 8) {
        // few scenarios (a logml / predict call): the wave solves them one after the other, every
        // row's dot product spread over the lanes, instead of one long serial loop on one lane
        for (int s = 0; s < g.D; ++s) {
            const double *ya = ya_base + (long)s * da;
            double *z = p.zbuf + ((long)item * g.D + s) * da;
            double quad = 0.0, quad_tail = 0.0, ldsum = 0.0, ld_tail = 0.0;
            for (int a = 0; a < da; ++a) {
                double part = 0.0;
                for (int pp = tid; pp < a; pp += 64) part += LA[a * da + pp] * z[pp];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
                const double e = (ya[a] - (g.n0 ? G[a * na + Y] : 0.0) - part) / LA[a * da + a];
                if (tid == 0) z[a] = e;
                __syncthreads();
                quad += e * e;
                ldsum += ldA[a];
                if (a < g.tail) { quad_tail += e * e; ld_tail += ldA[a]; }
            }
            const int nfull = g.n0 + da;
            if (tid == 0) {
                p.logml_full[(long)item * g.D + s] =
                    -0.5 * (q0 + quad) - (ld0 + ldsum) - 0.5 * nfull * LOG2PI;
                if (s == 0)
                    p.logml_base[item] =
                        -0.5 * (q0 + quad_tail) - (ld0 + ld_tail) - 0.5 * (g.n0 + g.tail) * LOG2PI;
            }
            if (p.mu) {
                double *mu = p.mu + ((long)item * g.D + s) * m;
                for (int i = tid; i < m; i += 64) {
                    double v = g.n0 ? G[(da + i) * na + Y] : 0.0;
                    for (int a = 0; a < da; ++a) v += VA[i * da + a] * z[a];
                    mu[i] = v;
                }
            }
        }
    } else {
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
    }
    if (tid == 0 && bad && p.info[item] == 0) p.info[item] = g.n0 + bad;
}

// ---------------------------------------------------------------------------------------
// gradient of the log marginal likelihood (HMC inside fit_smc! / mcmc_parameters!)
//   d logml / d theta_p = 1/2 sum_ij (alpha_i alpha_j - Kinv_ij) dK_ij / d theta_p
// The factorisation above ran with aux rows [I ; y'], so the aux block is W = [L^-T ; z'] and
//   Kinv = W_I W_I'  (MFMA Gram, upper-triangular W: k starts at the row tile),  alpha = W_I z.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void grad_kinv_kernel(JobGeom g, const double *L,
                                                           double *Kinv, int npairs) {
    const int item = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pr = blockIdx.x * 4 + wave;
    if (pr >= npairs) return;
    int I = (int)((sqrt(8.0 * pr + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= pr) ++I;
    while (I * (I + 1) / 2 > pr) --I;
    const int J = pr - I * (I + 1) / 2;   // I >= J
    const long ld = g.ld;
    const double *W = L + (long)item * g.item_stride + (long)g.n0 * ld;
    const int r16 = lane & 15, q = lane >> 4;
    double acc4[4][4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc4[a][b][r] = 0.0;
    // S'[jj][i] = sum_k W[64J + jj][k] W[64I + i][k]; W[a][k] = 0 for k < a, so k >= 64 I
    const double *pa = W + (long)(J * NB + r16) * ld + 2 * q;
    const double *pb = W + (long)(I * NB + r16) * ld + 2 * q;
    gemm_rows<4>(acc4, pa, pb, ld, I * NB, g.n0);
    double *Ko = Kinv + (long)item * g.n0 * g.n0;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const f64x4 d = to_d16(acc4[jt][it]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
                Ko[(long)(I * NB + 16 * it + r16) * g.n0 + J * NB + 16 * jt + q + 4 * s] = d[s];
        }
}

// K^-1 = W W' for long series: a workgroup takes a 2 x 2 block of 64 x 64 tiles and stages the
// four row tiles it needs (column tiles J0, J0+1 as the "panel", row tiles I0, I0+1) through LDS by
// LDS-DMA, 16 columns at a time, exactly as the fat step of the factorisation does (same layout,
// same swizzle, same mfma loop) — the wave-per-tile kernel above re-reads both row tiles of every
// tile from HBM (24.6 GB per call at n = 2048 x 64 items, 5.5 TB/s: it was bound by that).  k
// starts at 64 I0 for both row tiles; for I0 + 1 the first 64 columns are zeros of W (upper
// triangular), which add nothing.  Within a 16-column chunk the MFMAs take k in ascending groups
// of four (the fat step's order) where gemm_rows takes even then odd k of an 8-column stage: the
// two kernels agree to rounding, not bit for bit.  Block pairs with bi >= bj; the tile above the
// diagonal in a diagonal block is computed and dropped.
// Grid: 1-D, workgroups b and b + 8 share an XCD (round-robin dispatch), and all blocks of an item
// go to one XCD: the 136 blocks of an item at n = 2048 read its 18 MB of W thirteen times over
// (PMC: 200 MB of fetches per item, 3.2 TB/s) and only an L2 they share can absorb that.
__global__ __launch_bounds__(256, 2) void grad_kinv_lds_kernel(JobGeom g, const double *L,
                                                               double *Kinv, int nblk, int Bc,
                                                               double *alpha) {
    constexpr int ROWB = 128, BLKB = 8 * ROWB + 128, STAGE = 32 * BLKB;
    auto row_off = [](int row) { return (row >> 3) * BLKB + (row & 7) * ROWB; };
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const int wg = blockIdx.x;
    const int xcd = wg & 7, idx = wg >> 3;
    const int item = (idx / nblk) * 8 + xcd;
    const int pr = idx % nblk;
    if (item >= Bc) return;
    int bi = (int)((sqrt(8.0 * pr + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= pr) ++bi;
    while (bi * (bi + 1) / 2 > pr) --bi;
    const int bj = pr - bi * (bi + 1) / 2;   // bi >= bj
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ltile = wave >> 1, col = wave & 1;
    const int I0 = 2 * bi, J0 = 2 * bj;
    const int I = I0 + ltile, J = J0 + col;
    // a last, unpaired tile (odd nb0) is staged as a copy of its neighbour and not stored
    const bool valid = I < g.nb0 && J < g.nb0 && I >= J;
    const long ld = g.ld;
    const double *Wb = L + (long)item * g.item_stride + (long)g.n0 * ld;
    const int r16 = lane & 15, q = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(Wb), 0, (int)((long)g.n0 * ld * (long)sizeof(double)), 0x00020000);
    // stage rows: waves 0,1 the panel (column tiles J0, J0+1), waves 2,3 the row tiles I0, I0+1
    int src_tile = wave < 2 ? J0 + wave : I0 + (wave - 2);
    if (src_tile >= g.nb0) src_tile = g.nb0 - 1;
    const unsigned soff_base =
        (unsigned)__builtin_amdgcn_readfirstlane((int)((long)src_tile * NB * ld * 8));
    const unsigned row_step8 = (unsigned)(8 * ld * 8);
    const unsigned voff_even = (unsigned)(((lane >> 3) * ld + 2 * ((lane & 7) ^ ((lane >> 4) & 7))) * 8);
    const unsigned voff_odd = (unsigned)(((lane >> 3) * ld + 2 * ((lane & 7) ^ ((4 + (lane >> 4)) & 7))) * 8);
    auto stage = [&](int buf, int k) {
        const unsigned kb = (unsigned)k * 8u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            lds_ptr dst = (lds_ptr)(smem + buf * STAGE + (8 * wave + i) * BLKB);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, (i & 1) ? voff_odd : voff_even,
                                                     soff_base + i * row_step8 + kb, 0, 0);
        }
    };
    unsigned a_addr[4], b_addr[4][4];
    {
        const int arow = 64 * col + r16;
        const int akey = (r16 >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            a_addr[s] = (unsigned)(row_off(arow) + (((2 * s + (q >> 1)) ^ akey) << 4) + (q & 1) * 8);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = (r16 + 4 * r) & 15;
            const int brow = 128 + 64 * ltile + rr;
            const int bkey = (rr >> 1) & 7;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                b_addr[r][s] =
                    (unsigned)(row_off(brow) + (((2 * s + (q >> 1)) ^ bkey) << 4) + (q & 1) * 8);
        }
    }
    double acc4[4][4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc4[a][b][r] = 0.0;
    const int kbeg = I0 * NB;
    // columns beyond the real points are the identity padding of the last block: the rows of W that
    // matter are zero there, so the sum stops at the chunk that holds the last real column
    const int kend = min(g.n0, (g.n_real + LDS_KC - 1) / LDS_KC * LDS_KC);
    const int nchunks = max(kend - kbeg, LDS_KC) / LDS_KC;
    // Products that are known to be nothing are not issued (the wave still stages its rows and keeps
    // the barriers; its SIMD's other wave gets the matrix pipe): a wave whose tile lies above the
    // diagonal of a diagonal block or beyond the last tile, and — W being block upper triangular —
    // the first 64 columns of the k-range for the waves of row tile I0 + 1, whose rows are the
    // stored zeros of block (I0 + 1, I0) there.  Adding those zero products changed no bit.
    const int skip_chunks = __builtin_amdgcn_readfirstlane(!valid ? nchunks : (ltile == 1 ? NB / LDS_KC : 0));
    // alpha = W_I z for the rows of this block row's two row tiles, from the rows the workgroup
    // stages anyway (block pairs with bj = 0: one per block row; W[a][k] = 0 left of a's block
    // column, so the k-range of the block pair is the whole sum): thread (row, half) takes eight
    // of a chunk's sixteen columns.  The separate kernel read every row of W once more from HBM
    // (18.5 MB per item) beside this one and cost it 22 of its 713 ms.
    const bool do_alpha = alpha != nullptr && bj == 0;   // workgroup-uniform
    const int arow = tid >> 1, ahalf = tid & 1;          // LDS row 128 + arow: row arow of (I0, I0 + 1)
    const double *zrow = Wb + (long)g.n0 * ld;           // the data row of W
    const unsigned a_off = (unsigned)row_off(128 + arow);
    const int a_key = (arow >> 1) & 7;
    double asum = 0.0;
    stage(0, kbeg);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int cur = c & 1;
        if (c + 1 < nchunks) stage(cur ^ 1, kbeg + (c + 1) * LDS_KC);
        const char *buf = smem + cur * STAGE;
        // (the rows of tile I0 + 1 start at their own block column: what lies left of it is never
        // written — wave-uniform: waves 2, 3 hold those rows)
        if (do_alpha && (arow < NB || c >= NB / LDS_KC)) {
            const double *zc = zrow + kbeg + c * LDS_KC + 8 * ahalf;
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                const f64x2 w = *reinterpret_cast<const f64x2 *>(buf + a_off + (((4 * ahalf + pp) ^ a_key) << 4));
                const f64x2 zz = *reinterpret_cast<const f64x2 *>(zc + 2 * pp);
                asum = fma(w.x, zz.x, asum);
                asum = fma(w.y, zz.y, asum);
            }
        }
        if (c >= skip_chunks) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                double a[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    a[u] = *reinterpret_cast<const double *>(buf + a_addr[s] + u * 2 * BLKB);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    Rot4 br;
                    br.r0 = *reinterpret_cast<const double *>(buf + b_addr[0][s] + it * 2 * BLKB);
                    br.r1 = *reinterpret_cast<const double *>(buf + b_addr[1][s] + it * 2 * BLKB);
                    br.r2 = *reinterpret_cast<const double *>(buf + b_addr[2][s] + it * 2 * BLKB);
                    br.r3 = *reinterpret_cast<const double *>(buf + b_addr[3][s] + it * 2 * BLKB);
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) mfma16_as_4(acc4[jt][it], a[jt], br);
                }
            }
        }
        __syncthreads();
    }
    if (do_alpha) {
        asum += __shfl_xor(asum, 1, 64);
        const int trow = (I0 + (arow >> 6)) * NB + (arow & 63);
        if (ahalf == 0 && I0 + (arow >> 6) < g.nb0) alpha[(long)item * g.n0 + trow] = asum;
    }
    if (!valid) return;
    // The tile leaves as full 512-byte rows: sixteen rows at a time through a per-wave LDS tile (the
    // stage buffers are free: every wave passed the loop's last barrier), 32 store instructions of
    // 1 KiB instead of 64 that scatter 32-byte pieces over sixteen rows each
    // (profiles/r04/kinv_experiments.txt: the stores were 30 of the kernel's 710 ms).
    double *Ko = Kinv + (long)item * g.n0 * g.n0 + (long)(I * NB) * g.n0 + J * NB;
    constexpr int PITCH = NB + 2;   // doubles: rows stay 16-byte aligned, row groups on distinct banks
    double *tl = reinterpret_cast<double *>(smem) + wave * (16 * PITCH);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            const f64x4 d = to_d16(acc4[jt][it]);
#pragma unroll
            for (int s = 0; s < 4; ++s) tl[r16 * PITCH + 16 * jt + q + 4 * s] = d[s];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 2 * i + (lane >> 5), c2 = 2 * (lane & 31);
            const f64x2 v = *reinterpret_cast<const f64x2 *>(tl + row * PITCH + c2);
            *reinterpret_cast<f64x2 *>(Ko + (long)(16 * it + row) * g.n0 + c2) = v;
        }
    }
}

// alpha[a] = sum_k W[a][k] z[k] (z = the data row of W), quad = z'z; one wave per row
__global__ __launch_bounds__(256) void grad_alpha_kernel(JobGeom g, const double *L, double *alpha,
                                                         double *quad, int a_first) {
    const int item = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a = a_first + blockIdx.x * 4 + wave;   // a == n0: the quadratic form
    if (a > g.n0) return;
    const double *W = L + (long)item * g.item_stride + (long)g.n0 * g.ld;
    const double *wa = W + (long)a * g.ld, *z = W + (long)g.n0 * g.ld;
    // W is block upper triangular and what lies left of a row's diagonal block is never written
    // (nor read): row a starts at its own block column
    double s = 0.0;
    for (int k = (a < g.n0 ? (a / NB) * NB : 0) + lane; k < g.n0; k += 64) s += wa[k] * z[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) {
        if (a < g.n0) alpha[(long)item * g.n0 + a] = s;
        else quad[item] = s;
    }
}

// Reverse-mode sweep of the kernel tree per matrix element, contracted with
// w_ij = alpha_i alpha_j - Kinv_ij (lower triangle; the diagonal carries 1/2).
__global__ __launch_bounds__(256) void grad_contract_kernel(JobGeom g, const DevProgram *progs,
                                                            const double *t0, const double *Kinv,
                                                            const double *alpha, double *partials,
                                                            int ntri, DevSpec sp) {
    __shared__ DevProgram P;
    __shared__ double red[4][NGP_MAX_PARAMS + 1];
    const int item = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    load_program(&P, progs + item);
    __syncthreads();
    int r = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= tile) ++r;
    while (r * (r + 1) / 2 > tile) --r;
    const int c = tile - r * (r + 1) / 2;
    const int tx = tid & 63, ty = tid >> 6;
    const int col = c * NB + tx;
    const int np = P.n_params, nops = P.n_ops;
    double gacc[NGP_MAX_PARAMS + 1];
    for (int i = 0; i <= np; ++i) gacc[i] = 0.0;
    const double *Ki = Kinv + (long)item * g.n0 * g.n0;
    const double *al = alpha + (long)item * g.n0;
    if (col < g.n_real) {
        const double t2 = t0[col], ac = al[col];
        for (int rr = 0; rr < 16; ++rr) {
            const int row = r * NB + ty * 16 + rr;
            if (row >= g.n_real || col > row) continue;
            double w = al[row] * ac - Ki[(long)row * g.n0 + col];
            if (row == col) w *= 0.5;
            const double t1 = t0[row];
            const double d = fabs(t1 - t2);
            // ---- forward sweep: value of every node
            double val[NGP_MAX_OPS];
            for (int i = 0; i < nops; ++i) {
                const int op = P.ops[i], po = P.poff[i];
                double v;
                if (op == NGP_OP_CONSTANT) v = P.params[po];
                else if (op == NGP_OP_LINEAR)
                    v = P.params[po + 1] + P.params[po + 2] * (t1 - P.params[po]) * (t2 - P.params[po]);
                else if (op == NGP_OP_SQEXP) {
                    const double l = P.params[po];
                    v = P.params[po + 1] * exp(-0.5 * d * d / (sp.se_form ? l : l * l));
                } else if (op == NGP_OP_GAMMAEXP)
                    v = P.params[po + 2] * exp(-pow(d / P.params[po], P.params[po + 1]));
                else if (op == NGP_OP_PERIODIC) {
                    const double l = P.params[po], sn = sin(M_PI * d / P.params[po + 1]);
                    v = P.params[po + 2] * exp(-(sp.periodic_form ? 2.0 / l : 2.0 / (l * l)) * sn * sn);
                } else {
                    const double x = val[P.first[i]], y = val[i - 1];   // first-evaluated, second
                    if (op == NGP_OP_PLUS) v = x + y;
                    else if (op == NGP_OP_TIMES) v = x * y;
                    else {
                        const double kl = (op == NGP_OP_CHANGEPOINT) ? x : y;
                        const double kr = (op == NGP_OP_CHANGEPOINT) ? y : x;
                        const double g1 = cp_sigma(sp.cp_form, t1, P.params[po], P.params[po + 1]);
                        const double g2 = cp_sigma(sp.cp_form, t2, P.params[po], P.params[po + 1]);
                        v = g1 * kl * g2 + (1.0 - g1) * kr * (1.0 - g2);
                    }
                }
                val[i] = v;
            }
            // ---- reverse sweep: adjoint stack mirrors the evaluation stack
            double s0 = w, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
            for (int i = nops - 1; i >= 0; --i) {
                const int op = P.ops[i], po = P.poff[i];
                const double a = s0;
                s0 = s1; s1 = s2; s2 = s3; s3 = s4; s4 = s5; s5 = s6; s6 = s7;   // pop
                if (op == NGP_OP_CONSTANT) {
                    gacc[po] += a;
                } else if (op == NGP_OP_LINEAR) {
                    const double cc = P.params[po], a1 = t1 - cc, a2 = t2 - cc;
                    gacc[po] += a * P.params[po + 2] * (-a1 - a2);
                    gacc[po + 1] += a;
                    gacc[po + 2] += a * a1 * a2;
                } else if (op == NGP_OP_SQEXP) {
                    const double l = P.params[po], am = P.params[po + 1];
                    const double e = exp(-0.5 * d * d / (sp.se_form ? l : l * l));
                    gacc[po] += a * (sp.se_form ? am * e * 0.5 * d * d / (l * l)
                                                : am * e * d * d / (l * l * l));
                    gacc[po + 1] += a * e;
                } else if (op == NGP_OP_GAMMAEXP) {
                    const double l = P.params[po], gm = P.params[po + 1], am = P.params[po + 2];
                    const double rr_ = d / l, u = pow(rr_, gm), e = exp(-u);
                    gacc[po] += a * am * e * gm * u / l;
                    gacc[po + 1] += (d > 0.0) ? -a * am * e * u * log(rr_) : 0.0;
                    gacc[po + 2] += a * e;
                } else if (op == NGP_OP_PERIODIC) {
                    const double l = P.params[po], per = P.params[po + 1], am = P.params[po + 2];
                    const double ang = M_PI * d / per, sn = sin(ang), cs = cos(ang);
                    const double cq = sp.periodic_form ? 2.0 / l : 2.0 / (l * l);
                    const double e = exp(-cq * sn * sn);
                    gacc[po] += a * (sp.periodic_form ? am * e * 2.0 * sn * sn / (l * l)
                                                      : am * e * 4.0 * sn * sn / (l * l * l));
                    gacc[po + 1] += a * am * e * cq * 2.0 * sn * cs * M_PI * d / (per * per);
                    gacc[po + 2] += a * e;
                } else {
                    const double x = val[P.first[i]], y = val[i - 1];
                    double ax, ay;   // adjoints of the first-evaluated and the second operand
                    if (op == NGP_OP_PLUS) {
                        ax = a; ay = a;
                    } else if (op == NGP_OP_TIMES) {
                        ax = a * y; ay = a * x;
                    } else {
                        const bool nat = (op == NGP_OP_CHANGEPOINT);
                        const double kl = nat ? x : y, kr = nat ? y : x;
                        const double loc = P.params[po], sc = P.params[po + 1];
                        const double sgn = sp.cp_form ? 1.0 : -1.0;   // u = sgn (t - loc) / sc
                        const double u1 = sgn * (t1 - loc) / sc, u2 = sgn * (t2 - loc) / sc;
                        const double th1 = tanh(u1), th2 = tanh(u2);
                        const double g1 = 0.5 * (1.0 + th1), g2 = 0.5 * (1.0 + th2);
                        const double q1 = 0.5 * (1.0 - th1 * th1), q2 = 0.5 * (1.0 - th2 * th2);
                        const double d1l = q1 * (-sgn / sc), d2l = q2 * (-sgn / sc);
                        const double d1s = q1 * (-u1 / sc), d2s = q2 * (-u2 / sc);
                        gacc[po] += a * (d1l * kl * g2 + g1 * kl * d2l - d1l * kr * (1.0 - g2) -
                                         (1.0 - g1) * kr * d2l);
                        gacc[po + 1] += a * (d1s * kl * g2 + g1 * kl * d2s - d1s * kr * (1.0 - g2) -
                                             (1.0 - g1) * kr * d2s);
                        const double al_ = a * g1 * g2, ar_ = a * (1.0 - g1) * (1.0 - g2);
                        ax = nat ? al_ : ar_;
                        ay = nat ? ar_ : al_;
                    }
                    // push: the second operand (root at i-1) is visited next, so it goes on top
                    s7 = s5; s6 = s4; s5 = s3; s4 = s2; s3 = s1; s2 = s0; s1 = ax; s0 = ay;
                }
            }
            if (row == col) gacc[np] += w;   // d K / d noise = I (w already carries the 1/2)
        }
    }
    // ---- deterministic reduction: wave shuffles, then the four waves in order
    const int lane = tid & 63, wave = tid >> 6;
    for (int pidx = 0; pidx <= np; ++pidx) {
        double v = gacc[pidx];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wave][pidx] = v;
    }
    __syncthreads();
    if (tid <= np)
        partials[((long)item * ntri + tile) * (NGP_MAX_PARAMS + 1) + tid] =
            red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// The same contraction on lattice times: every transcendental of the tree comes from the per-item
// tables (tab / dtab by integer distance, sig by point), so the n^2/2 element loop is lookups and
// FMAs only.  ChangePoint: sigma = (1 + tanh u)/2 gives d sigma / du = 2 sigma (1 - sigma).
// LDSV: the node values of the forward sweep live in LDS (one column per thread) instead of a
// runtime-indexed private array, which hipcc puts in scratch — the kernel is bound by that scratch
// traffic (3.93 -> 2.96 ms at n = 2048, 64 items).  Needs programs of at most LDSV_OPS operators;
// the launcher falls back to the private-array instantiation otherwise.
constexpr int LDSV_OPS = 16;
template <bool LDSV>
__global__ __launch_bounds__(256) void grad_contract_lattice_kernel(JobGeom g, ChunkPtrs p,
                                                                    const double *Kinv,
                                                                    const double *alpha,
                                                                    double *partials, int ntri,
                                                                    int split, DevSpec sp,
                                                                    const int32_t *items) {
    __shared__ DevProgram P;
    __shared__ double red[4][NGP_MAX_PARAMS + 1];
    // split: workgroups per 64x64 tile (1, 2 or 4).  A thread walks 16 / split rows; a small
    // launch (few items, short series) is latency-bound on that walk, so it is cut into more,
    // shorter workgroups (158 -> 60 us for 64 particles at n = 150).
    const int item = items ? items[blockIdx.y] : (int)blockIdx.y;
    const int tile = blockIdx.x / split, sub = blockIdx.x % split;
    const int tid = threadIdx.x;
    const int nrows = 16 / split;
    load_program(&P, p.progs + item);
    __syncthreads();
    // per-operator constants of the derivative formulas, once per workgroup: the element loop
    // below then has no fp64 division (twelve of them per element before)
    __shared__ double cst[NGP_MAX_OPS][2];
    __shared__ double vals[LDSV ? LDSV_OPS : 1][256];
    for (int i = tid; i < P.n_ops; i += 256) {
        const int op = P.ops[i], po = P.poff[i];
        double c0 = 0.0, c1 = 0.0;
        if (op == NGP_OP_SQEXP) {
            const double l = P.params[po], am = P.params[po + 1];
            c0 = am * (sp.se_form ? 0.5 / (l * l) : 1.0 / (l * l * l));
        } else if (op == NGP_OP_GAMMAEXP) {
            c0 = P.params[po + 2] * P.params[po + 1] / P.params[po];
            c1 = P.params[po + 2];
        } else if (op == NGP_OP_PERIODIC) {
            const double l = P.params[po], per = P.params[po + 1], am = P.params[po + 2];
            const double cq = sp.periodic_form ? 2.0 / l : 2.0 / (l * l);
            c0 = am * (sp.periodic_form ? 2.0 / (l * l) : 4.0 / (l * l * l));
            c1 = am * cq * 2.0 * M_PI / (per * per);
        } else if (op == NGP_OP_CHANGEPOINT || op == OP_CP_SWAPPED) {
            c1 = 1.0 / P.params[po + 1];
            c0 = sp.cp_form ? c1 : -c1;            // u = c0 (t - loc)
        }
        cst[i][0] = c0;
        cst[i][1] = c1;
    }
    __syncthreads();
    int r = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= tile) ++r;
    while (r * (r + 1) / 2 > tile) --r;
    const int c = tile - r * (r + 1) / 2;
    const int tx = tid & 63, ty = tid >> 6;
    const int col = c * NB + tx;
    const int np = P.n_params, nops = P.n_ops;
    const int R = g.R, npts = g.npts;
    const double *tab = p.tab + (long)item * g.maxstat * R;
    const double *dt = p.dtab + (long)item * g.maxstat * 3 * R;
    const double *sig = p.sig + (long)item * g.maxcp * npts;
    double gacc[NGP_MAX_PARAMS + 1];
    for (int i = 0; i <= np; ++i) gacc[i] = 0.0;
    const double *Ki = Kinv + (long)item * g.n0 * g.n0;
    const double *al = alpha + (long)item * g.n0;
    if (col < g.n_real) {
        const double t2 = p.t0[col], ac = al[col];
        const int q2 = p.qpts[col];
        for (int rr = 0; rr < nrows; ++rr) {
            const int row = r * NB + ty * 16 + sub * nrows + rr;
            if (row >= g.n_real || col > row) continue;
            double w = al[row] * ac - Ki[(long)row * g.n0 + col];
            if (row == col) w *= 0.5;
            const double t1 = p.t0[row];
            const double d = fabs(t1 - t2);
            const int dq = abs(p.qpts[row] - q2);
            // ---- forward sweep: value of every node
            double vloc[LDSV ? 1 : NGP_MAX_OPS];
            auto val = [&](int i) -> double & { return LDSV ? vals[i][tid] : vloc[i]; };
            for (int i = 0; i < nops; ++i) {
                const int op = P.ops[i], po = P.poff[i];
                double v;
                if (op == NGP_OP_CONSTANT) v = P.params[po];
                else if (op == NGP_OP_LINEAR)
                    v = P.params[po + 1] + P.params[po + 2] * (t1 - P.params[po]) * (t2 - P.params[po]);
                else if (op < NGP_OP_PLUS) v = tab[(long)P.slot[i] * R + dq];
                else {
                    const double x = val(P.first[i]), y = val(i - 1);   // first-evaluated, second
                    if (op == NGP_OP_PLUS) v = x + y;
                    else if (op == NGP_OP_TIMES) v = x * y;
                    else {
                        const double kl = (op == NGP_OP_CHANGEPOINT) ? x : y;
                        const double kr = (op == NGP_OP_CHANGEPOINT) ? y : x;
                        const double g1 = sig[(long)P.slot[i] * npts + row];
                        const double g2 = sig[(long)P.slot[i] * npts + col];
                        v = g1 * kl * g2 + (1.0 - g1) * kr * (1.0 - g2);
                    }
                }
                val(i) = v;
            }
            // ---- reverse sweep: adjoint stack mirrors the evaluation stack
            double s0 = w, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
            for (int i = nops - 1; i >= 0; --i) {
                const int op = P.ops[i], po = P.poff[i];
                const double a = s0;
                s0 = s1; s1 = s2; s2 = s3; s3 = s4; s4 = s5; s5 = s6; s6 = s7;   // pop
                if (op == NGP_OP_CONSTANT) {
                    gacc[po] += a;
                } else if (op == NGP_OP_LINEAR) {
                    const double cc = P.params[po], a1 = t1 - cc, a2 = t2 - cc;
                    gacc[po] += a * P.params[po + 2] * (-a1 - a2);
                    gacc[po + 1] += a;
                    gacc[po + 2] += a * a1 * a2;
                } else if (op < NGP_OP_PLUS) {
                    const double *d0 = dt + (long)P.slot[i] * 3 * R + dq;
                    const double e = d0[0];
                    const double c0 = cst[i][0], c1 = cst[i][1];
                    if (op == NGP_OP_SQEXP) {
                        gacc[po] += a * e * d * d * c0;
                        gacc[po + 1] += a * e;
                    } else if (op == NGP_OP_GAMMAEXP) {
                        gacc[po] += a * c0 * d0[R];
                        gacc[po + 1] -= a * c1 * d0[2 * R];
                        gacc[po + 2] += a * e;
                    } else {
                        gacc[po] += a * c0 * d0[R];
                        gacc[po + 1] += a * c1 * d0[2 * R];
                        gacc[po + 2] += a * e;
                    }
                } else {
                    const double x = val(P.first[i]), y = val(i - 1);
                    double ax, ay;   // adjoints of the first-evaluated and the second operand
                    if (op == NGP_OP_PLUS) {
                        ax = a; ay = a;
                    } else if (op == NGP_OP_TIMES) {
                        ax = a * y; ay = a * x;
                    } else {
                        const bool nat = (op == NGP_OP_CHANGEPOINT);
                        const double kl = nat ? x : y, kr = nat ? y : x;
                        const double loc = P.params[po];
                        const double us = cst[i][0], isc = cst[i][1];   // u = us (t - loc), 1 / scale
                        const double u1 = us * (t1 - loc), u2 = us * (t2 - loc);
                        const double g1 = sig[(long)P.slot[i] * npts + row];
                        const double g2 = sig[(long)P.slot[i] * npts + col];
                        const double q1 = 2.0 * g1 * (1.0 - g1), q2_ = 2.0 * g2 * (1.0 - g2);
                        const double d1l = -q1 * us, d2l = -q2_ * us;
                        const double d1s = -q1 * u1 * isc, d2s = -q2_ * u2 * isc;
                        gacc[po] += a * (d1l * kl * g2 + g1 * kl * d2l - d1l * kr * (1.0 - g2) -
                                         (1.0 - g1) * kr * d2l);
                        gacc[po + 1] += a * (d1s * kl * g2 + g1 * kl * d2s - d1s * kr * (1.0 - g2) -
                                             (1.0 - g1) * kr * d2s);
                        const double al_ = a * g1 * g2, ar_ = a * (1.0 - g1) * (1.0 - g2);
                        ax = nat ? al_ : ar_;
                        ay = nat ? ar_ : al_;
                    }
                    s7 = s5; s6 = s4; s5 = s3; s4 = s2; s3 = s1; s2 = s0; s1 = ax; s0 = ay;
                }
            }
            if (row == col) gacc[np] += w;   // d K / d noise = I (w already carries the 1/2)
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
    for (int pidx = 0; pidx <= np; ++pidx) {
        double v = gacc[pidx];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wave][pidx] = v;
    }
    __syncthreads();
    if (tid <= np)
        partials[((long)item * ntri * split + blockIdx.x) * (NGP_MAX_PARAMS + 1) + tid] =
            red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// The lattice contraction for trees of at most NL leaves (2 NL - 1 nodes) with NOTHING
// runtime-indexed in private memory.  The runtime-indexed gacc[] of the kernel above goes to scratch
// (784 B per lane: 58 MB of HBM traffic per item and call, and every `gacc[po] +=` a dependent
// load-add-store); indexing the accumulators by node slot instead needs 3 x 16 of them and the
// unrolled sweeps then keep ~225 VGPRs + scratch.  Here the workgroup first splits the program
// into its LEAVES and its BINARY nodes (in postfix order each, which is a topological order), and
// both sweeps run leaf list / binary list separately, unrolled over the list ordinal:
//     forward:  leaves -> vals[node];  binaries ascending: vals[node] = op(vals[first], vals[node-1])
//     reverse:  vals[root] = w;  binaries descending: the adjoints of the two operands overwrite
//               their values (every node has one parent: its value is dead once the parent is
//               done);  leaves: a = vals[node], accumulate
// so the accumulators are ga[leaf ordinal][3] and gcp[binary ordinal][2] — 38 doubles for 15
// nodes, all static — and one LDS array [node][thread] carries values, then adjoints.
// `items`: the chunk's items whose trees have at most NL leaves (launch_grad_contract sorts the
// items into the instantiations by size: most trees of an ensemble are one to four leaves, and a
// launch sized for the largest tree of the batch would run all of them at its occupancy).
// f(std::integral_constant<int, I>) for I = FROM, FROM - 1, ..., 0: an unrolled loop by construction
// (where `#pragma unroll` is a request hipcc may decline, indices here ARE compile-time constants)

template <int FROM, class F>
__device__ __forceinline__ void static_for_down(F &&f) {
    if constexpr (FROM >= 0) {
        f(std::integral_constant<int, FROM>{});
        static_for_down<FROM - 1>(f);
    }
}

// NACC / PASS: trees of more than 8 leaves would need more accumulators than the register file holds
// beside the sweeps; they run the kernel several times (PASS = 0, 1, ...), every pass sweeping all
// nodes but accumulating only the leaves / binaries with ordinal in [PASS NACC, (PASS + 1) NACC) — the
// first pass writes the partial sums, the later ones add theirs (same thread, same address, stream
// order).  Twice the sweep arithmetic, still no scratch.
// DIAG (the Toeplitz gradient path, stationary trees on a regular series): the contraction runs
// over the n lattice distances instead of the n^2 / 2 elements — element d stands for the whole
// d-th diagonal, `Kinv` then holds its weight w[item][d] = sum_i (a_i a_(i-d) - Kinv_(i,i-d)) (the
// diagonal d = 0 already halved; toep_weights_kernel), evaluated at (row, col) = (d, 0); one
// distance per thread, blockIdx.x = block of 256 distances.
template <int NL, int NACC = NL, int PASS = 0, bool DIAG = false>
__global__ __launch_bounds__(256) void grad_contract_lists_kernel(JobGeom g, ChunkPtrs p,
                                                                  const double *Kinv,
                                                                  const double *alpha,
                                                                  double *partials, int ntri,
                                                                  int split, DevSpec sp,
                                                                  const int32_t *items, int tpw = 1) {
    constexpr int NBIN = NL - 1, NN = 2 * NL - 1;
    constexpr bool PREFETCH = NL <= 8;     // 4 NL + 2 NBIN more doubles in registers
    __shared__ DevProgram P;
    __shared__ double red[4][NGP_MAX_PARAMS + 1];
    __shared__ double cst[NN][2];
    // values, then adjoints, of the nodes: [node][thread] in LDS — except for trees of one or two
    // leaves (REGS), whose shape is fixed (leaf, leaf, operator: nodes 0, 1, 2): three registers, no
    // LDS round trip between the leaves, the operator and the adjoints of a row
    constexpr bool REGS = NL <= 2;
    __shared__ double vals[REGS ? 1 : NN][REGS ? 1 : 256];
    __shared__ unsigned leaf_dec[NL], bin_dec[NBIN > 0 ? NBIN : 1];
    const int item = items ? items[blockIdx.y] : (int)blockIdx.y;
    // a workgroup walks `tpw` consecutive tiles of its item (large launches: the program load, the
    // list decode and the final reduction are paid once per workgroup, a third of its life at one
    // tile) and leaves ONE row of partial sums
    const int tile_first = (int)(blockIdx.x / split) * tpw, sub = blockIdx.x % split;
    const int tid = threadIdx.x;
    const int nrows = 16 / split;
    load_program(&P, p.progs + item);
    for (int i = tid; i < 4 * (NGP_MAX_PARAMS + 1); i += 256) (&red[0][0])[i] = 0.0;
    __syncthreads();
    if (tid < 64) {   // one wave: node i -> its list and its constants
        const int i = tid;
        const bool live = i < P.n_ops;
        const int op = live ? P.ops[i] : 0, po = live ? P.poff[i] : 0;
        const bool leaf = live && op < NGP_OP_PLUS;
        const unsigned long long lm = __ballot(leaf), bm = __ballot(live && !leaf);
        const unsigned long long below = (1ull << i) - 1ull;
        // node | opcode | parameter offset | table / sigmoid slot; binaries: first operand in the top byte
        if (leaf)
            leaf_dec[__popcll(lm & below)] =
                (unsigned)i | ((unsigned)op << 5) | ((unsigned)po << 9) | ((unsigned)P.slot[i] << 17);
        else if (live)
            bin_dec[__popcll(bm & below)] = (unsigned)i | ((unsigned)op << 5) | ((unsigned)po << 9) |
                                            ((unsigned)P.slot[i] << 17) | ((unsigned)P.first[i] << 25);
        if (i < NN) {
            double c0 = 0.0, c1 = 0.0;
            if (op == NGP_OP_SQEXP) {
                const double l = P.params[po], am = P.params[po + 1];
                c0 = am * (sp.se_form ? 0.5 / (l * l) : 1.0 / (l * l * l));
            } else if (op == NGP_OP_GAMMAEXP) {
                c0 = P.params[po + 2] * P.params[po + 1] / P.params[po];
                c1 = P.params[po + 2];
            } else if (op == NGP_OP_PERIODIC) {
                const double l = P.params[po], per = P.params[po + 1], am = P.params[po + 2];
                const double cq = sp.periodic_form ? 2.0 / l : 2.0 / (l * l);
                c0 = am * (sp.periodic_form ? 2.0 / (l * l) : 4.0 / (l * l * l));
                c1 = am * cq * 2.0 * M_PI / (per * per);
            } else if (op == NGP_OP_CHANGEPOINT || op == OP_CP_SWAPPED) {
                c1 = 1.0 / P.params[po + 1];
                c0 = sp.cp_form ? c1 : -c1;            // u = c0 (t - loc)
            }
            cst[i][0] = c0;
            cst[i][1] = c1;
        }
    }
    __syncthreads();
    // a wave works on ONE row at a time (its 64 lanes are 64 columns): the row index is
    // wave-uniform, which hipcc cannot see in `tid >> 6` — said explicitly, everything that is a
    // function of the row alone (t0[row], qpts[row], alpha[row], the ChangePoint sigmoid of the row)
    // becomes a scalar load instead of a vector load that every lane repeats, and the table lookups
    // of an element no longer wait behind it (they were two dependent memory round trips per row)
    const int tx = tid & 63, ty = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int np = P.n_params;
    const int nops = __builtin_amdgcn_readfirstlane(P.n_ops);
    const int nl = (nops + 1) / 2, nbin = nops / 2;       // a binary tree: nl leaves, nl - 1 binaries
    const int R = g.R, npts = g.npts;
    const double *tab = p.tab + (long)item * g.maxstat * R;
    const double *dt = p.dtab + (long)item * g.maxstat * 3 * R;
    const double *sig = p.sig + (long)item * g.maxcp * npts;
    // The decoded lists.  Trees of one or two leaves (most items of an ensemble) read them ONCE into
    // scalar registers: every index below is a compile-time constant, so the two arrays are 2 NL - 1
    // SGPRs, never memory — read from LDS where they are used, every use is an LDS round trip on the
    // critical path of every row (the compiler barrier at the top of the row loop forbids keeping
    // them), three to four per node and row.  Larger trees keep the LDS reads: with the words in
    // registers hipcc hoists everything derived from them as well and spills SGPRs into VGPRs
    // (<8>: 232 -> 254 VGPRs, one wave per SIMD instead of two).
    constexpr bool HOIST = NL <= 2;
    unsigned ldv[HOIST ? NL : 1], bdv[HOIST && NBIN > 0 ? NBIN : 1];
    if constexpr (HOIST) {
        static_for_down<NL - 1>([&](auto lc) {
            constexpr int l = decltype(lc)::value;
            ldv[l] = l < nl ? (unsigned)__builtin_amdgcn_readfirstlane((int)leaf_dec[l]) : 0u;
        });
        static_for_down<NBIN - 1>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            bdv[b] = b < nbin ? (unsigned)__builtin_amdgcn_readfirstlane((int)bin_dec[b]) : 0u;
        });
    }
    auto LD = [&](int l) {
        if constexpr (HOIST) return ldv[l];
        else return (unsigned)__builtin_amdgcn_readfirstlane((int)leaf_dec[l]);
    };
    auto BD = [&](int b) {
        if constexpr (HOIST) return bdv[b];
        else return (unsigned)__builtin_amdgcn_readfirstlane((int)bin_dec[b]);
    };
    auto f_node = [](unsigned d) { return (int)(d & 31u); };
    auto f_op = [](unsigned d) { return (int)((d >> 5) & 15u); };
    auto f_po = [](unsigned d) { return (int)((d >> 9) & 255u); };
    auto f_slot = [](unsigned d) { return (int)((d >> 17) & 255u); };
    auto f_first = [](unsigned d) { return (int)(d >> 25); };
    double ga[NACC][3], gcp[NACC][2];
#pragma unroll
    for (int l = 0; l < NACC; ++l) ga[l][0] = ga[l][1] = ga[l][2] = gcp[l][0] = gcp[l][1] = 0.0;
    constexpr auto own = [](int ordinal) { return ordinal / NACC == PASS; };
    double gnoise = 0.0;
    const double *Ki = Kinv + (long)item * g.n0 * (DIAG ? 1 : g.n0);
    const double *al = alpha + (long)item * g.n0;
    // What depends on the row alone — its time, lattice coordinate and alpha — is loaded ONCE per
    // wave, lane rr holding the values of the wave's row rr, and handed to all lanes by v_readlane
    // where the row is processed.  Loaded inside the row loop (as until round 4) they were vector
    // loads that every lane repeats (the compiler barrier below forbids scalar loads: memory may
    // have changed), and the lattice coordinate stood between the row and its table lookups: two
    // dependent memory round trips per row where there is now one.
    for (int tile = tile_first; tile < (DIAG ? tile_first + 1 : min(tile_first + tpw, ntri)); ++tile) {
    int r = 0, c = 0;
    if constexpr (!DIAG) {
        r = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
        while ((r + 1) * (r + 2) / 2 <= tile) ++r;
        while (r * (r + 1) / 2 > tile) --r;
        c = tile - r * (r + 1) / 2;
    }
    const int col = DIAG ? 0 : c * NB + tx;
    const int row0 = DIAG ? 0 : r * NB + ty * 16 + sub * nrows;
    double t1_l = 0.0, al_l = 0.0;
    int q1_l = 0;
    if constexpr (!DIAG) {
        const int lrow = row0 + (tx < nrows ? tx : 0);      // < n0: inside every array
        t1_l = p.t0[lrow];
        q1_l = p.qpts[lrow];
        al_l = al[lrow];
    }
    // ChangePoint sigmoids (trees of up to four leaves): the column's value once per lane, the rows'
    // values once per wave (lane rr = row rr), instead of two loads per node and row
    constexpr bool SIGPRE = PREFETCH && NL <= 4 && NBIN > 0 && !DIAG;
    double sgc[SIGPRE ? NBIN : 1], sgr_l[SIGPRE ? NBIN : 1];
    if constexpr (SIGPRE) {
        const int lrow = row0 + (tx < nrows ? tx : 0), lcol = col < g.n0 ? col : 0;
        static_for_down<NBIN - 1>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            sgc[b] = sgr_l[b] = 0.0;
            if (b >= nbin) return;
            const int op = f_op(BD(b));
            if (op == NGP_OP_CHANGEPOINT || op == OP_CP_SWAPPED) {
                sgc[b] = sig[(long)f_slot(BD(b)) * npts + lcol];
                sgr_l[b] = sig[(long)f_slot(BD(b)) * npts + lrow];
            }
        });
    }
    if (col < g.n_real) {
        const double t2 = p.t0[col], ac = DIAG ? 0.0 : al[col];
        const int q2 = p.qpts[col];
        for (int rr = 0; rr < (DIAG ? 1 : nrows); ++rr) {
            const int row = DIAG ? (int)blockIdx.x * 256 + tid : row0 + rr;
            if (row >= g.n_real || col > row) continue;
            // nothing loop-invariant is to be hoisted out of this loop: with the sweeps unrolled
            // hipcc would keep every node's parameters, constants and table addresses in VGPRs
            // across the rows
            asm volatile("" ::: "memory");
            double w, t1;
            int q1;
            if constexpr (DIAG) {
                w = Ki[row];
                t1 = p.t0[row];
                q1 = p.qpts[row];
            } else {
                t1 = readlane_f64(t1_l, rr);
                q1 = __builtin_amdgcn_readlane(q1_l, rr);
                w = readlane_f64(al_l, rr) * ac - Ki[(long)row * g.n0 + col];
                if (row == col) w *= 0.5;
            }
            const double d = fabs(t1 - t2);
            const int dq = abs(q1 - q2);
            // ---- every table value of this element requested up front (PREFETCH): read where
            //      the sweeps use them, each leaf's lookups wait out their own round trip — four or
            //      five dependent memory latencies per element, which is what the kernel's time was
            //      (2.3 us per row of a wave at two leaves).  Together they cost one.
            double tv[PREFETCH ? NL : 1], td[PREFETCH ? NL : 1][3], sg[PREFETCH && NBIN ? NBIN : 1][2];
            if constexpr (PREFETCH) {
                static_for_down<NL - 1>([&](auto lc) {
                    constexpr int l = decltype(lc)::value;
                    if (l >= nl) return;
                    const int op = f_op(LD(l));
                    if (op > NGP_OP_LINEAR) {      // a stationary leaf: value and derivative factors
                        const long sl = f_slot(LD(l));
                        const double *d0 = dt + sl * 3 * R + dq;
                        tv[l] = tab[sl * R + dq];
                        td[l][0] = d0[0];
                        td[l][1] = d0[R];
                        td[l][2] = d0[2 * R];
                    }
                });
                static_for_down<NBIN - 1>([&](auto bc) {
                    constexpr int b = decltype(bc)::value;
                    if (b >= nbin) return;
                    const int op = f_op(BD(b));
                    if (op == NGP_OP_CHANGEPOINT || op == OP_CP_SWAPPED) {
                        if constexpr (SIGPRE) {
                            sg[b][0] = readlane_f64(sgr_l[b], rr);
                            sg[b][1] = sgc[b];
                        } else {
                            sg[b][0] = sig[(long)f_slot(BD(b)) * npts + row];
                            sg[b][1] = sig[(long)f_slot(BD(b)) * npts + col];
                        }
                    }
                });
            }
            double rv[3] = {0.0, 0.0, 0.0};
            // ---- forward: leaves, then binary nodes in postfix order
            static_for_down<NL - 1>([&](auto lc) {
                constexpr int l = decltype(lc)::value;
                if (l >= nl) return;
                const int op = f_op(LD(l)), po = f_po(LD(l));
                double v;
                if (op == NGP_OP_CONSTANT) v = P.params[po];
                else if (op == NGP_OP_LINEAR)
                    v = P.params[po + 1] + P.params[po + 2] * (t1 - P.params[po]) * (t2 - P.params[po]);
                else if constexpr (PREFETCH) v = tv[l];
                else v = tab[(long)f_slot(LD(l)) * R + dq];
                if constexpr (REGS) rv[l] = v;
                else vals[f_node(LD(l))][tid] = v;
            });
            static_for_down<NBIN - 1>([&](auto bc) {
                constexpr int b = NBIN - 1 - decltype(bc)::value;       // ascending: postfix order
                if (b >= nbin) return;
                const int op = f_op(BD(b)), nd = f_node(BD(b));
                const double x = REGS ? rv[0] : vals[f_first(BD(b))][tid], y = REGS ? rv[1] : vals[nd - 1][tid];
                double v;
                if (op == NGP_OP_PLUS) v = x + y;
                else if (op == NGP_OP_TIMES) v = x * y;
                else {
                    const double kl = (op == NGP_OP_CHANGEPOINT) ? x : y;
                    const double kr = (op == NGP_OP_CHANGEPOINT) ? y : x;
                    const double g1 = PREFETCH ? sg[b][0] : sig[(long)f_slot(BD(b)) * npts + row];
                    const double g2 = PREFETCH ? sg[b][1] : sig[(long)f_slot(BD(b)) * npts + col];
                    v = g1 * kl * g2 + (1.0 - g1) * kr * (1.0 - g2);
                }
                if constexpr (REGS) rv[2] = v;
                else vals[nd][tid] = v;
            });
            // ---- reverse: the root's adjoint is w; adjoints overwrite values on the way down
            if constexpr (REGS) {
                if (nops == 1) rv[0] = w;
                else rv[2] = w;
            } else {
                vals[nops - 1][tid] = w;
            }
            static_for_down<NBIN - 1>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                if (b >= nbin) return;
                const int op = f_op(BD(b)), nd = f_node(BD(b)), fi = f_first(BD(b));
                const double a = REGS ? rv[2] : vals[nd][tid];
                const double x = REGS ? rv[0] : vals[fi][tid], y = REGS ? rv[1] : vals[nd - 1][tid];
                double ax, ay;   // adjoints of the first-evaluated and the second operand
                if (op == NGP_OP_PLUS) {
                    ax = a; ay = a;
                } else if (op == NGP_OP_TIMES) {
                    ax = a * y; ay = a * x;
                } else {
                    const bool nat = (op == NGP_OP_CHANGEPOINT);
                    const double kl = nat ? x : y, kr = nat ? y : x;
                    const int po = f_po(BD(b));
                    const double loc = P.params[po];
                    const double us = cst[nd][0], isc = cst[nd][1];   // u = us (t - loc), 1 / scale
                    const double u1 = us * (t1 - loc), u2 = us * (t2 - loc);
                    const double g1 = PREFETCH ? sg[b][0] : sig[(long)f_slot(BD(b)) * npts + row];
                    const double g2 = PREFETCH ? sg[b][1] : sig[(long)f_slot(BD(b)) * npts + col];
                    const double q1 = 2.0 * g1 * (1.0 - g1), q2_ = 2.0 * g2 * (1.0 - g2);
                    const double d1l = -q1 * us, d2l = -q2_ * us;
                    const double d1s = -q1 * u1 * isc, d2s = -q2_ * u2 * isc;
                    if (own(b)) {
                        gcp[b % NACC][0] += a * (d1l * kl * g2 + g1 * kl * d2l -
                                                 d1l * kr * (1.0 - g2) - (1.0 - g1) * kr * d2l);
                        gcp[b % NACC][1] += a * (d1s * kl * g2 + g1 * kl * d2s -
                                                 d1s * kr * (1.0 - g2) - (1.0 - g1) * kr * d2s);
                    }
                    const double al_ = a * g1 * g2, ar_ = a * (1.0 - g1) * (1.0 - g2);
                    ax = nat ? al_ : ar_;
                    ay = nat ? ar_ : al_;
                }
                if constexpr (REGS) {
                    rv[0] = ax;
                    rv[1] = ay;
                } else {
                    vals[fi][tid] = ax;
                    vals[nd - 1][tid] = ay;
                }
            });
            static_for_down<NL - 1>([&](auto lc) {
                constexpr int l = decltype(lc)::value;
                if (l >= nl || !own(l)) return;
                const int op = f_op(LD(l)), po = f_po(LD(l)), nd = f_node(LD(l));
                const double a = REGS ? rv[l] : vals[nd][tid];
                if (op == NGP_OP_CONSTANT) {
                    ga[l % NACC][0] += a;
                } else if (op == NGP_OP_LINEAR) {
                    const double cc = P.params[po], a1 = t1 - cc, a2 = t2 - cc;
                    ga[l % NACC][0] += a * P.params[po + 2] * (-a1 - a2);
                    ga[l % NACC][1] += a;
                    ga[l % NACC][2] += a * a1 * a2;
                } else {
                    const double *d0 = dt + (long)f_slot(LD(l)) * 3 * R + dq;
                    const double e = PREFETCH ? td[l][0] : d0[0];
                    const double f1 = PREFETCH ? td[l][1] : d0[R], f2 = PREFETCH ? td[l][2] : d0[2 * R];
                    const double c0 = cst[nd][0], c1 = cst[nd][1];
                    if (op == NGP_OP_SQEXP) {
                        ga[l % NACC][0] += a * e * d * d * c0;
                        ga[l % NACC][1] += a * e;
                    } else if (op == NGP_OP_GAMMAEXP) {
                        ga[l % NACC][0] += a * c0 * f1;
                        ga[l % NACC][1] -= a * c1 * f2;
                        ga[l % NACC][2] += a * e;
                    } else {
                        ga[l % NACC][0] += a * c0 * f1;
                        ga[l % NACC][1] += a * c1 * f2;
                        ga[l % NACC][2] += a * e;
                    }
                }
            });
            if (PASS == 0 && row == col) gnoise += w;   // d K / d noise = I (w carries the 1/2)
        }
    }
    }   // tiles of this workgroup
    // ---- deterministic reduction: wave shuffles per (node, parameter), then the four waves in order
    const int lane = tid & 63, wave = tid >> 6;
    auto wave_sum = [&](double v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        return v;
    };
    static_for_down<NL - 1>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        if (l >= nl || !own(l)) return;
        const int op = f_op(LD(l));
        const int cnt = op == NGP_OP_CONSTANT ? 1 : (op == NGP_OP_SQEXP ? 2 : 3);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k >= cnt) break;
            const double v = wave_sum(ga[l % NACC][k]);
            if (lane == 0) red[wave][f_po(LD(l)) + k] = v;
        }
    });
    static_for_down<NBIN - 1>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        if (b >= nbin) return;
        const int op = f_op(BD(b));
        if ((op != NGP_OP_CHANGEPOINT && op != OP_CP_SWAPPED) || !own(b)) return;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double v = wave_sum(gcp[b % NACC][k]);
            if (lane == 0) red[wave][f_po(BD(b)) + k] = v;
        }
    });
    if (PASS == 0) {
        const double v = wave_sum(gnoise);
        if (lane == 0) red[wave][np] = v;
    }
    __syncthreads();
    if (tid <= np) {
        double *dst = partials + ((long)item * gridDim.x + blockIdx.x) * (NGP_MAX_PARAMS + 1) + tid;
        const double sum = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        if (PASS == 0) *dst = sum;
        else *dst += sum;      // a parameter of another pass adds 0.0: its bits do not change
    }
}

// ---------------------------------------------------------------------------------------
// The Toeplitz gradient path (stationary trees on a regular series).
// K is symmetric positive definite Toeplitz there, dK/dtheta depends on the lattice distance only,
// so  d logml / d theta = sum_d w(d) dk(d)/dtheta  with  w(d) = sum_i (a_i a_(i-d) - Kinv_(i,i-d))
// (halved at d = 0), and by the Gohberg-Semencul formula the diagonal sums of Kinv follow from its
// first column x = Kinv e_1 alone:
//     sum_i Kinv_(i,i-d) = (1/x_0) sum_(m=0)^(n-1-d) (n - d - m) (x_(m+d) x_m - x_(n-m) x_(n-m-d)),  x_n = 0.
// A = X Kinv for the two aux rows X = [y' ; e_1'] comes out of the ordinary factorisation and one
// backward sweep (aux_back_*): row 0 = a' (alpha), row 1 = x'.  n^3/3 flops instead of n^3, no W,
// no Kinv.  One workgroup per (item, block of 256 distances): each thread sums its distance in a
// fixed order (deterministic).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void toep_weights_kernel(JobGeom g, const double *A, double *wbuf) {
    extern __shared__ double sh[];      // a[n] | x[n + 1]
    const int item = blockIdx.y, n = g.n_real, tid = threadIdx.x;
    const double *a_g = A + (long)item * g.naux_pad * g.ld, *x_g = a_g + g.ld;
    double *a = sh, *x = sh + n;
    for (int i = tid; i < n; i += 256) {
        a[i] = a_g[i];
        x[i] = x_g[i];
    }
    if (tid == 0) x[n] = 0.0;
    __syncthreads();
    const int d = blockIdx.x * 256 + tid;
    if (d >= n) return;
    const double rx0 = 1.0 / x[0];
    double sa = 0.0, s1 = 0.0, s2 = 0.0;
    for (int m = 0; m < n - d; ++m) {
        const double wgt = (double)(n - d - m);
        sa += a[m + d] * a[m];
        s1 += wgt * (x[m + d] * x[m]);
        s2 += wgt * (x[n - m] * x[n - m - d]);
    }
    const double wv = sa - (s1 - s2) * rx0;
    wbuf[(long)item * g.n0 + d] = (d == 0) ? 0.5 * wv : wv;
}

// quad = z'z from the aux row that carries y' (row 0), before the backward sweep overwrites it
__global__ __launch_bounds__(256) void toep_quad_kernel(JobGeom g, const double *L, double *quad) {
    __shared__ double red[4];
    const int item = blockIdx.x, tid = threadIdx.x;
    const double *z = L + (long)item * g.item_stride + (long)g.n0 * g.ld;
    double s = 0.0;
    for (int i = tid; i < g.n_real; i += 256) s += z[i] * z[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) quad[item] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(128) void grad_reduce_kernel(JobGeom g, const DevProgram *progs,
                                                          const double *partials, const double *quad,
                                                          const double *logdet, double *grad,
                                                          double *logml, int ntri) {
    const int item = blockIdx.x, pidx = threadIdx.x;
    const int np = progs[item].n_params;
    if (pidx <= np) {
        // t ascending, as ever (the sum's bits do not depend on the launch); sixteen loads in
        // flight at a time — one dependent load per addition made this 0.2 ms of a 64-particle call
        const double *src = partials + (long)item * ntri * (NGP_MAX_PARAMS + 1) + pidx;
        double s = 0.0;
        int t = 0;
        for (; t + 16 <= ntri; t += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = src[(long)(t + u) * (NGP_MAX_PARAMS + 1)];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        for (; t < ntri; ++t) s += src[(long)t * (NGP_MAX_PARAMS + 1)];
        grad[(long)item * (NGP_MAX_PARAMS + 1) + pidx] = s;
    }
    if (pidx == 0)
        logml[item] = -0.5 * quad[item] - logdet[item] - 0.5 * g.n_real * 1.8378770664093454836;
}

// ---------------------------------------------------------------------------------------
// mixture sampling (ngp_mixture_sample)
// ---------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11):
// counter-based, so draw d of scenario s is a pure function of (seed, s, d) — no state to
// carry, any launch geometry gives the same stream.
struct Philox4 { unsigned x, y, z, w; };
__host__ __device__ inline Philox4 philox4x32_10(Philox4 c, unsigned k0, unsigned k1) {
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c.x, p1 = 0xCD9E8D57ull * c.z;
        const Philox4 n{(unsigned)(p1 >> 32) ^ c.y ^ k0, (unsigned)p1,
                        (unsigned)(p0 >> 32) ^ c.w ^ k1, (unsigned)p0};
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
// 53-bit uniform in (0, 1) from two 32-bit words
__host__ __device__ inline double u01(unsigned hi, unsigned lo) {
    const unsigned long long v = (((unsigned long long)hi << 32) | lo) >> 11;
    return ((double)v + 0.5) * (1.0 / 9007199254740992.0);
}

// in-place lower Cholesky of one m x m matrix per workgroup (row-major, upper part zeroed)
__global__ __launch_bounds__(256) void small_chol_kernel(double *A, int m, int32_t *info) {
    double *a = A + (long)blockIdx.x * m * m;
    __shared__ double piv;
    __shared__ int bad;
    const int tid = threadIdx.x;
    if (tid == 0) bad = 0;
    __syncthreads();
    for (int k = 0; k < m; ++k) {
        if (tid == 0) {
            const double akk = a[(long)k * m + k];
            if (!(akk > 0.0) && bad == 0) bad = k + 1;
            piv = sqrt(akk);
            a[(long)k * m + k] = piv;
        }
        __syncthreads();
        const double d = piv;
        for (int i = k + 1 + tid; i < m; i += 256) a[(long)i * m + k] /= d;
        __syncthreads();
        // trailing update, lower part: element (i, j), k < j <= i
        const int nt = m - k - 1;
        for (int e = tid; e < nt * nt; e += 256) {
            const int i = k + 1 + e / nt, j = k + 1 + e % nt;
            if (j <= i) a[(long)i * m + j] -= a[(long)i * m + k] * a[(long)j * m + k];
        }
        __syncthreads();
    }
    for (int e = tid; e < m * m; e += 256)
        if (e % m > e / m) a[e] = 0.0;
    if (tid == 0 && info) info[blockIdx.x] = bad;
}

// one thread per (scenario, draw).  seeds == nullptr: the S mixtures share their P components
// (mu [P][S][m], chol [P][m][m]) and one key, the scenario index is part of the counter.
// seeds != nullptr: S independent mixtures (mu [S][P][m], chol [S][P][m][m]); mixture s is keyed
// by seeds[s] with scenario counter 0, i.e. it draws exactly what a call with S = 1 and
// seed = seeds[s] draws.
__global__ __launch_bounds__(256) void mixture_sample_kernel(int P, int S, int m, const double *w,
                                                             const double *mu, const double *chol,
                                                             int draws, unsigned k0, unsigned k1,
                                                             const unsigned long long *seeds,
                                                             double *out, int32_t *comp) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)S * draws) return;
    const int s = (int)(idx / draws), d = (int)(idx % draws);
    unsigned cs = (unsigned)s;
    if (seeds) {
        k0 = (unsigned)seeds[s];
        k1 = (unsigned)(seeds[s] >> 32);
        cs = 0u;
    }
    // block 0: component pick by inverse CDF over the P weights of scenario s
    const Philox4 r0 = philox4x32_10(Philox4{(unsigned)d, cs, 0u, 0u}, k0, k1);
    const double u = u01(r0.x, r0.y);
    const double *ws = w + (long)s * P;
    int k = P - 1;
    double acc = 0.0;
    for (int i = 0; i < P; ++i) {
        acc += ws[i];
        if (u < acc) { k = i; break; }
    }
    if (comp) comp[idx] = k;
    // blocks 1..: four words -> one Box-Muller pair -> two normals
    const double *L = chol + ((seeds ? (long)s * P : 0l) + k) * m * m;
    const double *mk = mu + (seeds ? ((long)s * P + k) : ((long)k * S + s)) * m;
    double *o = out + idx * m;
    for (int i = 0; i < m; ++i) o[i] = mk[i];
    for (int j0 = 0; j0 < m; j0 += 2) {
        const Philox4 r = philox4x32_10(Philox4{(unsigned)d, cs, (unsigned)(1 + j0 / 2), 0u},
                                        k0, k1);
        const double u1 = u01(r.x, r.y), u2 = u01(r.z, r.w);
        const double rad = sqrt(-2.0 * log(u1)), ang = 2.0 * M_PI * u2;
        const double z0 = rad * cos(ang), z1 = rad * sin(ang);
        for (int i = j0; i < m; ++i) o[i] += L[(long)i * m + j0] * z0;          // column j0 of L
        if (j0 + 1 < m)
            for (int i = j0 + 1; i < m; ++i) o[i] += L[(long)i * m + j0 + 1] * z1;
    }
}

void launch_mixture_sample(int P, int S, int m, const double *w, const double *mu, double *chol,
                           int draws, uint64_t seed, const uint64_t *seeds, double *out,
                           int32_t *comp, int32_t *info, hipStream_t s) {
    const long mats = seeds ? (long)S * P : (long)P;
    hipLaunchKernelGGL(small_chol_kernel, dim3((unsigned)mats), dim3(256), 0, s, chol, m, info);
    const long n = (long)S * draws;
    hipLaunchKernelGGL(mixture_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, P,
                       S, m, w, mu, (const double *)chol, draws, (unsigned)seed,
                       (unsigned)(seed >> 32), (const unsigned long long *)seeds, out, comp);
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

__global__ void mfma4_composite_probe_kernel(const double *A, const double *Bm, double *Dout) {
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];
    const double b = Bm[(l >> 4) * 16 + (l & 15)];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    mfma16_as_4(acc, a, rot4(b));
    const f64x4 d = to_d16(acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) Dout[((l >> 4) + 4 * r) * 16 + (l & 15)] = d[r];
}

// D[32x32] = A[32x2] B[2x32] through one v_mfma_f32_32x32x2_f32 with the operand / result maps
// the mixed-precision k-loop assumes
__global__ void mfma_f32_probe_kernel(const float *A, const float *Bm, float *Dout) {
    const int l = threadIdx.x;
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * 2 + (l >> 5)], Bm[(l >> 5) * 32 + (l & 31)],
                                               acc, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 16; ++v)
        Dout[((v & 3) + 8 * (v >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[v];
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
void launch_tables(const JobGeom &g, const ChunkPtrs &p, int Bc, const DevSpec &sp, hipStream_t s) {
    if (g.n0 == 0 || !g.lattice) return;
    JobGeom gt = g;
    // short gradient jobs fill on the full program (launch_fill): no subtree tables behind the leaves'
    if (p.dtab && small_job(g, Bc)) gt.tab_sub = 0;
    hipLaunchKernelGGL(tables_kernel, dim3(Bc), dim3(256), 0, s, gt, p, sp);
}

void launch_fill(const JobGeom &g, const ChunkPtrs &p, int Bc, const DevSpec &sp, hipStream_t s,
                 bool aux_only) {
    if (g.n0 == 0) return;
    const int ntri = g.nb0 * (g.nb0 + 1) / 2;
    // Gradient jobs (aux rows [I ; y']): the identity block is NOT written — the column kernels
    // synthesise a tile of it the first time they meet it (chol_col*<.., IDENT>).  What is written:
    // the tile row that carries y' and the zero blocks (a, a - 1) just left of the block diagonal,
    // which the k-loops of a tile pair and of K^-1 = W W' read as part of their shared k-range.
    const int ntiles = ntri + (g.aux_identity ? g.nb0 + (g.nb0 - 1) : (g.naux_pad / NB) * g.nb0);
    const int off = aux_only ? ntri : 0;
    if (g.lattice) {
        const long nwg = (long)(ntiles - off) * Bc;
        const int split = nwg <= 1024 ? 4 : (nwg <= 2048 ? 2 : 1);
        // Short jobs are chains of launches a few microseconds long: ONE fill launch on the general
        // kernel (reduced / full programs; the chain and single-table kernels compute the same
        // values, operation for operation) instead of up to three per program shape plus the aux one
        const bool one_launch = small_job(g, Bc);
        if (one_launch && p.dtab && !aux_only) {
            ChunkPtrs q = p;   // gradient job: main tiles only (y' comes from the observations)
            q.fill_other = nullptr;
            const long nwg_m = (long)ntri * Bc;
            const int split_m = nwg_m <= 1024 ? 4 : (nwg_m <= 2048 ? 2 : 1);
            hipLaunchKernelGGL(fill_lattice_kernel<true>, dim3(ntri * split_m, Bc), dim3(256), 0, s, g, q,
                               ntri, 0, split_m, sp);
        } else if (p.dtab && p.fill_other && !aux_only) {
            // Gradient jobs: the main tiles through the kernels of the value jobs — one lookup for a
            // stationary tree, chain programs decoded once per thread for sixteen elements, the
            // rest on the reduced program — reading the subtree tables behind the per-leaf ones;
            // the values are those of the full program, operation for operation (keval_stat).  The
            // full-program interpreter decodes every node for every element from LDS: 8 us per item
            // at n = 2049 against 4.  The aux tiles (y', the zero blocks, e_1') hold no covariance.
            ChunkPtrs q = p;
            q.tab = p.tab + (size_t)g.tab_sub * g.R;
            q.dtab = nullptr;
            const long nwg_o = (long)ntri * p.n_fill_other;
            const int split_o = nwg_o <= 1024 ? 4 : (nwg_o <= 2048 ? 2 : 1);
            if (p.n_fill_other > 0)
                hipLaunchKernelGGL(fill_lattice_kernel<false>, dim3(ntri * split_o, p.n_fill_other),
                                   dim3(256), 0, s, g, q, ntri, 0, split_o, sp);
            if (p.n_fill_chain > 0) {
                const int tpw = (long)ntri * p.n_fill_chain >= 65536 ? 4 : 1;
                hipLaunchKernelGGL(fill_chain_kernel, dim3((ntri + tpw - 1) / tpw, p.n_fill_chain), dim3(256),
                                   0, s, g, q, ntri, sp, ntri, tpw);
            }
            if (p.n_fill_single > 0)
                hipLaunchKernelGGL(fill_single_kernel, dim3(ntri, p.n_fill_single), dim3(256), 0, s, g,
                                   q, ntri, sp);
            // (short jobs: chol_small_kernel takes y' from the observations and neither it nor
            // grad_kinv_small_kernel reads the zero blocks — one launch less in their chain)
            if (!small_job(g, Bc)) {
                ChunkPtrs a = p;
                a.fill_other = nullptr;
                const long nwg_a = (long)(ntiles - ntri) * Bc;
                const int split_a = nwg_a <= 1024 ? 4 : (nwg_a <= 2048 ? 2 : 1);
                hipLaunchKernelGGL(fill_lattice_kernel<true>, dim3((ntiles - ntri) * split_a, Bc), dim3(256),
                                   0, s, g, a, ntri, ntri, split_a, sp);
            }
        } else if (p.dtab) {
            ChunkPtrs q = p;
            q.fill_other = nullptr;
            hipLaunchKernelGGL(fill_lattice_kernel<true>, dim3((ntiles - off) * split, Bc), dim3(256),
                               0, s, g, q, ntri, off, split, sp);
        } else if (p.fill_other && !aux_only && !one_launch) {
            // staged value jobs: chain programs on their own kernel, the rest element by element
            if (p.n_fill_other > 0)
                hipLaunchKernelGGL(fill_lattice_kernel<false>, dim3(ntiles * split, p.n_fill_other),
                                   dim3(256), 0, s, g, p, ntri, 0, split, sp);
            if (p.n_fill_chain > 0) {
                const int tpw = (long)ntiles * p.n_fill_chain >= 65536 ? 4 : 1;
                hipLaunchKernelGGL(fill_chain_kernel, dim3((ntiles + tpw - 1) / tpw, p.n_fill_chain),
                                   dim3(256), 0, s, g, p, ntri, sp, ntiles, tpw);
            }
            if (p.n_fill_single > 0)
                hipLaunchKernelGGL(fill_single_kernel,
                                   dim3(g.toep ? ntiles - ntri + g.nb0 : ntiles, p.n_fill_single),
                                   dim3(256), 0, s, g, p, ntri, sp);
        } else {
            ChunkPtrs q = p;
            q.fill_other = nullptr;
            hipLaunchKernelGGL(fill_lattice_kernel<false>, dim3((ntiles - off) * split, Bc),
                               dim3(256), 0, s, g, q, ntri, off, split, sp);
        }
    }
    else
        hipLaunchKernelGGL(fill_kernel, dim3(ntiles - off, Bc), dim3(256), 0, s, g, p, ntri, off, sp);
}

// The product instantiation of the column sweep: NoProbe.  Weak, so that the diagnostic build
// (scripts/stamps/ngp_stamps.hip, linked beside this file into its own library) can put the
// stamping instantiation in their place; libngp.so contains these two and nothing of the probes.
// 1: chol_diag_wave_kernel (ngp_small_kernels.h), 0: chol_diag_kernel — a process-wide switch for
// same-box A/B runs (scripts/diag_form_ab.py), not part of the C-ABI
static std::atomic<int> g_diag_form{1};
constexpr int DIAG_WAVE_MAX_ITEMS = 512;
extern "C" void ngp_debug_set_diag_form(int form) { g_diag_form.store(form); }

__attribute__((weak)) void launch_chol_diag(const JobGeom &g, const ChunkPtrs &p, int Bc, int j,
                                            int k0, hipStream_t s) {
    // Small chunks (the 24- and 64-particle calls of a fit), where the launch is on the critical path
    // of the sweep: 42 -> 34 us at 64 items.  Large chunks keep chol_diag_kernel: there every
    // workgroup competes for its CU with three others and what counts is its total work, of which the
    // wave form — one wave factoring while three wait — has more (6,400 items: 376 -> 455 us per
    // launch).  Batch-invariant jobs never switch (the two forms differ in the last bits).
    if (g_diag_form.load(std::memory_order_relaxed) && !g.invariant && Bc <= DIAG_WAVE_MAX_ITEMS)
        launch_chol_diag_wave(g, p, Bc, j, k0, s);
    else
        launch_chol_diag_t<NoProbe>(g, p, Bc, j, k0, s);
}

__attribute__((weak)) void launch_chol_col(const JobGeom &g, const ChunkPtrs &p, int Bc, int j,
                                           int mode, int k0, hipStream_t s, const DevSpec *sp) {
    launch_chol_col_t<NoProbe>(g, p, Bc, j, mode, k0, s, sp);
}

void launch_aux_back(const JobGeom &g, const ChunkPtrs &p, const double *dinv_all, size_t mstep,
                     double *Aout, int accumulate, int Bc, int c, hipStream_t s) {
    const int ntl = (g.naux + NB - 1) / NB;
    hipLaunchKernelGGL(aux_back_solve_kernel, dim3(ntl, Bc), dim3(256), 0, s, g, p,
                       dinv_all + (size_t)c * mstep, Aout, accumulate, c);
    if (c > 0) {
        const int ngrp = (c + 3) / 4;
        if (g.naux <= 16)
            hipLaunchKernelGGL(aux_back_update_kernel<1>, dim3(ntl * ngrp, Bc), dim3(256), 0, s, g,
                               p, c);
        else
            hipLaunchKernelGGL(aux_back_update_kernel<4>, dim3(ntl * ngrp, Bc), dim3(256), 0, s, g,
                               p, c);
    }
}

void launch_kapply(const JobGeom &g, const ChunkPtrs &p, const double *A, const double *X,
                   double *R, int Bc, const DevSpec &sp, hipStream_t s) {
    // accumulators per thread and column: 12 aux rows at a time (the usual d + m + 1 = 11 is one pass)
    constexpr int NACC = 12;
    const dim3 grid(g.nb0, Bc, (g.naux + NACC - 1) / NACC);   // 64 columns per workgroup (CPT = 1)
    if (g.lattice) {
        hipLaunchKernelGGL((kapply_kernel<NACC, 2, KA_SINGLE>), grid, dim3(256), 0, s, g, p, A, X, R,
                           sp);
        hipLaunchKernelGGL((kapply_kernel<NACC, 1, KA_REDUCED>), grid, dim3(256), 0, s, g, p, A, X,
                           R, sp);
        hipLaunchKernelGGL((kapply_kernel<NACC, 1, KA_CHAIN>), grid, dim3(256), 0, s, g, p, A, X, R,
                           sp);
    } else {
        hipLaunchKernelGGL((kapply_kernel<NACC, 1, KA_DIRECT>), grid, dim3(256), 0, s, g, p, A, X,
                           R, sp);
    }
}

void launch_refine_gram(const JobGeom &g, const double *A, const double *X, const double *R,
                        double *S, double *T, double *U, double *G, double *delta, int Bc,
                        const int32_t *items, hipStream_t s) {
    const int npairs = g.naux * g.naux;
    hipLaunchKernelGGL(refine_gram_pairs_kernel, dim3((npairs + 3) / 4, Bc), dim3(256), 0, s, g, A,
                       X, R, S, T, U, items);
    hipLaunchKernelGGL(refine_gram_final_kernel, dim3(Bc), dim3(256), 0, s, g, (const double *)S,
                       (const double *)T, (const double *)U, G, delta, items);
}

void launch_aux_update(const JobGeom &g, const ChunkPtrs &p, int Bc, int j, hipStream_t s) {
    const int n = (g.naux_pad / NB) * (g.nb0 - 1 - j);
    if (n <= 0) return;
    hipLaunchKernelGGL(aux_update_kernel, dim3((n + 3) / 4, Bc), dim3(256), 0, s, g, p, j);
}

bool launch_mixed_order(const ChunkPtrs &p, unsigned *prev, int32_t *order, int Bc, hipStream_t s) {
    // one workgroup ranks the chunk in LDS (4 B per item): chunks beyond NGP_MIXED_ORDER_MAX items
    // keep their dispatch order — with that many items a launch no longer ends on a few heavy
    // ones.  Returns whether `order` may be used (the launch was accepted).
    if (Bc > NGP_MIXED_ORDER_MAX) return false;
    (void)hipGetLastError();
    hipLaunchKernelGGL(mixed_order_kernel, dim3(1), dim3(256), sizeof(unsigned) * (size_t)Bc, s,
                       (const unsigned *)p.mixcnt, prev, order, Bc);
    return hipGetLastError() == hipSuccess;
}

constexpr int DIAG_AHEAD_SPLIT_K = 512;
void launch_diag_ahead(const JobGeom &g, const ChunkPtrs &p, int Bc, int j, hipStream_t s) {
    // the split depends on the geometry only (not on the batch), so a given matrix is always
    // summed in the same order
    if (j * NB >= DIAG_AHEAD_SPLIT_K)
        hipLaunchKernelGGL(diag_ahead_kernel<4>, dim3(Bc), dim3(256), 0, s, g, p, j);
    else
        hipLaunchKernelGGL(diag_ahead_kernel<1>, dim3(Bc), dim3(64), 0, s, g, p, j);
}

void launch_grad_kinv(const JobGeom &g, const double *L, double *Kinv, double *alpha, double *quad,
                      int Bc, hipStream_t s, hipStream_t side, hipEvent_t fork, hipEvent_t join) {
    // long series (2 x 2 tile blocks staged through LDS): alpha = W_I z comes out of the K^-1 kernel
    // itself, from the rows it stages; the alpha kernel is left with the quadratic form z'z (one
    // wave per item).  Shorter series: alpha reads every row of W once and does not depend on K^-1:
    // it runs beside it on the side stream.
    const bool lds = g.nb0 >= 8;
    const int a_first = lds ? g.n0 : 0;
    const dim3 agrid(lds ? 1 : (g.n0 + 1 + 3) / 4, Bc);
    const bool beside = side && fork && join;
    if (beside) {
        (void)hipEventRecord(fork, s);
        (void)hipStreamWaitEvent(side, fork, 0);
        hipLaunchKernelGGL(grad_alpha_kernel, agrid, dim3(256), 0, side, g, L, alpha, quad, a_first);
        (void)hipEventRecord(join, side);
    }
    const int npairs = g.nb0 * (g.nb0 + 1) / 2;
    if (lds) {   // HBM traffic halves against the wave-per-tile form
        const int nb2 = (g.nb0 + 1) / 2, nblk = nb2 * (nb2 + 1) / 2;
        hipLaunchKernelGGL(grad_kinv_lds_kernel, dim3(nblk * ((Bc + 7) / 8 * 8)), dim3(256), 0, s, g, L,
                           Kinv, nblk, Bc, alpha);
    } else {
        hipLaunchKernelGGL(grad_kinv_kernel, dim3((npairs + 3) / 4, Bc), dim3(256), 0, s, g, L, Kinv,
                           npairs);
    }
    if (beside)
        (void)hipStreamWaitEvent(s, join, 0);
    else
        hipLaunchKernelGGL(grad_alpha_kernel, agrid, dim3(256), 0, s, g, L, alpha, quad, a_first);
}

void launch_grad_contract(const JobGeom &g, const ChunkPtrs &p, const double *Kinv,
                          const double *alpha, const double *quad, double *partials, double *grad,
                          double *logml, int Bc, const DevSpec &sp, hipStream_t s0,
                          const int32_t *items, const int32_t *bucket_counts, hipStream_t side,
                          hipEvent_t fork, hipEvent_t join) {
    const int ntri = g.nb0 * (g.nb0 + 1) / 2;
    int nparts = ntri;
    // size classes of a SMALL chunk alternate between the main and the side stream: each is a few
    // rounds of the chip with a ragged last one, and they touch different items
    const bool two = side && fork && join && items && Bc <= 512;
    int nlaunched = 0;
    hipStream_t s = s0;
    if (two) {
        (void)hipEventRecord(fork, s0);
        (void)hipStreamWaitEvent(side, fork, 0);
    }
    if (g.lattice && p.dtab) {
        const int split = grad_contract_split(ntri, Bc, g.invariant != 0);
        // large launches: four tiles per workgroup (not under ngp_set_batch_invariant: the grouping of
        // a thread's partial sums would follow the batch size)
        // items sorted by tree size (grad_bucket): 1, 2, 4, 8 leaves on the register-accumulator
        // kernel, up to 16 leaves in two passes of it, larger trees on the general kernel
        const int32_t *it = items;
        int32_t whole[GRAD_BUCKETS] = {};
        whole[grad_bucket(g.maxops)] = Bc;
        const int32_t *cnt = items ? bucket_counts : whole;
        // (... and only when every item of the chunk runs on the lists kernel: the kernel of the
        // largest trees keeps one tile per workgroup, and a chunk has ONE partial-sum layout)
        const int tpw = (!g.invariant && split == 1 && (long)ntri * Bc >= 65536 && cnt[GRAD_BUCKETS - 1] == 0) ? 4 : 1;
        const int ngrp = (ntri + tpw - 1) / tpw;
        nparts = ngrp * split;
        for (int bk = 0; bk < GRAD_BUCKETS; ++bk) {
            const int nb = cnt[bk];
            if (nb <= 0) continue;
            s = (two && (nlaunched++ & 1)) ? side : s0;
            const dim3 grid(ngrp * split, nb), blk(256);
#define NGP_LAUNCH_LISTS(...)                                                                    \
    hipLaunchKernelGGL((grad_contract_lists_kernel<__VA_ARGS__>), grid, blk, 0, s, g, p, Kinv, alpha, \
                       partials, ntri, split, sp, it, tpw)
            if (bk == 0) NGP_LAUNCH_LISTS(1);
            else if (bk == 1) NGP_LAUNCH_LISTS(2);
            else if (bk == 2) NGP_LAUNCH_LISTS(4);
            else if (bk == 3) NGP_LAUNCH_LISTS(8);
            else if (bk == 4) {   // 9 .. 16 leaves: two passes of eight accumulator sets
                NGP_LAUNCH_LISTS(16, 8, 0);
                NGP_LAUNCH_LISTS(16, 8, 1);
            } else if (g.maxops <= LDSV_OPS)
                hipLaunchKernelGGL(grad_contract_lattice_kernel<true>, grid, blk, 0, s, g, p, Kinv,
                                   alpha, partials, ntri, split, sp, it);
            else
                hipLaunchKernelGGL(grad_contract_lattice_kernel<false>, grid, blk, 0, s, g, p, Kinv,
                                   alpha, partials, ntri, split, sp, it);
#undef NGP_LAUNCH_LISTS
            if (it) it += nb;
        }
    } else {
        hipLaunchKernelGGL(grad_contract_kernel, dim3(ntri, Bc), dim3(256), 0, s0, g, p.progs, p.t0,
                           Kinv, alpha, partials, ntri, sp);
    }
    if (two) {
        (void)hipEventRecord(join, side);
        (void)hipStreamWaitEvent(s0, join, 0);
    }
    hipLaunchKernelGGL(grad_reduce_kernel, dim3(Bc), dim3(128), 0, s0, g, p.progs, partials, quad,
                       p.logdet, grad, logml, nparts);
}

// The Toeplitz gradient path after the factorisation and the backward sweep: weights per distance
// from A = [a' ; x'], then the 1-D contraction (DIAG instantiations, by tree-size bucket) and the
// final sum.  items / bucket_counts as in launch_grad_contract (null: one launch sized by maxops).
void launch_toep_grad(const JobGeom &g, const ChunkPtrs &p, const double *A, double *wbuf,
                      const double *quad, double *partials, double *grad, double *logml, int Bc,
                      const DevSpec &sp, hipStream_t s, const int32_t *items,
                      const int32_t *bucket_counts) {
    const int nd = (g.n_real + 255) / 256;
    const size_t lds = sizeof(double) * (2 * (size_t)g.n_real + 1);   // <= 128 KiB: n <= 8192 (ngp_grad_stage)
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void *)toep_weights_kernel,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(toep_weights_kernel, dim3(nd, Bc), dim3(256), lds, s, g, A, wbuf);
    int32_t whole[GRAD_BUCKETS] = {};
    whole[grad_bucket(g.maxops)] = Bc;
    const int32_t *cnt = items ? bucket_counts : whole;
    const int32_t *it = items;
    for (int bk = 0; bk < GRAD_BUCKETS; ++bk) {
        const int nb = cnt[bk];
        if (nb <= 0) continue;
        const dim3 grid(nd, nb), blk(256);
#define NGP_LAUNCH_DIAG(...)                                                                     \
    hipLaunchKernelGGL((grad_contract_lists_kernel<__VA_ARGS__, true>), grid, blk, 0, s, g, p, wbuf, \
                       wbuf, partials, nd, 1, sp, it)
        if (bk == 0) NGP_LAUNCH_DIAG(1, 1, 0);
        else if (bk == 1) NGP_LAUNCH_DIAG(2, 2, 0);
        else if (bk == 2) NGP_LAUNCH_DIAG(4, 4, 0);
        else if (bk == 3) NGP_LAUNCH_DIAG(8, 8, 0);
        else {   // 9 .. 32 leaves: passes of eight accumulator sets (NGP_MAX_OPS = 64 nodes)
            NGP_LAUNCH_DIAG(16, 8, 0);
            NGP_LAUNCH_DIAG(16, 8, 1);
        }
#undef NGP_LAUNCH_DIAG
        if (it) it += nb;
    }
    hipLaunchKernelGGL(grad_reduce_kernel, dim3(Bc), dim3(128), 0, s, g, p.progs, partials, quad,
                       p.logdet, grad, logml, nd);
}

void launch_toep_quad(const JobGeom &g, const double *L, double *quad, int Bc, hipStream_t s) {
    hipLaunchKernelGGL(toep_quad_kernel, dim3(Bc), dim3(256), 0, s, g, L, quad);
}

void launch_gram(const JobGeom &g, const double *L, double *G, int Bc, hipStream_t s) {
    if (g.n0 == 0) return;
    hipLaunchKernelGGL(gram_kernel, dim3(Bc), dim3(256), 0, s, g, L, G);
}

void launch_epilogue(const JobGeom &g, const EpiPtrs &p, const DevSpec &sp, hipStream_t s) {
    const size_t bytes = 8 * ((size_t)g.da * g.da + (size_t)g.m * g.da + (size_t)g.da);
    const int lds_work = bytes > 0 && bytes <= 60 * 1024;
    hipLaunchKernelGGL(epilogue_kernel, dim3(g.B), dim3(64), lds_work ? bytes : 0, s, g, p, sp,
                       lds_work);
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
    hipLaunchKernelGGL(mfma4_composite_probe_kernel, dim3(1), dim3(64), 0, s, A, Bm, Dout + 256);
}
void launch_mfma_f32_probe(const float *A, const float *Bm, float *Dout, hipStream_t s) {
    hipLaunchKernelGGL(mfma_f32_probe_kernel, dim3(1), dim3(64), 0, s, A, Bm, Dout);
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
