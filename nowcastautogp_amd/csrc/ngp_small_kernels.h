// ngp_small_kernels.h — short series (n0 <= 256): the whole factorisation of an item in ONE launch.
//
// The column sweep of ngp_col_kernels.h is a chain of dependent launches per 64-wide block column
// (chol_diag -> fat / thin / full step).  At the reference's everyday size — a few hundred points,
// 24-64 particles (docs/vignettes/getting-started.jl:266-268) — that chain IS the call: 13-14
// launches of 5-58 us, most of them a handful of workgroups (DESIGN.md sections 4.9, 4.15).  Here one
// 512-thread workgroup per item (8 waves, two per SIMD, 256 VGPRs each) keeps the matrix in
// REGISTERS as 16 x 16 blocks in the v_mfma_f64_16x16x4 C/D layout and sweeps it right-looking,
// 16 pivots per step:
//
//   wave 0 ("pivot wave")  owns nothing but the diagonal blocks' critical path: it applies the last
//            rank-16 update to diagonal block j+1, factors it (one row per lane, every broadcast a
//            DPP row_newbcast, a reciprocal-only dependency chain — no barrier inside a block),
//            inverts it and posts M = L_jj^-1 in MFMA operand order — all of it while the other
//            waves run the trailing update of step j;
//   waves 1..7 ("workers") own the blocks below the diagonal and the aux rows, twenty each,
//            round-robin over a column-major enumeration (the blocks still active at step j are a
//            suffix of it, so every step is balanced to one block).  A block is held TRANSPOSED,
//            T = (A_ik)': in the C/D layout lane (r, q), register s is T[4s+q][r] = A_ik[r][4s+q] —
//            which is at once the B operand of the solve  X' = M C'  (no LDS round trip,
//            ngp_kernels.hip header) and, written to LDS as it stands, the A and the B operand of
//            every trailing update (A_ik)' -= L_kj L_ij' (contiguous, conflict-free ds_read_b64).
//   The diagonal blocks live in LDS (row stride 17); the ones not next in line are updated by the
//   workers, one block each per step.
//
// Two LDS-only barriers per step; the step's critical path is the pivot wave's 16 x 16 block
// (1.3-2.3 us), not a launch.  Aux rows that do not fit the register file beside the main block are
// swept afterwards in the same launch against the finished factor (its panels come back from L2
// one step ahead, the M_j are still in LDS), on all 8 waves; a gradient job's identity rows —
// W_I = L^-T — are not swept at all: every block column of L^-1 is one wave's task, barrier-free
// (small_inverse_column).  grad_kinv_small_kernel (K^-1 and alpha of such a job, one wave per
// 16 x 16 block) and chol_diag_wave_kernel (the same pivot wave on the 64 x 64 diagonal blocks of
// the column sweep, for small chunks) live here too.
//
// What the launch leaves in the slab is what its consumers read — the aux rows W = X L^-T of a value
// job, of a gradient job the blocks of W_I on and right of the block diagonal and z' — plus logdet
// and info; L itself only when a later sweep of the launch reads it back.  Results differ from the
// column sweep in the last bits (other summation order); an item's bits do not depend on the batch
// it travels in.
#pragma once
#include <atomic>
#include <type_traits>

#include "ngp_col_kernels.h"

namespace ngp {

struct SmallLds {
    double *Minv;     // [16][4][64]   M_j = L_jj^-1 in operand order: [t][m + 16 c] = M[m][4t + c]
    double *Dg;       // [16][16 x 17] diagonal blocks (full symmetric) until they are factored
    double *diagL;    // [256]         the pivots d = diag(L)^2, for logdet
    int    *bad;      // first failed pivot + 1
    double *Panel;    // [npanel][4][64] the finished blocks of column j, operand order
};

// f(std::integral_constant<int, I>) for I = FROM .. TO - 1: unrolled by construction (the DPP
// controls below are instruction immediates)
template <int FROM, int TO, class F>
__device__ __forceinline__ void static_for_up(F &&f) {
    if constexpr (FROM < TO) {
        f(std::integral_constant<int, FROM>{});
        static_for_up<FROM + 1, TO>(f);
    }
}
// lane N of every 16-lane row in all lanes of that row.  (A DPP source needs two wait states
// after the VALU instruction that wrote it; inline asm is outside the compiler's hazard
// recogniser, hence the s_nop.)
template <int N>
__device__ __forceinline__ double row_bcast_f64(double v) {
    double o;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
                 : "=v"(o) : "v"(v), "n"(N));
    return o;
}
// acc += (lane N of src's row) * mul
template <int N, bool GUARD>
__device__ __forceinline__ void fmac_row_bcast(double &acc, double src, double mul) {
    if constexpr (GUARD)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc) : "v"(src), "v"(mul), "n"(N));
    else
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc) : "v"(src), "v"(mul), "n"(N));
}

// ---- the pivot wave: diagonal block jn -----------------------------------------------------
// (1) the rank-16 update of column jn - 1 (its panel block is in LDS), (2) one row per lane,
// right-looking over the 16 pivots with v_readlane broadcasts — per element the operations of the
// textbook loop a_rn -= l_rc l_nc, c ascending —, (3) L_jj to the slab and to LDS, (4) M = L_jj^-1
// column by column (lane c: forward substitution against broadcast LDS reads), posted in operand
// order.  Lanes 16..63 mirror lanes 0..15 and store nothing.
template <class Probe>
__device__ __forceinline__ void small_pivot_block(const SmallLds &L, double *S, long ld, int jn,
                                                  bool update, int lane, int &badl, Probe &probe,
                                                  double *lout = nullptr, long ldl = 0) {
    const int r16 = lane & 15, q = lane >> 4;
    double *dg = L.Dg + jn * (16 * SM_DSTR);
    if (update) {
        const double *pp = L.Panel + jn * 256 + lane;
        f64x4 d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double v = pp[t * 64];
            d = mfma64(v, v, d);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) dg[(q + 4 * s) * SM_DSTR + r16] -= d[s];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    probe.mark(8 * jn + 1);
    // a[]: row r16 of the block; x[]: column r16 of the inverse in the making.  Broadcasts are DPP
    // row_newbcast inside the 16-lane row (the four rows of the wave mirror each other):
    // a[n] -= l_r l_n is ONE v_fmac_f64_dpp, no SGPR traffic.
    //
    // A wave issues in order, so the sixteen pivots are one chain of dependent operations and that
    // chain — not the instruction count — is this wave's time (measured: giving the inverse to a
    // second wave halved the instructions and changed nothing).  The loop therefore factors
    // K_jj = Lu D Lu' with a UNIT lower triangular Lu: per pivot one reciprocal (v_rcp seed, two
    // Newton steps), the multipliers a_r / d, and the update a_rn -= (a_r / d) a_n.  No square root
    // is on the chain.  The inverse of Lu needs no division at all and rides along (once x_c is
    // final, Lu[n][c] x_c leaves every later entry, with the multipliers just formed).  After the
    // loop the sixteen 1 / sqrt(d) are formed side by side and M = L_jj^-1 = D^-1/2 Lu^-1 is a row
    // scaling.  L_jj itself is needed by nobody (the solves take M, logdet takes the d).
    double a[16], x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        a[c] = dg[r16 * SM_DSTR + c];
        x[c] = (c == r16) ? 1.0 : 0.0;
    }
    double dsel = 1.0;                               // d of the lane's own row
    double piv = row_bcast_f64<0>(a[0]);
    static_for_up<0, 16>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        double ri = __builtin_amdgcn_rcp(piv);
        ri = fma(fma(-piv, ri, 1.0), ri, ri);
        ri = fma(fma(-piv, ri, 1.0), ri, ri);
        const double acol = a[c];                    // a_r[c] (the pivot itself in lane c)
        const double lt = acol * ri, nlt = -lt;      // Lu[r][c]
        if (!(piv > 0.0) && badl == 0) badl = 16 * jn + c + 1;
        dsel = (r16 == c) ? piv : dsel;
        if constexpr (c + 1 < 16) {
            // column c + 1 first: the next pivot is then on its way while the rest is applied
            fmac_row_bcast<c + 1, true>(a[c + 1], acol, nlt);
            piv = row_bcast_f64<c + 1>(a[c + 1]);
            const double nxc = -x[c];
            fmac_row_bcast<c + 1, false>(x[c + 1], lt, nxc);
            static_for_up<c + 2, 16>([&](auto nn) {
                constexpr int n = decltype(nn)::value;
                fmac_row_bcast<n, false>(a[n], acol, nlt);
                fmac_row_bcast<n, false>(x[n], lt, nxc);
            });
        }
    });
    // rows of the inverse scaled by 1 / sqrt(d_i) (lane i holds d_i)
    {
        double dk, rs;
        sqrt_and_rcp(dsel, dk, rs);
        static_for_up<0, 16>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            const double f = row_bcast_f64<i>(rs);
            x[i] *= f;
            a[i] = (r16 >= i) ? a[i] * f : 0.0;      // L_jj = Lu D^1/2 (a[i] is still the unscaled column)
        });
    }
    if (lout && lane < 16) {                         // chol_diag_wave_kernel: the column sweep stores L_jj
#pragma unroll
        for (int c = 0; c < 16; c += 2) {
            f64x2 w;
            w.x = a[c];
            w.y = a[c + 1];
            *reinterpret_cast<f64x2 *>(lout + (long)r16 * ldl + c) = w;
        }
    }
    probe.mark(8 * jn + 2);
    if (lane < 16) {
        L.diagL[16 * jn + r16] = dsel;               // d = diag(L)^2, for logdet
        double *mo = L.Minv + jn * 256 + (r16 >> 2) * 64 + 16 * (r16 & 3);
#pragma unroll
        for (int c = 0; c < 16; c += 2) {
            f64x2 w;
            w.x = x[c];
            w.y = x[c + 1];
            *reinterpret_cast<f64x2 *>(mo + c) = w;
        }
    }
    probe.mark(8 * jn + 3);
}

// The barrier of the sweep's steps: LDS traffic only.  __syncthreads() also waits for every global
// access of the wave (s_waitcnt vmcnt(0)): the panel loads an aux-only sweep has in flight for the
// NEXT step would be waited for at every barrier, and so would any store.  Inside a sweep nothing
// one wave writes to the slab is read by another (between sweeps: __syncthreads()).
__device__ __forceinline__ void small_bar() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// a wave-uniform value the optimiser may not look through: what is derived from it is recomputed
// where it is used (a handful of scalar instructions) instead of being hoisted out of the sweep's
// loop for all twenty slots at once, which spilled four hundred SGPRs
__device__ __forceinline__ int opaque_sgpr(int v) {
    asm volatile("" : "+s"(v));
    return v;
}

// blocks of column k in the enumeration of a sweep: main rows below the diagonal, identity
// row-blocks that have joined (a <= k), dense aux row-blocks
__device__ __forceinline__ int small_col_count(const SmallSweep &sw, int nbe, int k) {
    const int cm = sw.main ? nbe - 1 - k : 0;
    const int ci = max(min(sw.i1, k + 1) - sw.i0, 0);
    return cm + ci + (sw.a1 - sw.a0);
}

template <bool MAIN, class Probe>
__device__ __forceinline__ void small_sweep(const JobGeom &g, const ChunkPtrs &p, int item,
                                            const SmallPlan &pl, const SmallSweep sw,
                                            const SmallLds &L, double *S, int tid, int lane, int wave,
                                            int &badl, Probe &probe) {
    const int nbe = pl.nbe;
    const int r16 = lane & 15, q = lane >> 4;
    const long ld = g.ld;
    const bool store_main = pl.nsweeps > 1;       // value jobs of one sweep: nobody reads L again
    if (MAIN && wave == 0) {
        // ---- the pivot wave ----
        small_pivot_block(L, S, ld, 0, false, lane, badl, probe);
        small_bar();
        for (int j = 0; j < nbe; ++j) {
            small_bar();                           // the workers' solves of column j
            probe.mark(8 * (j + 1));
            if (j + 1 < nbe) small_pivot_block(L, S, ld, j + 1, true, lane, badl, probe);
            small_bar();
            probe.mark(8 * (j + 1) + 4);
        }
        return;
    }

    // ---- workers ----
    const int nwork = MAIN ? SM_WAVES - 1 : SM_WAVES;
    const int widx = MAIN ? wave - 1 : wave;
    int nblk = 0;
    for (int k = 0; k < nbe; ++k) nblk += small_col_count(sw, nbe, k);
    const int nid = sw.i1 - sw.i0;
    // per slot: block column (-1: none) and, packed, its panel index | slab row / 16 << 8 | the
    // column it joins at << 16 (identity row-block a: column a — its blocks are zero before that
    // and have no panel entry)
    int bk[SM_NSLOT], bw[SM_NSLOT];
    f64x4 acc[SM_NSLOT];
    // (a wave's slots ascend through the column-major enumeration: the scan for a slot's column
    // carries on where the previous slot's ended — in registers; a table of column starts in LDS,
    // one lookup per probe, made this prologue 10 us)
    int sk = 0, scs = 0, scnt = small_col_count(sw, nbe, 0);
#pragma unroll
    for (int s = 0; s < SM_NSLOT; ++s) {
        const int idx = widx + nwork * s;
        int k = -1, row16 = 0, pi = 0, ident = 0, adiag = 0, join = 0;
        if (idx < nblk) {
            while (idx >= scs + scnt) {
                scs += scnt;
                ++sk;
                scnt = small_col_count(sw, nbe, sk);
            }
            k = sk;
            const int rem = idx - scs;
            const int cm = sw.main ? nbe - 1 - k : 0;
            const int ci = max(min(sw.i1, k + 1) - sw.i0, 0);
            if (rem < cm) {
                row16 = k + 1 + rem;
                pi = k + 1 + rem;
            } else if (rem < cm + ci) {
                const int a = sw.i0 + (rem - cm);
                row16 = g.n0 / 16 + a;
                pi = nbe + (a - sw.i0);
                ident = 1;
                adiag = (a == k) ? 1 : 0;
                join = a;
            } else {
                const int a = sw.a0 + (rem - cm - ci);
                row16 = g.n0 / 16 + a;
                pi = nbe + nid + (a - sw.a0);
            }
        }
        bk[s] = __builtin_amdgcn_readfirstlane(k);
        bw[s] = __builtin_amdgcn_readfirstlane(pi | (row16 << 8) | (join << 16));
        ident = __builtin_amdgcn_readfirstlane(ident);
        adiag = __builtin_amdgcn_readfirstlane(adiag);
        if (bk[s] >= 0 && !ident && pl.ident && ((bw[s] >> 8) & 255) >= g.n0 / 8) {
            // gradient jobs: the row-block that carries y' (slab rows 2 n0 ...) straight from the
            // observations — the fill's aux launch is not run for short jobs (launch_fill)
            const double *yv = p.y0 + (g.y_shared ? 0 : (long)item * g.n0) + 16 * bk[s] + q;
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[s][t] = (r16 == 0) ? yv[4 * t] : 0.0;
        } else if (bk[s] >= 0 && !ident) {
            // 32 bytes per lane, a whole 128-byte row per four lanes (the layout the sweep works in —
            // lane (r, q) holds columns q, 4 + q, 8 + q, 12 + q — would be four 8-byte loads per lane
            // over sixteen lines each: the address path, not the data, then sets the prologue's time);
            // turned into that layout through LDS below
            const double *src = S + (long)(16 * ((bw[s] >> 8) & 255) + (lane >> 2)) * ld + 16 * bk[s] +
                                4 * (lane & 3);
            acc[s] = *reinterpret_cast<const f64x4 *>(src);
            bw[s] |= 1 << 24;
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[s][t] = (adiag && r16 == 4 * t + q) ? 1.0 : 0.0;
        }
    }
    {
        // the wave's own 16 x 18 corner of the (still unused) panel storage: rows as loaded in,
        // the sweep's layout out (LDS is in order within a wave: no barrier)
        double *scr = L.Panel + wave * (16 * 18);
#pragma unroll
        for (int s = 0; s < SM_NSLOT; ++s)
            if (bw[s] >> 24) {
                *reinterpret_cast<f64x4 *>(scr + (lane >> 2) * 18 + 4 * (lane & 3)) = acc[s];
                const double *rd = scr + r16 * 18 + q;
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[s][t] = rd[4 * t];
            }
    }
    if (!MAIN) small_bar();                        // the panel storage is the steps' from here on
    // aux-only sweeps: the main panel of a step (at most 15 blocks) comes back from the slab (L2),
    // one step ahead, two blocks per wave
    f64x4 pf[2];
    if (!MAIN) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pfk = 1 + wave + SM_WAVES * u;
            if (pfk < nbe) {
                const double *src = S + (long)(16 * pfk + r16) * ld + q;
#pragma unroll
                for (int t = 0; t < 4; ++t) pf[u][t] = src[4 * t];
            }
        }
    }
    probe.mark(7);
    if (MAIN) small_bar();                         // M_0 is posted
    for (int j = 0; j < nbe; ++j) {
        probe.mark(8 * (j + 1));
        // ---- solves of column j:  X' = M_j C'
        if (!MAIN) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int pfk = j + 1 + wave + SM_WAVES * u;
                if (pfk < nbe) {
                    double *dst = L.Panel + pfk * 256 + lane;
#pragma unroll
                    for (int t = 0; t < 4; ++t) dst[t * 64] = pf[u][t];
                }
            }
        }
        const double *mi = L.Minv + j * 256 + lane;
#pragma unroll
        for (int s = 0; s < SM_NSLOT; ++s)
            if (bk[s] == j) {
                f64x4 d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int t = 0; t < 4; ++t) d = mfma64(mi[t * 64], acc[s][t], d);
                acc[s] = d;
                const int w = opaque_sgpr(bw[s]);
                double *dst = L.Panel + (w & 255) * 256 + lane;
#pragma unroll
                for (int t = 0; t < 4; ++t) dst[t * 64] = d[t];
                // to the slab in the background (the steps' barriers do not wait for stores): aux
                // rows always, L itself only when a later sweep reads it back
                if (store_main || ((w >> 8) & 255) >= g.n0 / 16) {
                    double *out = S + (long)(16 * ((w >> 8) & 255) + r16) * ld + 16 * j + q;
#pragma unroll
                    for (int t = 0; t < 4; ++t) out[4 * t] = d[t];
                }
            }
        probe.mark(8 * (j + 1) + 1);
        small_bar();
        probe.mark(8 * (j + 1) + 2);
        // ---- trailing update:  (A_ik)' -= L_kj L_ij'
        if (!MAIN) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int pfk = j + 2 + wave + SM_WAVES * u;
                if (pfk < nbe) {
                    const double *src = S + (long)(16 * pfk + r16) * ld + 16 * (j + 1) + q;
#pragma unroll
                    for (int t = 0; t < 4; ++t) pf[u][t] = src[4 * t];
                }
            }
        }
#pragma unroll
        for (int s = 0; s < SM_NSLOT; ++s)
            if (bk[s] > j && ((bw[s] >> 16) & 255) <= j) {
                const int w = opaque_sgpr(bw[s]), k = opaque_sgpr(bk[s]);
                const double *pa = L.Panel + k * 256 + lane;
                const double *pb = L.Panel + (w & 255) * 256 + lane;
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[s] = mfma64(-pa[t * 64], pb[t * 64], acc[s]);
            }
        if (MAIN) {
            // the diagonal blocks that are not next in line, at most two per worker
            for (int k = j + 2 + widx; k < nbe; k += nwork) {
                const double *pp = L.Panel + k * 256 + lane;
                f64x4 d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const double v = pp[t * 64];
                    d = mfma64(v, v, d);
                }
                double *dg = L.Dg + k * (16 * SM_DSTR);
#pragma unroll
                for (int s = 0; s < 4; ++s) dg[(q + 4 * s) * SM_DSTR + r16] -= d[s];
            }
        }
        probe.mark(8 * (j + 1) + 3);
        small_bar();
    }
    probe.mark(8 * (nbe + 1));
}

// ---- the inverse phase of a gradient job: W_I = L^-T --------------------------------------
// Block column j of W = L^-1 depends on nothing but L and the M_i:
//     W_jj = M_j,    W_ij = -M_i sum_(k = j)^(i - 1) L_ik W_kj    (i = j + 1 ...),
// so every column is ONE wave's task, start to end in its registers, with no barrier at all (the
// right-looking sweep of the identity rows took two per block column and twice the time).  What the
// slab gets is W_I block (j, i) = W_ij'.
// Layouts.  W_kj is held with its rows permuted, register s of lane (n, q) = W_kj[4 q + s][n]: as
// the B operand of L_ik W_kj its k-slot (s, q) then stands for column 4 q + s of L_ik — the four
// consecutive columns lane (m, q) fetches with ONE 32-byte load from the slab — and the transposed
// block goes out as one 32-byte store per lane.  The permutation costs nothing: the product
// M_i (sum) delivers its rows in the order the A operand's lanes ask for M_i's rows, so lane m
// reads row 4 (m & 3) + (m >> 2) of M_i from LDS.
// L comes back from the slab (L2) through a ring of eight blocks requested eight products ahead.
constexpr int SM_RING = 6;
constexpr int sm_tri_row(int p) {   // stage ii >= 1 of product p: ii (ii - 1) / 2 <= p < ii (ii + 1) / 2
    int ii = 1;
    while (ii * (ii + 1) / 2 <= p) ++ii;
    return ii;
}
struct SmallInvCol {
    const double *S;       // the item's slab
    const double *Minv;    // LDS: the M_i in operand order
    double *Wout;          // slab row 16 j of W_I, column 0
    long ld;
    int j, len, nb16, lane;
    f64x4 W[16];
    f64x4 ring[SM_RING];
};
template <int P>
__device__ __forceinline__ void small_inv_request(SmallInvCol &c) {
    constexpr int ii = sm_tri_row(P), kk = P - ii * (ii - 1) / 2;
    if constexpr (ii < 16)
        if (ii < c.len) {
            const int r16 = c.lane & 15, q = c.lane >> 4;
            const double *src = c.S + (long)(16 * (c.j + ii) + r16) * c.ld + 16 * (c.j + kk) + 4 * q;
            c.ring[P % SM_RING] = *reinterpret_cast<const f64x4 *>(src);
        }
}
template <int II>
__device__ __forceinline__ void small_inv_stage(SmallInvCol &c) {
    if constexpr (II < 16) {
        if (II >= c.len) return;
        const int r16 = c.lane & 15, q = c.lane >> 4;
        f64x4 s0 = {0.0, 0.0, 0.0, 0.0}, s1 = {0.0, 0.0, 0.0, 0.0};
        static_for_up<0, II>([&](auto kc) {
            constexpr int kk = decltype(kc)::value, P = II * (II - 1) / 2 + kk;
            const f64x4 a = c.ring[P % SM_RING];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (kk & 1) s1 = mfma64(a[s], c.W[kk][s], s1);
                else s0 = mfma64(a[s], c.W[kk][s], s0);
            }
            small_inv_request<P + SM_RING>(c);
        });
        if constexpr (II > 1) s0 += s1;
        // W_ij = -M_i (sum), rows in the permuted order
        const double *mi = c.Minv + (c.j + II) * 256 + 4 * (r16 & 3) + (r16 >> 2) + 16 * q;
        f64x4 d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < 4; ++t) d = mfma64(-mi[t * 64], s0[t], d);
        c.W[II] = d;
        *reinterpret_cast<f64x4 *>(c.Wout + (long)r16 * c.ld + 16 * (c.j + II) + 4 * q) = d;
        small_inv_stage<II + 1>(c);
    }
}
__device__ __forceinline__ void small_inverse_column(const SmallLds &L, double *S, long ld, int n0, int nbe,
                                                     int j, int lane) {
    const int r16 = lane & 15, q = lane >> 4;
    SmallInvCol c;
    c.S = S;
    c.Minv = L.Minv;
    c.Wout = S + (long)(n0 + 16 * j) * ld;
    c.ld = ld;
    c.j = j;
    c.len = nbe - j;
    c.nb16 = n0 / 16;
    c.lane = lane;
    static_for_up<0, SM_RING>([&](auto pc) { small_inv_request<decltype(pc)::value>(c); });
    // W_jj = M_j: register s of lane (n, q) = M_j[4 q + s][n] (operand order: [n >> 2][.. + 16 (n & 3)])
    c.W[0] = *reinterpret_cast<const f64x4 *>(L.Minv + j * 256 + (r16 >> 2) * 64 + 16 * (r16 & 3) + 4 * q);
    *reinterpret_cast<f64x4 *>(c.Wout + (long)r16 * ld + 16 * j + 4 * q) = c.W[0];
    small_inv_stage<1>(c);
}

template <class Probe = NoProbe>
__global__ __launch_bounds__(SM_THREADS) void chol_small_kernel(JobGeom g, ChunkPtrs p, SmallPlan pl) {
    extern __shared__ double sm_lds[];
    SmallLds L;
    L.Minv = sm_lds;
    L.Dg = L.Minv + 16 * 256;
    L.diagL = L.Dg + 16 * 16 * SM_DSTR;
    L.bad = reinterpret_cast<int *>(L.diagL + 256);
    L.Panel = L.diagL + 256 + 32;
    const int item = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const long ld = g.ld;
    const int nb16 = g.n0 / 16, nbe = pl.nbe;
    double *S = p.L + (long)item * g.item_stride;
    Probe probe;
    probe.begin(item, wave, lane);
    probe.mark(150);

    // (What the launch leaves in the slab is what its consumers read: the aux rows W of a value
    // job — gram_kernel —, of a gradient job the blocks of W_I = L^-T on and right of the block
    // diagonal of the data rows and z' — grad_kinv_small_kernel —, and L itself only when a later
    // sweep of this launch reads it back.  Nothing else of the column sweep's image is written.)
    if (tid == 0) {
        *L.bad = 0;
    }
    // the diagonal blocks to LDS
    for (int w = wave; w < nbe; w += SM_WAVES) {
        const double *src = S + (long)(16 * w + r16) * ld + 16 * w + q;
        double *dg = L.Dg + w * (16 * SM_DSTR) + r16 * SM_DSTR + q;
#pragma unroll
        for (int t = 0; t < 4; ++t) dg[4 * t] = src[4 * t];
    }
    small_bar();                                   // (the initialising stores drain in the background)

    int badl = 0;
    probe.mark(151);
    for (int si = 0; si < pl.nsweeps; ++si) {
        if (si > 0) __syncthreads();
        probe.sweep(si);
        if (pl.sw[si].main == 2) {
            for (int j = 0; j < nbe; ++j)
                if ((int)((pl.colwave >> (4 * j)) & 15) == wave) small_inverse_column(L, S, ld, g.n0, nbe, j, lane);
            probe.mark(8 * (nbe + 1));
        } else if (pl.sw[si].main) small_sweep<true>(g, p, item, pl, pl.sw[si], L, S, tid, lane, wave, badl, probe);
        else small_sweep<false>(g, p, item, pl, pl.sw[si], L, S, tid, lane, wave, badl, probe);
    }
    if (wave == 0 && lane == 0 && badl) *L.bad = badl;
    __syncthreads();
    // G = W W' of the aux rows, here instead of in a launch of its own (gram_kernel: 20 us of a
    // 24-item call's 130; this tail: 11): one pair of rows per thread, the serial dot product of
    // gram_kernel's short-row form — the same operations in the same order, so the same bits for the
    // same W.  (Four threads per pair, a quarter of the columns each, was slower: 15 us.)
    if (p.G) {
        const double *W = S + (long)g.n0 * ld;
        double *Go = p.G + (long)item * g.naux * g.naux;
        for (int e = tid; e < g.naux * g.naux; e += SM_THREADS) {
            const int a = e / g.naux, b = e % g.naux;
            if (b > a) continue;
            const double *wa = W + (long)a * ld, *wb = W + (long)b * ld;
            double s0 = 0.0, s1 = 0.0;
            for (int k = 0; k < g.n0; k += 2) {
                s0 += wa[k] * wb[k];
                s1 += wa[k + 1] * wb[k + 1];
            }
            const double sv = s0 + s1;
            Go[a * g.naux + b] = sv;
            Go[b * g.naux + a] = sv;
        }
    }
    // logdet = sum of log diag(L) over the data rows (padding rows are identity)
    if (wave == 0) {
        double s = 0.0;
        for (int i = lane; i < 16 * nbe; i += 64)
            if (i < g.n_real) s += 0.5 * log(L.diagL[i]);      // diagL holds the pivots d = diag(L)^2
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) {
            // written, not accumulated: this launch is the whole factorisation of the item, so a
            // re-run needs no memset of logdet | info before it (one link less in the chain)
            p.logdet[item] = s;
            p.info[item] = *L.bad;
        }
    }
    probe.sweep(0);
    probe.mark(152);
}

// ---------------------------------------------------------------------------------------
// K^-1 = W_I W_I' and alpha = W_I z of a short gradient job (n0 <= 256), behind chol_small_kernel.
// grad_kinv_kernel gives a whole 64 x 64 tile pair to one wave: ten waves per item at n0 = 256,
// each a k-loop of up to 256 on one SIMD (29 us of matrix-core time), 72 us per call with the
// alpha launch beside it.  Here a wave takes ONE 16 x 16 block (a >= b) of the data rows — up to
// 136 waves per item, k from 16 a (W_I is upper triangular) to the end of the data columns —
// with its operands straight from L2 in v_mfma_f64_16x16x4 operand order (lane (m, q): W[16a + m]
// [k + q]), sixteen k-slices (64 columns) requested together.  The diagonal blocks' waves also sum
// alpha for their sixteen rows from the A operands they hold anyway; block (0, 0)'s wave adds z'z.
// Only what the contraction reads is written: rows and columns < n_real, col <= row.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grad_kinv_small_kernel(JobGeom g, const double *L, double *Kinv,
                                                              double *alpha, double *quad, int nbe) {
    const int item = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pr = blockIdx.x * 4 + wave;
    if (pr >= nbe * (nbe + 1) / 2) return;
    int a = (int)((sqrt(8.0 * pr + 1.0) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) / 2 <= pr) ++a;
    while (a * (a + 1) / 2 > pr) --a;
    const int b = pr - a * (a + 1) / 2;   // a >= b
    const long ld = g.ld;
    const int r16 = lane & 15, q = lane >> 4;
    const double *W = L + (long)item * g.item_stride + (long)g.n0 * ld;
    const double *z = W + (long)g.n0 * ld;
    // a Gram product may take its k in any order as long as both operands agree: lane (m, q) reads
    // FOUR consecutive columns (32 bytes) of its row per 16-column group — slice s of the group pairs
    // k-lane q with column 4 q + s — so a 64-column chunk is 8 + 8 wide loads instead of 32 narrow
    // ones (the narrow form spent the kernel in the address path: 37 us per call)
    const double *pa = W + (long)(16 * a + r16) * ld + 4 * q;
    const double *pb = W + (long)(16 * b + r16) * ld + 4 * q;
    const int kend = 16 * nbe;
    const bool diag = a == b;
    f64x4 d0 = {0.0, 0.0, 0.0, 0.0}, d1 = {0.0, 0.0, 0.0, 0.0};
    double al = 0.0;
    for (int k0 = 16 * a; k0 < kend; k0 += 64) {
        // past the data columns a group is re-read from the last valid place and multiplied by
        // zero (uniform code, every load of the chunk in flight at once)
        f64x4 av[4], bv[4], zv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = min(k0 + 16 * u, kend - 16);
            av[u] = *reinterpret_cast<const f64x4 *>(pa + k);
            bv[u] = *reinterpret_cast<const f64x4 *>(pb + k);
            if (diag) zv[u] = *reinterpret_cast<const f64x4 *>(z + k + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool live = k0 + 16 * u < kend;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double x = live ? av[u][s] : 0.0;
                if (s & 1) d1 = mfma64(x, bv[u][s], d1);
                else d0 = mfma64(x, bv[u][s], d0);
                if (diag) al = fma(x, zv[u][s], al);
            }
        }
    }
    double *Ko = Kinv + (long)item * g.n0 * g.n0;
#pragma unroll
    for (int s = 0; s < 4; ++s) Ko[(long)(16 * a + q + 4 * s) * g.n0 + 16 * b + r16] = d0[s] + d1[s];
    if (diag) {
        al += __shfl_xor(al, 16, 64);
        al += __shfl_xor(al, 32, 64);
        if (q == 0) alpha[(long)item * g.n0 + 16 * a + r16] = al;
    }
    if (pr == 0) {
        double s = 0.0;
        for (int i = lane; i < g.n_real; i += 64) s += z[i] * z[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) quad[item] = s;
    }
}

// ---------------------------------------------------------------------------------------
// chol_diag, second form: the 64 x 64 diagonal block of block column j of the column sweep, factored
// and inverted the way chol_small_kernel does it — as 4 x 4 blocks of 16, wave 1 the pivot wave
// (no barrier inside a block, DPP broadcasts, unit-factor chain), the six blocks below the
// diagonal in the registers of the waves that staged them, two LDS-only barriers per 16 pivots;
// M = L_jj^-1 by block columns (one wave each), written in the strip order the column kernels'
// epilogues read.  chol_diag_kernel retires four pivots per two barriers with one active lane
// factoring each 4 x 4 block and then runs the inverse as six dependent tile steps: 17.7 + 9 us of
// its 38 at small batches (profiles/r04/chol_diag_phases_24_items.txt).
// Staging (C_jj = K_jj - L_j L_j' over the pending columns) is chol_diag_kernel's, except that the
// off-diagonal quadrant is formed TRANSPOSED (rows 0..31 x columns 32..63): a block in the C/D
// layout is then already the transposed block the sweep works on (ngp_small_kernels.h header), so
// every block stays in the wave that staged it.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void chol_diag_wave_kernel(JobGeom g, ChunkPtrs p, int j, int k0) {
    __shared__ double Dg[4 * 16 * SM_DSTR];   // the four diagonal 16 x 16 blocks (full symmetric)
    __shared__ double Minv[4 * 256];          // M_i = L_ii^-1, operand order
    __shared__ double Pan[4 * 256];           // the solved blocks of the current block column, by row block
    __shared__ double Lblk[6 * 256];          // all six blocks of L below the diagonal: (i, k) at i (i - 1) / 2 + k
    __shared__ double dL[64];
    __shared__ int misc[2];
    const int item = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long ld = g.ld;
    double *Lit = p.L + (long)item * g.item_stride;
    double *Lj = Lit + (long)j * NB * ld;     // rows of block j
    const int kmax = j * NB;
    const int r16 = lane & 15, q = lane >> 4;
    SmallLds L;
    L.Minv = Minv;
    L.Dg = Dg;
    L.diagL = dL;
    L.bad = misc + 1;
    L.Panel = Pan;
    NoProbe probe;
    // ---- staging.  wave 0: rows / columns 0..31; wave 3: 32..63; wave 2: rows 0..31 x columns 32..63
    //      (the transpose of the quadrant below the diagonal); wave 1 (the pivot wave) stages nothing
    f64x4 blk[4];                             // wave 0 / 3: [0] (lo,lo) [1] (lo,hi) [3] (hi,hi); wave 2: all four
    const int wr = wave == 3 ? 1 : 0, wc = wave == 0 ? 0 : 1;
    if (wave != 1) {
        double acc4[2][2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc4[a][b][r] = 0.0;
        const double *pa = Lj + (long)(32 * wr + r16) * ld + 2 * q;
        const double *pb = Lj + (long)(32 * wc + r16) * ld + 2 * q;
        // the K tile in the D layout of the product, read from the part of the tile ON OR BELOW
        // the diagonal (what the steps before this one keep up to date): element (M, N) with M < N
        // is taken from (N, M)
        double kt[2][2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int M = 32 * wr + 16 * mt + q + 4 * s, N = 32 * wc + 16 * nt + r16;
                    kt[mt][nt][s] = Lj[(long)max(M, N) * ld + kmax + min(M, N)];
                }
        for (int kc = k0; kc < kmax; kc += 16) {
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
                for (int nt = 0; nt < 2; ++nt) {
                    const Rot4 br = rot4(b[nt][s]);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) mfma16_as_4(acc4[mt][nt], a[mt][s], br);
                }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f64x4 d = to_d16(acc4[mt][nt]);
#pragma unroll
                for (int s = 0; s < 4; ++s) blk[2 * mt + nt][s] = kt[mt][nt][s] - d[s];
            }
        if (wave != 2) {   // the two diagonal blocks of the quadrant to LDS (register s of lane (n, q): row q + 4 s, column n)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                double *dg = Dg + (2 * wr + b) * (16 * SM_DSTR);
#pragma unroll
                for (int s = 0; s < 4; ++s) dg[(q + 4 * s) * SM_DSTR + r16] = blk[3 * b][s];
            }
        }
    }
    if (tid == 0) misc[1] = 0;
    // the wave's blocks below the diagonal, as (row block, column block) of the 4 x 4 grid: the C/D
    // registers of block (k, i) ARE the transposed block (i, k) the sweep keeps
    //   wave 0: blk[1] = (0,1) -> (1,0);  wave 3: blk[1] = (2,3) -> (3,2);
    //   wave 2: blk[2 mt + nt] = (mt, 2 + nt) -> (2 + nt, mt)
    int si[4], sk[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        si[b] = -1;
        sk[b] = -1;
    }
    if (wave == 0) { si[1] = 1; sk[1] = 0; }
    if (wave == 3) { si[1] = 3; sk[1] = 2; }
    if (wave == 2) {
#pragma unroll
        for (int b = 0; b < 4; ++b) { si[b] = 2 + (b & 1); sk[b] = b >> 1; }
    }
    __syncthreads();
    int badl = 0;
    double *Ljj = Lj + kmax;                  // the 64 x 64 tile in the slab
    if (wave == 1) {
        // ---- the pivot wave (its own branch: its registers are the block it factors, not blocks it holds)
        small_pivot_block(L, Lit, ld, 0, false, lane, badl, probe, Ljj, ld);
        small_bar();
        for (int c = 0; c < 4; ++c) {
            small_bar();                      // the solves of block column c
            if (c + 1 < 4)
                small_pivot_block(L, Lit, ld, c + 1, true, lane, badl, probe,
                                  Ljj + (long)16 * (c + 1) * ld + 16 * (c + 1), ld);
            small_bar();
        }
    } else {
        small_bar();                          // M_0 is posted
        for (int c = 0; c < 4; ++c) {
            // ---- solves of block column c: X' = M_c C'
            const double *mi = Minv + c * 256 + lane;
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (sk[b] == c) {
                    f64x4 d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int t = 0; t < 4; ++t) d = mfma64(mi[t * 64], blk[b][t], d);
                    blk[b] = d;
                    double *pn = Pan + si[b] * 256 + lane;
                    double *lb = Lblk + (si[b] * (si[b] - 1) / 2 + c) * 256 + lane;
                    double *out = Ljj + (long)(16 * si[b] + r16) * ld + 16 * c + q;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        pn[t * 64] = d[t];
                        lb[t * 64] = d[t];
                        out[4 * t] = d[t];
                    }
                }
            small_bar();
            // ---- trailing update (the pivot wave is on the next diagonal block meanwhile)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (sk[b] > c) {
                    const double *pa = Pan + sk[b] * 256 + lane, *pb = Pan + si[b] * 256 + lane;
#pragma unroll
                    for (int t = 0; t < 4; ++t) blk[b] = mfma64(-pa[t * 64], pb[t * 64], blk[b]);
                }
            // the diagonal blocks that are not next in line: (c + 2) on wave 0, (c + 3) on wave 3
            const int kk = wave == 0 ? c + 2 : (wave == 3 ? c + 3 : 4);
            if (kk < 4) {
                const double *pp = Pan + kk * 256 + lane;
                f64x4 d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const double v = pp[t * 64];
                    d = mfma64(v, v, d);
                }
                double *dg = Dg + kk * (16 * SM_DSTR);
#pragma unroll
                for (int s = 0; s < 4; ++s) dg[(q + 4 * s) * SM_DSTR + r16] -= d[s];
            }
            small_bar();
        }
    }
    // ---- M = L_jj^-1 by block columns, wave w the column w:  W_ww = M_w,
    //      W_iw = -M_i sum_(k = w)^(i - 1) L_ik W_kw;  out in strip order (chol_diag_kernel):
    //      element (R, C) at  ((R >> 2) * 4 + (C >> 4)) * 64 + (R & 3) + 4 ((C & 15) >> 2) + 16 (C & 3)
    {
        double *dv = p.dinv + (long)item * (NB * NB);
        auto put = [&](int bi, int bj, const f64x4 &d) {   // register s of lane (n, q): row q + 4 s, column n
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int R = 16 * bi + q + 4 * s;
                dv[((R >> 2) * 4 + bj) * 64 + (R & 3) + 4 * (r16 >> 2) + 16 * (r16 & 3)] = d[s];
            }
        };
        const int w = wave;
        f64x4 W[4];
        {
            const double *mo = Minv + w * 256 + (r16 >> 2) * 64 + 16 * (r16 & 3) + q;
#pragma unroll
            for (int s = 0; s < 4; ++s) W[0][s] = mo[4 * s];
        }
        put(w, w, W[0]);
        const f64x4 zero = {0.0, 0.0, 0.0, 0.0};
        for (int bi = 0; bi < w; ++bi) put(bi, w, zero);     // the blocks above the diagonal in this column
#pragma unroll
        for (int ii = 1; ii < 4; ++ii) {
            const int i = w + ii;
            if (i < 4) {
                f64x4 sacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < ii; ++kk) {
                    const double *la = Lblk + (i * (i - 1) / 2 + (w + kk)) * 256 + lane;
#pragma unroll
                    for (int t = 0; t < 4; ++t) sacc = mfma64(la[t * 64], W[kk][t], sacc);
                }
                const double *mi = Minv + i * 256 + lane;
                f64x4 d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int t = 0; t < 4; ++t) d = mfma64(-mi[t * 64], sacc[t], d);
                W[ii] = d;
                put(i, w, d);
            }
        }
    }
    // the 16-blocks above the diagonal of L_jj are zero (the diagonal blocks' own upper parts are
    // written with them by the pivot wave)
    {
        const int c4 = tid & 15, rr = tid >> 4;          // four columns, sixteen row classes
        for (int row = rr; row < 64; row += 16)
            if ((c4 >> 2) > (row >> 4)) {
                double *dst = Ljj + (long)row * ld + 4 * c4;
                const f64x2 z = {0.0, 0.0};
                *reinterpret_cast<f64x2 *>(dst) = z;
                *reinterpret_cast<f64x2 *>(dst + 2) = z;
            }
    }
    if (wave == 1 && lane == 0 && badl) misc[1] = badl;
    __syncthreads();
    if (wave == 0) {
        double sl = 0.5 * log(dL[lane]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sl += __shfl_down(sl, off, 64);
        if (lane == 0) {
            p.logdet[item] += sl;
            const int b = misc[1];
            if (b && p.info[item] == 0) p.info[item] = kmax + b;
        }
    }
}

void launch_chol_diag_wave(const JobGeom &g, const ChunkPtrs &p, int Bc, int j, int k0, hipStream_t s) {
    hipLaunchKernelGGL(chol_diag_wave_kernel, dim3(Bc), dim3(256), 0, s, g, p, j, k0);
}

void launch_grad_kinv_small(const JobGeom &g, const double *L, double *Kinv, double *alpha, double *quad,
                            int Bc, hipStream_t s) {
    const int nbe = (g.n_real + 15) / 16, nblk = nbe * (nbe + 1) / 2;
    hipLaunchKernelGGL(grad_kinv_small_kernel, dim3((nblk + 3) / 4, Bc), dim3(256), 0, s, g, L, Kinv, alpha,
                       quad, nbe);
}

void launch_chol_small(const JobGeom &g, const ChunkPtrs &p, int Bc, const SmallPlan &pl,
                       hipStream_t s) {
    // more LDS than the 64 KiB a kernel gets unasked: once per device of the process (the attribute
    // belongs to the kernel's image on that device)
    static std::atomic<unsigned long long> attr_set{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        (void)hipFuncSetAttribute((const void *)chol_small_kernel<NoProbe>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  SM_LDS_FIXED + SM_MAX_PANEL * 2048);
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL(chol_small_kernel<NoProbe>, dim3(Bc), dim3(SM_THREADS), small_lds_bytes(pl), s, g, p, pl);
}

}  // namespace ngp
