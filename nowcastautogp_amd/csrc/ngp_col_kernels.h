// ngp_col_kernels.h — the column sweep of the blocked Cholesky: chol_diag_kernel and the
// chol_col kernels (FAT / THIN / FULL steps), templated on a probe policy.
//
// Probe policy.  The product (ngp_kernels.hip) instantiates every kernel here with NoProbe: all of
// its hooks are empty inline functions and it has no state, so the kernels compile exactly as if
// the hooks were not written.  The diagnostic translation unit scripts/stamps/ngp_stamps.hip
// instantiates the same kernels with a probe that records 100-MHz timestamps per wave
// (scripts/fat_phases.py); nothing of it is compiled into libngp.so.
#pragma once
#include <algorithm>

#include "ngp_mfma.h"

namespace ngp {

struct NoProbe {
    // slot <- the time now
    __device__ __forceinline__ void mark(int) {}
    // slot <- the time once `dep` has been computed (anchors the stamp in the instruction stream)
    template <class T>
    __device__ __forceinline__ void mark_after(int, T) {}
    // wait until every outstanding memory operation of the wave has retired
    __device__ __forceinline__ void drain() {}
    // record the wave's stamps if block column j is the one being watched
    __device__ __forceinline__ void emit_diag(int, int, int) {}
    __device__ __forceinline__ void emit_col(int, int, int, int, int, int, bool) {}
    // chol_small_kernel: a wave's identity / the sweep it is in
    __device__ __forceinline__ void begin(int, int, int) {}
    __device__ __forceinline__ void sweep(int) {}
};

// ---------------------------------------------------------------------------------------
// chol_diag: factor the 64x64 diagonal block of block column j
// ---------------------------------------------------------------------------------------
// d = sqrt(a) and r = 1/d for a pivot, on the serial critical path of the factorisation: one
// v_rsq_f64 seed, two Newton steps on the reciprocal root, one correction each for d and r
// (a dozen dependent FMAs instead of the ~40 of sqrt() followed by a division; both results
// within 1 ulp).  a <= 0 or NaN gives non-finite values; the caller flags the pivot.
__device__ __forceinline__ void sqrt_and_rcp(double a, double &d, double &r) {
    double y = __builtin_amdgcn_rsq(a);
    const double h = 0.5 * a;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    d = a * y;
    d = fma(fma(-d, d, a), 0.5 * y, d);
    r = fma(fma(-d, y, 1.0), y, y);
}

// Lower-triangular tiles live in LDS packed by rows (38 KB per workgroup in total, so four
// workgroups share a CU: the kernel is a chain of short dependent phases and gains from occupancy).
__device__ __forceinline__ int tri(int r, int c) { return ((r * (r + 1)) >> 1) + c; }   // r >= c
constexpr int TRI = NB * (NB + 1) / 2;

template <class Probe = NoProbe>
__global__ __launch_bounds__(256, 4) void chol_diag_kernel(JobGeom g, ChunkPtrs p, int j, int k0) {
    __shared__ double Mt[TRI];        // C_jj while it is staged, then M = L_jj^-1
    __shared__ double Lt[TRI];        // L_jj
    __shared__ double pan[NB][4];     // the 64x4 panel of the current round
    __shared__ double dblk[16];       // 4x4 factor of the current diagonal block, row-major
    __shared__ double dinvd[4];       // reciprocals of its diagonal
    __shared__ double rdg[NB];        // 1 / diag(L_jj)
    __shared__ double logs[NB];
    __shared__ double Ts[TB][TB + 1];
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
    Probe probe;
    probe.mark_after(0, lane);

    if (tid == 0) bad = 0;

    // ---- C_jj = K_jj - L_j L_j'   (each wave one 32x32 quadrant; the strictly upper one is
    //      never read by the factorisation and is skipped)
    if (!(wr == 0 && wc == 1)) {
        double acc4[2][2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc4[a][b][r] = 0.0;
        const double *pa = Lj + (long)(32 * wr + r16) * ld + 2 * q;
        const double *pb = Lj + (long)(32 * wc + r16) * ld + 2 * q;
        // the K tile itself, in the D layout of the product (loaded first: its latency hides
        // under the k-loop): register s of kt[mt][nt] is K[32wr+16mt+q+4s][32wc+16nt+r16]
        double kt[2][2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    kt[mt][nt][s] = Lj[(long)(32 * wr + 16 * mt + q + 4 * s) * ld + kmax +
                                       32 * wc + 16 * nt + r16];
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
        f64x4 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = to_d16(acc4[a][b]);
        // D layout: register s of acc[mt][nt] is S[M = 32wr+16mt+q+4s][N = 32wc+16nt+r16]
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int M = 32 * wr + 16 * mt + q + 4 * s, N = 32 * wc + 16 * nt + r16;
                    if (M >= N) Mt[tri(M, N)] = kt[mt][nt][s] - acc[mt][nt][s];
                }
    }
    __syncthreads();
    probe.mark_after(1, lane);

    // ---- right-looking Cholesky of the 64x64 tile, register-blocked: thread (bi, bj) owns the 4x4
    //      block rows 4bi.., cols 4bj...  Four pivots are retired per round (16 rounds, two barriers
    //      each): the diagonal thread factors its 4x4 block in registers, the threads of block
    //      column kb solve their block against it and post the 64x4 panel through LDS, the blocks
    //      to the right apply the rank-4 update in registers.  Per element the operations and
    //      their order are those of the textbook loop  a_ij -= l_ik l_jk,  k ascending — only the
    //      synchronisation is coarser (one pivot per barrier measured 38 us of this kernel's 63).
    const int bi = tid >> 4, bj = tid & 15;
    {
        double a[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int R = 4 * bi + r, C = 4 * bj + c;
                a[r][c] = (R >= C) ? Mt[tri(R, C)] : 0.0;
            }
        for (int kb = 0; kb < NB / 4; ++kb) {
            if (bi == kb && bj == kb) {
                double dks[4];
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    const double akk = a[kc][kc];
                    double dk, inv;
                    sqrt_and_rcp(akk, dk, inv);
                    if (!(akk > 0.0) && bad == 0) bad = 4 * kb + kc + 1;
                    dks[kc] = dk;
                    dinvd[kc] = inv;
                    rdg[4 * kb + kc] = inv;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r > kc) a[r][kc] *= inv;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c > kc) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (r >= c) a[r][c] -= a[r][kc] * a[c][kc];
                        }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (r == c) a[r][c] = dks[r];
                        if (c > r) a[r][c] = 0.0;
                        dblk[4 * r + c] = a[r][c];
                    }
            }
            __syncthreads();
            if (bj == kb && bi > kb) {
                // x[r][c] = (a[r][c] - sum_{p<c} x[r][p] L[c][p]) / L[c][c], p ascending
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double inv = dinvd[c];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        double v = a[r][c];
#pragma unroll
                        for (int pp = 0; pp < 4; ++pp)
                            if (pp < c) v -= a[r][pp] * dblk[4 * c + pp];
                        a[r][c] = v * inv;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) pan[4 * bi + r][c] = a[r][c];
            }
            __syncthreads();
            if (bj > kb && bi >= bj) {
                double li[4][4], lj[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp) {
                        li[r][pp] = pan[4 * bi + r][pp];
                        lj[r][pp] = pan[4 * bj + r][pp];
                    }
#pragma unroll
                for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < 4; ++c) a[r][c] -= li[r][pp] * lj[c][pp];
            }
        }
        // L_jj: to LDS for the inverse, and straight from the registers back to the factor
        // storage (strict upper part zero)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int R = 4 * bi + r;
            f64x2 lo, hi;
            lo.x = (R >= 4 * bj) ? a[r][0] : 0.0;
            lo.y = (R >= 4 * bj + 1) ? a[r][1] : 0.0;
            hi.x = (R >= 4 * bj + 2) ? a[r][2] : 0.0;
            hi.y = (R >= 4 * bj + 3) ? a[r][3] : 0.0;
            double *dst = Lj + (long)R * ld + kmax + 4 * bj;
            *reinterpret_cast<f64x2 *>(dst) = lo;
            *reinterpret_cast<f64x2 *>(dst + 2) = hi;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (R >= 4 * bj + c) Lt[tri(R, 4 * bj + c)] = a[r][c];
        }
    }
    __syncthreads();
    probe.mark_after(2, lane);
    if (tid < NB) logs[tid] = log(Lt[tri(tid, tid)]);

    // ---- M = L_jj^-1 (64 x 64, lower): the four 16x16 diagonal-block inverses by forward
    //      substitution (column c of block b per thread), then the six off-diagonal tiles by block
    //      recursion  M[ct][jt] = -M[ct][ct] (sum_{k=jt}^{ct-1} L[ct][k] M[k][jt]),  one tile at a
    //      time over 256 threads.  Mt is free now (C_jj went to registers before the factorisation).
    if (tid < NB) {
        const int b = tid >> 4, c = tid & 15;
        double x[TB];
#pragma unroll
        for (int i = 0; i < TB; ++i) {
            double sum = (i == c) ? 1.0 : 0.0;
            int off = tri(TB * b + i, TB * b);   // one address per row, pp by offset
            // tie the row's address to the previous result: otherwise hipcc issues all 120 LDS
            // reads up front and spills them (measured: 178 spilled VGPRs at 4 waves/SIMD)
            if (i > 0) asm volatile("" : "+v"(off) : "v"(x[i - 1]));
            const double *lrow = &Lt[off];
#pragma unroll
            for (int pp = 0; pp < i; ++pp) sum -= lrow[pp] * x[pp];
            x[i] = sum * rdg[TB * b + i];
        }
#pragma unroll
        for (int i = 0; i < TB; ++i)
            if (i >= c) Mt[tri(TB * b + i, TB * b + c)] = x[i];
    }
    __syncthreads();
    probe.mark_after(3, lane);
    {
        const int ra = tid >> 4, cb = tid & 15;
        for (int dist = 1; dist < NB / TB; ++dist)
            for (int ct = dist; ct < NB / TB; ++ct) {
                const int jt = ct - dist;
                double t = 0.0;
                for (int kt = jt; kt < ct; ++kt)
#pragma unroll
                    for (int pp = 0; pp < TB; ++pp) {
                        // M[16kt+pp][16jt+cb]: inside the diagonal block (kt == jt) zero above it
                        const double mv = Mt[tri(TB * kt + pp, TB * jt + cb)];
                        t += Lt[tri(TB * ct + ra, TB * kt + pp)] * ((kt > jt || pp >= cb) ? mv : 0.0);
                    }
                Ts[ra][cb] = t;
                __syncthreads();
                double mv = 0.0;
#pragma unroll
                for (int pp = 0; pp < TB; ++pp) {
                    const double md = Mt[tri(TB * ct + ra, TB * ct + pp)];
                    mv -= ((pp <= ra) ? md : 0.0) * Ts[pp][cb];
                }
                Mt[tri(TB * ct + ra, TB * jt + cb)] = mv;
                __syncthreads();
            }
    }
    probe.mark_after(4, lane);
    // strip order: strip (cb4, jt) is the A operand of v_mfma_f64_4x4x4 for output rows
    // 4 cb4 .. 4 cb4 + 3 against the 16 columns of tile jt: lane l holds
    // M[4 cb4 + (l & 3)][16 jt + 4 ((l >> 2) & 3) + (l >> 4)] — one coalesced 512-B load per strip
    {
        double *dv = p.dinv + (long)item * (NB * NB);
        for (int e = tid; e < NB * NB; e += 256) {
            const int strip = e >> 6, l = e & 63;
            const int cb4 = strip >> 2, jt = strip & 3;
            const int R = 4 * cb4 + (l & 3), C = 16 * jt + 4 * ((l >> 2) & 3) + (l >> 4);
            dv[e] = (R >= C) ? Mt[tri(R, C)] : 0.0;
        }
    }
    if (tid == 0) {
        double s = 0.0;
        for (int k = 0; k < NB; ++k) s += logs[k];
        p.logdet[item] += s;
        if (bad && p.info[item] == 0) p.info[item] = kmax + bad;
    }
    probe.mark_after(5, lane);
    probe.emit_diag(j, tid, item);
}

// ---------------------------------------------------------------------------------------
// chol_col: every row tile below the diagonal of block column j (and every aux tile).
//
//   C_rj -= L_r,[k0,k1) L_j,[k0,k1)'    then    L_rj = C_rj L_jj^-T
//
// One 64 x 64 tile per wave; the transposed tile S' = L_j L_r' is accumulated with the 4x4x4 MFMA
// form (A operand = rows of block j, B operand = the tile's rows) and consumed by the epilogue in
// that register layout.
//
// HBM traffic of a left-looking factorisation is one pass over all previous columns of every row
// tile per block column; at the 4x4x4 MFMA rate a 64-wide block column needs ~4.7 TB/s of that
// (measured: the kernel turned bandwidth-bound).  So block columns are processed in pairs:
//   FAT  step (j even):  chol_col_glds_kernel — column j is accumulated over k < 64 j and finished
//                        (solve + store) by one wave, while its sibling wave pre-accumulates
//                        column j+1 over the same k from the same LDS-staged rows and subtracts
//                        the partial sum from K in place;
//   THIN step (j odd):   chol_col_kernel — only k in [64 (j-1), 64 j) is left;
//   FULL step:           chol_col_kernel — single column over all k (last column of an odd count).
// Beside the fat launch, diag_ahead_kernel pre-accumulates the diagonal tile (j+2, j+2) over
// k < 64 j on a side stream, so chol_diag never runs a long k-loop on one workgroup.
// ---------------------------------------------------------------------------------------
// 8 k-values of NF 16-row fragments: lane (r16, q) holds rows 16u + r16, k = kc + 2q, 2q + 1
template <int NF>
struct Frag8 {
    f64x2 v[NF];
};
template <int NF>
__device__ __forceinline__ void load_frag8(Frag8<NF> &f, const double *p, long ld) {
#pragma unroll
    for (int u = 0; u < NF; ++u) f.v[u] = *reinterpret_cast<const f64x2 *>(p + (long)u * 16 * ld);
}
template <int NA>
__device__ __forceinline__ void mfma_frag8(double (&acc)[NA][4][4], const Frag8<NA> &a,
                                           const Frag8<4> &b) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const Rot4 br = rot4(b.v[it].x);
#pragma unroll
        for (int jt = 0; jt < NA; ++jt) mfma16_as_4(acc[jt][it], a.v[jt].x, br);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const Rot4 br = rot4(b.v[it].y);
#pragma unroll
        for (int jt = 0; jt < NA; ++jt) mfma16_as_4(acc[jt][it], a.v[jt].y, br);
    }
}

// acc[jt][it] += sum_{k in [k0,k1)} A[16jt + m][k] B[16it + n][k]; A rows are NA*16 consecutive
// rows at pa (lane offset applied by the caller), B rows the wave's 64 tile rows at pb.
// k1 - k0 is a multiple of 16: two 8-deep operand stages ping-pong.
template <int NA>
__device__ __forceinline__ void gemm_rows(double (&acc)[NA][4][4], const double *pa,
                                          const double *pb, long ld, int k0, int k1) {
    if (k1 <= k0) return;
    Frag8<NA> a0, a1;
    Frag8<4> b0, b1;
    load_frag8(a0, pa + k0, ld);
    load_frag8(b0, pb + k0, ld);
    for (int kc = k0; kc < k1; kc += 16) {
        load_frag8(a1, pa + kc + 8, ld);
        load_frag8(b1, pb + kc + 8, ld);
        mfma_frag8(acc, a0, b0);
        if (kc + 16 < k1) {
            load_frag8(a0, pa + kc + 16, ld);
            load_frag8(b0, pb + kc + 16, ld);
        }
        mfma_frag8(acc, a1, b1);
    }
}

// ---------------------------------------------------------------------------------------
// Epilogue on the 4x4x4 register layout (no accumulator layout conversion, no 16x16x4 MFMA).
//   acc4[jt][it][r] at lane l:  S'[jj = 16 jt + 4 ((l&15)>>2) + (l>>4)][i = 16 it + ((l&15) + 4r) & 15]
// L_rj = C_rj L_jj^-T  <=>  X' = M C'  with the full inverse M = L_jj^-1 from chol_diag.  An MFMA
// with A = strip (cb4, jt) of M and B = C'[jt][it][r] multiplies, in every lane group b, the 4x4
// block of M against the row-block b of C' — a partial sum of X'[4 cb4 ..][i] over the rows
// jj = 4b (mod 16).  The four partial sums of an element sit in four different (register, lane
// group) pairs; three whole-register DPP rotations line them up and they are added in a fixed
// order, written to a per-wave LDS tile and from there to HBM as full 512-byte rows.
// LDS: M strips (32 KiB, shared) + 16 tile rows x 64 columns per wave and pass, pitch 66 doubles.
// ---------------------------------------------------------------------------------------
constexpr int EPI_PITCH = 66;                       // doubles per LDS row (16-byte aligned rows)
constexpr int EPI_M_BYTES = NB * NB * 8;            // M strips, shared by the workgroup's tiles
constexpr int EPI_WAVE_BYTES = 16 * EPI_PITCH * 8;  // 16 tile rows x 64 columns per wave and pass
constexpr int EPI_LDS_BYTES = EPI_M_BYTES + 4 * EPI_WAVE_BYTES;   // 66,560 B

// all 256 threads: M (strip order, 32 KiB) global -> LDS; callers put a barrier after it
__device__ __forceinline__ void stage_mstrips(double *lds_m, const double *mstrips, int tid) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + 256 * i;   // f64x2 units
        reinterpret_cast<f64x2 *>(lds_m)[idx] = reinterpret_cast<const f64x2 *>(mstrips)[idx];
    }
}

// SHADOW (mixed-precision jobs): every finished row also goes out rounded to fp32 (Lr32, same
// indexing) and the largest magnitude of the tile to *tmax_out — the fat steps of later columns
// decide from those maxima which tile products may run on the fp32 matrix cores.
// probe slots 12 + 3 it .. 14 + 3 it: K' of pass `it` arrived, product + LDS done, stores issued
// SYNTH (gradient jobs): the identity block of the aux rows [I ; y'] is never written to memory
// by the fill; a tile of it that no step has touched yet is synthesised here instead of read —
// synth = 1: zeros, 2: the identity (the tile on the block diagonal), 0: read as usual.
// ksl (JobGeom::toep, structured items, a tile no step has touched yet): the tile was never
// written — K[i][jj] = ksl[63 + i - jj], the 127 table entries of the tile's lattice distances,
// staged in LDS by the caller (struct_slice); null: read the stored tile.
template <bool SHADOW = false, bool SYNTH = false, class Probe = NoProbe, bool KROWS = false>
__device__ __forceinline__ void solve_and_store_lds(const double (*acc4)[4][4], double *Lr,
                                                    const double *lds_m, long ld, int kmax,
                                                    int lane, double *buf, Probe &probe,
                                                    float *Lr32 = nullptr,
                                                    float *tmax_out = nullptr, int synth = 0,
                                                    const double *ksl = nullptr, int nit = 4) {
    double amax = 0.0;
    const int n16 = lane & 15, isub = lane >> 4;
    const int jj0 = 4 * (n16 >> 2) + isub;          // row of S' inside a 16-tile
    // 63 + i - jj = [15 + ((n16 + 4r) & 15) - jj0] + 16 (it - jt + 3): a per-lane base and a
    // compile-time offset
    const double *ksl0 = ksl ? ksl + 15 - jj0 : nullptr;
    f64x2 kvr[8];
    double kvp[4][4];
    const int hrow = lane >> 5, c2 = 2 * (lane & 31);
    // KROWS (the thin steps): a stored tile arrives as full 512-byte rows — eight loads of two rows
    // each, requested one pass ahead, where the register layout asks for sixteen loads of sixteen
    // 32-byte pieces — and is turned into the register layout through the wave's LDS tile at the
    // head of its pass (the tile is free then: the previous pass's rows have been read out of
    // it).  Thin steps 95.5 -> 91.7 ms (C3), 188.7 -> 179.8 (fitted gradient); in the fat steps
    // the LDS round trip costs what the row loads gain (+2 ms): they keep the register loads.
    // from_rows: wave-uniform, the tile is a stored one (not a table slice, not synthesised)
    bool from_rows = KROWS && ksl0 == nullptr;
    if constexpr (SYNTH) from_rows = from_rows && synth == 0;
    // the source is tested once per 16-row group, outside the element loops: tested per element it
    // puts every load into a basic block of its own and the sixteen loads of a group are no longer
    // issued together
    auto k_elems = [&](int it, double (&kv)[4][4]) {
        if (ksl0) {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    kv[jt][r] = ksl0[((n16 + 4 * r) & 15) + 16 * (it - jt + 3)];
            return;
        }
        if constexpr (SYNTH) {
            if (synth) {
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * it + ((n16 + 4 * r) & 15);
                        kv[jt][r] = (synth == 2 && i == 16 * jt + jj0) ? 1.0 : 0.0;
                    }
                return;
            }
        }
        if constexpr (!KROWS) {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * it + ((n16 + 4 * r) & 15);
                    kv[jt][r] = Lr[(long)i * ld + kmax + 16 * jt + jj0];
                }
        }
    };
    auto load_k = [&](int it) {   // one pass ahead
        if constexpr (KROWS) {
            if (!from_rows) return;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                kvr[i] = *reinterpret_cast<const f64x2 *>(Lr + (long)(16 * it + 2 * i + hrow) * ld + kmax + c2);
        } else {
            k_elems(it, kvp);
        }
    };
    auto take_k = [&](int it, double (&kv)[4][4]) {   // at the head of pass `it`
        if constexpr (KROWS) {
            if (!from_rows) {
                k_elems(it, kv);
                return;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<f64x2 *>(buf + (2 * i + hrow) * EPI_PITCH + c2) = kvr[i];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    kv[jt][r] = buf[((n16 + 4 * r) & 15) * EPI_PITCH + 16 * jt + jj0];
        } else {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) kv[jt][r] = kvp[jt][r];
        }
    };
    load_k(0);
#pragma unroll
    for (int it = 0; it < 4; ++it) {                // 16 tile rows per pass
        if (it >= nit) break;                       // an aux tile's zero rows: nothing to solve or store
        double c4[4][4];   // C' = K' - S' for this 16-row group of the tile
        take_k(it, c4);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) c4[jt][r] = c4[jt][r] - acc4[jt][it][r];
        probe.mark_after(12 + 3 * it, c4[3][3] + c4[0][0] + c4[1][2] + c4[2][1]);
        if (it + 1 < nit) load_k(it + 1);
        // 16 strip groups cb4 = 4 ct + cq (output rows 4 cb4 ..: C' tiles jt <= ct).  The M strips of
        // group cb4 + 1 are requested before the result of group cb4 is written to the LDS tile:
        // behind that write hipcc would not move them (same LDS array), and every group would
        // wait out an LDS round trip before its first MFMA (15 us of a tile's 22, fat_phases.py)
        double ms[2][4];
        ms[0][0] = lds_m[lane];
#pragma unroll
        for (int cb4 = 0; cb4 < 16; ++cb4) {
            const int ct = cb4 >> 2;
            if (cb4 + 1 < 16) {
#pragma unroll
                for (int jt = 0; jt <= ((cb4 + 1) >> 2); ++jt)
                    ms[(cb4 + 1) & 1][jt] = lds_m[((cb4 + 1) * 4 + jt) * 64 + lane];
            }
            double R[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int jt = 0; jt <= ct; ++jt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) R[r] = mfma4(ms[cb4 & 1][jt], c4[jt][r], R[r]);
            }
            // lane group b of R[r] holds the partial sum for tile rows 4 ((b + r) & 3) ..:
            // rotating R[r] by r lane groups lines all four partial sums up on the lanes
            // i = n16, where they are added (fixed order: deterministic)
            double x = R[0];
            x += dpp_f64<ROW_ROR4, 0xF>(R[1], R[1]);
            x += dpp_f64<ROW_ROR8, 0xF>(R[2], R[2]);
            x += dpp_f64<ROW_ROR12, 0xF>(R[3], R[3]);
            buf[n16 * EPI_PITCH + 4 * cb4 + isub] = x;   // X'[c = 4 cb4 + isub][i = 16 it + n16]
        }
        probe.mark_after(13 + 3 * it, it);
        // rows 16 it .. 16 it + 15 of the tile: 32 lanes x 16 B per 512-byte row
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const int idx = w * 64 + lane, row = idx >> 5, cp = idx & 31;
            const f64x2 v = *reinterpret_cast<const f64x2 *>(buf + row * EPI_PITCH + 2 * cp);
            *reinterpret_cast<f64x2 *>(Lr + (long)(16 * it + row) * ld + kmax + 2 * cp) = v;
            if constexpr (SHADOW) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x2 vf;
                vf.x = (float)v.x;
                vf.y = (float)v.y;
                *reinterpret_cast<f32x2 *>(Lr32 + (long)(16 * it + row) * ld + kmax + 2 * cp) = vf;
                amax = fmax(amax, fmax(fabs(v.x), fabs(v.y)));
            }
        }
        probe.mark_after(14 + 3 * it, it);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (SHADOW) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmax(amax, __shfl_xor(amax, off, 64));
        // rounded UP to fp32: the decision must never see a maximum smaller than the true one
        if (lane == 0) *tmax_out = __double2float_ru(amax);
    }
}

// tile[i][c0 + jj] -= S'[jj][i], straight from the 4x4x4 register layout.  The 16 elements a lane
// owns in a 16-row group are loaded together and the next group is requested before this one is
// written back: written as `*e -= x` per element, hipcc orders every load behind the previous
// store (64 dependent round trips per tile: 35 us of a fat workgroup's 270, scripts/fat_phases.py).
// fresh (gradient jobs): the tile is an untouched zero tile of the identity block — nothing is read
// ksl: the tile was never written (see solve_and_store_lds) — tile = ksl[63 + i - jj] - S'[jj][i]
__device__ __forceinline__ void subtract_in_place_perm(double *rows, long ld, int c0,
                                                       const double (*acc4)[4][4], int lane,
                                                       bool fresh = false,
                                                       const double *ksl = nullptr, int nit = 4) {
    const int n16 = lane & 15, isub = lane >> 4;
    const int jj0 = 4 * (n16 >> 2) + isub;
    double *base = rows + c0 + jj0;
    long roff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) roff[r] = (long)((n16 + 4 * r) & 15) * ld;
    if (ksl) {
        const double *ksl0 = ksl + 15 - jj0;
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    base[(long)(16 * it) * ld + roff[r] + 16 * jt] =
                        ksl0[((n16 + 4 * r) & 15) + 16 * (it - jt + 3)] - acc4[jt][it][r];
        return;
    }
    if (fresh) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    base[(long)(16 * it) * ld + roff[r] + 16 * jt] = -acc4[jt][it][r];
        return;
    }
    double v[2][4][4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[0][jt][r] = base[roff[r] + 16 * jt];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        if (it >= nit) break;       // an aux tile's zero rows stay as they are
        if (it + 1 < nit) {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    v[(it + 1) & 1][jt][r] = base[(long)(16 * (it + 1)) * ld + roff[r] + 16 * jt];
        }
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                base[(long)(16 * it) * ld + roff[r] + 16 * jt] = v[it & 1][jt][r] - acc4[jt][it][r];
    }
}

// The same update with the tile travelling as full 512-byte rows: the lane's sixteen values of a
// 16-row group go through the wave's LDS tile `tl` (16 rows x EPI_PITCH doubles: the layout the solve
// writes its rows from), every load and store instruction then covers two whole rows, where the
// register form above touches sixteen rows with 32 bytes each (64 + 64 such instructions per tile; the
// same change took 19 of 714 ms off the K^-1 kernel, profiles/r04/kinv_experiments.txt).  Operands
// and operation per element are the register form's: bit-identical.
__device__ __forceinline__ void subtract_in_place_rows(double *rows, long ld, int c0,
                                                       const double (*acc4)[4][4], int lane,
                                                       double *tl, bool fresh = false,
                                                       const double *ksl = nullptr, int nit = 4) {
    const int n16 = lane & 15, isub = lane >> 4;
    const int jj0 = 4 * (n16 >> 2) + isub;
    const int hrow = lane >> 5, c2 = 2 * (lane & 31);
    double *base = rows + c0 + c2;
    auto load_rows = [&](int it, f64x2 (&kv)[8]) {
        if (ksl) {   // tile[i][jj] = ksl[63 + i - jj]
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int d = 63 + 16 * it + 2 * i + hrow - c2;
                kv[i].x = ksl[d];
                kv[i].y = ksl[d - 1];
            }
        } else if (fresh) {
#pragma unroll
            for (int i = 0; i < 8; ++i) kv[i].x = kv[i].y = 0.0;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                kv[i] = *reinterpret_cast<const f64x2 *>(base + (long)(16 * it + 2 * i + hrow) * ld);
        }
    };
    f64x2 kv[2][8];
    load_rows(0, kv[0]);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        if (it >= nit) break;       // an aux tile's zero rows stay as they are
        if (it + 1 < nit) load_rows(it + 1, kv[(it + 1) & 1]);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                tl[((n16 + 4 * r) & 15) * EPI_PITCH + 16 * jt + jj0] = acc4[jt][it][r];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f64x2 a = *reinterpret_cast<const f64x2 *>(tl + (2 * i + hrow) * EPI_PITCH + c2);
            f64x2 o;
            o.x = kv[it & 1][i].x - a.x;
            o.y = kv[it & 1][i].y - a.y;
            *reinterpret_cast<f64x2 *>(base + (long)(16 * it + 2 * i + hrow) * ld) = o;
        }
    }
}

// JobGeom::toep: the 127 table entries a structured item's tile (rows row0.., columns col0.. of the
// main block, row0 >= col0 + 64) is regenerated from, into the wave's LDS slice:
// ksl[x] = tab[toep (row0 - col0 - 63 + x)]
__device__ __forceinline__ void struct_slice(double *ksl, const JobGeom &g, const ChunkPtrs &p,
                                             int item, int row0, int col0, int lane) {
    const double *tab = p.tab + (long)item * g.maxstat * g.R;
    const int d0 = row0 - col0 - 63;
    ksl[lane] = tab[(long)g.toep * (d0 + lane)];
    if (lane < 63) ksl[64 + lane] = tab[(long)g.toep * (d0 + 64 + lane)];
}
constexpr int TOEP_LDS_BYTES = 4 * 128 * 8;   // one slice per wave

struct ColStep {
    int j;        // block column being finished
    int k0;       // first k not yet accumulated into column j
    int nmain;    // main row tiles below the diagonal (r = j+1 ...); 0: aux tiles only
    int ntiles;   // nmain + aux tiles
    int groups;   // workgroups per item
    int splits;   // fat steps of small batches: workgroups that share one tile pair's k-range (1: none)
    int first_touch;   // FAT / FULL steps: no earlier step has touched the main tiles of this column
                       // (JobGeom::toep: the tiles of single-table items were never written)
    // mixed-precision jobs: a tile product runs in fp32 iff max|A| max|B| <= c32 (noise + jitter),
    // c32 = mixed_tau / (64 * 2^-24)
    double c32, jitter;
};
// index of a finished tile's maximum in ChunkPtrs::tmax (per item): row tiles 0..nb0-1 are the
// main block rows, nb0.. the aux tiles
__device__ __forceinline__ long tmax_index(const JobGeom &g, int row_tile, int col) {
    return (long)row_tile * g.nb0 + col;
}

// THIN and FULL steps: direct operand loads (short k-loops), one row tile per wave.
// IDENT (gradient jobs, aux rows [I ; y']): identity tile a joins from block column a on, its
// k-loop starts at 64 a, and a tile of the identity block that no step has written yet is
// synthesised, not read (the fill leaves the identity block out).
// TOEP (JobGeom::toep, the FULL step of column 0 of an odd block-column count): the tiles of
// single-table items come from the table slice; an instantiation of its own, so that the ordinary
// one keeps its registers (with the slice path compiled in it spilled 136 B per lane).
template <bool MIXED, bool IDENT = false, bool TOEP = false>
__global__ __launch_bounds__(256, 2) void chol_col_kernel(JobGeom g, ChunkPtrs p, int Bc,
                                                          ColStep st) {
    __shared__ __attribute__((aligned(16))) char epi[EPI_LDS_BYTES + (TOEP ? TOEP_LDS_BYTES : 0)];
    const int wg = blockIdx.x;
    const int xcd = wg & 7, idx = wg >> 3;   // blocks b and b+8 share an XCD (speed only)
    const int slot = (idx / st.groups) * 8 + xcd;
    const int grp = idx % st.groups;
    if (slot >= Bc) return;                  // whole workgroup, before any barrier
    const int item = p.items ? p.items[slot] : slot;   // refinement sweeps: compacted item list
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = grp * 4 + wave;             // row tile
    bool valid = tile < st.ntiles;
    // gradient jobs: aux rows are [I ; y'], so W = X L^-T is block upper triangular — identity
    // tile a is zero left of block column a (skip it while a > j) and its k-loop starts at 64 a
    int kbeg = st.k0;
    int synth = 0;
    if constexpr (IDENT) {
        if (valid && tile >= st.nmain) {
            const int a = tile - st.nmain;
            if (a < g.nb0) {
                if (a > st.j) valid = false;
                else if (a * NB > kbeg) kbeg = a * NB;
                synth = (a == st.j) ? 2 : 1;   // FULL step: the tile has not been touched
            }
        }
    }

    const int j = st.j;
    const long ld = g.ld;
    double *Lit = p.L + (long)item * g.item_stride;
    const int kmax = j * NB;
    const int r16 = lane & 15, q = lane >> 4;

    double acc4[4][4][4];  // [jt][it][r]: block-diagonal r of S' tile (jt, it), see mfma16_as_4
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc4[a][b][r] = 0.0;

    const int vt = valid ? tile : 0;
    const long rowbase = (vt < st.nmain) ? (long)(j + 1 + vt) * NB
                                         : (long)g.n0 + (long)(vt - st.nmain) * NB;
    double *Lr = Lit + rowbase * ld;
    const double *Lj = Lit + (long)j * NB * ld;            // A operand: rows of block j
    const double *pa = Lj + (long)r16 * ld + 2 * q;
    const double *pb = Lr + (long)r16 * ld + 2 * q;        // B operand: rows of this tile
    if (valid) gemm_rows<4>(acc4, pa, pb, ld, kbeg, kmax);
    // JobGeom::toep: the main tiles of a single-table item were never written (FULL step of
    // column 0: nothing has touched them) — their values come from the wave's table slice
    const double *ksl = nullptr;
    if constexpr (TOEP) {
        if (valid && tile < st.nmain && prog_structure(p.progs + item)) {
            double *sl = reinterpret_cast<double *>(epi + EPI_LDS_BYTES) + wave * 128;
            struct_slice(sl, g, p, item, (int)rowbase, kmax, lane);
            ksl = sl;
        }
    }
    stage_mstrips(reinterpret_cast<double *>(epi), p.dinv + (long)item * (NB * NB), tid);
    __syncthreads();
    if (!valid) return;
    NoProbe probe;
    if constexpr (MIXED) {
        const int rt = (tile < st.nmain) ? j + 1 + tile : g.nb0 + (tile - st.nmain);
        solve_and_store_lds<true>(
            acc4, Lr, reinterpret_cast<const double *>(epi), ld, kmax, lane,
            reinterpret_cast<double *>(epi + EPI_M_BYTES + wave * EPI_WAVE_BYTES), probe,
            p.L32 + (long)item * g.item_stride + rowbase * ld,
            p.tmax + (long)item * (g.nb0 + g.naux_pad / NB) * g.nb0 + tmax_index(g, rt, j));
    } else {
        solve_and_store_lds<false, IDENT>(
            acc4, Lr, reinterpret_cast<const double *>(epi), ld, kmax, lane,
            reinterpret_cast<double *>(epi + EPI_M_BYTES + wave * EPI_WAVE_BYTES), probe, nullptr,
            nullptr, synth, ksl);
    }
}

// ---------------------------------------------------------------------------------------
// THIN step (column j of a pair: only k in [64 (j-1), 64 j) is left).  One row tile per wave, as in
// chol_col_kernel, but nothing of the operand traffic waits on a register: the 64 x 64 block of
// panel rows all four waves multiply against and the M strips of the epilogue go HBM -> LDS by
// LDS-DMA at kernel entry, and the wave's own 64 x 64 operand block is requested four 8-deep
// stages ahead (the direct-load kernel, at 256 VGPRs, could keep two in flight and spent most of
// its k-loop waiting: MFMA pipe 42 % busy, profiles/r02/clock_C3.txt).  Same MFMA order as
// gemm_rows: results are bit-identical to the direct-load kernel.
// LDS: M strips 32 KiB | panel block 32 KiB (64 rows x 512 B; 16-byte piece p of row r sits in
// slot p ^ (r & 7), swizzled on the source address), reused for the per-wave epilogue tiles.
// ---------------------------------------------------------------------------------------
template <bool MIXED, class Probe = NoProbe, bool IDENT = false>
__global__ __launch_bounds__(256, 2) void chol_col_thin_kernel(JobGeom g, ChunkPtrs p, int Bc,
                                                               ColStep st) {
    constexpr int A_BYTES = NB * NB * 8;
    constexpr int TAIL = (4 * EPI_WAVE_BYTES > A_BYTES) ? 4 * EPI_WAVE_BYTES : A_BYTES;
    __shared__ __attribute__((aligned(1024))) char smem[EPI_M_BYTES + TAIL];
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const int wg = blockIdx.x;
    const int xcd = wg & 7, idx = wg >> 3;   // blocks b and b+8 share an XCD (speed only)
    const int slot = (idx / st.groups) * 8 + xcd;
    const int grp = idx % st.groups;
    if (slot >= Bc) return;                  // whole workgroup, before any barrier
    const int item = p.items ? p.items[slot] : slot;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = grp * 4 + wave;
    bool valid = tile < st.ntiles;
    const int j = st.j;
    const int kmax = j * NB, k0 = kmax - NB;
    // gradient jobs: identity tile a is zero left of block column a — tile a == j has nothing to
    // subtract (its k-range would start at 64 j), tiles a > j are still zero
    bool update = true;
    int synth = 0;     // the tile on the block diagonal (a == j) has not been touched: identity
    if constexpr (IDENT) {
        if (valid && tile >= st.nmain) {
            const int a = tile - st.nmain;
            if (a < g.nb0) {
                if (a > j) valid = false;
                else if (a * NB > k0) update = false;
                if (a == j) synth = 2;
            }
        }
    }
    const long ld = g.ld;
    double *Lit = p.L + (long)item * g.item_stride;
    const int r16 = lane & 15, q = lane >> 4;
    Probe probe;
    probe.mark_after(0, lane);

    // ---- LDS-DMA: panel block (rows 64 j .. 64 j + 63, columns k0 .. k0 + 63) and M strips ----
    {
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(
            Lit, 0, (int)(g.item_stride * (long)sizeof(double)), 0x00020000);
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
            p.dinv + (long)item * (NB * NB), 0, NB * NB * 8, 0x00020000);
        // instruction i of wave w: rows 2 (8 w + i), + 1; lane = (row bit, slot); piece = slot ^ key
        const int rbit = lane >> 5, sl = lane & 31;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int key = (2 * i + rbit) & 7;
            const unsigned voff = (unsigned)((rbit * ld + 2 * (sl ^ key)) * 8);
            const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(
                (int)((((long)kmax + 2 * (8 * wave + i)) * ld + k0) * 8));
            lds_ptr dst = (lds_ptr)(smem + EPI_M_BYTES + (8 * wave + i) * 1024);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, dst, 16, voff, soff, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            lds_ptr dst = (lds_ptr)(smem + (8 * wave + i) * 1024);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rm, dst, 16, (unsigned)(lane * 16),
                                                     (unsigned)((8 * wave + i) * 1024), 0, 0);
        }
    }

    double acc4[4][4][4];  // [jt][it][r], see mfma16_as_4
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc4[a][b][r] = 0.0;

    const int vt = valid ? tile : 0;
    const long rowbase = (vt < st.nmain) ? (long)(j + 1 + vt) * NB
                                         : (long)g.n0 + (long)(vt - st.nmain) * NB;
    double *Lr = Lit + rowbase * ld;
    const double *pb = Lr + (long)r16 * ld + k0 + 2 * q;   // B operand: rows of this tile
    const bool mult = valid && update;
    Frag8<4> b[4];
    if (mult) {
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) load_frag8(b[sg], pb + 8 * sg, ld);
    }
    __syncthreads();   // the LDS-DMA of every wave has landed (hipcc drains vmcnt ahead of it)
    probe.mark_after(1, lane);
    if (mult) {
        const char *Ab = smem + EPI_M_BYTES + r16 * 512;
        const int key = r16 & 7;
#pragma unroll
        for (int sg = 0; sg < 8; ++sg) {
            Frag8<4> a;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                a.v[u] = *reinterpret_cast<const f64x2 *>(Ab + u * (16 * 512) +
                                                          (((4 * sg + q) ^ key) << 4));
            mfma_frag8(acc4, a, b[sg & 3]);
            if (sg + 4 < 8) load_frag8(b[sg & 3], pb + 8 * (sg + 4), ld);
        }
    }
    probe.mark_after(2, acc4[0][0][0] + acc4[3][3][3]);
    __syncthreads();   // the panel block is dead: its LDS becomes the per-wave epilogue tiles
    probe.mark_after(3, lane);
    probe.mark_after(4, lane);
    if (!valid) return;
    double *buf = reinterpret_cast<double *>(smem + EPI_M_BYTES + wave * EPI_WAVE_BYTES);
    if constexpr (MIXED) {
        const int rt = (tile < st.nmain) ? j + 1 + tile : g.nb0 + (tile - st.nmain);
        solve_and_store_lds<true, false, Probe, true>(
            acc4, Lr, reinterpret_cast<const double *>(smem), ld, kmax, lane, buf, probe,
            p.L32 + (long)item * g.item_stride + rowbase * ld,
            p.tmax + (long)item * (g.nb0 + g.naux_pad / NB) * g.nb0 + tmax_index(g, rt, j));
    } else {
        // an aux tile with at most 16 real rows: its other rows are zero and stay zero (fat kernel)
        int nit = (tile >= st.nmain && g.naux - NB * (tile - st.nmain) <= 16) ? 1 : 4;
        if (IDENT && tile < st.nmain && g.n_real - NB * (j + 1 + tile) <= 16) nit = 1;   // padded last row tile
        solve_and_store_lds<false, IDENT, Probe, true>(acc4, Lr, reinterpret_cast<const double *>(smem), ld,
                                          kmax, lane, buf, probe, nullptr, nullptr, synth, nullptr,
                                          nit);
    }
    probe.mark_after(5, lane);
    probe.drain();
    probe.mark_after(6, lane);
    probe.emit_col(j, lane, wg, wave, item, tile, /*thin=*/true);
}

constexpr int LDS_KC = 16;   // k-depth of one staged chunk

// ---------------------------------------------------------------------------------------
// chol_col FAT step (the production path; carries all the long k-loops).  Same math and the
// same FAT / THIN / FULL schedule as the direct-load kernel above, different data movement:
//   * the workgroup stages each 16-deep k-chunk of its operand rows ONCE into LDS: A panel = rows
//     of blocks j and j+1 (128), B = 2 row tiles (128); wave = (tile, column);
//   * two LDS buffers, one barrier per chunk: chunk c+1 is in flight while chunk c is multiplied;
//   * the three rotated copies of every B fragment that the 4x4x4 MFMA form needs come from LDS by
//     address (row (n' + 4r) mod 16) instead of DPP moves: the k-loop is ds_read_b64 + MFMA only;
//   * the operand rows go HBM -> LDS directly (buffer_load_dwordx4 ... lds: no VGPR round trip, no
//     ds_write).  A register-staged predecessor (padded 136-B rows, ds_write2_b64) measured 11 %
//     slower on the same box.  (A 64 x 128 tile in ONE wave would give register-level reuse, but
//     needs 256 accumulators; hipcc then splits them across the AGPR/VGPR halves and moves them
//     every iteration — measured 2x slower — hence two sibling waves x 64 x 64.)  An LDS-DMA instruction writes 1 KiB contiguously (8 rows x 128 B here), so rows cannot
// be padded; bank conflicts are removed by an XOR swizzle applied on the SOURCE address
// (slot p of row r holds the 16-byte piece p ^ ((r>>1)&7)) and again on the read address
// (guide rule: linear destination + swizzled source + the same swizzle on the read).  The
// swizzled read address is not affine in the k-step, so each (fragment, k-step) has its own
// address register — which also keeps hipcc from fusing the reads into ds_read2_b64 (banked
// mod 32, inherently 2-way conflicting on 16-byte-granular layouts).
// ---------------------------------------------------------------------------------------
//
// MIXED (NGP_PREC_MIXED jobs, BASELINE config C5).  The k-range of a step is cut into 64-wide
// k-tiles; a k-tile runs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, operands from the
// fp32 shadow L32, twice the fp64 rate and half the bytes) iff for all four operand tiles of the
// workgroup   max|A| max|B| <= c32 (noise + jitter)   — then the rounding error of that product,
// 64 * 2^-24 * max|A| max|B|, stays below mixed_tau of the smallest pivot the matrix can have
// (every pivot of K + (noise + jitter) I is >= noise + jitter).  All other k-tiles, the
// accumulators, the K tile, the solve and the stored factor are fp64.  The fp32 k-tiles go first
// into 64 fp32 accumulators, which are handed to the fp64 accumulators through LDS (one
// transposition of the 32x32 C/D layout into the 4x4x4 composite layout), then the fp64 k-tiles
// follow on top: one accumulator set is live at a time.  Same LDS image for both passes: a staged
// chunk is 128 B per row — 16 doubles or 32 floats.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));


//
// SPLITK (small batches, late block columns): with a few dozen items a fat step of a late column is
// a handful of workgroups per item, each with a k-loop of a thousand columns and more — a fraction
// of the CUs busy for hundreds of microseconds (64 items at n = 2048, j = 30: 64 workgroups, 290 us
// for 73 us of arithmetic).  st.splits workgroups then share one tile pair: each accumulates a
// contiguous piece of the k-range and leaves its four accumulator tiles in p.splitk_part
// (SPLITK = 1); a second launch with one workgroup per tile pair (SPLITK = 2) adds the pieces in
// piece order and runs the epilogue.  (One launch with the last workgroup to arrive doing the sum
// was measured first: the device-scope release it needs writes the L2 back per workgroup and cost
// more than the second launch.)
template <bool MIXED, class Probe = NoProbe, bool IDENT = false, int SPLITK = 0>
__global__ __launch_bounds__(256, 2) void chol_col_glds_kernel(JobGeom g, ChunkPtrs p, int Bc,
                                                               ColStep st) {
    // 32 LDS-DMA blocks (8 rows x 128 B) per buffer, each followed by a 128-B gap: row groups of
    // a fragment are then 2304 B apart, which hipcc cannot fuse into ds_read2st64_b64 (offsets
    // must be multiples of 512 B; fused reads bank mod 32 and measured 50 % conflict cycles),
    // and (bit3 ^ bit0, bits 3..1) of the row index still select distinct bank groups.
    constexpr int ROWB = 128, BLKB = 8 * ROWB + 128, STAGE = 32 * BLKB;   // 36 KiB per buffer
    auto row_off = [](int row) { return (row >> 3) * BLKB + (row & 7) * ROWB; };
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
    typedef __attribute__((address_space(3))) void *lds_ptr;

    const int wg = blockIdx.x;
    const int xcd = wg & 7, idx = wg >> 3;       // blocks b and b+8 share an XCD (speed only)
    const int per_item = SPLITK == 1 ? st.groups * st.splits : st.groups;
    const int slot = (idx / per_item) * 8 + xcd;
    const int grp = SPLITK == 1 ? (idx % per_item) / st.splits : idx % st.groups;
    const int piece = SPLITK == 1 ? (idx % per_item) % st.splits : 0;
    if (slot >= Bc) return;                      // whole workgroup, before any barrier
    // mixed-precision batches: items with the most fp64 tile products are dispatched first
    // (mixed_order_kernel), so that a launch does not end on its slowest workgroups
    const int item = (MIXED && p.order) ? p.order[slot] : slot;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ltile = wave >> 1, col = wave & 1;
    const int tile0 = grp * 2;
    const int tile = tile0 + ltile;
    bool valid = tile < st.ntiles;

    const int j = st.j;
    Probe probe;
    probe.mark(0);
    const long ld = g.ld;
    double *Lit = p.L + (long)item * g.item_stride;
    const int kmax = j * NB;
    // JobGeom::toep: the main tiles of a single-table item below the block diagonal were never
    // written; a fat step is the first to touch its two columns' tiles (the sibling wave of the
    // first row tile works on the diagonal tile (j+1, j+1), which IS stored: it carries the noise)
    bool structured = false;
    if constexpr (!MIXED && !IDENT) {
        if (g.toep && st.first_touch && valid && tile < st.nmain && !(col && tile == 0))
            structured = prog_structure(p.progs + item) != 0;
    }
    // gradient jobs (aux rows [I ; y']): identity tile a is zero left of block column a.  The two
    // tiles of a workgroup share the staged k-range, so it starts at the smaller of their starts;
    // a workgroup whose tiles are all still zero leaves before the first barrier.
    int kbeg = st.k0;
    int synth = 0;     // identity-block tiles are untouched when a fat step reaches them
    if constexpr (IDENT) {
        const int a = tile - st.nmain;
        if (tile < st.ntiles && a >= 0 && a < g.nb0) synth = (a == j) ? 2 : 1;
    }
    if (IDENT && tile0 >= st.nmain) {
        // workgroup-uniform: first k of each of the two tiles (kmax + 1: nothing to do)
        int kfirst = kmax + 1;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int a = tile0 + u - st.nmain;
            if (tile0 + u >= st.ntiles) continue;
            const int ks = (a >= g.nb0) ? st.k0 : (a > j) ? kmax + 1 : max(st.k0, a * NB);
            kfirst = min(kfirst, ks);
        }
        if (kfirst > kmax) return;
        kbeg = kfirst;
        const int a = tile - st.nmain;
        if (valid && a < g.nb0 && a > j) valid = false;
    }
    // 16-row groups of the wave's tile that hold real rows: 4 for main tiles and full aux tiles; an
    // aux tile with at most 16 real rows (see the k-loop) works on its first group only; a wave
    // without a tile (an odd tile count, an identity tile of a gradient job that is still zero)
    // stages its rows and keeps the barriers but multiplies nothing
    // (only gradient jobs have such waves in numbers — the identity tiles that are still zero; in
    // the value kernel the third loop cost 0.5 % and saved nothing)
    int nit = 4;
    if constexpr (!MIXED && SPLITK == 0) {
        if (IDENT && !valid) nit = 0;
        else if (valid && tile >= st.nmain && g.naux - NB * (tile - st.nmain) <= 16) nit = 1;
        // gradient jobs pad the main block to a multiple of 64 with identity rows: the last row tile
        // of a series of 2048 + 1 points has one real row, the other 63 are zero left of the diagonal
        else if (IDENT && valid && tile < st.nmain && g.n_real - NB * (j + 1 + tile) <= 16) nit = 1;
    }
    const int r16 = lane & 15, q = lane >> 4;
    auto tile_row0 = [&](int t) -> long {
        if (t >= st.ntiles) t = st.ntiles - 1;
        return (t < st.nmain) ? (long)(j + 1 + t) * NB : (long)g.n0 + (long)(t - st.nmain) * NB;
    };
    // ---- staging: wave w fills stage rows [64w, 64w+64): waves 0,1 the A panel (blocks j, j+1),
    //      waves 2,3 the two row tiles; instruction i covers rows 8i..8i+7 (lane>>3) x 8 pieces
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        Lit, 0, (int)(g.item_stride * (long)sizeof(double)), 0x00020000);
    long src_row0;
    if (wave < 2) src_row0 = (long)j * NB + 64 * wave;
    else src_row0 = tile_row0(tile0 + (wave - 2));
    const unsigned soff_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(src_row0 * ld * 8));
    const unsigned row_step8 = (unsigned)(8 * ld * 8);
    // piece fetched by this lane = (lane&7) ^ key(row), key = (row>>1)&7 = (4i + (lane>>4)) & 7
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
    // ---- operand read addresses: row * 128 + ((piece ^ key) * 16) + (k&1) * 8, piece = 2s + (q>>1)
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

    // one 16-deep fp64 chunk in buffer `buf`: 256 mfma4 per wave, for a tile that works on fewer
    // than its four 16-row groups; full tiles take mult_chunk_pipelined (ngp_mfma.h).  (Requesting ALL
    // operands of k-step s+1 before the MFMAs of k-step s — two register sets — measured 1 % slower.)
    auto mult64 = [&](const char *buf, auto nit_c) {
        constexpr int NIT = decltype(nit_c)::value;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            double a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                a[u] = *reinterpret_cast<const double *>(buf + a_addr[s] + u * 2 * BLKB);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                Rot4 br;
                br.r0 = *reinterpret_cast<const double *>(buf + b_addr[0][s] + it * 2 * BLKB);
                br.r1 = *reinterpret_cast<const double *>(buf + b_addr[1][s] + it * 2 * BLKB);
                br.r2 = *reinterpret_cast<const double *>(buf + b_addr[2][s] + it * 2 * BLKB);
                br.r3 = *reinterpret_cast<const double *>(buf + b_addr[3][s] + it * 2 * BLKB);
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) mfma16_as_4(acc4[jt][it], a[jt], br);
            }
        }
    };

    if constexpr (SPLITK == 2) {
        // finishing launch of a split-k step: no k-loop, the pieces are added below
    } else if constexpr (!MIXED) {
        int nchunks = (kmax - kbeg) / LDS_KC;
        if constexpr (SPLITK == 1) {   // this workgroup's piece of the k-range, whole chunks
            const int per = (nchunks + st.splits - 1) / st.splits;
            const int c0 = min(piece * per, nchunks);
            nchunks = min(c0 + per, nchunks) - c0;
            kbeg += c0 * LDS_KC;
        }
        if (nchunks > 0) {
            stage(0, kbeg);
            __syncthreads();   // hipcc drains the LDS-DMA (vmcnt(0)) ahead of the barrier
            // an aux tile whose real rows fit its first 16-row group (the 11 rows of a predictive
            // job: appended points, forecast dates, y'; the single y' row of a gradient job)
            // multiplies that group only — the other 48 rows are zero and stay zero.  Two loops,
            // not one loop with two bodies: that form spilled 400 B per lane.
            if (IDENT && nit == 0) {
                for (int c = 0; c < nchunks; ++c) {
                    if (c + 1 < nchunks) stage((c & 1) ^ 1, kbeg + (c + 1) * LDS_KC);
                    __syncthreads();
                }
            } else if (nit == 1) {
                for (int c = 0; c < nchunks; ++c) {
                    const int cur = c & 1;
                    if (c + 1 < nchunks) stage(cur ^ 1, kbeg + (c + 1) * LDS_KC);
                    mult64(smem + cur * STAGE, std::integral_constant<int, 1>{});
                    __syncthreads();
                }
            } else {
                for (int c = 0; c < nchunks; ++c) {
                    const int cur = c & 1;
                    if (c + 1 < nchunks) stage(cur ^ 1, kbeg + (c + 1) * LDS_KC);
                    mult_chunk_pipelined<2 * BLKB>(acc4, smem + cur * STAGE, a_addr, b_addr);
                    __syncthreads();
                    if (c == 0) probe.mark(1);
                    if (c == nchunks / 2) probe.mark(2);
                }
            }
        }
    } else {
        // ---- which k-tiles may run in fp32 (workgroup-uniform; every wave evaluates it) ----
        const int nkt = j;                          // k-tiles 0 .. j-1 (<= 128)
        const int nbt = g.nb0 + g.naux_pad / NB;
        const float *tm = p.tmax + (long)item * nbt * g.nb0;
        const int t1 = min(tile0 + 1, st.ntiles - 1);
        const int rt0 = (tile0 < st.nmain) ? j + 1 + tile0 : g.nb0 + (tile0 - st.nmain);
        const int rt1 = (t1 < st.nmain) ? j + 1 + t1 : g.nb0 + (t1 - st.nmain);
        const double lim = st.c32 * (p.progs[item].noise + st.jitter);
        unsigned long long m32lo, m32hi;
        {
            bool c[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kt = lane + 64 * h;
                c[h] = false;
                if (kt < nkt) {
                    const float ta = fmaxf(tm[tmax_index(g, j, kt)], tm[tmax_index(g, j + 1, kt)]);
                    const float tb = fmaxf(tm[tmax_index(g, rt0, kt)], tm[tmax_index(g, rt1, kt)]);
                    c[h] = (double)ta * (double)tb <= lim;
                }
            }
            m32lo = __ballot(c[0]);
            m32hi = __ballot(c[1]);
        }
        const unsigned long long inlo = nkt >= 64 ? ~0ull : ((1ull << nkt) - 1ull);
        const unsigned long long inhi = nkt <= 64 ? 0ull : (nkt >= 128 ? ~0ull : ((1ull << (nkt - 64)) - 1ull));
        unsigned long long m64lo = inlo & ~m32lo, m64hi = inhi & ~m32hi;
        const int n32 = __popcll(m32lo) + __popcll(m32hi);
        const int n64 = __popcll(m64lo) + __popcll(m64hi);
        if (tid == 0 && p.mixcnt) {
            const unsigned nv = (unsigned)(2 * min(2, st.ntiles - tile0));   // wave tiles that count
            atomicAdd(p.mixcnt + 2 * item, nv * (unsigned)n32);
            atomicAdd(p.mixcnt + 2 * item + 1, nv * (unsigned)n64);
        }
        auto pop = [](unsigned long long &lo, unsigned long long &hi) -> int {
            if (lo) {
                const int b = __builtin_ctzll(lo);
                lo &= lo - 1;
                return b;
            }
            const int b = __builtin_ctzll(hi);
            hi &= hi - 1;
            return 64 + b;
        };

        if (n32 > 0) {
            // ---- fp32 pass: 32-deep chunks of the shadow rows, v_mfma_f32_32x32x2_f32 ----
            const __amdgpu_buffer_rsrc_t rsrc32 = __builtin_amdgcn_make_buffer_rsrc(
                p.L32 + (long)item * g.item_stride, 0, (int)(g.item_stride * (long)sizeof(float)),
                0x00020000);
            const unsigned soff32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(src_row0 * ld * 4));
            const unsigned row_step32 = (unsigned)(8 * ld * 4);
            const unsigned v32_even = (unsigned)((lane >> 3) * ld * 4 + 16 * ((lane & 7) ^ ((lane >> 4) & 7)));
            const unsigned v32_odd = (unsigned)((lane >> 3) * ld * 4 + 16 * ((lane & 7) ^ ((4 + (lane >> 4)) & 7)));
            auto stage32 = [&](int buf, int k) {
                const unsigned kb = (unsigned)k * 4u;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    lds_ptr dst = (lds_ptr)(smem + buf * STAGE + (8 * wave + i) * BLKB);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc32, dst, 16,
                                                             (i & 1) ? v32_odd : v32_even,
                                                             soff32 + i * row_step32 + kb, 0, 0);
                }
            };
            // lane (r32 = lane & 31, h = lane >> 5) holds rows r32 of its two A and two B blocks; of
            // the 32 k-values of a chunk it reads pieces 2u + h (k = 8u + 4h .. + 3), u = 0..3, and
            // feeds element e of piece u to MFMA step (u, e): both operands use the same k-map
            const int r32 = lane & 31, h = lane >> 5;
            unsigned a32[2][4], b32[2][4];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int arow = 64 * col + 32 * blk + r32;
                const int brow = 128 + 64 * ltile + 32 * blk + r32;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a32[blk][u] = (unsigned)(row_off(arow) + (((2 * u + h) ^ ((arow >> 1) & 7)) << 4));
                    b32[blk][u] = (unsigned)(row_off(brow) + (((2 * u + h) ^ ((brow >> 1) & 7)) << 4));
                }
            }
            // two-level accumulation: the fp32 accumulators carry ONE k-tile (32 MFMA steps) and are
            // then added into fp64 partial sums of the same 32x32 layout.  Left in fp32 across all
            // k-tiles, the accumulator itself grows to the size of K (its rounding error, 2^-24 of
            // THAT, is what broke pivots at n >= 4096 — profiles/r02/README.md).
            f32x16 acc32[2][2];
            double acc64p[2][2][16];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        acc32[a][b][v] = 0.f;
                        acc64p[a][b][v] = 0.0;
                    }
            unsigned long long lo = m32lo, hi = m32hi;
            const int nch = 2 * n32;
            int kt_cur = pop(lo, hi), sub = 1;
            stage32(0, kt_cur * NB);
            __syncthreads();
            for (int c = 0; c < nch; ++c) {
                const int cur = c & 1;
                if (c + 1 < nch) {
                    if (sub == 2) {
                        kt_cur = pop(lo, hi);
                        sub = 0;
                    }
                    stage32(cur ^ 1, kt_cur * NB + 32 * sub);
                    ++sub;
                }
                const char *buf = smem + cur * STAGE;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    f32x4 av[2], bv[2];
#pragma unroll
                    for (int blk = 0; blk < 2; ++blk) {
                        av[blk] = *reinterpret_cast<const f32x4 *>(buf + a32[blk][u]);
                        bv[blk] = *reinterpret_cast<const f32x4 *>(buf + b32[blk][u]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int b = 0; b < 2; ++b)
                                acc32[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                    av[a][e], bv[b][e], acc32[a][b], 0, 0, 0);
                }
                if (c & 1) {   // a k-tile is two chunks: its sum moves to the fp64 level
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
#pragma unroll
                            for (int v = 0; v < 16; ++v) {
                                acc64p[a][b][v] += (double)acc32[a][b][v];
                                acc32[a][b][v] = 0.f;
                            }
                }
                __syncthreads();
            }
            // ---- hand-over: D[m = 32a + (v&3) + 8(v>>2) + 4h][n = 32b + r32] of the 32x32 layout
            //      -> LDS tile [jj = m][i = n] (per wave, 32 rows at a time, pitch 65 doubles)
            //      -> acc4[jt][it][r]
            constexpr int P64 = 65;
            double *t64 = reinterpret_cast<double *>(smem) + wave * (32 * P64);
            static_assert(4 * 32 * P64 * 8 <= 2 * STAGE, "hand-over tiles must fit the stage buffers");
            const int jj0 = 4 * (r16 >> 2) + q;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        t64[((v & 3) + 8 * (v >> 2) + 4 * h) * P64 + 32 * b + r32] = acc64p[a][b][v];
                __syncthreads();
#pragma unroll
                for (int jl = 0; jl < 2; ++jl)
#pragma unroll
                    for (int it = 0; it < 4; ++it)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc4[2 * a + jl][it][r] =
                                t64[(16 * jl + jj0) * P64 + 16 * it + ((r16 + 4 * r) & 15)];
                __syncthreads();   // before the next half / the fp64 pass overwrites the tiles
            }
        }
        if (n64 > 0) {
            unsigned long long lo = m64lo, hi = m64hi;
            const int nch = 4 * n64;
            int kt_cur = pop(lo, hi), sub = 1;
            stage(0, kt_cur * NB);
            __syncthreads();
            for (int c = 0; c < nch; ++c) {
                const int cur = c & 1;
                if (c + 1 < nch) {
                    if (sub == 4) {
                        kt_cur = pop(lo, hi);
                        sub = 0;
                    }
                    stage(cur ^ 1, kt_cur * NB + LDS_KC * sub);
                    ++sub;
                }
                mult_chunk_pipelined<2 * BLKB>(acc4, smem + cur * STAGE, a_addr, b_addr);
                __syncthreads();
            }
        }
    }
    if constexpr (SPLITK == 1) {
        // ---- hand in this piece: register e of lane l at [e][l] (512-byte rows) ----
        double *mine = p.splitk_part +
                       ((((long)slot * SPLITK_SLOTS + (long)grp * st.splits + piece) * 4 + wave) *
                        (64 * 64)) + lane;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[((jt * 4 + it) * 4 + r) * 64] = acc4[jt][it][r];
        return;
    }
    if constexpr (SPLITK == 2) {
        // ---- the pieces of this tile pair, added piece by piece (ascending: deterministic), the
        //      64 loads of a piece in flight together ----
        const double *part = p.splitk_part +
                             ((((long)slot * SPLITK_SLOTS + (long)grp * st.splits) * 4 + wave) *
                              (64 * 64)) + lane;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc4[jt][it][r] = part[((jt * 4 + it) * 4 + r) * 64];
        for (int q2 = 1; q2 < st.splits; ++q2) {
            const double *pq = part + (long)q2 * 4 * (64 * 64);
            double v[4][4][4];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[jt][it][r] = pq[((jt * 4 + it) * 4 + r) * 64];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc4[jt][it][r] += v[jt][it][r];
        }
    }
    // the staging buffers are free (every wave passed the k-loop's last barrier): M strips go to
    // LDS for both tiles of the workgroup
    static_assert(EPI_LDS_BYTES + TOEP_LDS_BYTES <= 2 * STAGE,
                  "epilogue LDS must fit the stage buffers");
    probe.mark(3);
    const double *ksl = nullptr;
    if (structured) {
        double *sl = reinterpret_cast<double *>(smem + EPI_LDS_BYTES) + wave * 128;
        struct_slice(sl, g, p, item, (int)tile_row0(tile), col ? kmax + NB : kmax, lane);
        ksl = sl;
    }
    stage_mstrips(reinterpret_cast<double *>(smem), p.dinv + (long)item * (NB * NB), tid);
    __syncthreads();
    probe.mark(4);
    if (!valid) return;

    double *Lr = Lit + tile_row0(tile) * ld;
    if (col) {
        subtract_in_place_rows(Lr, ld, (j + 1) * NB, acc4, lane,
                               reinterpret_cast<double *>(smem + EPI_M_BYTES + wave * EPI_WAVE_BYTES),
                               IDENT && synth != 0, ksl, nit);
    } else if constexpr (MIXED) {
        const int rt = (tile < st.nmain) ? j + 1 + tile : g.nb0 + (tile - st.nmain);
        solve_and_store_lds<true>(
            acc4, Lr, reinterpret_cast<const double *>(smem), ld, kmax, lane,
            reinterpret_cast<double *>(smem + EPI_M_BYTES + wave * EPI_WAVE_BYTES), probe,
            p.L32 + (long)item * g.item_stride + tile_row0(tile) * ld,
            p.tmax + (long)item * (g.nb0 + g.naux_pad / NB) * g.nb0 + tmax_index(g, rt, j));
    } else {
        solve_and_store_lds<false, IDENT>(
            acc4, Lr, reinterpret_cast<const double *>(smem), ld, kmax, lane,
            reinterpret_cast<double *>(smem + EPI_M_BYTES + wave * EPI_WAVE_BYTES), probe, nullptr,
            nullptr, synth, ksl, nit);
    }
    probe.mark(5);
    probe.drain();       // stores retired (vmcnt(0))
    probe.mark(6);
    probe.emit_col(j, lane, wg, wave, item, tile, /*thin=*/false);
}

// ---------------------------------------------------------------------------------------
// launchers, on the probe policy the kernels are instantiated with
// ---------------------------------------------------------------------------------------
template <class Probe>
void launch_chol_diag_t(const JobGeom &g, const ChunkPtrs &p, int Bc, int j, int k0, hipStream_t s) {
    hipLaunchKernelGGL(chol_diag_kernel<Probe>, dim3(Bc), dim3(256), 0, s, g, p, j, k0);
}

template <class Probe>
void launch_chol_col_t(const JobGeom &g, const ChunkPtrs &p, int Bc, int j, int mode, int k0,
                     hipStream_t s, const DevSpec *sp) {
    ColStep st{};
    const bool mixed = p.L32 != nullptr && sp != nullptr;
    if (mixed) {
        st.c32 = sp->mixed_tau / (64.0 * 5.9604644775390625e-08);   // 64 * 2^-24
        st.jitter = sp->jitter;
    }
    st.j = j;
    st.k0 = k0;
    st.nmain = (mode == COL_AUX) ? 0 : g.nb0 - 1 - j;
    st.ntiles = st.nmain + g.naux_pad / NB;
    if (st.ntiles <= 0) return;
    const int bpad = (Bc + 7) / 8 * 8;
    const dim3 blk(256);
    st.first_touch = (mode == COL_FAT || (mode == COL_FULL && k0 == 0 && j == 0)) ? 1 : 0;
    if (mode == COL_FAT) {
        st.groups = (st.ntiles + 1) / 2;
        const dim3 grid(st.groups * bpad);
        // small batches: late columns (few tile pairs, long k-loops) are cut along k so that the
        // launch fills the chip; every piece keeps at least 8 staged chunks
        int splits = 1;
        if (!mixed && !g.aux_identity && p.splitk_part) {
            const int nchunks = j * NB / LDS_KC;
            // 384, not the 512 workgroups the chip holds: chol_diag's successor tile (diag_ahead,
            // 4 waves of 224 VGPRs) is resident beside this launch and a second round costs more
            // than the split saves (measured at 64 items, n = 2048)
            splits = std::min(std::min(8, SPLITK_SLOTS / std::max(st.groups, 1)),
                              std::min(nchunks / 8, 384 / std::max(st.groups * Bc, 1)));
        }
        st.splits = std::max(splits, 1);
        if (splits >= 2) {
            hipLaunchKernelGGL((chol_col_glds_kernel<false, Probe, false, 1>),
                               dim3(st.groups * splits * bpad), blk, 0, s, g, p, Bc, st);
            hipLaunchKernelGGL((chol_col_glds_kernel<false, Probe, false, 2>), grid, blk, 0, s, g, p,
                               Bc, st);
        } else if (mixed)
            hipLaunchKernelGGL((chol_col_glds_kernel<true, Probe>), grid, blk, 0, s, g, p, Bc, st);
        else if (g.aux_identity)
            hipLaunchKernelGGL((chol_col_glds_kernel<false, Probe, true>), grid, blk, 0, s, g, p, Bc, st);
        else
            hipLaunchKernelGGL((chol_col_glds_kernel<false, Probe>), grid, blk, 0, s, g, p, Bc, st);
    } else if (mode == COL_THIN && k0 == j * NB - NB && j > 0) {
        st.groups = (st.ntiles + 3) / 4;
        const dim3 grid(st.groups * bpad);
        if (mixed)
            hipLaunchKernelGGL((chol_col_thin_kernel<true, Probe>), grid, blk, 0, s, g, p, Bc, st);
        else if (g.aux_identity)
            hipLaunchKernelGGL((chol_col_thin_kernel<false, Probe, true>), grid, blk, 0, s, g, p, Bc, st);
        else
            hipLaunchKernelGGL((chol_col_thin_kernel<false, Probe>), grid, blk, 0, s, g, p, Bc, st);
    } else {
        st.groups = (st.ntiles + 3) / 4;
        const dim3 grid(st.groups * bpad);
        if (mixed)
            hipLaunchKernelGGL(chol_col_kernel<true>, grid, blk, 0, s, g, p, Bc, st);
        else if (g.aux_identity)
            hipLaunchKernelGGL((chol_col_kernel<false, true>), grid, blk, 0, s, g, p, Bc, st);
        else if (g.toep && st.first_touch)
            hipLaunchKernelGGL((chol_col_kernel<false, false, true>), grid, blk, 0, s, g, p, Bc, st);
        else
            hipLaunchKernelGGL(chol_col_kernel<false>, grid, blk, 0, s, g, p, Bc, st);
    }
}

}  // namespace ngp
