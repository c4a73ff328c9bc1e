// ngp_internal.h — shared between the kernel TU (ngp_kernels.hip) and the host/C-ABI TU
// (ngp_api.hip).  gfx950 only; no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "../../include/ngp.h"

namespace ngp {

constexpr int NB = 64;        // block-column width of the left-looking factorisation
constexpr int TB = 16;        // MFMA tile edge (v_mfma_f64_16x16x4_f64)
constexpr int DEV_STACK = 8;  // register-resident evaluation stack depth on the device
constexpr int SPLITK_SLOTS = 16;   // (tile pair, piece) slots per item in ChunkPtrs::splitk_part

// device opcode = host opcode, plus CP with its two operands in swapped stack order
// (the host reorders children so the evaluation stack never exceeds DEV_STACK)
constexpr int OP_CP_SWAPPED = 9;
// reduced program only: a whole stationary subtree (no Linear, no ChangePoint below it) read from
// its table by lattice distance
constexpr int OP_TABLE = 10;
constexpr int RLEAF_TABLE = 1, RLEAF_LINEAR = 2;   // DevProgram::rops >> 4
constexpr int MAX_TABLES = NGP_MAX_OPS / 2;   // a tree of NGP_MAX_OPS nodes has at most that many leaves

// One particle's kernel, flattened for the device.  Copied into LDS by every workgroup that
// needs it; ops are uniform across the workgroup so the interpreter never diverges.
struct DevProgram {
    int32_t n_ops;
    int32_t n_params;
    double  noise;               // observation-noise variance (jitter is in the spec)
    uint8_t ops[NGP_MAX_OPS];
    uint8_t slot[NGP_MAX_OPS];   // per op: table slot (stationary leaf) or sigmoid slot (ChangePoint)
    uint8_t first[NGP_MAX_OPS];  // binary ops: index of the operand evaluated first (the second is i-1)
    uint8_t poff[NGP_MAX_OPS];   // per op: offset of its parameters in params[]
    double  params[NGP_MAX_PARAMS];
    // Reduced program for value jobs on lattice times: every maximal stationary subtree is ONE
    // table leaf (its value depends on |t1 - t2| only, so tables_kernel evaluates the subtree once
    // per lattice distance — the same operations in the same order as element by element).
    // A binary operation whose second operand is a leaf carries that leaf with it (high nibble of
    // rops: RLEAF_TABLE / RLEAF_LINEAR, rleaf = its table slot / parameter offset): the operation
    // then works on the top of the stack in place — a left-deep tree never moves the stack at all.
    // Same operands, same operation: the value is bit-identical to the unfused evaluation.
    int32_t n_rops, n_tab;
    int32_t rchain, rpad_;              // rchain: the reduced program pushes once (its first
                                        // instruction) and every other instruction carries its leaf
    uint8_t rops[NGP_MAX_OPS];          // low nibble: NGP_OP_LINEAR, OP_TABLE, Plus / Times / ChangePoint (+ swapped)
    uint8_t rslot[NGP_MAX_OPS];         // OP_TABLE: table slot; ChangePoint: sigmoid slot (= slot[] of the op)
    uint8_t rpoff[NGP_MAX_OPS];         // Linear / ChangePoint: offset of the parameters
    uint8_t rleaf[NGP_MAX_OPS];         // fused leaf: table slot (RLEAF_TABLE) or parameter offset (RLEAF_LINEAR)
    uint8_t tb_first[MAX_TABLES];       // table k tabulates ops[tb_first[k] .. tb_last[k]] (a postfix
    uint8_t tb_last[MAX_TABLES];        //  range of the full program = one subtree)
};
static_assert(sizeof(DevProgram) % 8 == 0, "DevProgram is copied as 8-byte words");
// the whole tree is stationary: its reduced program is ONE table lookup (fill_single_kernel)
__host__ __device__ inline bool prog_single_table(const DevProgram *P) {
    return P->n_rops == 1 && P->rops[0] == OP_TABLE;
}
// Structured items (JobGeom::toep: their tiles below the block diagonal are never stored, the
// column kernels regenerate them from 127 numbers in LDS):
//   1  the whole tree is stationary        K_ik = tab[toep |i - k|]            (Toeplitz)
//   0  anything else: stored
// (A single Linear leaf, K_ik = bias + amp (t_i - c)(t_k - c), was built as a second kind — 17 % of
// the bench ensemble, fill 35 -> 29 ms — and taken out again: with a second kind in the epilogue
// the fat kernel, stored tiles included, ran 4-5 % slower, whichever way the two kinds shared the
// code; profiles/r03/README.md.)
__host__ __device__ inline int prog_structure(const DevProgram *P) {
    return (P->n_rops == 1 && P->rops[0] == OP_TABLE) ? 1 : 0;
}
struct DevSpec {
    int32_t se_form, periodic_form, cp_form, precision;
    double  jitter;
    double  mixed_tau;   // NGP_PREC_MIXED: see ngp_spec in include/ngp.h
};

// Geometry of one staged job on the device (all items share times; see ngp_api.hip).
struct JobGeom {
    int32_t B;         // items (kernels)
    int32_t n0;        // main block: multiple of NB training points
    int32_t nb0;       // n0 / NB
    int32_t da;        // appended rows: tail (n - n0) + d nowcast points
    int32_t tail;      // n - n0: appended rows that belong to the base data
    int32_t m;         // forecast points
    int32_t naux;      // da + m + 1 (last aux row carries y)
    int32_t naux_pad;  // naux rounded up to a multiple of NB
    int32_t D;         // scenarios
    int32_t d;         // nowcast points per scenario
    int32_t noise_on_new;
    int32_t y_shared;  // 1: one base y / ya for all items
    int32_t lattice;   // 1: all times sit on a lattice t = tmin + q h  -> table-driven fill
    int32_t R;         // lattice table length (max |q_i - q_j| + 1)
    int32_t npts;      // n0 + da + m points that carry a time
    int32_t maxstat;   // stationary-leaf table slots per item
    int32_t maxcp;     // ChangePoint sigmoid slots per item
    int32_t n_real;    // main-block points that are data; rows/cols beyond are identity padding
    int32_t aux_identity;  // 1: aux rows are [I_n0 ; y'] (gradient path: W = L^-T)
    int32_t maxops;    // longest program of the batch (gradient jobs: picks the contraction kernel)
    int32_t aux_e1;    // 1: the Toeplitz gradient path — aux rows [y' ; e_1'] (the row after y' is e_1')
    int32_t toep;      // > 0: the main-block points sit on the lattice at a constant stride (q_i = q_0
                       // +- toep i), so K of a stationary tree is Toeplitz: K_ik = tab[toep |i - k|].  The
                       // fill then writes only the diagonal tiles and the aux rows of structured items
                       // (prog_structure) and the column kernels regenerate a tile from 128 numbers in
                       // LDS where they would have read the stored tile (staged fp64 value jobs; else 0)
    int32_t tab_sub;   // gradient jobs: table slot of the first SUBTREE table (the reduced program's tables
                       // follow the per-leaf ones, maxstat counts both); value jobs: 0
    int32_t invariant; // ngp_set_batch_invariant: nothing about an item's arithmetic may depend on
                       // the size of the batch it travels in (no split-k of small chunks, gradient
                       // routing and contraction shapes by item / geometry only, the epilogue never
                       // on the resident tables of a single-chunk job)
    int32_t short_series;  // ngp_set_short_series_path: n0 <= 256 is factorised in one launch
    int32_t pad_;
    double  h;         // lattice step
    int64_t ld;        // row stride of the factor storage (= n0)
    int64_t item_stride;  // elements per item in the factor storage
};

struct ChunkPtrs {
    double       *L;      // [Bc][(n0 + naux_pad) x n0] factor + aux rows, row-major
    double       *dinv;   // [Bc][64 strips x 64 lanes] L_jj^-1 of the current step, MFMA strip order
    const DevProgram *progs;  // [Bc] (already offset to the chunk)
    const DevProgram *progs_src;  // resident gradient jobs with new parameters: the programs in page-locked
                              // HOST memory — tables_kernel reads them from there and leaves the device copy
                              // (progs) for the kernels behind it, instead of a copy ahead of the chain; else null
    const double *t0;     // [n0]
    const double *taux;   // [da + m] times of the aux rows (appended then forecast)
    const double *y0;     // [Bc or 1][n0] (already offset to the chunk when per item)
    double       *logdet; // [Bc]
    int32_t      *info;   // [Bc]
    double       *tab;    // [Bc][maxstat][R]   stationary-leaf values by lattice distance
    double       *sig;    // [Bc][maxcp][npts]  ChangePoint sigmoids by point
    const int32_t *qpts;  // [npts] lattice coordinate of every point (t0 then taux)
    double       *dtab;   // [Bc][maxstat][3][R] gradient jobs on a lattice: per stationary leaf the
                          // unscaled factor e and the two parameter-derivative factors (else null)
    // ---- NGP_PREC_MIXED jobs only (all null otherwise) ----
    float        *L32;    // [Bc][item_stride] fp32 shadow of the finished off-diagonal tiles of L
                          // and of the aux rows W (operands of the fp32 tile products)
    float        *tmax;   // [Bc][nb0 + naux_pad/64][nb0] max |.| of every finished 64x64 tile
                          // (row tile, block column); aux tiles follow the nb0 main row tiles
    unsigned     *mixcnt; // [Bc][2] tile products of the fat steps that ran in fp32 / in fp64
    double       *auxX;   // [Bc][naux_pad][n0] the fill also leaves the aux rows X here
    const int32_t *order; // mixed fat steps: dispatch order of the items (heaviest first) or null
    // staged value jobs on a lattice: the chunk's items split by the shape of their reduced program
    // (job-wide indices, ascending; item of the chunk = entry - fill_base); null: every item goes
    // through the general fill
    const int32_t *fill_chain, *fill_other, *fill_single;
    int32_t n_fill_chain, n_fill_other, n_fill_single, fill_base;
    double       *G;      // short value jobs: [Bc][naux x naux] — chol_small_kernel leaves the Gram matrix
                          // of the aux rows itself (gram_kernel's arithmetic, one launch less); else null
    const int32_t *items; // refinement sweeps: the items (indices into the chunk) a launch works
                          // on, Bc = their count; null: all of them in order
    // small chunks only (null otherwise): split-k fat steps, see chol_col_glds_kernel<.., SPLITK>
    double       *splitk_part;  // [Bc][SPLITK_SLOTS][4 waves][64 x 64] accumulator tiles of the pieces
};

struct EpiPtrs {
    const DevProgram *progs;  // [B]
    const double *taux;       // [da + m]
    const double *G;          // [B][naux x naux]
    const double *ya;         // [B or 1][D][da]
    const double *logdet;     // [B]
    int32_t      *info;       // [B]
    double       *work;       // [B][work_stride]
    double       *zbuf;       // [B][D][da]
    double       *logml_base; // [B]
    double       *logml_full; // [B][D]
    double       *mu;         // [B][D][m]
    double       *sigma;      // [B][m][m]
    int64_t       work_stride;
    // lattice tables of the aux points when they are still resident (single-chunk jobs), else null:
    // the small Schur blocks are then lookups instead of fp64 transcendental evaluations
    const double *tab, *sig;  // [B][maxstat][R], [B][maxcp][npts]
    const int32_t *qpts;      // [npts]
};

// ---- short series in one launch (ngp_small_kernels.h) ------------------------------------
constexpr int SM_WAVES = 8, SM_THREADS = 64 * SM_WAVES;   // two waves per SIMD: 256 VGPRs each
constexpr int SM_NSLOT = 20;           // register blocks per wave (8 VGPRs each)
constexpr int SM_DSTR = 17;            // row stride of a diagonal block in LDS (doubles)
constexpr int SM_MAX_PANEL = 34;       // panel blocks of one step: main rows + the sweep's aux row-blocks
constexpr int SM_MAX_SWEEPS = 4;
constexpr int SM_MAX_ITEMS = 4096;      // larger chunks fill the chip on the column sweep (measured: 24 x n = 208 scenarios x particles up to 4,096 items win here, 8,192 gradient items lose)

// One sweep over the block columns.  Aux row-blocks taken along: identity rows [i0, i1) (gradient
// jobs: row-block a is e_(16a..16a+15)', it joins at column a and stays upper triangular) and dense
// rows [a0, a1) (slab rows n0 + 16 a ...).
struct SmallSweep {
    int32_t main;          // 1: factorise the main block; 0: aux rows only, against the stored factor;
                           // 2: the inverse phase of a gradient job — W_I = L^-T block column by block
                           // column of L^-1, every column a wave's own task (SmallPlan::colwave)
    int32_t i0, i1, a0, a1;
};
struct SmallPlan {
    int32_t nbe;           // 16-blocks of the main block that hold data: ceil(n_real / 16)
    int32_t nsweeps;
    int32_t ident;         // gradient job (aux rows [I ; y'])
    int32_t npanel;        // panel blocks to reserve in LDS
    uint64_t colwave;      // inverse phase: 4 bits per block column j — the wave that computes it
    SmallSweep sw[SM_MAX_SWEEPS];
};

constexpr int SM_LDS_FIXED = 8 * (16 * 256 + 16 * 16 * SM_DSTR + 256 + 32);   // M_j | diagonal blocks | pivots | flags
inline int small_lds_bytes(const SmallPlan &pl) { return SM_LDS_FIXED + pl.npanel * 2048; }

// The plan of a geometry, or false when the column sweep has to do it (n0 > 256, structured
// storage, the Toeplitz gradient path, more aux rows than four sweeps hold).
inline bool small_plan(const JobGeom &g, SmallPlan *pl) {
    if (g.n0 <= 0 || g.nb0 > 4 || g.aux_e1 || g.toep) return false;
    const int nb16 = g.n0 / 16, nbe = (g.n_real + 15) / 16;
    if (nbe < 1 || nbe > nb16) return false;
    const int cap_main = (SM_WAVES - 1) * SM_NSLOT, cap_aux = SM_WAVES * SM_NSLOT;
    SmallPlan p{};
    p.nbe = nbe;
    p.ident = g.aux_identity ? 1 : 0;
    int used = nbe * (nbe - 1) / 2;
    if (used > cap_main) return false;
    int ns = 0, npanel = nbe;
    if (g.aux_identity) {
        const int ytile = nb16;                           // slab rows 2 n0 ...: y'
        if (used + nbe > cap_main) return false;          // (120 + 16 <= 140)
        p.sw[ns++] = SmallSweep{1, 0, 0, ytile, ytile + 1};
        p.sw[ns++] = SmallSweep{2, 0, nbe, 0, 0};
        npanel = std::max(npanel, nbe + 1);
        // Block column j of L^-1 costs (nbe - j)(nbe - j - 1) / 2 block products, all on one wave.
        // Longest first, each to the wave whose SIMD (waves w and w + 4 share one) carries least;
        // of that SIMD's two waves the less loaded one.
        int load[SM_WAVES] = {};
        for (int j = 0; j < nbe; ++j) {
            int best = 0;
            for (int w = 1; w < SM_WAVES; ++w) {
                const int sb = load[best % 4] + load[best % 4 + 4], sw_ = load[w % 4] + load[w % 4 + 4];
                if (sw_ < sb || (sw_ == sb && load[w] < load[best])) best = w;
            }
            load[best] += (nbe - j) * (nbe - j - 1) / 2 + 1;
            p.colwave |= (uint64_t)best << (4 * j);
        }
    } else {
        const int nba = (g.naux + 15) / 16;
        int a = std::min(nba, (cap_main - used) / nbe);
        a = std::min(a, SM_MAX_PANEL - nbe);
        p.sw[ns++] = SmallSweep{1, 0, 0, 0, a};
        npanel = std::max(npanel, nbe + a);
        while (a < nba) {
            if (ns == SM_MAX_SWEEPS) return false;
            const int b = std::min(nba, a + std::min(cap_aux / nbe, SM_MAX_PANEL - nbe));
            p.sw[ns++] = SmallSweep{0, 0, 0, a, b};
            npanel = std::max(npanel, nbe + b - a);
            a = b;
        }
    }
    if (npanel > SM_MAX_PANEL) return false;
    p.nsweeps = ns;
    p.npanel = std::max(npanel, 9);     // (the waves' 16 x 18 staging corners of the prologue: 8 x 2,304 B)
    *pl = p;
    return true;
}

// the rule every caller shares: the geometry qualifies and the chunk is not one that fills the chip
// many times over (batch-invariant jobs: the geometry alone decides)
inline bool small_job(const JobGeom &g, int Bc, SmallPlan *pl = nullptr) {
    SmallPlan tmp;
    return g.short_series && small_plan(g, pl ? pl : &tmp) && (Bc <= SM_MAX_ITEMS || g.invariant);
}

// flops the launch executes per item (factor + the aux rows' solves and updates), for the profile
inline double small_flops(const JobGeom &g, const SmallPlan &pl) {
    const double n = 16.0 * pl.nbe;
    double f = n * n * n / 3.0;
    if (pl.ident) f += n * n * n / 3.0 + n * n;
    else f += (double)g.naux * n * n;
    return f;
}

void launch_chol_small(const JobGeom &g, const ChunkPtrs &p, int Bc, const SmallPlan &pl, hipStream_t s);
// K^-1 = W_I W_I', alpha = W_I z and z'z of a short gradient job in one launch (16 x 16 blocks)
void launch_grad_kinv_small(const JobGeom &g, const double *L, double *Kinv, double *alpha, double *quad,
                            int Bc, hipStream_t s);

// ---- launchers implemented in ngp_kernels.hip ------------------------------------------
void launch_tables(const JobGeom &g, const ChunkPtrs &p, int Bc, const DevSpec &sp, hipStream_t s);
void launch_fill(const JobGeom &g, const ChunkPtrs &p, int Bc, const DevSpec &sp, hipStream_t s,
                 bool aux_only = false);
void launch_chol_diag(const JobGeom &g, const ChunkPtrs &p, int Bc, int j, int k0, hipStream_t s);
enum { COL_FULL = 0, COL_FAT = 1, COL_THIN = 2, COL_AUX = 3 };
// sp: only mixed_tau / jitter are read, and only when p.L32 is set (mixed-precision job)
void launch_chol_col(const JobGeom &g, const ChunkPtrs &p, int Bc, int j, int mode, int k0,
                     hipStream_t s, const DevSpec *sp = nullptr);
// ---- Gram refinement of NGP_PREC_MIXED (G = X K^-1 X' against the fp64 covariance) ----
// A_c = C_c L_cc^-1, then C_j -= A_c L_(c,j) for j < c, c descending: C (aux rows of the slab,
// [Bc][naux_pad][n0] at rows n0.. of p.L) becomes C (L L')^-1 ... see aux_back_kernel
void launch_aux_back(const JobGeom &g, const ChunkPtrs &p, const double *dinv_all, size_t mstep,
                     double *Aout, int accumulate, int Bc, int c, hipStream_t s);
// R = X - A K  (K re-evaluated tile by tile from the kernel trees / lattice tables)
void launch_kapply(const JobGeom &g, const ChunkPtrs &p, const double *A, const double *X,
                   double *R, int Bc, const DevSpec &sp, hipStream_t s);
// G = sym(A X' + R A'); delta[2b] = max |R A'| relative to sqrt(G_aa G_bb), delta[2b+1] = max_a |R_a|/|X_a|
void launch_refine_gram(const JobGeom &g, const double *A, const double *X, const double *R,
                        double *S, double *T, double *U, double *G, double *delta, int Bc,
                        const int32_t *items, hipStream_t s);
void launch_mfma_f32_probe(const float *A, const float *Bm, float *Dout, hipStream_t s);
void launch_diag_ahead(const JobGeom &g, const ChunkPtrs &p, int Bc, int j, hipStream_t s);
// order <- items by fp64 tile products since the last call, most first (prev: [Bc] snapshot)
constexpr int NGP_MIXED_ORDER_MAX = 8192;   // items mixed_order_kernel ranks in LDS (32 KiB)
bool launch_mixed_order(const ChunkPtrs &p, unsigned *prev, int32_t *order, int Bc, hipStream_t s);
void launch_aux_update(const JobGeom &g, const ChunkPtrs &p, int Bc, int j, hipStream_t s);
// side / fork / join: when given, alpha is computed beside K^-1 on the side stream
void launch_grad_kinv(const JobGeom &g, const double *L, double *Kinv, double *alpha, double *quad,
                      int Bc, hipStream_t s, hipStream_t side = nullptr, hipEvent_t fork = nullptr,
                      hipEvent_t join = nullptr);
// items / bucket_counts: the chunk's items (chunk-local indices) sorted by tree size into
// GRAD_BUCKETS groups — at most 1, 2, 4, 8, 16 leaves, larger — and the size of every group;
// null: one launch for the whole chunk, sized by g.maxops
constexpr int GRAD_BUCKETS = 6;
inline int grad_bucket(int n_ops) {
    return n_ops <= 1 ? 0 : n_ops <= 3 ? 1 : n_ops <= 7 ? 2 : n_ops <= 15 ? 3 : n_ops <= 31 ? 4 : 5;
}
void launch_grad_contract(const JobGeom &g, const ChunkPtrs &p, const double *Kinv,
                          const double *alpha, const double *quad, double *partials, double *grad,
                          double *logml, int Bc, const DevSpec &sp, hipStream_t s,
                          const int32_t *items = nullptr, const int32_t *bucket_counts = nullptr,
                          hipStream_t side = nullptr, hipEvent_t fork = nullptr,
                          hipEvent_t join = nullptr);
// workgroups per 64x64 tile of the gradient contraction: small launches are cut finer
// (batch-invariant jobs: by the geometry alone — the split decides how a thread groups its rows,
// i.e. the order of a partial sum)
inline int grad_contract_split(long ntri, long Bc, bool invariant = false) {
    if (invariant) return ntri <= 36 ? 4 : 1;
    return ntri * Bc <= 1024 ? 4 : (ntri * Bc <= 2048 ? 2 : 1);
}
void launch_toep_grad(const JobGeom &g, const ChunkPtrs &p, const double *A, double *wbuf,
                      const double *quad, double *partials, double *grad, double *logml, int Bc,
                      const DevSpec &sp, hipStream_t s, const int32_t *items = nullptr,
                      const int32_t *bucket_counts = nullptr);
void launch_toep_quad(const JobGeom &g, const double *L, double *quad, int Bc, hipStream_t s);
void launch_gram(const JobGeom &g, const double *L, double *G, int Bc, hipStream_t s);
void launch_epilogue(const JobGeom &g, const EpiPtrs &p, const DevSpec &sp, hipStream_t s);
void launch_cov(const DevProgram *progs, int B, const double *t1, int n1, const double *t2,
                int n2, int add_diag, double *out, const DevSpec &sp, hipStream_t s);
void launch_mixture_sample(int P, int S, int m, const double *w, const double *mu, double *chol,
                           int draws, uint64_t seed, const uint64_t *seeds, double *out,
                           int32_t *comp, int32_t *info, hipStream_t s);
void launch_mfma_bench(double *out, int iters, int blocks, hipStream_t s);
void launch_mfma_bench_detail(unsigned long long *stamps, int iters, int blocks, hipStream_t s);
void launch_mfma_layout_probe(const double *A, const double *Bm, double *Dout, hipStream_t s);
void launch_stream_write(double *dst, int64_t n, hipStream_t s);
void launch_stream_copy(double *dst, const double *src, int64_t n, hipStream_t s);

}  // namespace ngp
