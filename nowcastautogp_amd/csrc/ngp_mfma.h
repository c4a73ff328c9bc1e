// ngp_mfma.h — the fp64 matrix-core idioms shared by every MFMA kernel of the library (gfx950).
#pragma once
#include "ngp_internal.h"

namespace ngp {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f64x4 mfma64(double a, double b, f64x4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// lane `lane`'s value of v in every lane (two v_readlane_b32; `lane` wave-uniform)
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------
// The fast fp64 matrix path.  Measured on MI355X (profiles/r01/ubench_mfma_f64.log):
//   v_mfma_f64_16x16x4_f64   ~100 cycles per SIMD slot (>=2 waves/SIMD)  -> 49.5 TFLOP/s ceiling
//   v_mfma_f64_4x4x4_4b_f64   16.5 cycles for 512 flop, one wave suffices -> 75 TFLOP/s
// The 4x4x4 form takes the SAME operand registers as the 16x16x4 form (A lane = m + 16 k,
// B lane = n + 16 k) but produces only the four diagonal 4x4 blocks of the 16x16 product
// (probed: D lane n' + 16 i = D[m = 4 (n'>>2) + i][n'], cbsz/abid ignored for f64).  Rotating the
// B operand left by 4 r lanes inside each 16-lane row (DPP row_ror:16-4r; probed:
// row_ror:n is dst[i] = src[(i - n) mod 16]) makes instruction r produce block-diagonal r:
//     acc[r] lane (n', i)  =  D[m = 4 (n'>>2) + i][n = (n' + 4 r) mod 16]
// so four of them (66 cycles) equal one 16x16x4 MFMA (100-138 cycles).  to_d16() gathers the
// four accumulators back into the 16x16x4 C/D register layout with bank-masked DPP moves, so
// everything downstream of the k-loop is unchanged.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double mfma4(double a, double b, double c) {
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_f64(double old, double src) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xF,
                                               BANK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xF,
                                               BANK, false);
    return __hiloint2double(hi, lo);
}
constexpr int ROW_ROR4 = 0x124, ROW_ROR8 = 0x128, ROW_ROR12 = 0x12C;

struct Rot4 {  // b and its three left-rotations by 4, 8, 12 lanes within each 16-lane row
    double r0, r1, r2, r3;
};
__device__ __forceinline__ Rot4 rot4(double b) {
    Rot4 o;
    o.r0 = b;
    o.r1 = dpp_f64<ROW_ROR12, 0xF>(b, b);
    o.r2 = dpp_f64<ROW_ROR8, 0xF>(b, b);
    o.r3 = dpp_f64<ROW_ROR4, 0xF>(b, b);
    return o;
}
// one 16x16x4 product as four 4x4x4 MFMAs
__device__ __forceinline__ void mfma16_as_4(double (&acc)[4], double a, const Rot4 &b) {
    acc[0] = mfma4(a, b.r0, acc[0]);
    acc[1] = mfma4(a, b.r1, acc[1]);
    acc[2] = mfma4(a, b.r2, acc[2]);
    acc[3] = mfma4(a, b.r3, acc[3]);
}
// One 16-deep chunk of the fat step's LDS image (ngp_col_kernels.h: chol_col_glds_kernel) into the
// wave's 64 x 64 tile — 256 4x4x4 MFMAs —, the LDS reads software-pipelined: the operands of row
// group 0 of k-step s + 1 (and its A values) are requested under the last sixteen MFMAs of k-step
// s, the operands of row groups 1..3 under the first sixteen of their own step.  A wave that has
// its SIMD to itself (its partner in an epilogue) no longer waits out an LDS round trip per
// k-step (+8 VGPRs).  a_addr[s] / b_addr[r][s]: the swizzled read addresses of k-step s (rotation
// r); GRP: bytes between the 16-row groups of a fragment.  Per accumulator the k-steps arrive in
// ascending order, as in the plain loop: bit-identical to it.  (Carried across the chunk's
// barrier too — the barrier before the last sixteen MFMAs, the next chunk's first operands
// requested behind it — it was slower: profiles/r04/lds_read_pipeline.txt.)
template <int GRP>
__device__ __forceinline__ void mult_chunk_pipelined(double (&acc4)[4][4][4], const char *buf,
                                                     const unsigned (&a_addr)[4],
                                                     const unsigned (&b_addr)[4][4]) {
    auto ld_a = [&](int s, double (&a)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            a[u] = *reinterpret_cast<const double *>(buf + a_addr[s] + u * GRP);
    };
    auto ld_b = [&](int s, int it, Rot4 &br) {
        br.r0 = *reinterpret_cast<const double *>(buf + b_addr[0][s] + it * GRP);
        br.r1 = *reinterpret_cast<const double *>(buf + b_addr[1][s] + it * GRP);
        br.r2 = *reinterpret_cast<const double *>(buf + b_addr[2][s] + it * GRP);
        br.r3 = *reinterpret_cast<const double *>(buf + b_addr[3][s] + it * GRP);
    };
    double a[4];
    Rot4 b0;
    ld_a(0, a);
    ld_b(0, 0, b0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        Rot4 b1, b2, b3;
        ld_b(s, 1, b1);
        ld_b(s, 2, b2);
        ld_b(s, 3, b3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) mfma16_as_4(acc4[jt][0], a[jt], b0);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) mfma16_as_4(acc4[jt][1], a[jt], b1);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) mfma16_as_4(acc4[jt][2], a[jt], b2);
        __builtin_amdgcn_sched_barrier(0);
        double an[4];
        Rot4 b0n;
        if (s < 3) {
            ld_a(s + 1, an);
            ld_b(s + 1, 0, b0n);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) mfma16_as_4(acc4[jt][3], a[jt], b3);
        if (s < 3) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = an[u];
            b0 = b0n;
        }
    }
}
// four block-diagonal accumulators -> the 16x16x4 C/D layout (reg s, lane (n, q): D[q + 4s][n])
__device__ __forceinline__ f64x4 to_d16(const double (&c)[4]) {
    f64x4 o;
    {
        double x = c[0];
        x = dpp_f64<ROW_ROR4, 0x2>(x, c[1]);
        x = dpp_f64<ROW_ROR8, 0x4>(x, c[2]);
        x = dpp_f64<ROW_ROR12, 0x8>(x, c[3]);
        o[0] = x;
    }
    {
        double x = c[0];
        x = dpp_f64<ROW_ROR4, 0x4>(x, c[1]);
        x = dpp_f64<ROW_ROR8, 0x8>(x, c[2]);
        x = dpp_f64<ROW_ROR12, 0x1>(x, c[3]);
        o[1] = x;
    }
    {
        double x = c[0];
        x = dpp_f64<ROW_ROR4, 0x8>(x, c[1]);
        x = dpp_f64<ROW_ROR8, 0x1>(x, c[2]);
        x = dpp_f64<ROW_ROR12, 0x2>(x, c[3]);
        o[2] = x;
    }
    {
        double x = c[0];
        x = dpp_f64<ROW_ROR4, 0x1>(x, c[1]);
        x = dpp_f64<ROW_ROR8, 0x2>(x, c[2]);
        x = dpp_f64<ROW_ROR12, 0x4>(x, c[3]);
        o[3] = x;
    }
    return o;
}

}  // namespace ngp
