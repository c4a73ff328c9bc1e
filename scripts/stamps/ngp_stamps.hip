// ngp_stamps.hip — DIAGNOSTIC translation unit (scripts/fat_phases.py); never part of libngp.so.
//
// Instantiates the column-sweep kernels of nowcastautogp_amd/csrc/ngp_col_kernels.h with a probe
// that records 100-MHz timestamps of every wave's phases and the CU / SIMD it ran on, and defines
// the two launchers the product declares weak, so that a library linked from
//     ngp_kernels.hip + ngp_api.hip + this file
// runs the stamping instantiation wherever the product runs the NoProbe one.  Built by
// scripts/fat_phases.py into build/libngp_stamps.so.
#include "../../nowcastautogp_amd/csrc/ngp_col_kernels.h"

namespace ngp {

constexpr int STAMP_WORDS = 32;
__device__ unsigned long long *ngp_stamps = nullptr;
__device__ unsigned int ngp_stamp_count = 0, ngp_stamp_cap = 0;
// which launch is watched: j (fat step of block column j), -j (thin step), 1000 + j (chol_diag)
__device__ int ngp_stamp_j = -1;

struct StampProbe {
    // slots 0..11: phases of the kernel; 12..29: the passes of solve_and_store_lds
    unsigned long long t[30] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    __device__ __forceinline__ void mark(int slot) { t[slot] = __builtin_amdgcn_s_memrealtime(); }
    // not before `dep` is computed, ordered with the memory operations around it
    __device__ __forceinline__ void mark_after(int slot, double dep) {
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t[slot]) : "v"(dep) : "memory");
    }
    __device__ __forceinline__ void mark_after(int slot, int dep) {
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t[slot]) : "v"(dep) : "memory");
    }
    __device__ __forceinline__ void drain() { __builtin_amdgcn_s_waitcnt(0); }
    __device__ __forceinline__ void emit_diag(int j, int tid, int item) {
        if (j != ngp_stamp_j - 1000 || tid != 0) return;
        const unsigned idx = atomicAdd(&ngp_stamp_count, 1u);
        if (idx >= ngp_stamp_cap) return;
        unsigned long long *o = ngp_stamps + (size_t)idx * STAMP_WORDS;
        o[0] = 0;
        o[1] = (unsigned long long)item;
#pragma unroll
        for (int i = 0; i < 6; ++i) o[2 + i] = t[i];
    }
    __device__ __forceinline__ void emit_col(int j, int lane, int wg, int wave, int item, int tile,
                                             bool thin) {
        if (j != (thin ? -ngp_stamp_j : ngp_stamp_j) || lane != 0) return;
        const unsigned idx = atomicAdd(&ngp_stamp_count, 1u);
        if (idx >= ngp_stamp_cap) return;
        unsigned long long *o = ngp_stamps + (size_t)idx * STAMP_WORDS;
        const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        o[0] = ((unsigned long long)xcc << 32) | hwid;
        o[1] = ((unsigned long long)wg << 8) | (unsigned)wave;
#pragma unroll
        for (int i = 0; i < 7; ++i) o[2 + i] = t[i];
        o[9] = (unsigned long long)item;
        o[10] = (unsigned long long)tile;
        o[11] = t[7];
#pragma unroll
        for (int i = 0; i < 18; ++i) o[12 + i] = t[12 + i];
    }
};

// strong definitions: they take the place of the product's weak NoProbe launchers at link time
void launch_chol_diag(const JobGeom &g, const ChunkPtrs &p, int Bc, int j, int k0, hipStream_t s) {
    launch_chol_diag_t<StampProbe>(g, p, Bc, j, k0, s);
}
void launch_chol_col(const JobGeom &g, const ChunkPtrs &p, int Bc, int j, int mode, int k0,
                     hipStream_t s, const DevSpec *sp) {
    launch_chol_col_t<StampProbe>(g, p, Bc, j, mode, k0, s, sp);
}

}  // namespace ngp

extern "C" int ngp_dbg_stamps_begin(int j, unsigned cap) {
    unsigned long long *buf = nullptr;
    if (hipMalloc(&buf, (size_t)cap * ngp::STAMP_WORDS * 8) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, (size_t)cap * ngp::STAMP_WORDS * 8);
    unsigned zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(ngp::ngp_stamps), &buf, sizeof(buf));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(ngp::ngp_stamp_count), &zero, sizeof(zero));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(ngp::ngp_stamp_cap), &cap, sizeof(cap));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(ngp::ngp_stamp_j), &j, sizeof(j));
    return 0;
}
extern "C" long ngp_dbg_stamps_fetch(unsigned long long *out, unsigned cap) {
    (void)hipDeviceSynchronize();
    unsigned n = 0;
    unsigned long long *buf = nullptr;
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(ngp::ngp_stamp_count), sizeof(n));
    (void)hipMemcpyFromSymbol(&buf, HIP_SYMBOL(ngp::ngp_stamps), sizeof(buf));
    if (n > cap) n = cap;
    (void)hipMemcpy(out, buf, (size_t)n * ngp::STAMP_WORDS * 8, hipMemcpyDeviceToHost);
    int off = -1;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(ngp::ngp_stamp_j), &off, sizeof(off));
    (void)hipFree(buf);
    return (long)n;
}
