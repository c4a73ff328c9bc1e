// Pin the semantics of v_mov_b32 dpp row_ror:n and bank_mask on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int BANK>
__global__ void k(int *out) {
    int l = threadIdx.x;
    out[l] = __builtin_amdgcn_update_dpp(-1, l, CTRL, 0xF, BANK, false);
}
template <int CTRL, int BANK>
void run(const char *name) {
    int *d, h[64];
    hipMalloc(&d, 256);
    hipLaunchKernelGGL((k<CTRL, BANK>), dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("%s:", name);
    for (int i = 0; i < 32; ++i) printf(" %d", h[i]);
    printf("\n");
    hipFree(d);
}
int main() {
    run<0x124, 0xF>("row_ror:4  bank=F");
    run<0x128, 0xF>("row_ror:8  bank=F");
    run<0x12C, 0xF>("row_ror:12 bank=F");
    run<0x124, 0x2>("row_ror:4  bank=2");
    run<0x104, 0xF>("row_shl:4  bank=F");
    return 0;
}
