// Checks on the GPU at hand what the DPP wave shifts do (wave_shl:1 = lane i reads lane i + 1, wave_shr:1 = lane i reads
// lane i - 1, lanes without a source read 0 with bound_ctrl): prints OK or the lanes that differ.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* p) {
    const unsigned v = p[threadIdx.x];
    p[64 + threadIdx.x] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);   // wave_shl:1
    p[128 + threadIdx.x] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);  // wave_shr:1
}
int main() {
    unsigned h[192], *d;
    for (int i = 0; i < 64; i++) h[i] = 1000 + i;
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 2;
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    int bad = 0;
    for (int i = 0; i < 64; i++) {
        const unsigned dn = i < 63 ? 1000 + i + 1 : 0, up = i > 0 ? 1000 + i - 1 : 0;
        if (h[64 + i] != dn) { printf("wave_shl lane %d: %u want %u\n", i, h[64 + i], dn); bad++; }
        if (h[128 + i] != up) { printf("wave_shr lane %d: %u want %u\n", i, h[128 + i], up); bad++; }
    }
    puts(bad ? "DIFFERS" : "OK");
    return bad != 0;
}
