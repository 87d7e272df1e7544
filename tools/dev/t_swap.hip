// dev probe: lane semantics of v_permlane32_swap / v_permlane16_swap on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
    unsigned v = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(v, v + 100, false, false);
    auto q = __builtin_amdgcn_permlane16_swap(v, v + 100, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
    out[128 + threadIdx.x] = q[0];
    out[192 + threadIdx.x] = q[1];
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[4] = {"swap32 ret[0] (vdst)", "swap32 ret[1] (src0)", "swap16 ret[0] (vdst)", "swap16 ret[1] (src0)"};
    for (int a = 0; a < 4; a++) {
        printf("%s:", names[a]);
        for (int i = 0; i < 64; i++) printf(" %u", h[a * 64 + i]);
        printf("\n");
    }
    return 0;
}
