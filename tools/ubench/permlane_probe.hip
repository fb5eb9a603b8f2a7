// Microbenchmark / semantics check: v_permlane32_swap_b32 (gfx950) as the cross-half exchange of a wave64 (lane j <-> lane j + 32)
// in place of __shfl_xor(v, 32) = ds_bpermute_b32 (an LDS round trip).  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/pl tools/ubench/permlane_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* p) {
    float v = (float)threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    p[threadIdx.x] = __uint_as_float(r[0]);
    p[64 + threadIdx.x] = __uint_as_float(r[1]);
    p[128 + threadIdx.x] = __shfl_xor(v, 32, 64);
}
int main() {
    float* d; hipMalloc(&d, 192 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int a = 0; a < 3; ++a) { printf("%s:", a == 0 ? "r[0]" : a == 1 ? "r[1]" : "shfl_xor32"); for (int i = 0; i < 64; i += 9) printf(" [%d]=%g", i, h[a * 64 + i]); printf("\n"); }
    return 0;
}
