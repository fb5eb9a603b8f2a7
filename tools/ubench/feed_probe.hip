// Microbenchmark: per-CU operand feed rate into LDS, L2-resident data, GEMM-like access (256-row panels, 128 B per row per
// k-step, row pitch = K * 2 bytes).  Variant 0: global_load_lds_dwordx4 (LDS-DMA).  Variant 1: global_load_dwordx4 into
// registers + ds_write_b128.  One 512-thread workgroup per CU; all workgroups of an XCD read the same two panels.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VAR, int INFLIGHT>
__global__ __launch_bounds__(512) void feed(const char* base, long panel_bytes, int pitch, int ksteps, int iters, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7;
    const char* pa = base + (long)xcd * 2 * panel_bytes;        // A panel, B panel right behind
    // piece p (0..63) of a stage: operand p >> 5, rows ((p & 31) * 8 .. +8), 128 B per row: lane -> row (lane >> 3), 16 B chunk (lane & 7)
    int acc = 0;
    for (int it = 0; it < iters; ++it) {
        for (int k = 0; k < ksteps; ++k) {
            char* stage = smem + (k & 1) * 65536;
            if (VAR == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int p = wave * 8 + i;
                    const char* src = pa + (long)(p >> 5) * panel_bytes + (long)((p & 31) * 8 + (lane >> 3)) * pitch + k * 128 + (lane & 7) * 16;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(stage + p * 1024), 16, 0, 0);
                }
                if (INFLIGHT == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                else { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
                __builtin_amdgcn_s_barrier();
            } else {
                uint4 r[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int p = wave * 8 + i;
                    const char* src = pa + (long)(p >> 5) * panel_bytes + (long)((p & 31) * 8 + (lane >> 3)) * pitch + k * 128 + (lane & 7) * 16;
                    r[i] = *(const uint4*)src;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int p = wave * 8 + i;
                    *(uint4*)(stage + p * 1024 + lane * 16) = r[i];
                }
                __syncthreads();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc = *(int*)(smem + tid * 4);
    if (acc == 0x12345678) sink[0] = acc;
}

int main() {
    const int K = 768, pitch = K * 2, rows = 256, ksteps = K / 64;
    const long panel = (long)rows * pitch;
    char* buf; int* sink;
    CHECK(hipMalloc(&buf, 16 * panel)); CHECK(hipMemset(buf, 1, 16 * panel)); CHECK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char* name) {
        CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        const int iters = 200;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), 131072, 0, (const char*)buf, panel, pitch, ksteps, iters, sink);
            hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)iters * ksteps * 65536;
            if (rep) printf("%-44s %8.1f us  %6.1f GB/s per CU  (%5.1f B/clk @2.4GHz)  chip %5.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e9,
                            bytes / (ms * 1e-3) / 2.4e9, bytes * 256 / (ms * 1e-3) / 1e12);
        }
    };
    run(feed<0, 0>, "LDS-DMA dwordx4, wait all per step");
    run(feed<0, 1>, "LDS-DMA dwordx4, one step in flight");
    run(feed<1, 0>, "global_load_dwordx4 -> VGPR -> ds_write_b128");
    return 0;
}
