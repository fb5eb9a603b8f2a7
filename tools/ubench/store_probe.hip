// Microbenchmark: per-CU global store rate for the GEMM epilogue's access patterns.  One 512-thread workgroup per CU writes
// 256 x 256 bf16 tiles (128 KB) of an (M x N) row-major output, 16 bytes per lane per store:
//   P8  : a wave instruction covers 8 rows x 128 B   (wave-private 64-column slices: the shipped epilogue)
//   P2  : a wave instruction covers 2 rows x 512 B   (block-wide rows of 256 columns)
//   P1  : a wave instruction covers 1 KB contiguous  (tile stored as a compact block: upper bound)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int PAT>
__global__ __launch_bounds__(512) void store_k(char* out, int tiles_n, int ntiles, long pitch) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint4 v = make_uint4(tid, 1, 2, 3);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        char* base = out + (long)(t / tiles_n) * 256 * pitch + (long)(t % tiles_n) * 512;
        if (PAT == 8) {           // wave w: rows [128 (w >> 2), +128), columns [64 (w & 3), +64): 16 instructions of 8 rows
            char* wb = base + (long)(wave >> 2) * 128 * pitch + (wave & 3) * 128;
#pragma unroll
            for (int i = 0; i < 16; ++i) *(uint4*)(wb + (long)(i * 8 + (lane >> 3)) * pitch + (lane & 7) * 16) = v;
        } else if (PAT == 2) {    // wave w: rows [32 w, +32): 16 instructions of 2 rows x 512 B
            char* wb = base + (long)wave * 32 * pitch;
#pragma unroll
            for (int i = 0; i < 16; ++i) *(uint4*)(wb + (long)(i * 2 + (lane >> 5)) * pitch + (lane & 31) * 16) = v;
        } else {                  // compact: tile t at out + t * 128 KB
            char* wb = out + (long)t * 131072 + wave * 16384;
#pragma unroll
            for (int i = 0; i < 16; ++i) *(uint4*)(wb + i * 1024 + lane * 16) = v;
        }
    }
}

int main() {
    const int M = 45056, N = 3072;                     // 176 x 12 tiles
    const long pitch = (long)N * 2;
    const int tiles_n = N / 256, ntiles = (M / 256) * tiles_n;
    char* out; CHECK(hipMalloc(&out, (long)M * pitch));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, int grid, const char* name) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, out, tiles_n, ntiles, pitch);
            hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)ntiles * 131072;
            if (rep == 2) printf("%-44s grid %4d: %7.1f us  %6.2f TB/s chip  %6.1f GB/s per CU\n", name, grid, ms * 1e3, bytes / (ms * 1e-3) / 1e12,
                                 bytes / (grid < 256 ? grid : 256) / (ms * 1e-3) / 1e9);
        }
    };
    for (int grid : {16, 64, 256}) {
        run(store_k<8>, grid, "8 rows x 128 B per instruction (shipped)");
        run(store_k<2>, grid, "2 rows x 512 B per instruction");
        run(store_k<1>, grid, "1 KB contiguous per instruction");
    }
    return 0;
}
