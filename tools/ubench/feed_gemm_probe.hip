// Microbenchmark 2: the operand feed of the NT GEMM (LDS-DMA only, no MFMA, no LDS reads) with the GEMM's own tile mapping:
// M x N output in 256 x 256 tiles, XCD-aware contiguous runs, A panel (256 rows x K) and B panel per tile, 128 B per row per
// k-step.  Prefetch depth as a template parameter: stages of SBYTES in flight.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// BK: k elements per stage (64 -> 64 KB stage, 32 -> 32 KB); NS: stages in LDS; AHEAD: stages issued ahead of the consumer
template <int BK, int NS, int AHEAD>
__global__ __launch_bounds__(512) void feed(const char* A, const char* B, int M, int N, int K, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = N / 256, tiles_m = (M + 255) / 256, nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    { int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3; bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
    const int m0 = (bid / tiles_n) * 256, n0 = (bid % tiles_n) * 256;
    const long pitch = (long)K * 2;
    constexpr int SB = 512 * BK * 2;                      // stage bytes (A 256 rows + B 256 rows)
    constexpr int PIECES = SB / 1024 / 8;                 // per wave
    constexpr int RPP = 1024 / (BK * 2);                  // rows per 1 KB piece
    constexpr int CPR = BK * 2 / 16;                      // 16 B chunks per row
    const int nk = K / BK;
    auto issue = [&](int kt) {
        char* st = smem + (kt % NS) * SB;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int p = wave * PIECES + i;              // 0 .. SB/1024
            const int op = p >= SB / 2048;
            const int pr = op ? p - SB / 2048 : p;
            int row = pr * RPP + lane / CPR;
            const char* base = op ? B : A;
            int grow = (op ? n0 : m0) + row;
            if (!op && grow >= M) grow = M - 1;
            const char* src = base + grow * pitch + (long)kt * BK * 2 + (lane % CPR) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(st + p * 1024), 16, 0, 0);
        }
    };
    for (int s = 0; s < AHEAD && s < nk; ++s) issue(s);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + AHEAD < nk) {
            issue(kt + AHEAD);
            // stage kt must have landed: AHEAD stages stay in flight
            if (AHEAD * PIECES == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (AHEAD * PIECES == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (AHEAD * PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (AHEAD * PIECES == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (AHEAD * PIECES == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    __syncthreads();
    int acc = *(int*)(smem + tid * 4);
    if (acc == 0x12345678) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 45120, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
    char *A, *B; int* sink;
    CHECK(hipMalloc(&A, (long)M * K * 2)); CHECK(hipMalloc(&B, (long)N * K * 2)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(A, 1, (long)M * K * 2)); CHECK(hipMemset(B, 1, (long)N * K * 2));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nwg = ((M + 255) / 256) * (N / 256);
    printf("M %d N %d K %d: %d tiles, %.1f tiles per CU\n", M, N, K, nwg, nwg / 256.0);
    auto run = [&](auto kern, int lds, const char* name) {
        CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), lds, 0, (const char*)A, (const char*)B, M, N, K, sink);
            hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)nwg * 512 * K * 2;
            if (rep == 2) printf("%-52s %8.1f us  %6.1f GB/s per CU  chip %5.2f TB/s  (MFMA time of this GEMM at peak: %.1f us)\n", name, ms * 1e3,
                                 bytes / 256 / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12, 2.0 * M * N * K / 2.5e15 * 1e6);
        }
    };
    run(feed<64, 2, 1>, 131072, "64 KB stages, 2 in LDS, 1 ahead (shipped kernel)");
    run(feed<32, 4, 1>, 131072, "32 KB stages, 4 in LDS, 1 ahead");
    run(feed<32, 4, 2>, 131072, "32 KB stages, 4 in LDS, 2 ahead");
    run(feed<32, 4, 3>, 131072, "32 KB stages, 4 in LDS, 3 ahead");
    run(feed<64, 2, 2>, 131072, "64 KB stages, 2 in LDS, 2 ahead (no consumer: upper bound)");
    return 0;
}
