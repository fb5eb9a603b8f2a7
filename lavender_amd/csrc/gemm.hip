// bf16 MFMA GEMM for gfx950 with fused epilogues.  One kernel template covers the three operand
// layouts the LAVENDER step needs:
//   NT  C[M,N] = A[M,K] . B[N,K]^T      forward linears (x . W^T)
//   NN  C[M,N] = A[M,K] . B[K,N]        input gradients (dY . W)
//   TN  C[M,N] = A[K,M]^T . B[K,N]      weight gradients (dY^T . X), split over the contraction
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16.
// K-contiguous operands sit in LDS as [row][64] with an XOR swizzle of the 16-B slot (conflict-free
// ds_read_b128); contraction-strided operands sit as [16-col subtile][k][16] and are read with
// ds_read_b64_tr_b16 (hardware transpose).  The accumulator tile is staged through LDS (fp32) so the
// epilogue (bias / GELU / dropout / drop-path scale / residual / column sums) runs on 8-wide row
// chunks with 16-byte global accesses.
#include "common.h"
#include "../../include/lavender_hip.h"

#define BM 128
#define BN 128
#define BKT 64
#define NT_ 256
#define CSTRIDE 132
#define GEMM_LDS_BYTES (BM * CSTRIDE * 4)

struct GemmArgs {
    const bf16_t* A; const bf16_t* B; void* C;
    long lda, ldb, ldc;
    int M, N, K;
    int k_per_split;
    lav_gemm_epilogue e;
    uint32_t drop_thresh;
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ s16x4 tr_read(const char* lds_base, int off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_base + off));
}

// ---- global -> registers (one 128x64 or 64x128 operand tile = 4 x 16 B per thread) -------------
template <bool KCONTIG>
__device__ __forceinline__ void tile_load(uint4 (&r)[4], const bf16_t* __restrict__ P, long ld, int o0, int O,
                                          int k0, int kend, int tid, const float* keep, int keep_rpg) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int c = tid + NT_ * j;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (KCONTIG) {
            int row = c >> 3, slot = c & 7;
            int go = o0 + row, gk = k0 + slot * 8;
            if (go < O && gk < kend) v = *(const uint4*)(P + (long)go * ld + gk);
        } else {
            int krow = c >> 4, n8 = c & 15;
            int gk = k0 + krow, go = o0 + n8 * 8;
            bool ok = gk < kend && go < O;
            if (ok && keep) ok = keep[gk / keep_rpg] != 0.f;
            if (ok) v = *(const uint4*)(P + (long)gk * ld + go);
        }
        r[j] = v;
    }
}

template <bool KCONTIG>
__device__ __forceinline__ void tile_store(const uint4 (&r)[4], char* lds, int tid) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int c = tid + NT_ * j;
        int off;
        if (KCONTIG) {
            int row = c >> 3, slot = c & 7;
            off = row * 128 + ((slot ^ (row & 7)) << 4);
        } else {
            int krow = c >> 4, n8 = c & 15;
            int sub = n8 >> 1, half = n8 & 1, ks = krow >> 5, kk = krow & 31;
            int pos = (kk & 3) | (((kk >> 3) & 3) << 2) | (((kk >> 2) & 1) << 4);
            off = sub * 2048 + ks * 1024 + ((pos ^ (sub & 3)) << 5) + half * 16;
        }
        *(uint4*)(lds + off) = r[j];
    }
}

// ---- LDS -> MFMA fragment: 16 rows (or cols) x 32 k; lane (i = l&15, g = l>>4) gets k = 8g..8g+7 ----
template <bool KCONTIG>
__device__ __forceinline__ bf16x8 frag_read(const char* lds, int t16, int ks, int lane) {
    if (KCONTIG) {
        int row = t16 * 16 + (lane & 15);
        int slot = ks * 4 + (lane >> 4);
        return *(const bf16x8*)(lds + row * 128 + ((slot ^ (row & 7)) << 4));
    } else {
        int i = lane & 15, g = lane >> 4, r = i >> 2, c = i & 3;
        int base = t16 * 2048 + ks * 1024 + c * 8;
        int sw = t16 & 3;
        s16x4 lo = tr_read(lds, base + (((r + 4 * g) ^ sw) << 5));
        s16x4 hi = tr_read(lds, base + (((16 + r + 4 * g) ^ sw) << 5));
        union { struct { s16x4 a, b; } s; bf16x8 v; } u;
        u.s.a = lo; u.s.b = hi;
        return u.v;
    }
}

// KG = number of 4-wave groups per block.  KG == 2 (weight gradients): the two groups walk alternate k-tiles of
// the SAME output tile with private LDS stages and merge their accumulators through LDS -- twice the waves per
// CU without doubling the number of fp32-atomic output tiles (the epilogue atomics are what caps split-K).
template <bool AK, bool BK, int KG>
__global__ __launch_bounds__(NT_ * KG) void gemm_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const int grp = KG == 1 ? 0 : (threadIdx.x >> 8);
    char* smem = smem_all + grp * 65536;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap: each XCD (block b -> XCD b % 8) walks a contiguous run of tiles, n fastest
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nk = (kend - kbeg + BKT - 1) / BKT;


    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // TN only: bias gradient = sum over the contraction of A's columns, computed on the matrix cores against a
    // ones fragment by the wn==0 waves of the n0==0 blocks (every column of the product equals the row sum)
    const bool do_rowsum = !AK && g.e.rowsum_a != nullptr && n0 == 0 && wn == 0;
    f32x4 acc1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc1[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
    {
        union { uint4 u; bf16x8 b; } o; o.u = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u); ones = o.b;
    }

    uint4 ra[4], rb[4];
    const float* keep = g.e.k_keep;
    const int nit = (nk + KG - 1) / KG;                    // iterations per group (tiles kt = it*KG + grp; overrun tiles read as zeros)
    if (nk > 0) {
        tile_load<AK>(ra, g.A, g.lda, m0, g.M, kbeg + grp * BKT, kend, tid, keep, g.e.k_rows_per_group);
        tile_load<BK>(rb, g.B, g.ldb, n0, g.N, kbeg + grp * BKT, kend, tid, keep, g.e.k_rows_per_group);
        tile_store<AK>(ra, smem, tid);
        tile_store<BK>(rb, smem + 16384, tid);
    }
    __syncthreads();
    for (int it = 0; it < nit; ++it) {
        const int cur = it & 1;
        char* la = smem + cur * 32768;
        char* lb = la + 16384;
        if (it + 1 < nit) {
            const int k0n = kbeg + ((it + 1) * KG + grp) * BKT;
            tile_load<AK>(ra, g.A, g.lda, m0, g.M, k0n, kend, tid, keep, g.e.k_rows_per_group);
            tile_load<BK>(rb, g.B, g.ldb, n0, g.N, k0n, kend, tid, keep, g.e.k_rows_per_group);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = frag_read<AK>(la, wm * 4 + i, ks, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = frag_read<BK>(lb, wn * 4 + j, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            if (!AK && do_rowsum) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc1[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, acc1[i], 0, 0, 0);
            }
        }
        if (it + 1 < nit) {
            tile_store<AK>(ra, smem + (cur ^ 1) * 32768, tid);
            tile_store<BK>(rb, smem + (cur ^ 1) * 32768 + 16384, tid);
        }
        __syncthreads();
    }

    if (!AK && do_rowsum && (lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
                if (row < g.M) atomicAdd(g.e.rowsum_a + row, acc1[i][r] * g.e.alpha);
            }
    }
    // ---- stage the accumulator tile through LDS (fp32, row stride 132 floats) ----------------------
    float* cl = (float*)smem_all;
    if (KG == 1 || grp == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int row = wm * 64 + i * 16 + (lane >> 4) * 4 + r;
                    int col = wn * 64 + j * 16 + (lane & 15);
                    cl[row * CSTRIDE + col] = acc[i][j][r];
                }
    }
    __syncthreads();
    if (KG == 2) {
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int row = wm * 64 + i * 16 + (lane >> 4) * 4 + r;
                        int col = wn * 64 + j * 16 + (lane & 15);
                        cl[row * CSTRIDE + col] += acc[i][j][r];
                    }
        }
        __syncthreads();
    }

    const lav_gemm_epilogue& e = g.e;
    const int cc = tid & 15;
    const int gcol = n0 + cc * 8;
    float csum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float bias[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int ncols = min(8, g.N - gcol);              // <=0: chunk outside
    if (e.bias && ncols > 0) {
#pragma unroll
        for (int x = 0; x < 8; ++x)
            if (x < ncols) bias[x] = e.bias[gcol + x];
    }
    const bool full = ncols == 8;
    const int etid = threadIdx.x;                           // all KG*256 threads share the epilogue
#pragma unroll 1
    for (int j = 0; j < 8 / KG; ++j) {
        const int row = (etid >> 4) + 16 * KG * j;
        const int grow = m0 + row;
        if (grow >= g.M || ncols <= 0) continue;
        float v[8];
        *(float4*)&v[0] = *(const float4*)&cl[row * CSTRIDE + cc * 8];
        *(float4*)&v[4] = *(const float4*)&cl[row * CSTRIDE + cc * 8 + 4];
#pragma unroll
        for (int x = 0; x < 8; ++x) v[x] = v[x] * e.alpha + bias[x];
        if (e.preact) {
            bf16_t* p = (bf16_t*)e.preact + (long)grow * e.ldp + gcol;
            if (full) *(uint4*)p = pack8(v);
            else for (int x = 0; x < ncols; ++x) p[x] = f2bf(v[x]);
        }
        if (e.act == 1) {
#pragma unroll
            for (int x = 0; x < 8; ++x) v[x] = gelu_f(v[x]);
        }
        if (e.gelu_in) {
            const bf16_t* p = (const bf16_t*)e.gelu_in + (long)grow * e.ldg + gcol;
            float h[8];
            if (full) { uint4 u = *(const uint4*)p; unpack8(u, h); }
            else for (int x = 0; x < 8; ++x) h[x] = x < ncols ? bf2f(p[x]) : 0.f;
#pragma unroll
            for (int x = 0; x < 8; ++x) v[x] *= gelu_grad_f(h[x]);
        }
        if (e.dropout_p > 0.f) {
            const float inv = 1.0f / (1.0f - e.dropout_p);
#pragma unroll
            for (int x = 0; x < 8; ++x)
                v[x] = lav_keep(e.seed, (uint64_t)grow * (uint64_t)g.N + (uint64_t)(gcol + x), g.drop_thresh) ? v[x] * inv : 0.f;
        }
        if (e.row_scale) {
            const float s = e.row_scale[grow / e.rows_per_group];
#pragma unroll
            for (int x = 0; x < 8; ++x) v[x] *= s;
        }
        if (e.residual) {
            const bf16_t* p = (const bf16_t*)e.residual + (long)grow * e.ldr + gcol;
            float h[8];
            if (full) { uint4 u = *(const uint4*)p; unpack8(u, h); }
            else for (int x = 0; x < 8; ++x) h[x] = x < ncols ? bf2f(p[x]) : 0.f;
#pragma unroll
            for (int x = 0; x < 8; ++x) v[x] += h[x];
        }
        if (e.colsum) {
#pragma unroll
            for (int x = 0; x < 8; ++x) csum[x] += v[x];
        }
        if (e.out_mode == 0) {
            bf16_t* p = (bf16_t*)g.C + (long)grow * g.ldc + gcol;
            if (full) *(uint4*)p = pack8(v);
            else for (int x = 0; x < ncols; ++x) p[x] = f2bf(v[x]);
        } else if (e.out_mode == 1) {
            float* p = (float*)g.C + (long)grow * g.ldc + gcol;
            if (full) { *(float4*)p = *(float4*)&v[0]; *(float4*)(p + 4) = *(float4*)&v[4]; }
            else for (int x = 0; x < ncols; ++x) p[x] = v[x];
        } else {
            float* p = (float*)g.C + (long)grow * g.ldc + gcol;
            for (int x = 0; x < ncols; ++x) atomicAdd(p + x, v[x]);
        }
    }
    if (e.colsum) {
        __syncthreads();
        float* red = (float*)smem_all;                   // [16*KG][128]
#pragma unroll
        for (int x = 0; x < 8; ++x) red[(etid >> 4) * 128 + cc * 8 + x] = csum[x];
        __syncthreads();
        if (etid < 128 && n0 + etid < g.N) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16 * KG; ++r) s += red[r * 128 + etid];
            atomicAdd(e.colsum + n0 + etid, s);
        }
    }
}

extern "C" int lav_gemm_bf16(void* stream, int layout, int M, int N, int K, const void* A, long lda, const void* B,
                             long ldb, void* C, long ldc, const lav_gemm_epilogue* epi, int splits) {
    LAV_REQUIRE(M > 0 && N > 0 && K > 0, "lav_gemm_bf16: empty problem M=%d N=%d K=%d", M, N, K);
    LAV_REQUIRE(layout >= 0 && layout <= 2, "lav_gemm_bf16: bad layout %d", layout);
    LAV_REQUIRE(A && B && C, "lav_gemm_bf16: null operand");
    LAV_REQUIRE((lda % 8) == 0 && (ldb % 8) == 0, "lav_gemm_bf16: lda/ldb must be multiples of 8 (16-byte rows)");
    LAV_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0,
                "lav_gemm_bf16: operands must be 16-byte aligned");
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    if (epi) g.e = *epi; else { g.e.alpha = 1.f; }
    if (g.e.alpha == 0.f) g.e.alpha = 1.f;
    if (splits < 1) splits = 1;
    LAV_REQUIRE(!g.e.rowsum_a || layout == 2, "lav_gemm_bf16: rowsum_a is only defined for layout 2 (TN)");
    LAV_REQUIRE(splits == 1 || g.e.out_mode == 2, "lav_gemm_bf16: split-K needs out_mode=2 (fp32 atomic accumulate)");
    LAV_REQUIRE(g.e.out_mode != 0 || (ldc % 8) == 0, "lav_gemm_bf16: bf16 output needs ldc %% 8 == 0");
    LAV_REQUIRE(g.e.out_mode == 0 || (ldc % 4) == 0, "lav_gemm_bf16: fp32 output needs ldc %% 4 == 0");
    int kps = ((K + splits - 1) / splits + BKT - 1) / BKT * BKT;
    g.k_per_split = kps;
    splits = (K + kps - 1) / kps;
    g.drop_thresh = lav_drop_thresh(g.e.dropout_p);
    int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    dim3 grid(tiles, 1, splits), block(NT_);
    hipStream_t s = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm_kernel<true, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
        hipFuncSetAttribute((const void*)gemm_kernel<true, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
        hipFuncSetAttribute((const void*)gemm_kernel<false, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipGetLastError();
        attr_set = true;
    }
    if (layout == 0) hipLaunchKernelGGL((gemm_kernel<true, true, 1>), grid, block, GEMM_LDS_BYTES, s, g);
    else if (layout == 1) hipLaunchKernelGGL((gemm_kernel<true, false, 1>), grid, block, GEMM_LDS_BYTES, s, g);
    else hipLaunchKernelGGL((gemm_kernel<false, false, 2>), grid, dim3(NT_ * 2), 131072, s, g);
    return lav_check_launch("lav_gemm_bf16");
}
