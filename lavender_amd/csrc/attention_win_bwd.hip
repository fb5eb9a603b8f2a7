// Fused backward of the shifted-window attention (window volume N <= 256, head_dim 32): dQ, dK, dV and the relative-
// position-bias gradient from ONE pass over the window -- S, P, dP and dS are computed once per (window, head).
//
// Why: the two-pass backward (win_dq_p + win_dkv_p, attention_win.hip) reads q/k/v/dO twice, redoes the softmax side twice
// and keeps the 65 536 dS sums of the bias gradient next to a second set of accumulators -- the dQ pass spills (rocprofv3:
// 571 MB fetched + 234 MB written per launch against ~210 + 41 MB algorithmic) and the pair costs 10 ms of an 82 ms step.
// The op is HBM-bound (122 flop/B at head_dim 32, ridge 312): what matters is touching every byte once.
//
// One 256-thread workgroup (4 waves, ONE per SIMD, so each may use the whole 512-register file) owns a (head, window
// position) and walks the batch.  Wave w owns key tiles {w, w+4} (their K/V rows as register-resident B operands, dK/dV
// accumulators) AND query tiles {w, w+4} (dQ accumulators).  Scores are computed queries x keys, so a lane holds ONE key:
//   step s = 0..7, for each own key tile kt: query tile qt = (kt + s) & 7
//     S = Q_qt K_kt^T, dP = dO_qt V_kt^T (A operands from LDS);  P = exp2(S sc + bias - lse_q);  dS = P (dP - delta_q)
//     dV_kt += dO_qt^T P,  dK_kt += Q_qt^T dS (transposing LDS reads),  bias-gradient sums += dS (plain register adds)
//     dS tile -> LDS slot [qt] as bf16 [key][query]                       (each query tile is written by exactly one wave)
//   barrier;  owner of query tile qt:  dQ_qt += K_kt'^T dS^T  with the slot read back through a transposing read.
// The cross-wave reduction over keys therefore happens inside the MFMA accumulators of the query owner; nothing is
// reduced with atomics or fp32 LDS traffic.  Inputs of window b+1 (Q, dO, K: 48 KB) stream into the other LDS buffer
// with direct-to-LDS loads spread over the steps of window b.
#include "attn_common.h"

#define LOG2E 1.4426950408889634f
#define HD 32

namespace {

constexpr int FB_IN = 49152;              // one input buffer: Q | dO | K, 256 rows x 64 B each
constexpr int FB_SL = 16384;              // one slot buffer: 8 dS tiles of 2 KB
constexpr int FB_OFF_SL = 2 * FB_IN;
constexpr int FB_OFF_LD = FB_OFF_SL + 2 * FB_SL;
constexpr int FB_OFF_TAB = FB_OFF_LD + 2 * 2048;
constexpr int FB_LDS = FB_OFF_TAB + 2048;

struct WinGeoF { int head, ws, b0, b1, type; };

__device__ __forceinline__ bool win_geo_f(const AttnArgs& a, int bsplit, WinGeoF& g) {
    int wg = blockIdx.x;
    const int bs = wg % bsplit; wg /= bsplit;
    g.ws = wg % a.nWs; g.head = wg / a.nWs;
    const int per = (a.d.B + bsplit - 1) / bsplit;
    g.b0 = bs * per; g.b1 = min(a.d.B, g.b0 + per);
    g.type = a.d.win_type[g.ws];
    return g.b0 < g.b1;
}

// A-operand fragment (32 rows = the 32 channels, 16 k = rows row0 .. row0+15 of the tile) from a [row][32 ch] bf16 tile with
// 64-byte rows whose 16-byte slots are XOR-swizzled like krow_off<32>: transposing reads with per-lane swizzled addresses
// (an 8-byte element never straddles a slot), so ONE LDS image serves the row-major A reads and these.
__device__ __forceinline__ bf16x8 tr_frag_swz(const char* tile, int row0, int lane) {
    const int i = lane & 15, dhalf = (lane >> 4) & 1, hi = lane >> 5;
    const int r = i >> 2, c = i & 3;
    const int dcol = 16 * dhalf + 4 * c;
    const int slot = dcol >> 3, sub = (dcol & 7) * 2;
    const int ra = row0 + 4 * hi + r, rb = ra + 8;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tile + krow_off<HD>(ra, slot) + sub));
    s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tile + krow_off<HD>(rb, slot) + sub));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi4;
    return u.v;
}

// dS slot tile: [32 keys][32 queries] bf16, 64-byte rows, 8-byte unit u of row j stored at u ^ (j & 7)
__device__ __forceinline__ int slot_off(int key, int unit) { return key * 64 + ((unit ^ (key & 7)) << 3); }

__device__ __forceinline__ bf16x8 tr_frag_slot(const char* slot, int key0, int lane) {
    const int i = lane & 15, dhalf = (lane >> 4) & 1, hi = lane >> 5;
    const int r = i >> 2, c = i & 3;
    const int unit = 4 * dhalf + c;
    const int ka = key0 + 4 * hi + r, kb = ka + 8;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(slot + slot_off(ka, unit)));
    s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(slot + slot_off(kb, unit)));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi4;
    return u.v;
}

// direct-to-LDS load as inline asm (per-lane 64-bit address): hidden from hipcc's scoreboard, which would otherwise drain
// every outstanding LDS-DMA (s_waitcnt vmcnt(0)) in front of each transposing read
__device__ __forceinline__ void dma16(unsigned lds_dst, const void* src) {
    unsigned keep_m0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep_m0) : "v"(src), "s"(lds_dst) : "memory");
}

}  // namespace

template <bool DBIAS>
__global__ __launch_bounds__(256) void win_bwd_fused(AttnArgs a, int bsplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    WinGeoF g;
    if (!win_geo_f(a, bsplit, g)) return;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, C = a.C, ld = 3 * C;
    const bf16_t* qkv = a.qkv;
    const float sc = a.d.scale * LOG2E;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    int* kcode = (int*)(smem + FB_OFF_TAB);
    int* srel_l = kcode + 256;
    {
        const int i = tid;
        kcode[i] = i < N ? (i / (a.d.cfg_ww * a.d.cfg_wh)) * a.cstride_d + ((i / a.d.cfg_ww) % a.d.cfg_wh) * a.cstride_h + (i % a.d.cfg_ww) : 0;
        const int rel = a.d.tok_table[g.ws * 256 + i];
        srel_l[i] = rel < 0 ? 0 : rel;       // padded slots read a valid row; their scores are masked (bias table / lse = +inf)
    }
    const int qrel_row = a.d.tok_table[g.ws * 256 + tid];           // this thread's query row (delta / lse producer)
    const bool row_ok = qrel_row >= 0;
    // own key rows (B operands) and own query rows (dQ stores)
    int krel[2]; bool k_ok[2];                                   // own key rows == own query rows (tiles wave, wave + 4)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        krel[h] = a.d.tok_table[g.ws * 256 + (wave + 4 * h) * 32 + j];
        k_ok[h] = krel[h] >= 0;
    }
    const int ntile = (N + 31) >> 5;
    const bf16_t* combT0 = (const bf16_t*)a.d.combT + ((long)(g.type * a.d.heads + g.head) * 64 * 64 + lane) * 16;
    __syncthreads();

    // ---- direct-to-LDS stream of a window's Q | dO | K rows: 48 one-KB pieces, 12 per wave, piece = 16 rows x 64 B ----
    auto dma_piece = [&](int b, int buf, int p) {                  // p = 0..11 (this wave's pieces)
        const int gp = wave * 12 + p;                              // 0..47
        const int ten = gp >> 4, piece = gp & 15;
        const int row = piece * 16 + (lane >> 2);
        const int slot = (lane & 3) ^ ((lane >> 4) & 3);           // logical 16-byte slot landing on physical slot lane & 3
        const long tok = (long)b * a.tps + srel_l[row];
        const bf16_t* src = ten == 1 ? a.dout + tok * C + g.head * HD + slot * 8
                                     : qkv + tok * ld + (ten == 2 ? C : 0) + g.head * HD + slot * 8;
        dma16(__builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * FB_IN + ten * 16384 + piece * 1024)), src);
    };
    // register-resident per-window rows: V (and K) of the own key tiles as B operands; O + lse of this thread's query row
    bf16x8 vf[2][2];
    uint4 o_nx[4]; float lse_nx;
    uint4 vn[2][2];
    auto load_rows = [&](int b) {
        const long base = (long)b * a.tps;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                vn[h][ks] = make_uint4(0, 0, 0, 0);
                if (k_ok[h]) vn[h][ks] = *(const uint4*)(qkv + (base + krel[h]) * ld + 2 * C + g.head * HD + ks * 16 + 8 * hi);
            }
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            o_nx[c4] = make_uint4(0, 0, 0, 0);
            if (row_ok) o_nx[c4] = *(const uint4*)(a.out + (base + qrel_row) * C + g.head * HD + c4 * 8);
        }
        lse_nx = row_ok ? a.lse[((long)(b * a.nWs + g.ws) * a.d.heads + g.head) * a.Npad + tid] : INFINITY;
    };

    // bias-gradient sums: dsa[h][s] = sum over windows of dS(queries of tile (kt + s) & 7, keys of own tile kt = wave + 4 h)
    f32x16 dsa[2][8];
    if (DBIAS) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) dsa[h][s][r] = 0.f;
    }

    // prologue: window b0 into buffer 0
    for (int p = 0; p < 12; ++p) dma_piece(g.b0, 0, p);
    load_rows(g.b0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int b = g.b0; b < g.b1; ++b) {
        const int cur = (b - g.b0) & 1;
        const char* Qs = smem + cur * FB_IN;
        const char* Gs = Qs + 16384;
        const char* Ks = Qs + 32768;
        float* qlse = (float*)(smem + FB_OFF_LD + cur * 2048);
        float* qdl = qlse + 256;
        const bool more = b + 1 < g.b1;
        // ---- per-window prologue: delta[q] = dO[q] . O[q], lse[q] -> LDS; own K / V rows -> B operands ----
        {
            float d = 0.f;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const uint4 gv = *(const uint4*)(Gs + krow_off<HD>(tid, c4));
                float gf[8], of[8];
                unpack8(gv, gf); unpack8(o_nx[c4], of);
#pragma unroll
                for (int e = 0; e < 8; ++e) d += gf[e] * of[e];
            }
            qlse[tid] = lse_nx;
            qdl[tid] = d;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) vf[h][ks] = as_bf16x8(vn[h][ks]);
        }
        f32x16 dq[2], dk[2], dv[2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dq[h][r] = 0.f; dk[h][r] = 0.f; dv[h][r] = 0.f; }
        __syncthreads();

#pragma unroll
        for (int s = 0; s < 8; ++s) {
            // opaque copy of the wave index per step: everything derived from (wave, s) -- tile offsets, table pointers -- is
            // recomputed here (a few scalar ops) instead of being hoisted out of the window loop for all 16 (step, tile) pairs,
            // which cost > 100 scalar and > 80 vector registers of spills
            int wv = wave;
            asm volatile("" : "+s"(wv));
            char* SL = smem + FB_OFF_SL + (s & 1) * FB_SL;
            // ---- phase 1: the two own key tiles against query tiles (kt + s) & 7 ----
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kt = wv + 4 * h;
                const int qt = (kt + s) & 7;
                if (kt >= ntile || qt >= ntile) continue;          // wave-uniform
                const int q0 = qt * 32;
                const bf16_t* cp = combT0 + (long)(kt * 8 + qt) * 1024;
                const uint4 c0 = *(const uint4*)cp, c1 = *(const uint4*)(cp + 8);
                f32x16 sv, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sv[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 qa = *(const bf16x8*)(Qs + krow_off<HD>(q0 + j, ks * 2 + hi));
                    const bf16x8 kb_ = *(const bf16x8*)(Ks + krow_off<HD>(kt * 32 + j, ks * 2 + hi));   // padded keys: masked by the bias table
                    sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kb_, sv, 0, 0, 0);
                    const bf16x8 ga = *(const bf16x8*)(Gs + krow_off<HD>(q0 + j, ks * 2 + hi));
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, vf[h][ks], dp, 0, 0, 0);
                }
                float c[16], pd[16], ds[16];
                unpack8(c0, c); unpack8(c1, c + 8);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int qb = q0 + 8 * r4 + 4 * hi;
                    const float4 l4 = *(const float4*)(qlse + qb);
                    const float4 d4 = *(const float4*)(qdl + qb);
                    const float ls[4] = {l4.x, l4.y, l4.z, l4.w};
                    const float dls[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = r4 * 4 + e;
                        // padded keys: the bias table holds -30000 there; padded queries: lse = +inf
                        const float p = fast_exp2(fmaf(sv[r], sc, c[r]) - ls[e]);
                        pd[r] = p;
                        ds[r] = p * (dp[r] - dls[e]);
                    }
                }
                if (DBIAS) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) dsa[h][s][r] += ds[r];
                }
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const bf16x8 pf = pack_frag(pd + 8 * sl), dsf = pack_frag(ds + 8 * sl);
                    const bf16x8 gt = tr_frag_swz(Gs, q0 + 16 * sl, lane);
                    dv[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt, pf, dv[h], 0, 0, 0);
                    const bf16x8 qt_ = tr_frag_swz(Qs, q0 + 16 * sl, lane);
                    dk[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt_, dsf, dk[h], 0, 0, 0);
                }
                // dS tile -> slot [qt] as [key][query]: this lane's key row, queries 8 m + 4 hi .. + 3 per 8-byte unit
                char* st = SL + qt * 2048;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    uint2 w;
                    w.x = pack2(ds[4 * m + 0], ds[4 * m + 1]);
                    w.y = pack2(ds[4 * m + 2], ds[4 * m + 3]);
                    *(uint2*)(st + slot_off(j, 2 * m + hi)) = w;
                }
                __builtin_amdgcn_sched_barrier(0);        // keep each tile pair's loads next to their use (register pressure)
            }
            // the next window's inputs trickle in: two pieces per step (steps 0..5)
            if (more && s < 6) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no compiler-visible load is outstanding across an invisible one
                dma_piece(b + 1, cur ^ 1, 2 * s);
                dma_piece(b + 1, cur ^ 1, 2 * s + 1);
            }
            if (more && s == 6) load_rows(b + 1);         // register rows of the next window (V of the own keys, O + lse of the own query row)
            __syncthreads();
            // ---- phase 2: own query tiles: dQ_qt += K_kt'^T dS^T, the tile written this step by the owner of kt' = (qt - s) & 7 ----
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int qt = wv + 4 * h;
                const int kt = (qt - s) & 7;
                if (qt >= ntile || kt >= ntile) continue;
                const char* st = SL + qt * 2048;
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const bf16x8 ktf = tr_frag_swz(Ks, kt * 32 + 16 * sl, lane);
                    const bf16x8 dst = tr_frag_slot(st, 16 * sl, lane);
                    dq[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dst, dq[h], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- stores: dQ rows of the own query tiles, dK / dV rows of the own key tiles ----
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (k_ok[h]) {
                bf16_t* op = a.dqkv + ((long)b * a.tps + krel[h]) * ld + g.head * HD;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    uint2 w;
                    w.x = pack2(dq[h][r4 * 4 + 0] * a.d.scale, dq[h][r4 * 4 + 1] * a.d.scale);
                    w.y = pack2(dq[h][r4 * 4 + 2] * a.d.scale, dq[h][r4 * 4 + 3] * a.d.scale);
                    *(uint2*)(op + 8 * r4 + 4 * hi) = w;
                }
            }
            if (k_ok[h]) {
                bf16_t* op = a.dqkv + ((long)b * a.tps + krel[h]) * ld + g.head * HD;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int d = 8 * r4 + 4 * hi;
                    uint2 w;
                    w.x = pack2(dk[h][r4 * 4 + 0] * a.d.scale, dk[h][r4 * 4 + 1] * a.d.scale);
                    w.y = pack2(dk[h][r4 * 4 + 2] * a.d.scale, dk[h][r4 * 4 + 3] * a.d.scale);
                    *(uint2*)(op + C + d) = w;
                    w.x = pack2(dv[h][r4 * 4 + 0], dv[h][r4 * 4 + 1]);
                    w.y = pack2(dv[h][r4 * 4 + 2], dv[h][r4 * 4 + 3]);
                    *(uint2*)(op + 2 * C + d) = w;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // next window's pieces + register rows have landed
        __syncthreads();
    }

    // ---- bias-gradient flush: registers -> wave-private LDS table (index = code(q) - code(k) + const) -> global ----
    // A lane holds ONE key; for a fixed accumulator register the 32 lanes of a half-wave share the query and have distinct
    // keys, so their 32 table indices are distinct: a plain read-modify-write per half is race-free.
    if (DBIAS) {
        float* dtbl_all = (float*)smem;                   // the input buffers are free now
        for (int r = tid; r < 4 * a.tbl_rows; r += 256) dtbl_all[r] = 0.f;
        __syncthreads();
        float* dtbl = dtbl_all + wave * a.tbl_rows;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kt = wave + 4 * h;
            if (kt >= ntile) continue;
            const int key = kt * 32 + j;
            const int kc = kcode[key];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int qt = (kt + s) & 7;
                if (qt >= ntile) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = qt * 32 + tile_row(r, hi);
                    const int idx = kcode[q] + a.tbl_const - kc;
                    const bool ok = key < N && q < N;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        if (hi == hh && ok) dtbl[idx] += dsa[h][s][r];
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
        }
        __syncthreads();
        for (int r = tid; r < a.tbl_rows; r += 256) {
            const float v = dtbl_all[r] + dtbl_all[a.tbl_rows + r] + dtbl_all[2 * a.tbl_rows + r] + dtbl_all[3 * a.tbl_rows + r];
            if (v != 0.f) atomicAdd(a.dbias + (long)r * a.d.heads + g.head, v);
        }
    }
}

int win_fused_bwd(void* stream, const AttnArgs& a, int bsplit) {
    size_t lds = FB_LDS;
    const size_t flush = (size_t)a.tbl_rows * 16;
    if (flush > lds) lds = flush;
    if (a.dbias) {
        (void)hipFuncSetAttribute((const void*)win_bwd_fused<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipGetLastError();
        hipLaunchKernelGGL((win_bwd_fused<true>), dim3(a.d.heads * a.nWs * bsplit), dim3(256), lds, (hipStream_t)stream, a, bsplit);
    } else {
        (void)hipFuncSetAttribute((const void*)win_bwd_fused<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipGetLastError();
        hipLaunchKernelGGL((win_bwd_fused<false>), dim3(a.d.heads * a.nWs * bsplit), dim3(256), lds, (hipStream_t)stream, a, bsplit);
    }
    return lav_check_launch("lav_attention_bwd(window, fused)");
}
