// Persistent shifted-window attention for gfx950 (window volume N <= 256, head_dim 32): the Swin hot loop.
//
// One 512-thread workgroup (8 waves, two per SIMD) per CU owns a fixed (head, window position) pair and walks the
// batch: the window position fixes the token-row offsets and the mask type, so
//   * token rows come from a per-geometry table read ONCE per workgroup (no div/mod chains in the loop),
//   * the (relative-position bias + shift mask + key padding) term of this wave's 32-query strip sits in
//     registers as packed bf16 for the whole kernel (forward) or streams from L2 (backward),
//   * K/V (or Q/dO) of the NEXT window are fetched into registers while the current window is computed and land
//     in the other half of a double-buffered LDS tile: HBM latency never sits on the critical path,
//   * the bias gradient (sum over windows of dS) is accumulated in registers across the whole batch walk and
//     flushed once per workgroup (LDS atomics -> 1521 global atomics), instead of per window.
// Wave w owns 32-row tile w of the 256-slot window (8 tiles).  Scores use the exp2 domain: the bias tables are
// pre-multiplied by log2(e).
#include "attn_common.h"
#include <stdlib.h>

#define LOG2E 1.4426950408889634f
#define HD 32

struct WinGeo {
    int head, ws, b0, b1, type;
};

__device__ __forceinline__ bool win_geo(const AttnArgs& a, int bsplit, WinGeo& g) {
    int wg = blockIdx.x;
    const int bs = wg % bsplit; wg /= bsplit;
    g.ws = wg % a.nWs; g.head = wg / a.nWs;
    const int per = (a.d.B + bsplit - 1) / bsplit;
    g.b0 = bs * per; g.b1 = min(a.d.B, g.b0 + per);
    g.type = a.d.win_type[g.ws];
    return g.b0 < g.b1;
}

__device__ __forceinline__ void unpack16(const uint4& u0, const uint4& u1, float* c) { unpack8(u0, c); unpack8(u1, c + 8); }

// ------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void win_fwd_p(AttnArgs a, int bsplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [2][K 16 KB | V 16 KB]
    WinGeo g;
    if (!win_geo(a, bsplit, g)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hi = lane >> 5;
    const int N = a.N, C = a.C, ld = 3 * C;
    const bf16_t* qkv = a.qkv;
    const float sc = a.d.scale * LOG2E;

    int srow[2], sslot[2], srel[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + 512 * i;
        srow[i] = c >> 2; sslot[i] = c & 3;
        srel[i] = a.d.tok_table[g.ws * 256 + srow[i]];
    }
    const int q = wave * 32 + j;
    const int qrel = a.d.tok_table[g.ws * 256 + q];
    const bool q_ok = qrel >= 0;
    const int ntile = (N + 31) >> 5;

    const bf16_t* comb = (const bf16_t*)a.d.comb + (((long)(g.type * a.d.heads + g.head) * 64 + wave * 8) * 64 + lane) * 16;

    uint4 kr[2], vr[2], qn[2];
    auto load = [&](int b) {
        const long base = (long)b * a.tps;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            kr[i] = make_uint4(0, 0, 0, 0); vr[i] = kr[i];
            if (srel[i] >= 0) {
                const bf16_t* p = qkv + (base + srel[i]) * ld + g.head * HD + sslot[i] * 8;
                kr[i] = *(const uint4*)(p + C);
                vr[i] = *(const uint4*)(p + 2 * C);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qn[ks] = make_uint4(0, 0, 0, 0);
            if (q_ok) qn[ks] = *(const uint4*)(qkv + (base + qrel) * ld + g.head * HD + ks * 16 + 8 * hi);
        }
    };
    auto stash = [&](int buf) {
        char* Ks = smem + buf * 32768;
        char* Vs = Ks + 16384;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(uint4*)(Ks + krow_off<HD>(srow[i], sslot[i])) = kr[i];
            *(uint4*)(Vs + vrow_off<HD>(srow[i], sslot[i])) = vr[i];
        }
    };
    load(g.b0);
    stash(0);
    __syncthreads();

    for (int b = g.b0; b < g.b1; ++b) {
        const int cur = (b - g.b0) & 1;
        const char* Ks = smem + cur * 32768;
        const char* Vs = Ks + 16384;
        const bf16x8 qf[2] = {as_bf16x8(qn[0]), as_bf16x8(qn[1])};
        if (b + 1 < g.b1) load(b + 1);

        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half * 4 >= ntile) break;
            uint4 cb[4][2];                              // this strip's (bias + mask) fragments, L2-resident
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                cb[t][0] = *(const uint4*)(comb + (half * 4 + t) * 1024);
                cb[t][1] = *(const uint4*)(comb + (half * 4 + t) * 1024 + 8);
            }
            f32x16 s[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
                const int k0 = (half * 4 + t) * 32;
                if (k0 >= N) continue;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 kf = *(const bf16x8*)(Ks + krow_off<HD>(k0 + j, ks * 2 + hi));
                    s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[t], 0, 0, 0);
                }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int kt = half * 4 + t;
                if (kt * 32 >= N) continue;
                float c[16];
                unpack16(cb[t][0], cb[t][1], c);
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[t][r] = fmaf(s[t][r], sc, c[r]); mx = fmaxf(mx, s[t][r]); }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = fast_exp2(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] *= alpha;
            m_run = m_new;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k0 = (half * 4 + t) * 32;
                if (k0 >= N) continue;
                float p[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) { p[r] = fast_exp2(s[t][r] - m_new); l_run += p[r]; }
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    bf16x8 pf = pack_frag(p + 8 * sl);
                    bf16x8 vf = tr_frag<HD>(Vs, k0 + 16 * sl, 0, lane);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o, 0, 0, 0);
                }
            }
        }
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv_l = l_tot > 0.f ? 1.f / l_tot : 0.f;
        if (q_ok) {
            const long row = (long)b * a.tps + qrel;
            bf16_t* op = a.o_w + row * C + g.head * HD;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                uint2 w;
                w.x = pack2(o[r4 * 4 + 0] * inv_l, o[r4 * 4 + 1] * inv_l);
                w.y = pack2(o[r4 * 4 + 2] * inv_l, o[r4 * 4 + 3] * inv_l);
                *(uint2*)(op + 8 * r4 + 4 * hi) = w;
            }
            // lse kept in the exp2 domain: log2(sum_k 2^v)
            if (a.lse && hi == 0) a.lse[((long)(b * a.nWs + g.ws) * a.d.heads + g.head) * a.Npad + q] = m_run + log2f(l_tot);
        }
        if (b + 1 < g.b1) stash(cur ^ 1);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------
// backward pass 1: dQ (+ bias-table gradient).  256 threads = 4 waves, ONE per SIMD, so each wave may use the whole
// 512-register file: wave w owns query tiles w and w+4 and keeps sum_windows dS for both strips (2 x 8 tiles x 16
// = 256 accumulator registers) resident across the batch walk.  The accumulation itself runs on the matrix cores:
// dsa += E . dS with E a one-hot (32 x 16) fragment, so it costs no VALU work and the sums live in AGPRs.
// LDS float atomics are avoided entirely (measured: a 64-lane ds_add_f32 retires in ~200 cycles): the final
// flush is a half-wave-masked read-modify-write into a wave-private LDS copy of the table.
// ------------------------------------------------------------------------------------------------------
// NW = 4: one wave per SIMD, two query tiles per wave (512 registers each).  NW = 8: two waves per SIMD, ONE query tile
// per wave (<= 256 registers: 128 for the resident dS sums), so the MFMA -> VALU -> MFMA chains of one wave are
// covered by the other; the two wave groups flush into the same four LDS tables one after the other.
#ifndef WIN_DQ_NOPREFETCH
#define WIN_DQ_NOPREFETCH 1
#endif
template <bool DBIAS, int NW>
__global__ __launch_bounds__(NW * 64) void win_dq_p(AttnArgs a, int bsplit, float* delta_out) {
    constexpr int NQ = 8 / NW;                           // query tiles per wave
    constexpr bool PREFETCH = NW == 4 && !(DBIAS && WIN_DQ_NOPREFETCH);
    constexpr int NTHR = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [2][K(A) 16K | K(tr) 16K | V(A) 16K] + kcode + 4 x dtbl
    WinGeo g;
    if (!win_geo(a, bsplit, g)) return;
    int* kcode = (int*)(smem + 2 * 49152);
    int* srel_l = kcode + 256;
    float* dtbl_all = (float*)(srel_l + 256);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hi = lane >> 5;
    const int N = a.N, C = a.C, ld = 3 * C;
    const bf16_t* qkv = a.qkv;
    const float sc = a.d.scale * LOG2E;

    if (DBIAS) for (int r = tid; r < 4 * a.tbl_rows; r += NTHR) dtbl_all[r] = 0.f;
    if (tid < 256) {
        const int i = tid;
        kcode[i] = i < N ? (i / (a.d.cfg_ww * a.d.cfg_wh)) * a.cstride_d + ((i / a.d.cfg_ww) % a.d.cfg_wh) * a.cstride_h + (i % a.d.cfg_ww) : 0;
        const int rel = a.d.tok_table[g.ws * 256 + i];
        srel_l[i] = rel < 0 ? 0 : rel;      // padded keys read a valid row: their scores are masked to -inf-like by the bias table
    }
    __syncthreads();
    int qrel[NQ]; bool q_ok[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        qrel[qi] = a.d.tok_table[g.ws * 256 + (wave + NW * qi) * 32 + j];
        q_ok[qi] = qrel[qi] >= 0;
    }
    const int ntile = (N + 31) >> 5;
    const bf16_t* comb0 = (const bf16_t*)a.d.comb + ((long)(g.type * a.d.heads + g.head) * 64 * 64 + lane) * 16;

    // one-hot A fragments: E_sl[i][k-slot (hi, e)] = 1 iff i == 16 sl + 8 (e >> 2) + 4 hi + (e & 3)
    bf16x8 onehot[2];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        float e8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) e8[e] = (j == 16 * sl + 8 * (e >> 2) + 4 * hi + (e & 3)) ? 1.f : 0.f;
        onehot[sl] = pack_frag(e8);
    }
    f32x16 dsa[NQ][8];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dsa[qi][kt][r] = 0.f;

    uint4 qn[NQ][2], gn[NQ][2], on[NQ][2];
    float lse_n[NQ];
    // K (two layouts) and V of window b go straight from HBM/L2 into LDS buffer `buf` (global_load_lds, 1 KB per
    // wave-instruction, no staging registers); the swizzle of the ds_read_b128 layout is applied to the SOURCE slot
    auto issue_kv = [&](int b, int buf) {
        const long base = (long)b * a.tps;
        char* B0 = smem + buf * 49152;
#pragma unroll
        for (int i = 0; i < 48 / NW; ++i) {
            const int t = wave * (48 / NW) + i;
            const int cpy = t >> 4, chunk = t & 15;
            const int rel = srel_l[chunk * 16 + (lane >> 2)];
            const int lslot = cpy == 1 ? (lane & 3) : ((lane & 3) ^ ((lane >> 4) & 3));
            const bf16_t* src = qkv + (base + rel) * ld + g.head * HD + lslot * 8 + (cpy == 2 ? 2 * C : C);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(B0 + cpy * 16384 + chunk * 1024), 16, 0, 0);
        }
    };
    auto load = [&](int b) {
        const long base = (long)b * a.tps;
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                qn[qi][ks] = make_uint4(0, 0, 0, 0); gn[qi][ks] = qn[qi][ks]; on[qi][ks] = qn[qi][ks];
                if (q_ok[qi]) {
                    const long row = base + qrel[qi];
                    const int off = g.head * HD + ks * 16 + 8 * hi;
                    qn[qi][ks] = *(const uint4*)(qkv + row * ld + off);
                    gn[qi][ks] = *(const uint4*)(a.dout + row * C + off);
                    on[qi][ks] = *(const uint4*)(a.out + row * C + off);
                }
            }
            lse_n[qi] = q_ok[qi] ? a.lse[((long)(b * a.nWs + g.ws) * a.d.heads + g.head) * a.Npad + (wave + NW * qi) * 32 + j] : 0.f;
        }
    };
    issue_kv(g.b0, 0);
    load(g.b0);
    __syncthreads();

    for (int b = g.b0; b < g.b1; ++b) {
        const int cur = (b - g.b0) & 1;
        const char* Ks = smem + cur * 49152;
        const char* Kv = Ks + 16384;
        const char* Vs = Ks + 32768;
        if (!PREFETCH && b > g.b0) load(b);
        bf16x8 qf[NQ][2], dof[NQ][2];
        float dl[NQ], lse[NQ];
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            float d = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                qf[qi][ks] = as_bf16x8(qn[qi][ks]); dof[qi][ks] = as_bf16x8(gn[qi][ks]);
                float gf[8], of[8];
                unpack8(gn[qi][ks], gf); unpack8(on[qi][ks], of);
#pragma unroll
                for (int e = 0; e < 8; ++e) d += gf[e] * of[e];
            }
            d += __shfl_xor(d, 32, 64);
            dl[qi] = d; lse[qi] = lse_n[qi];
            if (q_ok[qi] && hi == 0)
                delta_out[((long)(b * a.nWs + g.ws) * a.d.heads + g.head) * a.Npad + (wave + NW * qi) * 32 + j] = d;
        }
        if (b + 1 < g.b1) {
            issue_kv(b + 1, cur ^ 1);
            if (PREFETCH) load(b + 1);                    // register prefetch of the next window's q / dO / O rows
        }

#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            const int qt = wave + NW * qi;
            if (qt >= ntile) break;
            const bf16_t* comb = comb0 + (long)qt * 8 * 1024;
            f32x16 dq;
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) {
                if (kt >= ntile) break;
                const int k0 = kt * 32;
                const uint4 c0 = *(const uint4*)(comb + kt * 1024), c1 = *(const uint4*)(comb + kt * 1024 + 8);
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 kf = *(const bf16x8*)(Ks + krow_off<HD>(k0 + j, ks * 2 + hi));
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qi][ks], s, 0, 0, 0);
                    bf16x8 vf = *(const bf16x8*)(Vs + krow_off<HD>(k0 + j, ks * 2 + hi));
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[qi][ks], dp, 0, 0, 0);
                }
                float c[16], ds[16];
                unpack16(c0, c1, c);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = q_ok[qi] ? fast_exp2(fmaf(s[r], sc, c[r]) - lse[qi]) : 0.f;   // padded keys: c = -inf-like -> 0
                    ds[r] = p * (dp[r] - dl[qi]);
                }
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    bf16x8 dsf = pack_frag(ds + 8 * sl);
                    bf16x8 ktf = tr_frag<HD>(Kv, k0 + 16 * sl, 0, lane);
                    dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf, dq, 0, 0, 0);
                    if (DBIAS) dsa[qi][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(onehot[sl], dsf, dsa[qi][kt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);        // keep each tile's loads next to their use (register pressure)
            }
            if (q_ok[qi]) {
                bf16_t* op = a.dqkv + ((long)b * a.tps + qrel[qi]) * ld + g.head * HD;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    uint2 w;
                    w.x = pack2(dq[r4 * 4 + 0] * a.d.scale, dq[r4 * 4 + 1] * a.d.scale);
                    w.y = pack2(dq[r4 * 4 + 2] * a.d.scale, dq[r4 * 4 + 3] * a.d.scale);
                    *(uint2*)(op + 8 * r4 + 4 * hi) = w;
                }
            }
        }
        __syncthreads();                                  // drains the LDS-DMA of window b+1 (vmcnt) and fences buffer reuse
    }
    // ---- flush: registers -> wave-private LDS table (index = code(q) - code(k) + const) -> global --------------
    // Within one half-wave the 32 queries are distinct tokens and the key is fixed, so the 32 indices are distinct:
    // a plain read-modify-write per half is race-free.  The two halves (key, key + 4) are done one after the other.
    if (DBIAS) {
        float* dtbl = dtbl_all + (wave & 3) * a.tbl_rows;
#pragma unroll
        for (int grp = 0; grp < NW / 4; ++grp) {             // wave groups take turns on the four tables
            if (grp > 0) __syncthreads();
            if ((wave >> 2) != grp) continue;
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                const int qt = wave + NW * qi;
                if (qt >= ntile) break;
                const int qc = kcode[qt * 32 + j] + a.tbl_const;
#pragma unroll
                for (int kt = 0; kt < 8; ++kt) {
                    if (kt >= ntile) break;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int k = kt * 32 + tile_row(r, hi);
                        const int idx = qc - kcode[k];
                        const bool ok = q_ok[qi] && k < N;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            if (hi == h && ok) dtbl[idx] += dsa[qi][kt][r];
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                }
            }
        }
        __syncthreads();
        for (int r = tid; r < a.tbl_rows; r += NTHR) {
            const float v = dtbl_all[r] + dtbl_all[a.tbl_rows + r] + dtbl_all[2 * a.tbl_rows + r] + dtbl_all[3 * a.tbl_rows + r];
            if (v != 0.f) atomicAdd(a.dbias + (long)r * a.d.heads + g.head, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// backward pass 2: dK, dV.  Wave owns key tile `wave`; queries / dO of the window are staged in LDS.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void win_dkv_p(AttnArgs a, int bsplit, const float* delta_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][Q(A) | Q(tr) | dO(A) | dO(tr) 16K each | lse 1K | delta 1K]
    WinGeo g;
    if (!win_geo(a, bsplit, g)) return;
    constexpr int BUF = 65536 + 2048;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hi = lane >> 5;
    const int N = a.N, C = a.C, ld = 3 * C;
    const bf16_t* qkv = a.qkv;
    const float sc = a.d.scale * LOG2E;

    int srow[2], sslot[2], srel[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + 512 * i;
        srow[i] = c >> 2; sslot[i] = c & 3;
        srel[i] = a.d.tok_table[g.ws * 256 + srow[i]];
    }
    const int key = wave * 32 + j;
    const int krel = a.d.tok_table[g.ws * 256 + key];
    const bool k_ok = krel >= 0;
    const int ntile = (N + 31) >> 5;
    const bool wave_on = wave < ntile;
    const bf16_t* combT = (const bf16_t*)a.d.combT + (((long)(g.type * a.d.heads + g.head) * 64 + wave * 8) * 64 + lane) * 16;
    const int srel_q = tid < 256 ? a.d.tok_table[g.ws * 256 + tid] : -1;

    uint4 qr[2], gr[2], kn[2], vn[2];
    float lse_r = 0.f, dl_r = 0.f;
    auto load = [&](int b) {
        const long base = (long)b * a.tps;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            qr[i] = make_uint4(0, 0, 0, 0); gr[i] = qr[i];
            if (srel[i] >= 0) {
                const long row = base + srel[i];
                qr[i] = *(const uint4*)(qkv + row * ld + g.head * HD + sslot[i] * 8);
                gr[i] = *(const uint4*)(a.dout + row * C + g.head * HD + sslot[i] * 8);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kn[ks] = make_uint4(0, 0, 0, 0); vn[ks] = kn[ks];
            if (k_ok) {
                const bf16_t* p = qkv + (base + krel) * ld + g.head * HD + ks * 16 + 8 * hi;
                kn[ks] = *(const uint4*)(p + C);
                vn[ks] = *(const uint4*)(p + 2 * C);
            }
        }
        if (tid < 256) {
            const long li = ((long)(b * a.nWs + g.ws) * a.d.heads + g.head) * a.Npad + tid;
            lse_r = srel_q >= 0 ? a.lse[li] : 0.f;
            dl_r = srel_q >= 0 ? delta_in[li] : 0.f;
        }
    };
    auto stash = [&](int buf) {
        char* B0 = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(uint4*)(B0 + krow_off<HD>(srow[i], sslot[i])) = qr[i];
            *(uint4*)(B0 + 16384 + vrow_off<HD>(srow[i], sslot[i])) = qr[i];
            *(uint4*)(B0 + 32768 + krow_off<HD>(srow[i], sslot[i])) = gr[i];
            *(uint4*)(B0 + 49152 + vrow_off<HD>(srow[i], sslot[i])) = gr[i];
        }
        if (tid < 256) { ((float*)(B0 + 65536))[tid] = lse_r; ((float*)(B0 + 65536 + 1024))[tid] = dl_r; }
    };
    load(g.b0);
    stash(0);
    __syncthreads();

    for (int b = g.b0; b < g.b1; ++b) {
        const int cur = (b - g.b0) & 1;
        const char* Qs = smem + cur * BUF;
        const char* Qv = Qs + 16384;
        const char* Gs = Qs + 32768;
        const char* Gv = Qs + 49152;
        const float* qlse = (const float*)(Qs + 65536);
        const float* qdl = qlse + 256;
        const bf16x8 kf[2] = {as_bf16x8(kn[0]), as_bf16x8(kn[1])};
        const bf16x8 vf[2] = {as_bf16x8(vn[0]), as_bf16x8(vn[1])};
        if (b + 1 < g.b1) load(b + 1);

        if (wave_on) {
            f32x16 dk, dv;
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
#pragma unroll
            for (int qt = 0; qt < 8; ++qt) {
                if (qt >= ntile) break;
                const int q0 = qt * 32;
                const uint4 c0 = *(const uint4*)(combT + qt * 1024), c1 = *(const uint4*)(combT + qt * 1024 + 8);
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 qa = *(const bf16x8*)(Qs + krow_off<HD>(q0 + j, ks * 2 + hi));
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], s, 0, 0, 0);
                    bf16x8 ga = *(const bf16x8*)(Gs + krow_off<HD>(q0 + j, ks * 2 + hi));
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, vf[ks], dp, 0, 0, 0);
                }
                float c[16], pd[16], ds[16];
                unpack16(c0, c1, c);
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
                        // padded queries: combT = 0 but lse slot = 0 and Q = dO = 0 -> mask explicitly
                        const float p = (k_ok && qb + e < N) ? fast_exp2(fmaf(s[r], sc, c[r]) - ls[e]) : 0.f;
                        pd[r] = p;
                        ds[r] = p * (dp[r] - dls[e]);
                    }
                }
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    bf16x8 pf = pack_frag(pd + 8 * sl), dsf = pack_frag(ds + 8 * sl);
                    bf16x8 gt = tr_frag<HD>(Gv, q0 + 16 * sl, 0, lane);
                    dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt, pf, dv, 0, 0, 0);
                    bf16x8 qt_ = tr_frag<HD>(Qv, q0 + 16 * sl, 0, lane);
                    dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt_, dsf, dk, 0, 0, 0);
                }
            }
            if (k_ok) {
                bf16_t* op = a.dqkv + ((long)b * a.tps + krel) * ld + g.head * HD;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int d = 8 * r4 + 4 * hi;
                    uint2 w;
                    w.x = pack2(dk[r4 * 4 + 0] * a.d.scale, dk[r4 * 4 + 1] * a.d.scale);
                    w.y = pack2(dk[r4 * 4 + 2] * a.d.scale, dk[r4 * 4 + 3] * a.d.scale);
                    *(uint2*)(op + C + d) = w;
                    w.x = pack2(dv[r4 * 4 + 0], dv[r4 * 4 + 1]);
                    w.y = pack2(dv[r4 * 4 + 2], dv[r4 * 4 + 3]);
                    *(uint2*)(op + 2 * C + d) = w;
                }
            }
        }
        if (b + 1 < g.b1) stash(cur ^ 1);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------
// fragment-ordered (bias + shift mask + key padding) tables, exp2 domain (values multiplied by log2 e)
//   comb [type][head][qt][kt][lane][r]: query = qt*32 + (lane&31), key = kt*32 + tile_row(r, lane>>5)
//   combT[type][head][kt][qt][lane][r]: key   = kt*32 + (lane&31), query = qt*32 + tile_row(r, lane>>5)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_bias_kernel(AttnArgs a, bf16_t* comb, bf16_t* combT) {
    const long total = (long)a.d.n_types * a.d.heads * 64 * 64 * 16;
    const lav_attn_desc& d = a.d;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int r = idx & 15, lane = (idx >> 4) & 63, t1 = (idx >> 10) & 7, t0 = (idx >> 13) & 7;
        const int head = (idx >> 16) % d.heads, type = (idx >> 16) / d.heads;
        const int jj = lane & 31, hi = lane >> 5;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const int q = which == 0 ? t0 * 32 + jj : t1 * 32 + tile_row(r, hi);
            const int k = which == 0 ? t1 * 32 + tile_row(r, hi) : t0 * 32 + jj;
            float v;
            if (k >= a.N) v = -30000.f;
            else if (q >= a.N) v = 0.f;
            else {
                // index decode with the CONFIGURED (h, w) extents: relative_position_index[:N, :N] (video_swin.py:153)
                const int qw = q % d.cfg_ww, qh = (q / d.cfg_ww) % d.cfg_wh, qd = q / (d.cfg_ww * d.cfg_wh);
                const int kw = k % d.cfg_ww, kh = (k / d.cfg_ww) % d.cfg_wh, kd = k / (d.cfg_ww * d.cfg_wh);
                const int bi = (qd - kd) * a.cstride_d + (qh - kh) * a.cstride_h + (qw - kw) + a.tbl_const;
                v = d.bias_table[(long)bi * d.heads + head];
                if (d.type_region[type * 256 + q] != d.type_region[type * 256 + k]) v += -100.0f;
            }
            (which == 0 ? comb : combT)[idx] = f2bf(v * LOG2E);
        }
    }
}

extern "C" int lav_attention_build_bias(void* stream, const lav_attn_desc* d) {
    AttnArgs a; int problems = 0;
    if (int rc = attn_setup(d, a, problems)) return rc;
    LAV_REQUIRE(d->mode == 0 && d->comb && d->combT, "lav_attention_build_bias: window mode with comb/combT buffers required");
    long total = (long)d->n_types * d->heads * 64 * 64 * 16;
    int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(build_bias_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, (bf16_t*)d->comb, (bf16_t*)d->combT);
    return lav_check_launch("lav_attention_build_bias");
}

static int pick_bsplit(const AttnArgs& a) {
    const int base = a.d.heads * a.nWs;
    int bs = (256 + base - 1) / base;
    if (bs > a.d.B) bs = a.d.B;
    if (bs < 1) bs = 1;
    return bs;
}

template <typename Kn>
static void big_lds(Kn k, size_t bytes) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    (void)hipGetLastError();
}

int win_persistent_fwd(void* stream, const AttnArgs& a) {
    const int bs = pick_bsplit(a);
    const size_t lds = 2 * 32768;
    hipLaunchKernelGGL(win_fwd_p, dim3(a.d.heads * a.nWs * bs), dim3(512), lds, (hipStream_t)stream, a, bs);
    return lav_check_launch("lav_attention_fwd(window, persistent)");
}

// LAV_WIN_BWD_FUSED=1 selects the single-kernel backward of attention_win_bwd.hip (measured slower on every Swin-B stage:
// 10.9 vs 8.9 ms per step); the default is the two-pass form below (dQ + bias gradient, then dK / dV)
static const bool lav_win_bwd_fused = getenv("LAV_WIN_BWD_FUSED") ? atoi(getenv("LAV_WIN_BWD_FUSED")) != 0 : false;

int win_persistent_bwd(void* stream, const AttnArgs& a, float* delta) {
    const int bs = pick_bsplit(a);
    if (lav_win_bwd_fused) return win_fused_bwd(stream, a, bs);
    const size_t lds1 = 2 * 49152 + 2048 + (size_t)a.tbl_rows * 16;
    // with the bias-table gradient the resident dS sums need the 512-register budget of one wave per SIMD (the 8-wave
    // variant spills in its inner loop: 375 vs 309 us on the stage-2 shape); without it two waves per SIMD win (202 vs 230 us)
    if (a.dbias) {
        big_lds((win_dq_p<true, 4>), lds1);
        hipLaunchKernelGGL((win_dq_p<true, 4>), dim3(a.d.heads * a.nWs * bs), dim3(256), lds1, (hipStream_t)stream, a, bs, delta);
    } else {
        big_lds((win_dq_p<false, 8>), lds1);
        hipLaunchKernelGGL((win_dq_p<false, 8>), dim3(a.d.heads * a.nWs * bs), dim3(512), lds1, (hipStream_t)stream, a, bs, delta);
    }
    const size_t lds2 = 2 * (65536 + 2048);
    big_lds(win_dkv_p, lds2);
    hipLaunchKernelGGL(win_dkv_p, dim3(a.d.heads * a.nWs * bs), dim3(512), lds2, (hipStream_t)stream, a, bs, (const float*)delta);
    return lav_check_launch("lav_attention_bwd(window, persistent)");
}
