// Shifted-window attention for LARGE windows on gfx950 (256 < N <= 768 tokens, head_dim 32): the Swin-L 384^2 shape
// (12 x 12 x 5 = 720-token windows, video_swin.py:46-106 with window_size (8, 12, 12)).  Same rules as attention_win.hip /
// attention_seq.hip, but a window no longer fits the "whole problem as A-operand bias fragments" scheme (a 768 x 768 bias tile
// set is 1.2 MB per head and mask type), so:
//   * K and V (forward, dQ pass) or Q and dO (dK / dV pass) of a (window, head) problem arrive by global_load_lds into one
//     slot-swizzled 48 KB LDS image each, through the window's token rows (roll / partition are address arithmetic: the rows of
//     the un-rolled token tensor are computed once per problem into LDS).  The image serves both the ds_read_b128 fragment reads
//     and the transposing ds_read_b64_tr_b16 reads.
//   * relative-position bias: the REACHABLE part of this head's table column (a window clamped to 5 of the configured 8 frames
//     reaches 9 x 23 x 23 = 4761 of the 7935 entries; pre-multiplied by log2 e) sits in LDS and is gathered per
//     score element with index code(q) - code(k) + const -- ONE integer add per element, the per-token codes come from LDS as
//     int4 broadcasts.  The shift mask (region(q) != region(k) -> -100) is only evaluated for windows that straddle a region
//     boundary (block-uniform flag): 49 of 64 windows of a shifted 8 x 8 layout skip it.
//   * one 512-thread workgroup per CU (124-143 KB of LDS) walks items (head, window, query part); the 8 waves take the 32-row
//     tiles of the part round-robin.  With few problems (late stages) a problem is split into 3 query parts so that 23 tiles
//     fill 3 x 8 wave slots instead of 3 rounds of 8.
//   * bias-table gradient: its own kernel (winl_dbias, issued by the engine on the weight-gradient stream).  A first version
//     accumulated it in the dQ pass with one ds_add_f32 per score element into an LDS copy of the table column: 69 of 86 ms per
//     cfg4 step (LDS float atomics retire ~3 cycles per LANE).  Now a workgroup owns (head, 128 queries x 256 keys) of the
//     window plane, walks a slice of ALL windows of the batch with double-buffered operand DMA, recomputes dS for its block
//     and accumulates it DENSE in MFMA accumulators (one-hot fragments); one flush per workgroup: dense matrix through LDS,
//     per-offset-class box sums, one global atomic per class.
// Scores use the exp2 domain; lse is log2(sum_k 2^v), stored [window * heads + head][Npad]; delta follows it (backward scratch).
#include "attn_common.h"
#include <stdlib.h>

#define LOG2E 1.4426950408889634f
#define HD 32
#define WL_ROWS 768
#define WL_IMG (WL_ROWS * 64)

__device__ __forceinline__ void wl_dma16(unsigned lds_dst, const void* sbase, unsigned voff) {
    unsigned keep_m0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep_m0) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}
__device__ __forceinline__ void wl_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// transposing A-operand read (32 d-values x 16 rows) from a K-type (slot-swizzled) 64-byte-row image
__device__ __forceinline__ bf16x8 wl_tr_frag(const char* tile, int row_base, int lane) {
    const int i = lane & 15, dhalf = (lane >> 4) & 1, hi = lane >> 5;
    const int r = i >> 2, c = i & 3;
    const int dcol = 16 * dhalf + 4 * c;
    const int slot = dcol >> 3, sub = (dcol & 7) * 2;
    const int row0 = row_base + 4 * hi + r;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tile + krow_off<32>(row0, slot) + sub));
    s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tile + krow_off<32>(row0 + 8, slot) + sub));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi4;
    return u.v;
}

union WFrag { uint4 u; bf16x8 b; };
#define WZERO16 {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}
__device__ __forceinline__ float wmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// A workgroup keeps ONE head for its whole walk (table column loaded once, bias gradient flushed once): workgroup b serves head
// b % heads and takes that head's (window, query part) items bi, bi + Gh, ... where Gh = workgroups of the head.
struct WlTab { int lo, n, mc; };     // first reachable table row, reachable rows, code of the last window slot
struct WlWalk { int head, first, step, count; };
__device__ __forceinline__ WlWalk wl_walk(const AttnArgs& a, int nwin, int QS) {
    WlWalk w;
    const int heads = a.d.heads;
    w.head = blockIdx.x % heads; w.first = blockIdx.x / heads;
    w.step = ((int)gridDim.x - w.head + heads - 1) / heads;
    w.count = nwin * QS;
    return w;
}
struct WlItem { int win, head, part; long p; };
__device__ __forceinline__ WlItem wl_item(const AttnArgs& a, const WlWalk& k, int wi, int QS) {
    WlItem w;
    w.part = wi % QS; w.win = wi / QS; w.head = k.head;
    w.p = (long)w.win * a.d.heads + w.head;
    return w;
}

// win_token() split into its per-slot part (three divisions, hoisted out of the window loops: a thread keeps its slots) and its
// per-window part (uniform: scalar divisions) -- same results as attn_common.h:win_token.
struct WlSlot { int di, hi, wi, code; };
struct WlOrg { int b, od, oh, ow; };
__device__ __forceinline__ WlSlot wl_slot(const AttnArgs& a, int i) {
    const lav_attn_desc& d = a.d;
    WlSlot s;
    s.wi = i % d.ww; const int t2 = i / d.ww;
    s.hi = t2 % d.wh; s.di = t2 / d.wh;
    s.code = (i / (d.cfg_wh * d.cfg_ww)) * a.cstride_d + ((i / d.cfg_ww) % d.cfg_wh) * a.cstride_h + (i % d.cfg_ww);
    return s;
}
__device__ __forceinline__ WlOrg wl_origin(const AttnArgs& a, int win) {
    WlOrg o;
    const int wwi = win % a.nWw; int t = win / a.nWw;
    const int whi = t % a.nWh; t /= a.nWh;
    const int wdi = t % a.nWd; o.b = t / a.nWd;
    o.od = wdi * a.d.wd; o.oh = whi * a.d.wh; o.ow = wwi * a.d.ww;
    return o;
}
__device__ __forceinline__ int wl_row(const AttnArgs& a, const WlOrg& o, const WlSlot& s, int& region) {
    const lav_attn_desc& d = a.d;
    const int pd = o.od + s.di, ph = o.oh + s.hi, pw = o.ow + s.wi;           // coordinates in the rolled tensor
    int sd_ = pd + d.sd, sh_ = ph + d.sh, sw_ = pw + d.sw;                    // roll(-shift): rolled[p] = x[(p+s) % n]
    if (sd_ >= d.D) sd_ -= d.D;
    if (sh_ >= d.H) sh_ -= d.H;
    if (sw_ >= d.W) sw_ -= d.W;
    const int rd = d.sd ? (pd >= d.D - d.wd) + (pd >= d.D - d.sd) : 0;
    const int rh = d.sh ? (ph >= d.H - d.wh) + (ph >= d.H - d.sh) : 0;
    const int rw = d.sw ? (pw >= d.W - d.ww) + (pw >= d.W - d.sw) : 0;
    region = rd * 9 + rh * 3 + rw;
    return ((o.b * d.D + sd_) * d.H + sh_) * d.W + sw_;
}

// Per-problem token geometry into LDS: rows[i] = token row of window slot i (slots >= N repeat slot N - 1: never used unmasked),
// kcn[i] = mc - code(i) (bias index of (q, k) in the LDS copy = code(q) + kcn[k]), kr[i] = shift region.  Returns (block-uniform) whether the window holds more than one region.
__device__ __forceinline__ bool wl_geometry(const AttnArgs& a, int win, int nslot, int* rows, int* kr, int tid, const WlSlot (&sl)[2]) {
    const WlOrg o = wl_origin(a, win);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + u * 512;
        if (i < nslot) { int reg; rows[i] = wl_row(a, o, sl[u], reg); kr[i] = reg; }
    }
    __syncthreads();
    if (!(a.d.sd | a.d.sh | a.d.sw)) return false;
    const int r0 = kr[0];
    int diff = 0;
    for (int i = tid; i < a.N; i += 512) diff |= kr[i] != r0;
    return __syncthreads_or(diff) != 0;
}

// DMA of one operand image (nslot rows of 64 B, 16 rows per 1-KB piece) through the window's token rows
__device__ __forceinline__ void wl_dma_image(unsigned lds_img, const bf16_t* base, int ld, int col0, const int* rows, int npiece, int wave, int lane) {
    for (int t = wave; t < npiece; t += 8) {
        const int row = rows[t * 16 + (lane >> 2)];
        const int lslot = (lane & 3) ^ ((lane >> 4) & 3);
        wl_dma16(lds_img + t * 1024, base, ((unsigned)row * (unsigned)ld + (unsigned)(col0 + lslot * 8)) * 2u);
    }
}

__device__ __forceinline__ void wl_load_table(const AttnArgs& a, int head, float* tbl, int tid, const WlTab& tb) {
    for (int r = tid; r < tb.n; r += 512) tbl[r] = a.d.bias_table[(long)(tb.lo + r) * a.d.heads + head] * LOG2E;
}

// (scaled score + bias + shift mask) of one 32 x 32 S^T tile held as [key r][query j]: in place, exp2 domain
template <bool LAST>
__device__ __forceinline__ void wl_bias_rows_keys(f32x16& s, float sc, const float* tbl, const int* kcn, const int* kr, int key0, int hi,
                                                  int q_code, int q_reg, bool multi, int N) {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int kb = key0 + 8 * r4 + 4 * hi;
        const int4 c4 = *(const int4*)(kcn + kb);
        const int cs[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) s[4 * r4 + e] = fmaf(s[4 * r4 + e], sc, tbl[q_code + cs[e]]);
        if (multi) {
            const int4 g4 = *(const int4*)(kr + kb);
            const int gs[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) s[4 * r4 + e] += gs[e] != q_reg ? -100.0f * LOG2E : 0.f;
        }
        if (LAST) {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (kb + e >= N) s[4 * r4 + e] = -INFINITY;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// forward.  LDS: K image | V image | rows | kcn | kr | table column.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void winl_fwd(AttnArgs a, int nwin, int QS, WlTab tb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + WL_IMG;
    int* rows = (int*)(smem + 2 * WL_IMG);
    int* kcn = rows + WL_ROWS;
    int* kr = kcn + WL_ROWS;
    float* tbl = (float*)(kr + WL_ROWS);
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, C = a.C, ld = 3 * C, nt = a.nqt;
    const float sc = a.d.scale * LOG2E;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const WlWalk walk = wl_walk(a, nwin, QS);
    bf16x8 ones0;
    {
        float e8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) e8[e] = j == 0 ? 1.f : 0.f;
        ones0 = pack_frag(e8);
    }
    wl_load_table(a, walk.head, tbl, tid, tb);
    WlSlot slots[2];                                         // this thread's window slots tid, tid + 512: constant over the walk
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + u * 512;
        slots[u] = wl_slot(a, min(i, N - 1));
        if (i < nt * 32) kcn[i] = tb.mc - slots[u].code;
    }
    for (int wi = walk.first; wi < walk.count; wi += walk.step) {
        const WlItem w = wl_item(a, walk, wi, QS);
        __syncthreads();                                     // every wave is done with the previous item's LDS
        const bool multi = wl_geometry(a, w.win, nt * 32, rows, kr, tid, slots);
        wl_dma_image(lds0, a.qkv, ld, C + w.head * HD, rows, nt * 2, wave, lane);
        wl_dma_image(lds0 + WL_IMG, a.qkv, ld, 2 * C + w.head * HD, rows, nt * 2, wave, lane);
        wl_dma_wait();
        __syncthreads();

        for (int qt = w.part + QS * wave; qt < nt; qt += QS * 8) {
            const int q = qt * 32 + j;
            const bool q_ok = q < N;
            const long qrow = rows[q];
            const int q_code = tb.mc - kcn[q], q_reg = kr[q];
            bf16x8 qf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { WFrag f; f.u = *(const uint4*)(a.qkv + qrow * ld + w.head * HD + ks * 16 + 8 * hi); qf[ks] = f.b; }
            f32x16 o = WZERO16, lacc = WZERO16;
            float m_run = -INFINITY;
            f32x16 sa, sb;
            auto qk = [&](int t, f32x16& s) {
                const f32x16 z = WZERO16;
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Ks + krow_off<HD>(t * 32 + j, hi)), qf[0], z, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Ks + krow_off<HD>(t * 32 + j, 2 + hi)), qf[1], s, 0, 0, 0);
            };
            auto soft = [&](int t, f32x16& s) {
                const bf16x8 vf0 = wl_tr_frag(Vs, t * 32, lane), vf1 = wl_tr_frag(Vs, t * 32 + 16, lane);
                if (t == nt - 1) wl_bias_rows_keys<true>(s, sc, tbl, kcn, kr, t * 32, hi, q_code, q_reg, multi, N);
                else wl_bias_rows_keys<false>(s, sc, tbl, kcn, kr, t * 32, hi, q_code, q_reg, multi, N);
                float mx = wmax3(s[0], s[1], s[2]);
#pragma unroll
                for (int r = 3; r < 15; r += 2) mx = wmax3(mx, s[r], s[r + 1]);
                mx = fmaxf(mx, s[15]);
                mx = xhalf_max(mx);
                if (__any(mx > m_run)) {
                    const float m_new = fmaxf(m_run, mx);
                    const float alpha = fast_exp2(m_run - m_new);
                    lacc[0] *= alpha;
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] *= alpha;
                    m_run = m_new;
                }
                const float nm = -m_run;                     // finite: key 0 of tile 0 is never masked to -inf
                uint32_t pk[8];
#pragma unroll
                for (int r = 0; r < 16; r += 2) pk[r >> 1] = pack2(fast_exp2(s[r] + nm), fast_exp2(s[r + 1] + nm));
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    WFrag pf; pf.u = make_uint4(pk[4 * sl], pk[4 * sl + 1], pk[4 * sl + 2], pk[4 * sl + 3]);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sl ? vf1 : vf0, pf.b, o, 0, 0, 0);
                    lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones0, pf.b, lacc, 0, 0, 0);
                }
            };
            qk(0, sa);
#pragma unroll 1
            for (int t = 0; t < nt; t += 2) {
                if (t + 1 < nt) qk(t + 1, sb);
                soft(t, sa);
                if (t + 1 < nt) {
                    if (t + 2 < nt) qk(t + 2, sa);
                    soft(t + 1, sb);
                }
            }
            const float l_tot = lower_half(lacc[0]);
            const float inv_l = l_tot > 0.f ? 1.f / l_tot : 0.f;
            if (q_ok) {
                bf16_t* op = a.o_w + qrow * C + w.head * HD;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    uint2 v;
                    v.x = pack2(o[r4 * 4 + 0] * inv_l, o[r4 * 4 + 1] * inv_l);
                    v.y = pack2(o[r4 * 4 + 2] * inv_l, o[r4 * 4 + 3] * inv_l);
                    *(uint2*)(op + 8 * r4 + 4 * hi) = v;
                }
                if (a.lse && hi == 0) a.lse[w.p * a.Npad + q] = m_run + log2f(l_tot);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// backward pass 1: dQ and delta[q] = sum_d dO[q,d] O[q,d] (stored behind the lse).
// LDS: K image | V image | rows | kcn | kr | table column.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void winl_dq(AttnArgs a, int nwin, int QS, WlTab tb, float* delta_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + WL_IMG;
    int* rows = (int*)(smem + 2 * WL_IMG);
    int* kcn = rows + WL_ROWS;
    int* kr = kcn + WL_ROWS;
    float* tbl = (float*)(kr + WL_ROWS);
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, C = a.C, ld = 3 * C, nt = a.nqt;
    const float sc = a.d.scale * LOG2E;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const WlWalk walk = wl_walk(a, nwin, QS);
    wl_load_table(a, walk.head, tbl, tid, tb);
    WlSlot slots[2];                                         // this thread's window slots tid, tid + 512: constant over the walk
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + u * 512;
        slots[u] = wl_slot(a, min(i, N - 1));
        if (i < nt * 32) kcn[i] = tb.mc - slots[u].code;
    }
    for (int wi = walk.first; wi < walk.count; wi += walk.step) {
        const WlItem w = wl_item(a, walk, wi, QS);
        __syncthreads();
        const bool multi = wl_geometry(a, w.win, nt * 32, rows, kr, tid, slots);
        wl_dma_image(lds0, a.qkv, ld, C + w.head * HD, rows, nt * 2, wave, lane);
        wl_dma_image(lds0 + WL_IMG, a.qkv, ld, 2 * C + w.head * HD, rows, nt * 2, wave, lane);
        wl_dma_wait();
        __syncthreads();

        for (int qt = w.part + QS * wave; qt < nt; qt += QS * 8) {
            const int q = qt * 32 + j;
            const bool q_ok = q < N;
            const long qrow = rows[q];
            const int q_code = tb.mc - kcn[q], q_reg = kr[q];
            bf16x8 qf[2], dof[2];
            float dl = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                WFrag fq, fg, fo;
                fq.u = *(const uint4*)(a.qkv + qrow * ld + w.head * HD + ks * 16 + 8 * hi);
                fg.u = *(const uint4*)(a.dout + qrow * C + w.head * HD + ks * 16 + 8 * hi);
                fo.u = *(const uint4*)(a.out + qrow * C + w.head * HD + ks * 16 + 8 * hi);
                qf[ks] = fq.b; dof[ks] = fg.b;
                float gf[8], of[8];
                unpack8(fg.u, gf); unpack8(fo.u, of);
#pragma unroll
                for (int e = 0; e < 8; ++e) dl = fmaf(gf[e], of[e], dl);
            }
            dl = xhalf_add(dl);
            const float nl = q_ok ? -a.lse[w.p * a.Npad + q] : -INFINITY;     // padded query: P = 0
            if (q_ok && hi == 0) delta_out[w.p * a.Npad + q] = dl;
            f32x16 ndl;
#pragma unroll
            for (int r = 0; r < 16; ++r) ndl[r] = -dl;
            f32x16 dq = WZERO16;
#pragma unroll 1
            for (int t = 0; t < nt; ++t) {
                const f32x16 z = WZERO16;
                f32x16 s, dp;
                {
                    const int o0 = krow_off<HD>(t * 32 + j, hi), o1 = krow_off<HD>(t * 32 + j, 2 + hi);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Ks + o0), qf[0], z, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Vs + o0), dof[0], ndl, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Ks + o1), qf[1], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Vs + o1), dof[1], dp, 0, 0, 0);
                }
                const bf16x8 kt0 = wl_tr_frag(Ks, t * 32, lane), kt1 = wl_tr_frag(Ks, t * 32 + 16, lane);
                if (t == nt - 1) wl_bias_rows_keys<true>(s, sc, tbl, kcn, kr, t * 32, hi, q_code, q_reg, multi, N);
                else wl_bias_rows_keys<false>(s, sc, tbl, kcn, kr, t * 32, hi, q_code, q_reg, multi, N);
                float ds[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) ds[r] = fast_exp2(s[r] + nl) * dp[r];
                uint32_t dk[8];
#pragma unroll
                for (int r = 0; r < 16; r += 2) dk[r >> 1] = pack2(ds[r], ds[r + 1]);
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    WFrag df; df.u = make_uint4(dk[4 * sl], dk[4 * sl + 1], dk[4 * sl + 2], dk[4 * sl + 3]);
                    dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sl ? kt1 : kt0, df.b, dq, 0, 0, 0);
                }
            }
            if (q_ok) {
                bf16_t* op = a.dqkv + qrow * ld + w.head * HD;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    uint2 v;
                    v.x = pack2(dq[r4 * 4 + 0] * a.d.scale, dq[r4 * 4 + 1] * a.d.scale);
                    v.y = pack2(dq[r4 * 4 + 2] * a.d.scale, dq[r4 * 4 + 3] * a.d.scale);
                    *(uint2*)(op + 8 * r4 + 4 * hi) = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// backward pass 2: dK, dV.  A wave owns 32-key tiles and loops over the queries.
// LDS: Q image | dO image | rows | kcn | kr | lse (+inf for padded queries) | delta | table column.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void winl_dkv(AttnArgs a, int nwin, int QS, WlTab tb, const float* delta_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;
    char* Gs = smem + WL_IMG;
    int* rows = (int*)(smem + 2 * WL_IMG);
    int* kcn = rows + WL_ROWS;
    int* kr = kcn + WL_ROWS;
    float* qlse = (float*)(kr + WL_ROWS);
    float* qdl = qlse + WL_ROWS;
    float* tbl = qdl + WL_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, C = a.C, ld = 3 * C, nt = a.nqt;
    const float sc = a.d.scale * LOG2E;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const WlWalk walk = wl_walk(a, nwin, QS);
    wl_load_table(a, walk.head, tbl, tid, tb);
    WlSlot slots[2];                                         // this thread's window slots tid, tid + 512: constant over the walk
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + u * 512;
        slots[u] = wl_slot(a, min(i, N - 1));
        if (i < nt * 32) kcn[i] = tb.mc - slots[u].code;
    }
    for (int wi = walk.first; wi < walk.count; wi += walk.step) {
        const WlItem w = wl_item(a, walk, wi, QS);
        __syncthreads();
        for (int k = tid; k < nt * 32; k += 512) {
            qlse[k] = k < N ? a.lse[w.p * a.Npad + k] : INFINITY;      // padded query rows: P = exp2(v - inf) = 0
            qdl[k] = k < N ? delta_in[w.p * a.Npad + k] : 0.f;
        }
        const bool multi = wl_geometry(a, w.win, nt * 32, rows, kr, tid, slots);
        wl_dma_image(lds0, a.qkv, ld, w.head * HD, rows, nt * 2, wave, lane);
        wl_dma_image(lds0 + WL_IMG, a.dout, C, w.head * HD, rows, nt * 2, wave, lane);
        wl_dma_wait();
        __syncthreads();

        for (int kt = w.part + QS * wave; kt < nt; kt += QS * 8) {
            const int key = kt * 32 + j;
            const bool k_ok = key < N;
            const long krow = rows[key];
            const int kb = tb.mc + kcn[key], k_reg = kr[key];        // bias index = kb - kcn[q]
            bf16x8 kf[2], vf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                WFrag fk, fv;
                fk.u = *(const uint4*)(a.qkv + krow * ld + C + w.head * HD + ks * 16 + 8 * hi);
                fv.u = *(const uint4*)(a.qkv + krow * ld + 2 * C + w.head * HD + ks * 16 + 8 * hi);
                kf[ks] = fk.b; vf[ks] = fv.b;
            }
            const float k_add = k_ok ? 0.f : -INFINITY;
            f32x16 dk = WZERO16, dv = WZERO16;
#pragma unroll 1
            for (int qt = 0; qt < nt; ++qt) {
                const int q0 = qt * 32;
                const f32x16 z = WZERO16;
                f32x16 s, dp;
                {
                    const int o0 = krow_off<HD>(q0 + j, hi), o1 = krow_off<HD>(q0 + j, 2 + hi);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Qs + o0), kf[0], z, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Gs + o0), vf[0], z, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Qs + o1), kf[1], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Gs + o1), vf[1], dp, 0, 0, 0);
                }
                const bf16x8 gt0 = wl_tr_frag(Gs, q0, lane), gt1 = wl_tr_frag(Gs, q0 + 16, lane);
                const bf16x8 qt0 = wl_tr_frag(Qs, q0, lane), qt1 = wl_tr_frag(Qs, q0 + 16, lane);
                uint32_t pk[8], dsk[8];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int qb = q0 + 8 * r4 + 4 * hi;
                    const int4 c4 = *(const int4*)(kcn + qb);
                    const float4 l4 = *(const float4*)(qlse + qb);
                    const float4 d4 = *(const float4*)(qdl + qb);
                    const int cs[4] = {c4.x, c4.y, c4.z, c4.w};
                    const float ls[4] = {l4.x, l4.y, l4.z, l4.w};
                    const float dls[4] = {d4.x, d4.y, d4.z, d4.w};
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(s[4 * r4 + e], sc, tbl[kb - cs[e]]) + k_add;
                    if (multi) {
                        const int4 g4 = *(const int4*)(kr + qb);
                        const int gs[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += gs[e] != k_reg ? -100.0f * LOG2E : 0.f;
                    }
                    float pv[4], dsv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pv[e] = fast_exp2(v[e] - ls[e]);
                        dsv[e] = pv[e] * (dp[4 * r4 + e] - dls[e]);
                    }
                    pk[2 * r4] = pack2(pv[0], pv[1]); pk[2 * r4 + 1] = pack2(pv[2], pv[3]);
                    dsk[2 * r4] = pack2(dsv[0], dsv[1]); dsk[2 * r4 + 1] = pack2(dsv[2], dsv[3]);
                }
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    WFrag pf, df;
                    pf.u = make_uint4(pk[4 * sl], pk[4 * sl + 1], pk[4 * sl + 2], pk[4 * sl + 3]);
                    df.u = make_uint4(dsk[4 * sl], dsk[4 * sl + 1], dsk[4 * sl + 2], dsk[4 * sl + 3]);
                    dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sl ? gt1 : gt0, pf.b, dv, 0, 0, 0);
                    dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sl ? qt1 : qt0, df.b, dk, 0, 0, 0);
                }
            }
            if (k_ok) {
                bf16_t* op = a.dqkv + krow * ld + w.head * HD;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int d = 8 * r4 + 4 * hi;
                    uint2 v;
                    v.x = pack2(dk[r4 * 4 + 0] * a.d.scale, dk[r4 * 4 + 1] * a.d.scale);
                    v.y = pack2(dk[r4 * 4 + 2] * a.d.scale, dk[r4 * 4 + 3] * a.d.scale);
                    *(uint2*)(op + C + d) = v;
                    v.x = pack2(dv[r4 * 4 + 0], dv[r4 * 4 + 1]);
                    v.y = pack2(dv[r4 * 4 + 2], dv[r4 * 4 + 3]);
                    *(uint2*)(op + 2 * C + d) = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// relative-position-bias gradient: dtable[code(q) - code(k) + const, head] += sum over windows and batch of dS[q, k].
// Workgroup = (head, query block qb of 128, key block kb of 256, window slice); wave = (query strip qs of the block, key half kh):
// 4 key tiles, 64 accumulators.  Per window: K / V rows of the key block, Q / dO rows of the query block, lse and delta by DMA
// into one of two buffers (the next window's under this one's MFMAs); geometry (token rows, regions) is arithmetic.
// LDS: max([2][K 16 KB | V 16 KB | Q 8 KB | dO 8 KB | lse 1 KB | delta 1 KB | rowsK | rowsQ | krK | krQ], flush matrix
// 128 x 257 floats) | kcnK | kcnQ | table column.
// ------------------------------------------------------------------------------------------------------
#define WLB_BUF (16384 * 2 + 8192 * 2 + 2048 + 384 * 4 * 2)
#define WLB_DLD 257
#define WLB_MAIN (2 * WLB_BUF > 128 * WLB_DLD * 4 ? 2 * WLB_BUF : 128 * WLB_DLD * 4)

__global__ __launch_bounds__(512) void winl_dbias(AttnArgs a, int nwin, int WS, WlTab tb, const float* delta_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* kcnK = (int*)(smem + WLB_MAIN);
    int* kcnQ = kcnK + 256;
    float* tbl = (float*)(kcnQ + 128);
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qs = wave & 3, kh = wave >> 2;
    const int N = a.N, C = a.C, ld = 3 * C, nt = a.nqt;
    const float sc = a.d.scale * LOG2E;
    const int nqb = (nt + 3) / 4, nkb = (nt + 7) / 8;
    int it = blockIdx.x;
    const int ws = it % WS; it /= WS;
    const int kb = it % nkb; it /= nkb;
    const int qb = it % nqb;
    const int head = it / nqb;
    const int per = (nwin + WS - 1) / WS;
    const int w_beg = ws * per, w_end = min(nwin, w_beg + per);
    if (w_beg >= w_end) return;
    const int q_lo = qb * 128, k_lo = kb * 256;
    const int qt = qb * 4 + qs;
    const bool strip_on = qt < nt;                           // wave-uniform
    const int q = qt * 32 + j;
    const bool q_ok = q < N;

    wl_load_table(a, head, tbl, tid, tb);
    // fixed slots of this thread: the one whose region it publishes (keys 0..255, then queries 0..127 of the block) and the three
    // whose rows its DMA lanes fetch (two key pieces, one query piece) -- rows stay in registers, no LDS hand-over
    const WlSlot s_reg = wl_slot(a, min(tid < 256 ? k_lo + tid : q_lo + (tid & 127), N - 1));
    if (tid < 256) kcnK[tid] = tb.mc - s_reg.code;
    else if (tid < 384) kcnQ[tid - 256] = tb.mc - s_reg.code;
    const WlSlot s_k0 = wl_slot(a, min(k_lo + wave * 32 + (lane >> 2), N - 1));
    const WlSlot s_k1 = wl_slot(a, min(k_lo + wave * 32 + 16 + (lane >> 2), N - 1));
    const WlSlot s_q = wl_slot(a, min(q_lo + wave * 16 + (lane >> 2), N - 1));
    bf16x8 onehot[2];                                        // E_sl[i][k-slot (hi, e)] = 1 iff i == 16 sl + 8 (e >> 2) + 4 hi + (e & 3)
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        float e8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) e8[e] = (j == 16 * sl + 8 * (e >> 2) + 4 * hi + (e & 3)) ? 1.f : 0.f;
        onehot[sl] = pack_frag(e8);
    }
    f32x16 dsa[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dsa[t][r] = 0.f;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto issue = [&](int win, int buf) {                     // regions of the block's slots into LDS, operands by DMA
        const WlOrg o = wl_origin(a, win);
        int* rk = (int*)(smem + buf * WLB_BUF + 51200);
        int reg;
        (void)wl_row(a, o, s_reg, reg);
        if (tid < 384) rk[384 + tid] = reg;
        const unsigned d0 = lds0 + buf * WLB_BUF;
        const int lslot = (lane & 3) ^ ((lane >> 4) & 3);
        {
            const unsigned off0 = ((unsigned)wl_row(a, o, s_k0, reg) * (unsigned)ld + (unsigned)(C + head * HD + lslot * 8)) * 2u;
            const unsigned off1 = ((unsigned)wl_row(a, o, s_k1, reg) * (unsigned)ld + (unsigned)(C + head * HD + lslot * 8)) * 2u;
            wl_dma16(d0 + wave * 2048, a.qkv, off0);
            wl_dma16(d0 + 16384 + wave * 2048, a.qkv, off0 + 2 * C);
            wl_dma16(d0 + wave * 2048 + 1024, a.qkv, off1);
            wl_dma16(d0 + 16384 + wave * 2048 + 1024, a.qkv, off1 + 2 * C);
        }
        {
            const unsigned row = (unsigned)wl_row(a, o, s_q, reg);
            wl_dma16(d0 + 32768 + wave * 1024, a.qkv, (row * (unsigned)ld + (unsigned)(head * HD + lslot * 8)) * 2u);
            wl_dma16(d0 + 40960 + wave * 1024, a.dout, (row * (unsigned)C + (unsigned)(head * HD + lslot * 8)) * 2u);
        }
        const long lo = ((long)win * a.d.heads + head) * a.Npad;
        const unsigned el = (unsigned)min(q_lo + lane * 4, a.Npad - 4) * 4u;
        if (wave == 0) wl_dma16(d0 + 49152, a.lse + lo, el);
        if (wave == 1) wl_dma16(d0 + 50176, delta_in + lo, el);
    };
    issue(w_beg, 0);
    wl_dma_wait();
    __syncthreads();

    for (int win = w_beg; win < w_end; ++win) {
        const int cur = (win - w_beg) & 1;
        const char* Ks = smem + cur * WLB_BUF;
        const char* Vs = Ks + 16384;
        const char* Sq = Ks + 32768 + qs * 2048;
        const char* Sg = Ks + 40960 + qs * 2048;
        const float* lse_l = (const float*)(Ks + 49152);
        const float* dl_l = (const float*)(Ks + 50176);
        const int* krK = (const int*)(Ks + 51200) + 384;
        const int* krQ = krK + 256;
        if (win + 1 < w_end) issue(win + 1, cur ^ 1);
        if (strip_on) {
            const int ql = qs * 32 + j;
            bf16x8 qf[2], dof[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int off = krow_off<HD>(j, ks * 2 + hi);
                qf[ks] = *(const bf16x8*)(Sq + off); dof[ks] = *(const bf16x8*)(Sg + off);
            }
            // lse / delta: the DMA only clamps source elements past Npad, which no active strip reads
            const float nl = q_ok ? -lse_l[ql] : -INFINITY;
            const float nd = q_ok ? -dl_l[ql] : 0.f;
            const int q_code = tb.mc - kcnQ[ql], q_reg = krQ[ql];
            // shift mask needed?  wave-uniform: the strip's queries and the half's keys all in one region
            bool multi = false;
            if (a.d.sd | a.d.sh | a.d.sw) {
                const int r0 = krQ[qs * 32];
                multi = __any(q_reg != r0 || krK[kh * 128 + lane] != r0 || krK[kh * 128 + 64 + lane] != r0);
            }
            f32x16 ndl;
#pragma unroll
            for (int r = 0; r < 16; ++r) ndl[r] = nd;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int tl = kh * 4 + i, kt = kb * 8 + tl;
                if (kt >= nt) break;                         // wave-uniform
                const f32x16 z = WZERO16;
                f32x16 s, dp;
                {
                    const int o0 = krow_off<HD>(tl * 32 + j, hi), o1 = krow_off<HD>(tl * 32 + j, 2 + hi);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Ks + o0), qf[0], z, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Vs + o0), dof[0], ndl, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Ks + o1), qf[1], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Vs + o1), dof[1], dp, 0, 0, 0);
                }
                if (kt == nt - 1) wl_bias_rows_keys<true>(s, sc, tbl, kcnK, krK, tl * 32, hi, q_code, q_reg, multi, N - k_lo);
                else wl_bias_rows_keys<false>(s, sc, tbl, kcnK, krK, tl * 32, hi, q_code, q_reg, multi, N - k_lo);
                uint32_t dk[8];
#pragma unroll
                for (int r = 0; r < 16; r += 2) dk[r >> 1] = pack2(fast_exp2(s[r] + nl) * dp[r], fast_exp2(s[r + 1] + nl) * dp[r + 1]);
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    WFrag df; df.u = make_uint4(dk[4 * sl], dk[4 * sl + 1], dk[4 * sl + 2], dk[4 * sl + 3]);
                    dsa[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(onehot[sl], df.b, dsa[i], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);           // one tile's S / dP live at a time (the unrolled tiles otherwise interleave and spill)
            }
        }
        wl_dma_wait();
        __syncthreads();
    }
    // ---- flush: dense [128 q][256 k] matrix over the idle operand buffers, then per-class box sums ------------------------
    float* Dm = (float*)smem;
    if (strip_on && q_ok) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tl = kh * 4 + i;
            if (kb * 8 + tl >= nt) break;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = tl * 32 + tile_row(r, hi);
                if (k_lo + kl < N) Dm[(qs * 32 + j) * WLB_DLD + kl] = dsa[i][r];
            }
        }
    }
    __syncthreads();
    // window slot i < N  <->  (d, h, w) with the CONFIGURED (h, w) extents (relative_position_index[:N, :N], video_swin.py:153)
    const int ch = a.d.cfg_wh, cw = a.d.cfg_ww, chw = ch * cw;
    const int dv = (N + chw - 1) / chw;
    const int nw_ = 2 * cw - 1, nh_ = 2 * ch - 1;
    const int ncls = (2 * dv - 1) * nh_ * nw_;
    const int row_first = q_lo / cw, row_last = min((q_lo + 127) / cw, dv * ch - 1);
    for (int c = tid; c < ncls; c += 512) {
        const int dw = c % nw_ - (cw - 1), dh = (c / nw_) % nh_ - (ch - 1), dd = c / (nw_ * nh_) - (dv - 1);   // offset = q - k
        const int koff = dd * chw + dh * cw + dw;
        const int w0 = max(0, dw), w1 = min(cw, cw + dw);
        const int d0 = max(0, dd), d1 = min(dv, dv + dd), h0 = max(0, dh), h1 = min(ch, ch + dh);
        float part[2] = {0.f, 0.f};
        for (int rr = row_first; rr <= row_last; ++rr) {
            const int qd = rr / ch, qhh = rr - qd * ch;
            if (qd < d0 || qd >= d1 || qhh < h0 || qhh >= h1) continue;
            const int qrow = rr * cw;
            for (int qw = w0; qw < w1; ++qw) {
                const int qq = qrow + qw, ql = qq - q_lo, kk = qq - koff - k_lo;
                if ((unsigned)ql < 128u && qq < N && (unsigned)kk < 256u && kk + k_lo < N) part[qw & 1] += Dm[ql * WLB_DLD + kk];
            }
        }
        const float sum = part[0] + part[1];
        if (sum != 0.f) atomicAdd(a.dbias + (long)(dd * a.cstride_d + dh * a.cstride_h + dw + a.tbl_const) * a.d.heads + head, sum);
    }
}

// ------------------------------------------------------------------------------------------------------
template <typename Kn>
static void wl_lds(Kn k, size_t bytes) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    (void)hipGetLastError();
}

static const bool lav_winl_on = getenv("LAV_WINL") ? atoi(getenv("LAV_WINL")) != 0 : true;   // probe hook: 0 = generic kernels

static int lav_winl_parts = 0;                             // test hook: force the number of query parts per problem (0 = by problem count)
extern "C" int lav_winl_select(int parts) { const int old = lav_winl_parts; lav_winl_parts = parts; return old; }

static WlTab wl_tab(const AttnArgs& a) {
    const lav_attn_desc& d = a.d;
    const int i = a.N - 1;                                   // codes grow with the slot index: the last slot has the largest
    WlTab t;
    t.mc = (i / (d.cfg_wh * d.cfg_ww)) * a.cstride_d + ((i / d.cfg_ww) % d.cfg_wh) * a.cstride_h + (i % d.cfg_ww);
    t.lo = a.tbl_const - t.mc;
    t.n = 2 * t.mc + 1;
    return t;
}

bool winl_supported(const AttnArgs& a) {
    if (lav_winl_parts < 0) return false;                     // test hook: lav_winl_select(-1) routes large windows to the generic kernels
    if (!(lav_winl_on && a.d.mode == 0 && !a.d.comb && a.N <= WL_ROWS)) return false;
    const double qkv_bytes = (double)a.d.B * a.tps * 3.0 * a.C * 2.0;             // DMA offsets are 32-bit
    const WlTab t = wl_tab(a);
    return qkv_bytes < 4.0e9 && t.lo >= 0 && t.lo + t.n <= a.tbl_rows && (size_t)t.n * 4 + 2 * WL_IMG + 5 * WL_ROWS * 4 + 512 <= 160 * 1024 &&
           (size_t)t.n * 4 + WLB_MAIN + 384 * 4 + 512 <= 160 * 1024;
}

static int wl_parts(int problems) { return lav_winl_parts > 0 ? lav_winl_parts : problems >= 1024 ? 1 : 3; }

int winl_fwd_launch(void* stream, const AttnArgs& a, int nwin) {
    hipStream_t s = (hipStream_t)stream;
    const WlTab tb = wl_tab(a);
    const int QS = wl_parts(nwin * a.d.heads), items = nwin * a.d.heads * QS;
    const size_t lds = 2 * WL_IMG + 3 * WL_ROWS * 4 + (size_t)tb.n * 4;
    wl_lds(winl_fwd, lds);
    hipLaunchKernelGGL(winl_fwd, dim3(items < 256 ? items : 256), dim3(512), lds, s, a, nwin, QS, tb);
    return lav_check_launch("lav_attention_fwd(large window)");
}

int winl_dbias_launch(void* stream, const AttnArgs& a, int nwin, const float* delta);

int winl_bwd_launch(void* stream, const AttnArgs& a, int nwin, float* delta) {
    hipStream_t s = (hipStream_t)stream;
    const WlTab tb = wl_tab(a);
    const int QS = wl_parts(nwin * a.d.heads), items = nwin * a.d.heads * QS;
    const dim3 grid(items < 256 ? items : 256);
    const size_t lds1 = 2 * WL_IMG + 3 * WL_ROWS * 4 + (size_t)tb.n * 4;
    const size_t lds2 = 2 * WL_IMG + 5 * WL_ROWS * 4 + (size_t)tb.n * 4;
    wl_lds(winl_dq, lds1);
    hipLaunchKernelGGL(winl_dq, grid, dim3(512), lds1, s, a, nwin, QS, tb, delta);
    wl_lds(winl_dkv, lds2);
    hipLaunchKernelGGL(winl_dkv, grid, dim3(512), lds2, s, a, nwin, QS, tb, (const float*)delta);
    if (a.dbias) return winl_dbias_launch(stream, a, nwin, delta);
    return lav_check_launch("lav_attention_bwd(large window)");
}

// needs lse (forward) and delta (written by winl_dq on the same problems)
int winl_dbias_launch(void* stream, const AttnArgs& a, int nwin, const float* delta) {
    const WlTab tb = wl_tab(a);
    const int nqb = (a.nqt + 3) / 4, nkb = (a.nqt + 7) / 8;
    const int base = a.d.heads * nqb * nkb;
    int WS = (512 + base - 1) / base;
    if (WS > nwin) WS = nwin;
    if (WS < 1) WS = 1;
    const size_t lds = WLB_MAIN + 384 * 4 + (size_t)tb.n * 4;
    wl_lds(winl_dbias, lds);
    hipLaunchKernelGGL(winl_dbias, dim3(base * WS), dim3(512), lds, (hipStream_t)stream, a, nwin, WS, tb, delta);
    return lav_check_launch("lav_attention_bwd_bias(large window)");
}
