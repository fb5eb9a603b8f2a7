// Shared pieces of the attention kernels (forward, dQ pass, dK/dV pass).
//
// MFMA shape: v_mfma_f32_32x32x16_bf16.  Operand lane maps (lane l, hi = l>>5, j = l&31):
//   A[i=j][k=8hi..8hi+7], B[k=8hi..8hi+7][j], C/D[row=(r&3)+8(r>>2)+4hi][col=j], r in [0,16).
// All three kernels compute score tiles "transposed" so that the softmax row (one query, or one key in
// the dK/dV pass) is lane-local: a lane owns one column j and 16 rows of each 32x32 tile.
#pragma once
#include "common.h"
#include "../../include/lavender_hip.h"

struct AttnArgs {
    lav_attn_desc d;
    const bf16_t* qkv; const bf16_t* out; const bf16_t* dout;
    bf16_t* o_w; float* lse; bf16_t* dqkv; float* dbias;
    int N;            // tokens per problem (window volume or sequence length)
    int Npad;         // lse row stride
    int nqt;          // ceil(N/32)
    int nWd, nWh, nWw;
    int C;            // heads*head_dim
    int tbl_rows, tbl_const, cstride_d, cstride_h;   // bias-table geometry
    int R;            // 32-row tiles per wave
    uint32_t thresh;
};

__device__ __forceinline__ int tile_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

struct TokInfo { int row; int code; int region; };

// window w, in-window index i  ->  global token row (un-rolled tensor), bias code, shift region
__device__ __forceinline__ TokInfo win_token(const AttnArgs& a, int win, int i) {
    const lav_attn_desc& d = a.d;
    int wwi = win % a.nWw, t = win / a.nWw;
    int whi = t % a.nWh; t /= a.nWh;
    int wdi = t % a.nWd, b = t / a.nWd;
    int wi = i % d.ww, t2 = i / d.ww;
    int hi = t2 % d.wh, di = t2 / d.wh;
    int pd = wdi * d.wd + di, ph = whi * d.wh + hi, pw = wwi * d.ww + wi;     // coordinates in the rolled tensor
    int sd_ = pd + d.sd, sh_ = ph + d.sh, sw_ = pw + d.sw;                    // roll(-shift): rolled[p] = x[(p+s) % n]
    if (sd_ >= d.D) sd_ -= d.D;
    if (sh_ >= d.H) sh_ -= d.H;
    if (sw_ >= d.W) sw_ -= d.W;
    TokInfo o;
    o.row = ((b * d.D + sd_) * d.H + sh_) * d.W + sw_;
    o.code = di * a.cstride_d + hi * a.cstride_h + wi;
    int rd = d.sd ? (pd >= d.D - d.wd) + (pd >= d.D - d.sd) : 0;
    int rh = d.sh ? (ph >= d.H - d.wh) + (ph >= d.H - d.sh) : 0;
    int rw = d.sw ? (pw >= d.W - d.ww) + (pw >= d.W - d.sw) : 0;
    o.region = rd * 9 + rh * 3 + rw;
    return o;
}

// K-type LDS tile: rows of HD bf16 (d contiguous), 16-byte slots XOR-swizzled so that the 32x32 A-operand
// read (32 rows x one slot) is bank-conflict-free.
template <int HD>
__device__ __forceinline__ int krow_off(int row, int slot) {
    if (HD == 32) return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4);
    return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
}

__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) {
    union { uint4 u; bf16x8 b; } x; x.u = v; return x.b;
}
__device__ __forceinline__ bf16x8 pack_frag(const float* p) {
    union { uint4 u; bf16x8 b; } x; x.u = pack8(p); return x.b;
}
__device__ __forceinline__ bf16x8 frag2(uint2 lo, uint2 hi) {
    union { uint4 u; bf16x8 b; } x; x.u = make_uint4(lo.x, lo.y, hi.x, hi.y); return x.b;
}
