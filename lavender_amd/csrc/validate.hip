// fp32-I/O VALIDATION kernels (tier T2 of SURVEY.md section 8c; BASELINE.json north_star: "MLM logits within 1e-3
// of reference").  The production path stores activations in bf16 and feeds bf16 operands to the matrix cores; no
// such implementation can meet 1e-3 on logits of rms 0.5 after 36 blocks.  This file is the same forward path with
// fp32 activations end to end: GEMMs on the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32 (an fmaf chain,
// 157 TFLOP/s peak -- speed is irrelevant here), attention / embeddings in plain fp32 VALU code, LayerNorm through
// the fp32 I/O mode of lav_layernorm_fwd.  Forward only, eval mode (no dropout / drop-path); selected with
// args.validate_fp32 / LAV_FP32=1 (lavender_amd/validate.py).  Also the only kernels that implement the zero-pad
// branch of a Swin block for EVERY geometry (video_swin.py:211-215,241-242), so they double as an on-device
// cross-check of the fast kernels.
#include "common.h"
#include <math.h>
#include "../../include/lavender_hip.h"

// ---- GEMM: C[M,N] = act(A[M,K] . B[N,K]^T + bias) + residual, all fp32 row-major -------------------------------
__global__ __launch_bounds__(256) void v_gemm_kernel(int M, int N, int K, const float* __restrict__ A, long lda,
                                                    const float* __restrict__ B, long ldb, float* __restrict__ C, long ldc,
                                                    const float* __restrict__ bias, int act, const float* __restrict__ res, long ldr) {
    __shared__ float As[64][17], Bs[64][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int lrow = tid >> 2, lc0 = (tid & 3) * 4;
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int gk = k0 + lc0 + e;
            As[lrow][lc0 + e] = (m0 + lrow < M && gk < K) ? A[(long)(m0 + lrow) * lda + gk] : 0.f;
            Bs[lrow][lc0 + e] = (n0 + lrow < N && gk < K) ? B[(long)(n0 + lrow) * ldb + gk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            const float a = As[wm * 32 + (lane & 31)][kk + (lane >> 5)];
            const float b = Bs[wn * 32 + (lane & 31)][kk + (lane >> 5)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = n0 + wn * 32 + (lane & 31);
    if (col >= N) return;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= M) continue;
        float v = acc[r] + bv;
        if (act == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));          // exact (erf) GELU
        else if (act == 2) v = fmaxf(v, 0.f);
        if (res) v += res[(long)row * ldr + col];
        C[(long)row * ldc + col] = v;
    }
}

extern "C" int lav_v_gemm_f32(void* stream, int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C,
                              long ldc, const float* bias, int act, const float* residual, long ldr) {
    LAV_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C, "lav_v_gemm_f32: bad arguments M=%d N=%d K=%d", M, N, K);
    hipLaunchKernelGGL(v_gemm_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, (hipStream_t)stream, M, N, K, A, lda, B, ldb,
                       C, ldc, bias, act, residual, ldr);
    return lav_check_launch("lav_v_gemm_f32");
}

// ---- attention, fp32, any geometry -----------------------------------------------------------------------------
// window mode: the token grid (D,H,W) is zero-padded to multiples of the window AFTER norm1 (video_swin.py:211-215): a
// padded token's q/k/v row is therefore the qkv bias (`pad_qkv`, 3C floats); roll, partition, reverse, roll back and the
// crop (:241-242) are index math.  The shift mask follows compute_mask over the PADDED grid (:290-305).
struct VAttn {
    lav_attn_desc d;
    const float* qkv; float* out; const float* pad_qkv;
    int N, C, Dp, Hp, Wp, nWd, nWh, nWw, cstride_d, cstride_h, tbl_const;
};

__global__ __launch_bounds__(256) void v_attn_kernel(VAttn a) {
    extern __shared__ float sm[];                          // [4 waves][Npad] scores | krow | kcode | kreg
    const lav_attn_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int prob = blockIdx.x / d.heads, head = blockIdx.x % d.heads;
    const int N = a.N, hd = d.head_dim, C = a.C, ld = 3 * C;
    const int Npad = (N + 63) & ~63;
    float* sc = sm + wave * Npad;
    int* krow = (int*)(sm + 4 * Npad);
    int* kcode = krow + Npad;
    int* kreg = kcode + Npad;
    for (int i = tid; i < N; i += 256) {
        int row, code = 0, reg = 0;
        if (d.mode == 0) {
            int wwi = prob % a.nWw, t = prob / a.nWw;
            int whi = t % a.nWh; t /= a.nWh;
            int wdi = t % a.nWd, b = t / a.nWd;
            int wi = i % d.ww, t2 = i / d.ww;
            int hi = t2 % d.wh, di = t2 / d.wh;
            int pd = wdi * d.wd + di, ph = whi * d.wh + hi, pw = wwi * d.ww + wi;      // coordinates in the rolled, padded grid
            int sd_ = (pd + d.sd) % a.Dp, sh_ = (ph + d.sh) % a.Hp, sw_ = (pw + d.sw) % a.Wp;   // roll(-shift)
            row = (sd_ < d.D && sh_ < d.H && sw_ < d.W) ? ((b * d.D + sd_) * d.H + sh_) * d.W + sw_ : -1;
            // the reference slices relative_position_index[:N, :N] of the CONFIGURED window (video_swin.py:153): in-window
            // index i is decoded with the configured (h, w) extents, whatever the clamped window is
            code = (i / (d.cfg_wh * d.cfg_ww)) * a.cstride_d + ((i / d.cfg_ww) % d.cfg_wh) * a.cstride_h + (i % d.cfg_ww);
            int rd = d.sd ? (pd >= a.Dp - d.wd) + (pd >= a.Dp - d.sd) : 0;
            int rh = d.sh ? (ph >= a.Hp - d.wh) + (ph >= a.Hp - d.sh) : 0;
            int rw = d.sw ? (pw >= a.Wp - d.ww) + (pw >= a.Wp - d.sw) : 0;
            reg = rd * 9 + rh * 3 + rw;
        } else {
            row = prob * N + i;
            reg = (d.key_mask && d.key_mask[(long)prob * N + i] == 0) ? 1 : 0;          // 1 = masked key
        }
        krow[i] = row; kcode[i] = code; kreg[i] = reg;
    }
    __syncthreads();
    for (int q = wave; q < N; q += 4) {
        const int qrow = krow[q];
        if (qrow < 0) continue;                            // padded query: cropped away (video_swin.py:241-242)
        const float* qp = a.qkv + (long)qrow * ld + head * hd;
        float mx = -INFINITY;
        for (int k = lane; k < Npad; k += 64) {
            float s = -INFINITY;
            if (k < N) {
                const int kr = krow[k];
                const float* kp = kr >= 0 ? a.qkv + (long)kr * ld + C + head * hd : a.pad_qkv + C + head * hd;
                float dot = 0.f;
                for (int e = 0; e < hd; ++e) dot = fmaf(qp[e] * d.scale, kp[e], dot);   // q * scale first, as the reference (:150-151)
                if (d.mode == 0) {
                    s = dot + d.bias_table[(long)(kcode[q] - kcode[k] + a.tbl_const) * d.heads + head];
                    if (kreg[q] != kreg[k]) s += -100.0f;
                } else {
                    s = kreg[k] ? -INFINITY : dot;
                    if (d.causal_from > 0 && k >= d.causal_from && (q < d.causal_from || k > q)) s = -INFINITY;   // seq2seq mask
                }
            }
            sc[k] = s;
            mx = fmaxf(mx, s);
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int k = lane; k < Npad; k += 64) {
            const float p = k < N ? expf(sc[k] - mx) : 0.f;
            sc[k] = p;
            sum += p;
        }
        sum = wave_sum(sum);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (lane < hd) {
            float o = 0.f;
            for (int k = 0; k < N; ++k) {
                const int kr = krow[k];
                const float* vp = kr >= 0 ? a.qkv + (long)kr * ld + 2 * C + head * hd : a.pad_qkv + 2 * C + head * hd;
                o = fmaf(sc[k], vp[lane], o);
            }
            a.out[(long)qrow * C + head * hd + lane] = o / sum;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

extern "C" int lav_v_attention_f32(void* stream, const lav_attn_desc* d, const float* qkv, float* out, const float* pad_qkv) {
    LAV_REQUIRE(d && qkv && out, "lav_v_attention_f32: null pointer");
    VAttn a;
    memset(&a, 0, sizeof(a));
    a.d = *d; a.qkv = qkv; a.out = out; a.pad_qkv = pad_qkv;
    a.C = d->heads * d->head_dim;
    LAV_REQUIRE(d->head_dim > 0 && d->head_dim <= 64, "lav_v_attention_f32: head_dim %d (max 64)", d->head_dim);
    int problems;
    if (d->mode == 0) {
        LAV_REQUIRE(d->wd > 0 && d->wh > 0 && d->ww > 0 && d->bias_table, "lav_v_attention_f32: bad window descriptor");
        a.N = d->wd * d->wh * d->ww;
        a.Dp = (d->D + d->wd - 1) / d->wd * d->wd; a.Hp = (d->H + d->wh - 1) / d->wh * d->wh; a.Wp = (d->W + d->ww - 1) / d->ww * d->ww;
        LAV_REQUIRE((a.Dp == d->D && a.Hp == d->H && a.Wp == d->W) || pad_qkv, "lav_v_attention_f32: padded geometry needs pad_qkv (the qkv bias)");
        a.nWd = a.Dp / d->wd; a.nWh = a.Hp / d->wh; a.nWw = a.Wp / d->ww;
        problems = d->B * a.nWd * a.nWh * a.nWw;
        a.cstride_h = 2 * d->cfg_ww - 1;
        a.cstride_d = (2 * d->cfg_wh - 1) * a.cstride_h;
        a.tbl_const = (d->cfg_wd - 1) * a.cstride_d + (d->cfg_wh - 1) * a.cstride_h + (d->cfg_ww - 1);
    } else {
        LAV_REQUIRE(d->n_seq > 0 && d->L > 0, "lav_v_attention_f32: bad sequence descriptor");
        a.N = d->L; problems = d->n_seq;
    }
    LAV_REQUIRE(a.N <= 2048, "lav_v_attention_f32: N=%d too long", a.N);
    const int Npad = (a.N + 63) & ~63;
    const size_t lds = (size_t)Npad * 7 * 4;
    if (lds > 65536) { (void)hipFuncSetAttribute((const void*)v_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); (void)hipGetLastError(); }
    hipLaunchKernelGGL(v_attn_kernel, dim3(problems * d->heads), dim3(256), lds, (hipStream_t)stream, a);
    return lav_check_launch("lav_v_attention_f32");
}

// ---- PatchEmbed3D im2col, fp32 rows (same column order as lav_patch_im2col) ---------------------------------------
__global__ __launch_bounds__(256) void v_im2col_kernel(const float* __restrict__ img, int B, int T, int H, int W, int frame_major,
                                                      float* __restrict__ out) {
    const int Hp = H >> 2, Wp = W >> 2;
    const long total = (long)B * T * Hp * Wp * 96;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int col = idx % 96; const long row = idx / 96;
        const int pw = row % Wp; long t1 = row / Wp;
        const int ph = t1 % Hp; t1 /= Hp;
        const int t = t1 % T, b = t1 / T;
        const int c = col >> 5, kt = (col >> 4) & 1, kh = (col >> 2) & 3, kw = col & 3;
        const int ts = t + kt;
        float v = 0.f;
        if (ts < T) {
            const long plane = frame_major ? (((long)b * T + ts) * 3 + c) : (((long)b * 3 + c) * T + ts);
            v = img[plane * H * W + (long)(4 * ph + kh) * W + 4 * pw + kw];
        }
        out[idx] = v;
    }
}

extern "C" int lav_v_im2col_f32(void* stream, const float* img, int B, int T, int H, int W, int frame_major, float* out) {
    LAV_REQUIRE(img && out && B > 0 && T > 0 && H % 4 == 0 && W % 4 == 0, "lav_v_im2col_f32: bad arguments");
    hipLaunchKernelGGL(v_im2col_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, img, B, T, H, W, frame_major, out);
    return lav_check_launch("lav_v_im2col_f32");
}

// ---- rows: out[r] = LayerNorm(src(r)) with the row built on the fly; one wave per row, Hd <= 1024 ------------------
// kind 0: video token assembly (model.py:69-85): rows (b,t,0) = emb_cls, (b,t,1+p) = feat[b,t,p]; + emb_pos + emb_len
// kind 1: BERT text embedding (model.py:125-129): word[id] + pos[x] + type0
struct VRows {
    int kind, B, T, hw, Hd, n, X;
    const float* feat; const float* cls; const float* pos; const float* len;
    const int64_t* ids; const float* word; const float* type0;
    const float* gamma; const float* beta; float eps;
    float* out; long seq_rows;
};

__global__ __launch_bounds__(256) void v_rows_kernel(VRows a) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int P = 1 + a.hw;
    const long rows = a.kind == 0 ? (long)a.B * a.T * P : (long)a.n * a.X;
    if (r >= rows) return;
    float v[16];
    float s = 0.f;
    long orow;
    if (a.kind == 0) {
        const int pp = r % P; const int bt = r / P; const int t = bt % a.T, b = bt / a.T;
        orow = (long)b * a.seq_rows + (long)t * P + pp;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = lane + 64 * i;
            v[i] = 0.f;
            if (c < a.Hd) {
                const float x = pp == 0 ? a.cls[c] : a.feat[((long)bt * a.hw + pp - 1) * a.Hd + c];
                v[i] = x + a.pos[(long)pp * a.Hd + c] + a.len[(long)t * a.Hd + c];
                s += v[i];
            }
        }
    } else {
        const int xp = r % a.X;
        const long id = a.ids[r];
        orow = r;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = lane + 64 * i;
            v[i] = 0.f;
            if (c < a.Hd) { v[i] = a.word[id * a.Hd + c] + a.type0[c] + a.pos[(long)xp * a.Hd + c]; s += v[i]; }
        }
    }
    const float mean = wave_sum(s) / a.Hd;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int c = lane + 64 * i; if (c < a.Hd) { const float dd = v[i] - mean; q += dd * dd; } }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / a.Hd + a.eps);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        if (c < a.Hd) a.out[orow * a.Hd + c] = (v[i] - mean) * rstd * a.gamma[c] + a.beta[c];
    }
}

extern "C" int lav_v_video_embed_f32(void* stream, int B, int T, int hw, int Hd, const float* feat, const float* emb_cls,
                                     const float* emb_pos, const float* emb_len, const float* gamma, const float* beta, float eps,
                                     float* out, long seq_rows) {
    LAV_REQUIRE(B > 0 && T > 0 && hw > 0 && Hd > 0 && Hd <= 1024 && feat && out, "lav_v_video_embed_f32: bad arguments");
    VRows a; memset(&a, 0, sizeof(a));
    a.kind = 0; a.B = B; a.T = T; a.hw = hw; a.Hd = Hd; a.feat = feat; a.cls = emb_cls; a.pos = emb_pos; a.len = emb_len;
    a.gamma = gamma; a.beta = beta; a.eps = eps; a.out = out; a.seq_rows = seq_rows;
    const long rows = (long)B * T * (1 + hw);
    hipLaunchKernelGGL(v_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    return lav_check_launch("lav_v_video_embed_f32");
}

extern "C" int lav_v_text_embed_f32(void* stream, int n, int X, int Hd, const int64_t* ids, const float* word, const float* pos,
                                    const float* type0, const float* gamma, const float* beta, float eps, float* out) {
    LAV_REQUIRE(n > 0 && X > 0 && Hd > 0 && Hd <= 1024 && ids && out, "lav_v_text_embed_f32: bad arguments");
    VRows a; memset(&a, 0, sizeof(a));
    a.kind = 1; a.n = n; a.X = X; a.Hd = Hd; a.ids = ids; a.word = word; a.pos = pos; a.type0 = type0;
    a.gamma = gamma; a.beta = beta; a.eps = eps; a.out = out;
    const long rows = (long)n * X;
    hipLaunchKernelGGL(v_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    return lav_check_launch("lav_v_text_embed_f32");
}

// ---- row gather, fp32 (src_row < 0 -> zeros) -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void v_gather_kernel(long n_rows, int C, const float* src, long lds_, const int32_t* src_row,
                                                      float* dst, long ldd) {
    const long total = n_rows * C;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = idx % C; const long r = idx / C;
        const int sr = src_row[r];
        dst[r * ldd + c] = sr >= 0 ? src[(long)sr * lds_ + c] : 0.f;
    }
}

extern "C" int lav_v_gather_rows_f32(void* stream, int n_rows, int C, const float* src, long lds_, const int32_t* src_row,
                                     float* dst, long ldd) {
    LAV_REQUIRE(n_rows > 0 && C > 0 && src && src_row && dst, "lav_v_gather_rows_f32: bad arguments");
    hipLaunchKernelGGL(v_gather_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, (long)n_rows, C, src, lds_, src_row, dst, ldd);
    return lav_check_launch("lav_v_gather_rows_f32");
}
