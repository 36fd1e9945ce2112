// xq_f32.hip — fp32 forward kernels of the encoder / decoder layers: the REFERENCE-PARITY path (inference, no autograd).
//
// BASELINE.json asks for reconstructions within 1e-4 (fp32) of the reference CPU path.  The training kernels compute in bf16
// (as the reference does under autocast) and cannot sit on that path; these do: every matrix product is an exact fp32 fma
// chain on v_mfma_f32_32x32x2_f32 (1/16 of the bf16 MFMA rate — irrelevant at the batch sizes parity is checked at), norms
// and softmax use IEEE expf / sqrtf / division.  With them tests/test_model_parity.py runs the whole tokenizer on hand-written
// kernels end to end.
//   conv2d_f32_kernel    : implicit-GEMM convolution, NHWC, kernel 1x1 / 3x3, stride 1 / 2, explicit top/left padding (bottom /
//                          right implied by the output size: the (0,1,0,1) pad of Downsample, xqgan_model.py:697-704), optional
//                          nearest-2x upsampling of the input folded into the gather (Upsample, :682-686), any channel counts
//                          (conv_in 3 -> 128, conv_out 128 -> 3); with H = W = 1 it is nn.Linear (ViT blocks, patch embedding).
//   attention_f32_kernel : softmax(q k^T * scale) v for one (batch, head, query) per wave — the ViT's SDPA
//                          (vision_transformer.py:175-195) and the CNN AttnBlock (xqgan_model.py:635-659, one head of C dims).
//   groupnorm_silu_f32   : GroupNorm(32, eps 1e-6) [+ x * sigmoid(x)] (xqgan_model.py:662-672), two-pass statistics in double.
//   gemm_f32_tn_kernel   : dW = g^T x of nn.Linear (round 4): with conv2d_f32_kernel on W / W^T the three products of a Linear layer's
//                          fp32 TRAINING step, so that the fp32 leg of the gradient-parity tests exercises hand-written kernels.
#include "xq_common.hpp"
#include "xq_internal.hpp"
#include "../../include/xq_ops.h"

using namespace xq;

namespace {

struct ConvF32 {
    const float *X, *Wp, *bias;
    float *Y;
    int B, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW, stride, pad_t, pad_l, up;   // up = 1: the conv sees the 2x nearest-upsampled input
    long M;        // B * Ho * Wo
    int K;         // KH * KW * Cin
};

// 64 output pixels x 64 output channels per block, 4 waves of 32 x 32, K step 16
__global__ __launch_bounds__(256) void conv2d_f32_kernel(const ConvF32 p) {
    __shared__ float As[64][17], Bs[64][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long m0 = (long)blockIdx.x * 64;
    const int n0 = blockIdx.y * 64;
    const int Hl = p.up ? 2 * p.Hi : p.Hi, Wl = p.up ? 2 * p.Wi : p.Wi;     // logical input size

    // staging: thread -> row r = tid / 4, k offsets (tid % 4) * 4 .. + 3
    const int r = tid >> 2, kq = (tid & 3) * 4;
    const long m = m0 + r;
    const bool mok = m < p.M;
    const long mm = mok ? m : 0;
    const int ox = (int)(mm % p.Wo), oy = (int)((mm / p.Wo) % p.Ho);
    const long b = mm / ((long)p.Wo * p.Ho);
    const int n = n0 + r;
    const bool nok = n < p.Cout;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;

    for (int k0 = 0; k0 < p.K; k0 += 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + kq + j;
            float a = 0.0f, w = 0.0f;
            if (k < p.K) {
                const int tap = k / p.Cin, c = k - tap * p.Cin;
                const int ky = tap / p.KW, kx = tap - ky * p.KW;
                const int yy = oy * p.stride + ky - p.pad_t, xx = ox * p.stride + kx - p.pad_l;
                if (mok && yy >= 0 && yy < Hl && xx >= 0 && xx < Wl)
                    a = p.X[((b * p.Hi + (p.up ? yy >> 1 : yy)) * p.Wi + (p.up ? xx >> 1 : xx)) * p.Cin + c];
                if (nok) w = p.Wp[(long)n * p.K + k];
            }
            As[r][kq + j] = a;
            Bs[r][kq + j] = w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            const float a = As[wm * 32 + (lane & 31)][kk + (lane >> 5)];
            const float w = Bs[wn * 32 + (lane & 31)][kk + (lane >> 5)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // D[i = pixel][j = channel]: j = lane & 31, i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int co = n0 + wn * 32 + (lane & 31);
    if (co < p.Cout) {
        const float bv = p.bias ? p.bias[co] : 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const long mo = m0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
            if (mo < p.M) p.Y[mo * p.Cout + co] = acc[q] + bv;
        }
    }
}

// one wave per (batch b, head h, query i): scores over all keys, softmax, weighted sum of the values.
// q / k / v: element offset of (b, token, head) = b * batch_stride + token * tok_stride + h * hd; out row = (b * N + i) * (H * hd) + h * hd
__global__ __launch_bounds__(256) void attention_f32_kernel(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                                                            int B, int N, int H, int hd, long batch_stride, long tok_stride, float scale,
                                                            float *__restrict__ out, float *__restrict__ lse) {
    extern __shared__ float sm[];          // [4 waves][hd + N]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long item = (long)blockIdx.x * 4 + wave;            // (b, h, i) flattened, i fastest
    if (item >= (long)B * H * N) return;                       // whole wave exits together (no block barrier below)
    float *qs = sm + (size_t)wave * (hd + N), *ps = qs + hd;
    const int i = (int)(item % N), h = (int)((item / N) % H);
    const long b = item / ((long)N * H);
    const float *qp = q + b * batch_stride + (long)i * tok_stride + (long)h * hd;
    for (int d = lane; d < hd; d += 64) qs[d] = qp[d];
    __builtin_amdgcn_wave_barrier();
    // scores
    float mx = -__builtin_inff();
    for (int j = lane; j < N; j += 64) {
        const float *kp = k + b * batch_stride + (long)j * tok_stride + (long)h * hd;
        float s = 0.0f;
        for (int d = 0; d < hd; ++d) s = __builtin_fmaf(qs[d], kp[d], s);
        s *= scale;
        ps[j] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.0f;
    for (int j = lane; j < N; j += 64) {
        const float e = expf(ps[j] - mx);
        ps[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_wave_barrier();
    const float inv = 1.0f / sum;
    if (lse != nullptr && lane == 0) lse[(b * H + h) * (long)N + i] = mx + logf(sum);      // for the backward kernels: p_ij = exp(s_ij - lse_i)
    float *op = out + ((b * N + i) * (long)H + h) * hd;
    for (int d = lane; d < hd; d += 64) {
        float o = 0.0f;
        const float *vp = v + b * batch_stride + (long)h * hd + d;
        for (int j = 0; j < N; ++j) o = __builtin_fmaf(ps[j] * inv, vp[(long)j * tok_stride], o);
        op[d] = o;
    }
}

// Backward of the same attention in fp32 (round 4: the fp32 TRAINING leg of the parity tests), head dim <= 64, lane = channel d.
// With p_ij = exp(scale q_i.k_j - lse_i), delta_i = dO_i.O_i, dP_ij = dO_i.v_j, dS_ij = p_ij (dP_ij - delta_i):
//   attention_f32_bwd_q_kernel  : one wave per (b, h, query i), keys in ascending order:  dq_i = scale sum_j dS_ij k_j;  writes delta_i
//   attention_f32_bwd_kv_kernel : one wave per (b, h, key j), queries in ascending order: dk_j = scale sum_i dS_ij q_i,  dv_j = sum_i p_ij dO_i
// Every dot product is a butterfly sum over the 64 lanes, every output one ascending chain: deterministic, no atomics.  ~40 wave
// instructions per (query, key) pair and direction — a parity path (milliseconds at the sizes the goldens use), not a fast one.
__device__ __forceinline__ float f32_bfly_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void attention_f32_bwd_q_kernel(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                                                                  const float *__restrict__ out, const float *__restrict__ dout,
                                                                  const float *__restrict__ lse, int B, int N, int H, int hd, long batch_stride,
                                                                  long tok_stride, float scale, float *__restrict__ dq, float *__restrict__ delta) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= (long)B * H * N) return;
    const int i = (int)(item % N), h = (int)((item / N) % H);
    const long b = item / ((long)N * H);
    const bool on = lane < hd;
    const long qoff = b * batch_stride + (long)i * tok_stride + (long)h * hd + lane;
    const long ooff = ((b * N + i) * (long)H + h) * hd + lane;
    const float qd = on ? q[qoff] : 0.0f, dod = on ? dout[ooff] : 0.0f, od = on ? out[ooff] : 0.0f;
    const float dl = f32_bfly_sum(dod * od), ls = lse[(b * H + h) * (long)N + i];
    if (lane == 0) delta[(b * H + h) * (long)N + i] = dl;
    float acc = 0.0f;
    const float *kp = k + b * batch_stride + (long)h * hd + lane, *vp = v + b * batch_stride + (long)h * hd + lane;
    for (int j = 0; j < N; ++j) {
        const float kd = on ? kp[(long)j * tok_stride] : 0.0f, vd = on ? vp[(long)j * tok_stride] : 0.0f;
        const float sc = f32_bfly_sum(qd * kd) * scale;
        const float pj = expf(sc - ls);
        const float ds = pj * (f32_bfly_sum(dod * vd) - dl);
        acc = __builtin_fmaf(ds, kd, acc);
    }
    if (on) dq[qoff] = acc * scale;
}

__global__ __launch_bounds__(256) void attention_f32_bwd_kv_kernel(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                                                                   const float *__restrict__ dout, const float *__restrict__ lse,
                                                                   const float *__restrict__ delta, int B, int N, int H, int hd, long batch_stride,
                                                                   long tok_stride, float scale, float *__restrict__ dk, float *__restrict__ dv) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= (long)B * H * N) return;
    const int j = (int)(item % N), h = (int)((item / N) % H);
    const long b = item / ((long)N * H);
    const bool on = lane < hd;
    const long koff = b * batch_stride + (long)j * tok_stride + (long)h * hd + lane;
    const float kd = on ? k[koff] : 0.0f, vd = on ? v[koff] : 0.0f;
    float ak = 0.0f, av = 0.0f;
    const float *qp = q + b * batch_stride + (long)h * hd + lane;
    const float *dop = dout + (b * N * (long)H + h) * hd + lane;
    const float *lp = lse + (b * H + h) * (long)N, *dp = delta + (b * H + h) * (long)N;
    for (int i = 0; i < N; ++i) {
        const float qd = on ? qp[(long)i * tok_stride] : 0.0f, dod = on ? dop[(long)i * H * hd] : 0.0f;
        const float sc = f32_bfly_sum(qd * kd) * scale;
        const float pi = expf(sc - lp[i]);
        const float ds = pi * (f32_bfly_sum(dod * vd) - dp[i]);
        av = __builtin_fmaf(pi, dod, av);
        ak = __builtin_fmaf(ds, qd, ak);
    }
    if (on) { dk[koff] = ak * scale; dv[koff] = av; }
}

// one block per (sample, group): x [B][HW][C] -> y, statistics in double
__global__ __launch_bounds__(256) void groupnorm_silu_f32_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                                 int HW, int C, int G, float eps, int silu, float *__restrict__ y) {
    __shared__ double red[256];
    const int b = blockIdx.y, g = blockIdx.x, cg = C / G;
    const float *xb = x + (long)b * HW * C + g * cg;
    float *yb = y + (long)b * HW * C + g * cg;
    const long n = (long)HW * cg;
    double s = 0.0;
    for (long e = threadIdx.x; e < n; e += 256) s += (double)xb[(e / cg) * C + (e % cg)];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    const double mean = red[0] / (double)n;
    __syncthreads();
    s = 0.0;
    for (long e = threadIdx.x; e < n; e += 256) { const double d = (double)xb[(e / cg) * C + (e % cg)] - mean; s += d * d; }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    const float rstd = (float)(1.0 / sqrt(red[0] / (double)n + (double)eps));
    const float mu = (float)mean;
    for (long e = threadIdx.x; e < n; e += 256) {
        const int c = (int)(e % cg);
        const long o = (e / cg) * C + c;
        float pre = (xb[o] - mu) * rstd;
        pre = pre * (w ? w[g * cg + c] : 1.0f) + (bias ? bias[g * cg + c] : 0.0f);
        yb[o] = silu ? pre / (1.0f + expf(-pre)) : pre;
    }
}

// Weight gradient of nn.Linear in fp32 (round 4: the fp32 TRAINING leg of the parity tests runs its products on these kernels — forward
// and data gradient are conv2d_f32_kernel with the weight / its transpose):  C[i][j] = sum_m A[m][i] * B[m][j], A [M][Na] = the output
// gradient, B [M][Nb] = the layer input, C [Na][Nb] = dW.  64 x 64 outputs per block, 4 waves of 32 x 32, 16 rows of m per LDS stage; the
// sum over m is ONE ascending fp32 fma chain per output (v_mfma_f32_32x32x2_f32 consumes m, m + 1 per instruction, in order): deterministic,
// and the order a sequential CPU loop over the rows would use.  Blocks are not split over m — this is a parity path, not a fast one.
__global__ __launch_bounds__(256) void gemm_f32_tn_kernel(const float *__restrict__ A, const float *__restrict__ Bm, long M, int Na, int Nb,
                                                          float *__restrict__ C) {
    __shared__ float As[16][68], Bs[16][68];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    const int r = tid >> 4, cq = (tid & 15) * 4;      // staging: row r of the 16, columns cq .. cq + 3
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
    for (long m0 = 0; m0 < M; m0 += 16) {
        const long m = m0 + r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ia = i0 + cq + e, jb = j0 + cq + e;
            As[r][cq + e] = (m < M && ia < Na) ? A[m * Na + ia] : 0.0f;
            Bs[r][cq + e] = (m < M && jb < Nb) ? Bm[m * Nb + jb] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            const float a = As[kk + (lane >> 5)][wi * 32 + (lane & 31)];
            const float b = Bs[kk + (lane >> 5)][wj * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // D[i][j]: j = lane & 31, i = (q & 3) + 8 (q >> 2) + 4 (lane >> 5)
    const int j = j0 + wj * 32 + (lane & 31);
    if (j < Nb) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = i0 + wi * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
            if (i < Na) C[(long)i * Nb + j] = acc[q];
        }
    }
}

__global__ __launch_bounds__(256) void pack_conv_weights_f32_kernel(const float *__restrict__ W, int Cout, int Cin, int KH, int KW,
                                                                    float *__restrict__ Wp) {
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)Cout * Cin * KH * KW;
    if (o >= total) return;
    // Wp[n][(ky * KW + kx) * Cin + c] = W[n][c][ky][kx]
    const int c = (int)(o % Cin);
    const int tap = (int)((o / Cin) % (KH * KW));
    const long n = o / ((long)Cin * KH * KW);
    Wp[o] = W[((n * Cin + c) * KH + tap / KW) * KW + tap % KW];
}

}  // namespace

extern "C" int xq_conv2d_f32_pack_weights(const float *w_oihw, int Cout, int Cin, int KH, int KW, float *w_packed, xq_stream_t stream) {
    const char *fn = "xq_conv2d_f32_pack_weights";
    if (Cout < 1 || Cin < 1 || KH < 1 || KW < 1 || !w_oihw || !w_packed) return xq_set_error(XQ_EINVAL, "%s: bad argument", fn);
    const long total = (long)Cout * Cin * KH * KW;
    hipLaunchKernelGGL(pack_conv_weights_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin, KH,
                       KW, w_packed);
    return xq_check_launch(fn);
}

extern "C" int xq_conv2d_f32_nhwc(const float *x, const float *w_packed, const float *bias, int B, int Hi, int Wi, int Cin, int Cout, int KH,
                                  int KW, int stride, int pad_top, int pad_left, int Ho, int Wo, int upsample2x, float *y, xq_stream_t stream) {
    const char *fn = "xq_conv2d_f32_nhwc";
    if (B < 0 || Hi < 1 || Wi < 1 || Cin < 1 || Cout < 1 || Ho < 1 || Wo < 1) return xq_set_error(XQ_EINVAL, "%s: bad shape", fn);
    if (!((KH == 1 && KW == 1) || (KH == 3 && KW == 3)) || (stride != 1 && stride != 2) || pad_top < 0 || pad_left < 0)
        return xq_set_error(XQ_EINVAL, "%s: kernel 1x1 / 3x3, stride 1 / 2 only", fn);
    if (B == 0) return XQ_OK;
    if (!x || !w_packed || !y) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    ConvF32 p{x, w_packed, bias, y, B, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW, stride, pad_top, pad_left, upsample2x ? 1 : 0, (long)B * Ho * Wo,
              KH * KW * Cin};
    const long gx = (p.M + 63) / 64;
    if (gx > 0x7fffffffL) return xq_set_error(XQ_EINVAL, "%s: too many pixels", fn);
    hipLaunchKernelGGL(conv2d_f32_kernel, dim3((unsigned)gx, (unsigned)((Cout + 63) / 64)), dim3(256), 0, (hipStream_t)stream, p);
    return xq_check_launch(fn);
}

extern "C" int xq_gemm_f32_tn(const float *a, const float *b, int64_t M, int Na, int Nb, float *c, xq_stream_t stream) {
    const char *fn = "xq_gemm_f32_tn";
    if (M < 0 || Na < 1 || Nb < 1) return xq_set_error(XQ_EINVAL, "%s: bad shape", fn);
    if (!c || (M > 0 && (!a || !b))) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    hipLaunchKernelGGL(gemm_f32_tn_kernel, dim3((unsigned)((Na + 63) / 64), (unsigned)((Nb + 63) / 64)), dim3(256), 0, (hipStream_t)stream, a, b,
                       (long)M, Na, Nb, c);
    return xq_check_launch(fn);
}

extern "C" int xq_attention_f32_lse(const float *q, const float *k, const float *v, int B, int N, int H, int hd, int64_t batch_stride,
                                    int64_t token_stride, float scale, float *out, float *lse, xq_stream_t stream);
extern "C" int xq_attention_f32(const float *q, const float *k, const float *v, int B, int N, int H, int hd, int64_t batch_stride,
                                int64_t token_stride, float scale, float *out, xq_stream_t stream) {
    return xq_attention_f32_lse(q, k, v, B, N, H, hd, batch_stride, token_stride, scale, out, nullptr, stream);
}

extern "C" int xq_attention_f32_backward(const float *q, const float *k, const float *v, const float *out, const float *dout, const float *lse,
                                         int B, int N, int H, int hd, int64_t batch_stride, int64_t token_stride, float scale, float *dq,
                                         float *dk, float *dv, float *delta, xq_stream_t stream) {
    const char *fn = "xq_attention_f32_backward";
    if (B < 0 || N < 1 || H < 1 || hd < 1 || hd > 64) return xq_set_error(XQ_EINVAL, "%s: bad shape (head dim <= 64)", fn);
    if (B == 0) return XQ_OK;
    if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || !delta) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const long items = (long)B * H * N;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(attention_f32_bwd_q_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, q, k, v, out, dout, lse, B, N, H, hd,
                       (long)batch_stride, (long)token_stride, scale, dq, delta);
    hipLaunchKernelGGL(attention_f32_bwd_kv_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, q, k, v, dout, lse, delta, B, N, H, hd,
                       (long)batch_stride, (long)token_stride, scale, dk, dv);
    return xq_check_launch(fn);
}

extern "C" int xq_attention_f32_lse(const float *q, const float *k, const float *v, int B, int N, int H, int hd, int64_t batch_stride,
                                    int64_t token_stride, float scale, float *out, float *lse, xq_stream_t stream) {
    const char *fn = "xq_attention_f32";
    if (B < 0 || N < 1 || H < 1 || hd < 1) return xq_set_error(XQ_EINVAL, "%s: bad shape", fn);
    if (B == 0) return XQ_OK;
    if (!q || !k || !v || !out) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const size_t lds = (size_t)4 * (hd + N) * sizeof(float);
    if (lds > 64 * 1024) return xq_set_error(XQ_EINVAL, "%s: hd + N = %ld exceeds the 4096-float LDS row", fn, (long)(hd + N));
    const long items = (long)B * H * N;
    hipLaunchKernelGGL(attention_f32_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), lds, (hipStream_t)stream, q, k, v, B, N, H, hd,
                       (long)batch_stride, (long)token_stride, scale, out, lse);
    return xq_check_launch(fn);
}

extern "C" int xq_groupnorm_silu_f32(const float *x, const float *w, const float *bias, int B, int HW, int C, int G, float eps, int silu, float *y,
                                     xq_stream_t stream) {
    const char *fn = "xq_groupnorm_silu_f32";
    if (B < 0 || HW < 1 || C < 1 || G < 1 || C % G) return xq_set_error(XQ_EINVAL, "%s: bad shape", fn);
    if (B == 0) return XQ_OK;
    if (!x || !y) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    hipLaunchKernelGGL(groupnorm_silu_f32_kernel, dim3(G, B), dim3(256), 0, (hipStream_t)stream, x, w, bias, HW, C, G, eps, silu, y);
    return xq_check_launch(fn);
}
