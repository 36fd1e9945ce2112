// xq_attn.hip — multi-head self-attention (head_dim 64, no mask, no dropout) on the packed qkv projection, forward and
// backward, bf16 MFMA with fp32 softmax statistics (gfx950).
//
// Replaces F.scaled_dot_product_attention in the reference's ViT blocks (tokenizer/tokenizer_image/dino_enc/
// vision_transformer.py:175-195: qkv.reshape(B,N,3,H,hd).permute(2,0,3,1,4) -> SDPA -> transpose(1,2).reshape(B,N,C)) and in
// the frozen DINO-S trunk of the discriminator (discriminator_dino.py:28).  The kernels read q/k/v straight out of the
// packed (B, N, 3, H, 64) projection and write (B, N, H*64) / the packed gradient, so the permute copies and the
// gradient concat of the library path do not exist.
//
// Layouts (v_mfma_f32_32x32x16_bf16: A[i][k] lane = i, 8 consecutive k per lane half; B[k][j] lane = j; D[i][j] col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5)):
//  * forward / dQ kernels keep one query per lane: S^T = K Q^T (A = K rows from LDS, B = Q in registers), so softmax row
//    statistics are per-lane scalars (+ one cross-half shuffle), P stays in registers and is directly the B operand of
//    O^T = V^T P^T (resp. dQ^T = K^T dS^T); the register order of P defines the key order of the contraction, the A operand
//    (V^T / K^T) is read in the same order out of the row-major LDS tile with the gfx950 transpose read ds_read_b64_tr_b16.
//  * the dK/dV kernel keeps one key per lane: S = Q K^T (A = Q rows from LDS, B = K in registers), P / dS are the B operands
//    of dV^T = dO^T P and dK^T = Q^T dS (A = dO^T / Q^T through transpose reads); lse and delta come per register from LDS.
// ds_read_b64_tr_b16 (probed in tools/ubench/ds_read_tr.hip): in each group of 16 lanes, lane i receives element (i & 3) of
// the 8-byte chunks addressed by lanes 4j + (i >> 2), j = 0..3 -> with lane x pointing at row (x >> 2), columns 4*(x & 3)..+3
// of a row-major tile, lane i gets column i of rows 0..3: four consecutive contraction indices at a fixed output row.
// S and dP are recomputed in both backward kernels (7 tile products in total, no atomics, no N x N buffer).
#include "xq_common.hpp"
#include "xq_internal.hpp"
#include "../../include/xq_ops.h"

using namespace xq;

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bfv2 __attribute__((ext_vector_type(2)));
typedef float fv2 __attribute__((ext_vector_type(2)));

static constexpr int AT_RP = 72;   // row-major tile pitch in bf16 (144 B: conflict-free ds_read_b128 fragments)

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    fv2 v = {a, b};
    bfv2 r = __builtin_convertvector(v, bfv2);   // v_cvt_pk_bf16_f32 (RNE)
    return __builtin_bit_cast(unsigned, r);
}

__device__ __forceinline__ bf16x8 pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    uint4 u = make_uint4(pack_bf16(a0, a1), pack_bf16(a2, a3), pack_bf16(a4, a5), pack_bf16(a6, a7));
    return __builtin_bit_cast(bf16x8, u);
}

__device__ __forceinline__ bf16x8 cat4(bf16x4 lo, bf16x4 hi) {
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// blocks that share one (batch, head) land on the same XCD (round-robin dispatch: XCD = block id mod 8), so its K/V (or
// Q/dO) rows are fetched into one L2 only
__device__ __forceinline__ bool map_block(int L, int nper, int G, int &g, int &i) {
    const int xcd = L & 7, slot = L >> 3;
    const int gl = slot / nper;
    i = slot - gl * nper;
    g = gl * 8 + xcd;
    return g < G;
}

typedef short s4lds __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x4 lds_tr4(const short *p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4lds __attribute__((address_space(3))) *)p);
}
// A-operand fragment of T^T for a row-major tile T (pitch AT_RP): rows (contraction index) ROW0 + {4hh..4hh+3, 8+4hh..},
// output rows = columns COL0 + (lane & 31).  trp = T + (4*hh + ((lane&15)>>2))*AT_RP + 16*((lane>>4)&1) + 4*(lane&3).
#define AT_TFRAG(TRP, ROW0, COL0) cat4(lds_tr4((TRP) + (ROW0) * AT_RP + (COL0)), lds_tr4((TRP) + ((ROW0) + 8) * AT_RP + (COL0)))

// row index clamped by the caller: the load is unconditional and the value is zeroed afterwards (a "cond ? *p : zero" select
// makes hipcc pick between a global and a private pointer and issue flat loads through scratch)
__device__ __forceinline__ uint4 load16_or_zero(const short *p, bool ok) {
    uint4 v = *reinterpret_cast<const uint4 *>(p);
    if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
    return v;
}

// v_max3_f32 without the operand canonicalisation that fmaxf() carries under IEEE mode (57 -> 16 instructions per tile)
__device__ __forceinline__ float max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

#define AT_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16((A), (B), (C), 0, 0, 0)
// A (uint2: this lane's 4 bf16 of column group k) and B (group k + 1) -> one 16-byte store per lane: lanes 0-31 columns 8 k .. 8 k + 7, lanes
// 32-63 columns 8 k + 8 .. 8 k + 15 (P already carries the upper half-wave's + 16 bytes).  Executed by ALL lanes; OK guards only the store.
#define AT_STORE16_SWAPPED(P, A, B, OK)                                                         \
    do {                                                                                        \
        auto rx_ = __builtin_amdgcn_permlane32_swap((A).x, (B).x, false, false);                \
        auto ry_ = __builtin_amdgcn_permlane32_swap((A).y, (B).y, false, false);                \
        if (OK) *reinterpret_cast<uint4 *>(P) = make_uint4(rx_[0], ry_[0], rx_[1], ry_[1]);     \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// forward: out[b, n, h, :] = softmax(q k^T * scale) v ; lse[b, h, n] = log sum exp (natural log, scaled scores)
// block = 128 queries of one (b, h) (4 waves x 32), loop over 64-key tiles: K row-major + V transposed in LDS, 2 buffers.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 4) void attn_fwd_kernel(const short *__restrict__ qkv, int B, int N, int H, float c, short *__restrict__ out,
                                                       float *__restrict__ lse, int nqb) {
    __shared__ __attribute__((aligned(16))) short Ks[2][64 * AT_RP];
    __shared__ __attribute__((aligned(16))) short Vs[2][64 * AT_RP];
    int g, qb;
    if (!map_block(blockIdx.x, nqb, B * H, g, qb)) return;
    const int b = g / H, h = g - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hh = lane >> 5;
    const int troff = (4 * hh + ((lane & 15) >> 2)) * AT_RP + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const long RS = 3L * H * 64;
    const short *base = qkv + (long)b * N * RS + h * 64;
    const short *kbase = base + H * 64, *vbase = base + 2 * H * 64;

    // the block's four 32-query groups rotate over the waves (= SIMDs) with the (batch, head) index: N = 128 j + 1 (class
    // token) leaves a block with ONE live group per (batch, head), which would otherwise always load SIMD 0
    const int wq = (wave + g) & 3;
    const int qn = qb * 128 + wq * 32 + li;
    const int qc = qn < N ? qn : N - 1;
    bf16x8 qf0, qf1, qf2, qf3;
    {
        const short *qp = base + (long)qc * RS + 8 * hh;
        qf0 = *reinterpret_cast<const bf16x8 *>(qp);
        qf1 = *reinterpret_cast<const bf16x8 *>(qp + 16);
        qf2 = *reinterpret_cast<const bf16x8 *>(qp + 32);
        qf3 = *reinterpret_cast<const bf16x8 *>(qp + 48);
    }

    // staging: 16-byte chunks of K and V, key = tid/8 [+32], part = tid%8, both row-major
    const int kkey = tid >> 3, kpart = tid & 7;
    uint4 rk0, rk1, rv0, rv1;
#define FW_LOAD(KV0)                                                                                                   \
    {                                                                                                                  \
        const int k0_ = (KV0) + kkey, k1_ = k0_ + 32;                                                                  \
        rk0 = load16_or_zero(kbase + (long)(k0_ < N ? k0_ : 0) * RS + 8 * kpart, k0_ < N);                              \
        rk1 = load16_or_zero(kbase + (long)(k1_ < N ? k1_ : 0) * RS + 8 * kpart, k1_ < N);                              \
        rv0 = load16_or_zero(vbase + (long)(k0_ < N ? k0_ : 0) * RS + 8 * kpart, k0_ < N);                              \
        rv1 = load16_or_zero(vbase + (long)(k1_ < N ? k1_ : 0) * RS + 8 * kpart, k1_ < N);                              \
    }
#define FW_STORE(BUF)                                                                                                  \
    {                                                                                                                  \
        *reinterpret_cast<uint4 *>(Ks[BUF] + kkey * AT_RP + 8 * kpart) = rk0;                                          \
        *reinterpret_cast<uint4 *>(Ks[BUF] + (kkey + 32) * AT_RP + 8 * kpart) = rk1;                                   \
        *reinterpret_cast<uint4 *>(Vs[BUF] + kkey * AT_RP + 8 * kpart) = rv0;                                          \
        *reinterpret_cast<uint4 *>(Vs[BUF] + (kkey + 32) * AT_RP + 8 * kpart) = rv1;                                   \
    }

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.0f; o1[r] = 0.0f; }
    // softmax bookkeeping per query (= per lane): m_run is the reference exponent (log2 units), only moved when a tile's
    // maximum exceeds it by more than 2^8 ("lazy rescale": P <= 2^8 stays exact in bf16/fp32 and the O / l rescale becomes a
    // rare wave-uniform branch); the row sums come out of the matrix pipe (ones x P^T) instead of 32 VALU adds per tile.
    float m_run = -INFINITY, l_run = 0.0f;
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const bool wave_live = qb * 128 + wq * 32 < N;   // waves whose 32 queries are all padding only help with staging

    const int ntiles = (N + 63) / 64;
    FW_LOAD(0)
    FW_STORE(0)
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1, kv0 = t * 64;
        if (t + 1 < ntiles) FW_LOAD(kv0 + 64)
        if (wave_live) {
            const short *K = Ks[cur], *V = Vs[cur] + troff;
            const bool wide = kv0 + 32 < N;   // the second 32 keys of the tile hold at least one real key
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f32x16 s0, s1;
            {
                const short *ka = K + li * AT_RP + 8 * hh, *kb = ka + 32 * AT_RP;
                s0 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(ka), qf0, zero16);
                s0 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(ka + 16), qf1, s0);
                s0 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(ka + 32), qf2, s0);
                s0 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(ka + 48), qf3, s0);
                if (wide) {
                    s1 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(kb), qf0, zero16);
                    s1 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(kb + 16), qf1, s1);
                    s1 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(kb + 32), qf2, s1);
                    s1 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(kb + 48), qf3, s1);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s1[r] = -INFINITY;
                }
            }
            if (kv0 + 64 > N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= N) s0[r] = -INFINITY;
                    if (key + 32 >= N) s1[r] = -INFINITY;
                }
            }
            float mx = max3(s0[0], s0[1], s0[2]);
            mx = max3(mx, s0[3], s0[4]);
            mx = max3(mx, s0[5], s0[6]);
            mx = max3(mx, s0[7], s0[8]);
            mx = max3(mx, s0[9], s0[10]);
            mx = max3(mx, s0[11], s0[12]);
            mx = max3(mx, s0[13], s0[14]);
            mx = max3(mx, s0[15], s1[0]);
            mx = max3(mx, s1[1], s1[2]);
            mx = max3(mx, s1[3], s1[4]);
            mx = max3(mx, s1[5], s1[6]);
            mx = max3(mx, s1[7], s1[8]);
            mx = max3(mx, s1[9], s1[10]);
            mx = max3(mx, s1[11], s1[12]);
            mx = max3(mx, s1[13], s1[14]);
            mx = max3(mx, s1[15], s1[15]);
            mx = max3(mx, __shfl_xor(mx, 32), mx);
            const float cand = mx * c;
            if (__any(cand > m_run + 8.0f)) {
                const float mn = fmaxf(m_run, cand);
                const float alpha = __builtin_amdgcn_exp2f(m_run - mn);   // first tile: exp2(-inf) = 0 on o = l = 0
                m_run = mn;
                l_run *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) s0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], c, -m_run));
            const bf16x8 p00 = pack8(s0[0], s0[1], s0[2], s0[3], s0[4], s0[5], s0[6], s0[7]);
            const bf16x8 p01 = pack8(s0[8], s0[9], s0[10], s0[11], s0[12], s0[13], s0[14], s0[15]);
            // A = V^T rows d (lane), key slots {4hh..4hh+3, 8+4hh..} of each 16-key step: the order P's registers hold
            o0 = AT_MFMA(AT_TFRAG(V, 0, 0), p00, o0);
            o1 = AT_MFMA(AT_TFRAG(V, 0, 32), p00, o1);
            f32x16 ls = AT_MFMA(ones, p00, zero16);
            o0 = AT_MFMA(AT_TFRAG(V, 16, 0), p01, o0);
            o1 = AT_MFMA(AT_TFRAG(V, 16, 32), p01, o1);
            ls = AT_MFMA(ones, p01, ls);
            if (wide) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], c, -m_run));
                const bf16x8 p10 = pack8(s1[0], s1[1], s1[2], s1[3], s1[4], s1[5], s1[6], s1[7]);
                const bf16x8 p11 = pack8(s1[8], s1[9], s1[10], s1[11], s1[12], s1[13], s1[14], s1[15]);
                o0 = AT_MFMA(AT_TFRAG(V, 32, 0), p10, o0);
                o1 = AT_MFMA(AT_TFRAG(V, 32, 32), p10, o1);
                ls = AT_MFMA(ones, p10, ls);
                o0 = AT_MFMA(AT_TFRAG(V, 48, 0), p11, o0);
                o1 = AT_MFMA(AT_TFRAG(V, 48, 32), p11, o1);
                ls = AT_MFMA(ones, p11, ls);
            }
            l_run += ls[0];   // every row of ones x P^T is the same: sum over the tile's keys of this lane's query
        }
        if (t + 1 < ntiles) FW_STORE(cur ^ 1)
        __syncthreads();
    }
#undef FW_LOAD
#undef FW_STORE

    const float inv = 1.0f / l_run;
    {
        // lane i holds columns 8 k + 0..3, lane i + 32 columns 8 k + 4..7 of query i: the half-waves exchange register quads
        // (v_permlane32_swap, cdna_hip_programming.md T21) and every lane stores 16 contiguous bytes — 4 instead of 8 store instructions
        const bool okq = qn < N;
        short *op = out + ((long)b * N + (okq ? qn : 0)) * (H * 64) + h * 64 + 8 * hh;
#define FW_W(O, R4) make_uint2(pack_bf16(O[4 * (R4)] * inv, O[4 * (R4) + 1] * inv), pack_bf16(O[4 * (R4) + 2] * inv, O[4 * (R4) + 3] * inv))
#pragma unroll
        for (int r4 = 0; r4 < 4; r4 += 2) {
            uint2 a = FW_W(o0, r4), c2 = FW_W(o0, r4 + 1);
            AT_STORE16_SWAPPED(op + 8 * r4, a, c2, okq);
            a = FW_W(o1, r4); c2 = FW_W(o1, r4 + 1);
            AT_STORE16_SWAPPED(op + 32 + 8 * r4, a, c2, okq);
        }
#undef FW_W
        if (okq && hh == 0) lse[((long)b * H + h) * N + qn] = (m_run + __builtin_amdgcn_logf(l_run)) * 0.6931471805599453f;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dQ: one query per lane (as the forward).  Per 64-key tile: S^T = K Q^T, dP^T = V dO^T, dS^T = P (dP - delta) scale,
// dQ^T += K^T dS^T.  LDS per buffer: K and V row-major (K^T fragments through transpose reads).
// ---------------------------------------------------------------------------------------------------------------------
// Round 5: delta[b, h, n] = sum_d dO * O is computed HERE, in the prologue — each lane holds 32 of its query's 64 dO values already (the B operand
// of dP^T = V dO^T), the matching O values come in with four more 16-byte loads, the two half-rows meet in one cross-half shuffle — and
// written out for the dK/dV kernel, which runs after this one: the stand-alone attn_delta_kernel (pure VALU, 1.0 ms per train step in 36
// launches, 82.8 % of its cycles waiting: profiles/r04_attn_pmc_sq.txt) is gone from the backward pass.
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(const short *__restrict__ qkv, const short *__restrict__ out, const short *__restrict__ dout,
                                                          const float *__restrict__ lse, float *__restrict__ delta, int B, int N, int H,
                                                          float scale, short *__restrict__ dqkv, int nqb) {
    __shared__ __attribute__((aligned(16))) short Ks[2][64 * AT_RP];
    __shared__ __attribute__((aligned(16))) short Vs[2][64 * AT_RP];
    int g, qb;
    if (!map_block(blockIdx.x, nqb, B * H, g, qb)) return;
    const int b = g / H, h = g - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hh = lane >> 5;
    const int troff = (4 * hh + ((lane & 15) >> 2)) * AT_RP + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const long RS = 3L * H * 64;
    const short *base = qkv + (long)b * N * RS + h * 64;
    const short *kbase = base + H * 64, *vbase = base + 2 * H * 64;
    const float c = scale * 1.4426950408889634f;

    const int wq = (wave + g) & 3;      // query groups rotate over the SIMDs (see attn_fwd_kernel)
    const int qn = qb * 128 + wq * 32 + li;
    const int qc = qn < N ? qn : N - 1;
    bf16x8 qf0, qf1, qf2, qf3, df0, df1, df2, df3;
    {
        const short *qp = base + (long)qc * RS + 8 * hh;
        qf0 = *reinterpret_cast<const bf16x8 *>(qp);
        qf1 = *reinterpret_cast<const bf16x8 *>(qp + 16);
        qf2 = *reinterpret_cast<const bf16x8 *>(qp + 32);
        qf3 = *reinterpret_cast<const bf16x8 *>(qp + 48);
        const short *dp = dout + ((long)b * N + qc) * (H * 64) + h * 64 + 8 * hh;
        df0 = *reinterpret_cast<const bf16x8 *>(dp);
        df1 = *reinterpret_cast<const bf16x8 *>(dp + 16);
        df2 = *reinterpret_cast<const bf16x8 *>(dp + 32);
        df3 = *reinterpret_cast<const bf16x8 *>(dp + 48);
    }
    const float lse2 = lse[((long)b * H + h) * N + qc] * 1.4426950408889634f;
    float dq_;
    {
        const short *op_ = out + ((long)b * N + qc) * (H * 64) + h * 64 + 8 * hh;
        const uint4 o0_ = *reinterpret_cast<const uint4 *>(op_), o1_ = *reinterpret_cast<const uint4 *>(op_ + 16);
        const uint4 o2_ = *reinterpret_cast<const uint4 *>(op_ + 32), o3_ = *reinterpret_cast<const uint4 *>(op_ + 48);
        float acc_ = 0.0f;
#define DQ_DOT(O_, D_)                                                                                                                  \
        {                                                                                                                               \
            const uint4 d_ = __builtin_bit_cast(uint4, D_);                                                                             \
            acc_ = fmaf(bf_lo(O_.x), bf_lo(d_.x), acc_); acc_ = fmaf(bf_hi(O_.x), bf_hi(d_.x), acc_);                                   \
            acc_ = fmaf(bf_lo(O_.y), bf_lo(d_.y), acc_); acc_ = fmaf(bf_hi(O_.y), bf_hi(d_.y), acc_);                                   \
            acc_ = fmaf(bf_lo(O_.z), bf_lo(d_.z), acc_); acc_ = fmaf(bf_hi(O_.z), bf_hi(d_.z), acc_);                                   \
            acc_ = fmaf(bf_lo(O_.w), bf_lo(d_.w), acc_); acc_ = fmaf(bf_hi(O_.w), bf_hi(d_.w), acc_);                                   \
        }
        DQ_DOT(o0_, df0) DQ_DOT(o1_, df1) DQ_DOT(o2_, df2) DQ_DOT(o3_, df3)
#undef DQ_DOT
        dq_ = acc_ + __shfl_xor(acc_, 32);
        if (hh == 0 && qn < N) delta[((long)b * H + h) * N + qn] = dq_;
    }

    const int kkey = tid >> 3, kpart = tid & 7;
    uint4 rk0, rk1, rv0, rv1;
#define DQ_LOAD(KV0)                                                                                                   \
    {                                                                                                                  \
        const int k0_ = (KV0) + kkey, k1_ = k0_ + 32;                                                                  \
        rk0 = load16_or_zero(kbase + (long)(k0_ < N ? k0_ : 0) * RS + 8 * kpart, k0_ < N);                   \
        rk1 = load16_or_zero(kbase + (long)(k1_ < N ? k1_ : 0) * RS + 8 * kpart, k1_ < N);                   \
        rv0 = load16_or_zero(vbase + (long)(k0_ < N ? k0_ : 0) * RS + 8 * kpart, k0_ < N);                   \
        rv1 = load16_or_zero(vbase + (long)(k1_ < N ? k1_ : 0) * RS + 8 * kpart, k1_ < N);                   \
    }
#define DQ_STORE(BUF)                                                                                                  \
    {                                                                                                                  \
        *reinterpret_cast<uint4 *>(Ks[BUF] + kkey * AT_RP + 8 * kpart) = rk0;                                          \
        *reinterpret_cast<uint4 *>(Ks[BUF] + (kkey + 32) * AT_RP + 8 * kpart) = rk1;                                   \
        *reinterpret_cast<uint4 *>(Vs[BUF] + kkey * AT_RP + 8 * kpart) = rv0;                                          \
        *reinterpret_cast<uint4 *>(Vs[BUF] + (kkey + 32) * AT_RP + 8 * kpart) = rv1;                                   \
    }

    f32x16 a0, a1;   // dQ^T, d rows 0..31 / 32..63
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.0f; a1[r] = 0.0f; }

    const bool wave_live = qb * 128 + wq * 32 < N;   // waves whose 32 queries are all padding only help with staging
    const int ntiles = (N + 63) / 64;
    DQ_LOAD(0)
    DQ_STORE(0)
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1, kv0 = t * 64;
        if (t + 1 < ntiles) DQ_LOAD(kv0 + 64)
        if (wave_live) {
        const short *K = Ks[cur], *V = Vs[cur], *T = Ks[cur] + troff;
        f32x16 s0, s1, p0, p1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.0f; s1[r] = 0.0f; p0[r] = 0.0f; p1[r] = 0.0f; }
        {
            const short *ka = K + li * AT_RP + 8 * hh, *kb = ka + 32 * AT_RP;
            const short *va = V + li * AT_RP + 8 * hh, *vb = va + 32 * AT_RP;
            s0 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(ka), qf0, s0);
            s1 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(kb), qf0, s1);
            p0 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(va), df0, p0);
            p1 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(vb), df0, p1);
            s0 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(ka + 16), qf1, s0);
            s1 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(kb + 16), qf1, s1);
            p0 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(va + 16), df1, p0);
            p1 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(vb + 16), df1, p1);
            s0 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(ka + 32), qf2, s0);
            s1 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(kb + 32), qf2, s1);
            p0 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(va + 32), df2, p0);
            p1 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(vb + 32), df2, p1);
            s0 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(ka + 48), qf3, s0);
            s1 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(kb + 48), qf3, s1);
            p0 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(va + 48), df3, p0);
            p1 = AT_MFMA(*reinterpret_cast<const bf16x8 *>(vb + 48), df3, p1);
        }
        // keys beyond N were staged as zeros: their dS is finite and multiplies K^T = 0
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], c, -lse2)) * (p0[r] - dq_);   // the softmax scale is applied once, to dQ
            s1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], c, -lse2)) * (p1[r] - dq_);
        }
        const bf16x8 d00 = pack8(s0[0], s0[1], s0[2], s0[3], s0[4], s0[5], s0[6], s0[7]);
        const bf16x8 d01 = pack8(s0[8], s0[9], s0[10], s0[11], s0[12], s0[13], s0[14], s0[15]);
        const bf16x8 d10 = pack8(s1[0], s1[1], s1[2], s1[3], s1[4], s1[5], s1[6], s1[7]);
        const bf16x8 d11 = pack8(s1[8], s1[9], s1[10], s1[11], s1[12], s1[13], s1[14], s1[15]);
        {
            a0 = AT_MFMA(AT_TFRAG(T, 0, 0), d00, a0);
            a1 = AT_MFMA(AT_TFRAG(T, 0, 32), d00, a1);
            a0 = AT_MFMA(AT_TFRAG(T, 16, 0), d01, a0);
            a1 = AT_MFMA(AT_TFRAG(T, 16, 32), d01, a1);
            a0 = AT_MFMA(AT_TFRAG(T, 32, 0), d10, a0);
            a1 = AT_MFMA(AT_TFRAG(T, 32, 32), d10, a1);
            a0 = AT_MFMA(AT_TFRAG(T, 48, 0), d11, a0);
            a1 = AT_MFMA(AT_TFRAG(T, 48, 32), d11, a1);
        }
        }
        if (t + 1 < ntiles) DQ_STORE(cur ^ 1)
        __syncthreads();
    }
#undef DQ_LOAD
#undef DQ_STORE

    {
        uint2 w0[4], w1[4];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            w0[r4] = make_uint2(pack_bf16(a0[4 * r4] * scale, a0[4 * r4 + 1] * scale), pack_bf16(a0[4 * r4 + 2] * scale, a0[4 * r4 + 3] * scale));
            w1[r4] = make_uint2(pack_bf16(a1[4 * r4] * scale, a1[4 * r4 + 1] * scale), pack_bf16(a1[4 * r4 + 2] * scale, a1[4 * r4 + 3] * scale));
        }
        const bool okq = qn < N;
        short *op = dqkv + ((long)b * N + (okq ? qn : 0)) * RS + h * 64 + 8 * hh;
#pragma unroll
        for (int r4 = 0; r4 < 4; r4 += 2) {      // 16-byte stores through the half-wave exchange (see attn_fwd_kernel)
            AT_STORE16_SWAPPED(op + 8 * r4, w0[r4], w0[r4 + 1], okq);
            AT_STORE16_SWAPPED(op + 32 + 8 * r4, w1[r4], w1[r4 + 1], okq);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dK, dV: one key per lane (block = 128 keys of one (b, h), 4 waves x 32).  Per 32-query tile: S = Q K^T, dP = dO V^T,
// P = exp2(S c - lse2), dS = P (dP - delta) scale, dV^T += dO^T P, dK^T += Q^T dS.
// LDS per buffer: Q, dO row-major [32][72]; Q, dO transposed [64][36]; lse2, delta [32].
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void attn_bwd_dkdv_kernel(const short *__restrict__ qkv, const short *__restrict__ dout,
                                                            const float *__restrict__ lse, const float *__restrict__ delta, int B, int N, int H,
                                                            float scale, short *__restrict__ dqkv, int nkb) {
    __shared__ __attribute__((aligned(16))) short Qs[2][32 * AT_RP];
    __shared__ __attribute__((aligned(16))) short Os[2][32 * AT_RP];
    __shared__ __attribute__((aligned(16))) float Ls[2][32];
    __shared__ __attribute__((aligned(16))) float Ds[2][32];
    int g, kb;
    if (!map_block(blockIdx.x, nkb, B * H, g, kb)) return;
    const int b = g / H, h = g - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hh = lane >> 5;
    const int troff = (4 * hh + ((lane & 15) >> 2)) * AT_RP + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const long RS = 3L * H * 64;
    const int OSr = H * 64;
    const short *base = qkv + (long)b * N * RS + h * 64;
    const short *dob = dout + (long)b * N * OSr + h * 64;
    const float *lseb = lse + ((long)b * H + h) * N, *delb = delta + ((long)b * H + h) * N;
    const float c = scale * 1.4426950408889634f;

    const int wk = (wave + g) & 3;      // key groups rotate over the SIMDs (see attn_fwd_kernel)
    const int kn = kb * 128 + wk * 32 + li;
    const bool wave_live = kb * 128 + wk * 32 < N;   // waves whose 32 keys are all padding only help with staging
    bf16x8 kf0, kf1, kf2, kf3, vf0, vf1, vf2, vf3;
    {
        const bool ok = kn < N;
        const short *kp = base + (long)(ok ? kn : 0) * RS + H * 64 + 8 * hh;
        const short *vp = kp + H * 64;
        kf0 = __builtin_bit_cast(bf16x8, load16_or_zero(kp, ok));
        kf1 = __builtin_bit_cast(bf16x8, load16_or_zero(kp + 16, ok));
        kf2 = __builtin_bit_cast(bf16x8, load16_or_zero(kp + 32, ok));
        kf3 = __builtin_bit_cast(bf16x8, load16_or_zero(kp + 48, ok));
        vf0 = __builtin_bit_cast(bf16x8, load16_or_zero(vp, ok));
        vf1 = __builtin_bit_cast(bf16x8, load16_or_zero(vp + 16, ok));
        vf2 = __builtin_bit_cast(bf16x8, load16_or_zero(vp + 32, ok));
        vf3 = __builtin_bit_cast(bf16x8, load16_or_zero(vp + 48, ok));
    }

    // staging: one 16-byte chunk of Q and of dO per thread (query = tid/8, part = tid%8); threads 0..31 fetch lse, 32..63 delta
    const int sq = tid >> 3, sp = tid & 7;
    uint4 rq, ro;
    float rl = 0.0f;
#define KV_LOAD(Q0)                                                                                                    \
    {                                                                                                                  \
        const int q_ = (Q0) + sq;                                                                                      \
        rq = load16_or_zero(base + (long)(q_ < N ? q_ : 0) * RS + 8 * sp, q_ < N);                          \
        ro = load16_or_zero(dob + (long)(q_ < N ? q_ : 0) * OSr + 8 * sp, q_ < N);                          \
        if (tid < 64) {                                                                                                \
            const int ql_ = (Q0) + li;                                                                                 \
            rl = ql_ < N ? (hh == 0 ? lseb[ql_] * 1.4426950408889634f : delb[ql_]) : 0.0f;                              \
        }                                                                                                              \
    }
#define KV_STORE(BUF)                                                                                                  \
    {                                                                                                                  \
        *reinterpret_cast<uint4 *>(Qs[BUF] + sq * AT_RP + 8 * sp) = rq;                                                \
        *reinterpret_cast<uint4 *>(Os[BUF] + sq * AT_RP + 8 * sp) = ro;                                                \
        if (tid < 32) Ls[BUF][li] = rl;                                                                                \
        else if (tid < 64) Ds[BUF][li] = rl;                                                                           \
    }

    f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk0[r] = 0.0f; dk1[r] = 0.0f; dv0[r] = 0.0f; dv1[r] = 0.0f; }

    const int ntiles = (N + 31) / 32;
    KV_LOAD(0)
    KV_STORE(0)
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) KV_LOAD(t * 32 + 32)
        if (wave_live) {
            const short *qa = Qs[cur] + li * AT_RP + 8 * hh, *oa = Os[cur] + li * AT_RP + 8 * hh;
            f32x16 s, p;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.0f; p[r] = 0.0f; }
            s = AT_MFMA(*reinterpret_cast<const bf16x8 *>(qa), kf0, s);
            p = AT_MFMA(*reinterpret_cast<const bf16x8 *>(oa), vf0, p);
            s = AT_MFMA(*reinterpret_cast<const bf16x8 *>(qa + 16), kf1, s);
            p = AT_MFMA(*reinterpret_cast<const bf16x8 *>(oa + 16), vf1, p);
            s = AT_MFMA(*reinterpret_cast<const bf16x8 *>(qa + 32), kf2, s);
            p = AT_MFMA(*reinterpret_cast<const bf16x8 *>(oa + 32), vf2, p);
            s = AT_MFMA(*reinterpret_cast<const bf16x8 *>(qa + 48), kf3, s);
            p = AT_MFMA(*reinterpret_cast<const bf16x8 *>(oa + 48), vf3, p);
            // register r <-> query (r&3) + 8*(r>>2) + 4*hh of the tile; padded queries have Q = dO = 0 and lse2 = delta = 0:
            // their P is finite and multiplies dO^T = 0 / their dS multiplies Q^T = 0
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 l4 = *reinterpret_cast<const float4 *>(&Ls[cur][8 * r4 + 4 * hh]);
                const float4 d4 = *reinterpret_cast<const float4 *>(&Ds[cur][8 * r4 + 4 * hh]);
                float e;
                e = __builtin_amdgcn_exp2f(fmaf(s[4 * r4 + 0], c, -l4.x)); s[4 * r4 + 0] = e; p[4 * r4 + 0] = e * (p[4 * r4 + 0] - d4.x);
                e = __builtin_amdgcn_exp2f(fmaf(s[4 * r4 + 1], c, -l4.y)); s[4 * r4 + 1] = e; p[4 * r4 + 1] = e * (p[4 * r4 + 1] - d4.y);
                e = __builtin_amdgcn_exp2f(fmaf(s[4 * r4 + 2], c, -l4.z)); s[4 * r4 + 2] = e; p[4 * r4 + 2] = e * (p[4 * r4 + 2] - d4.z);
                e = __builtin_amdgcn_exp2f(fmaf(s[4 * r4 + 3], c, -l4.w)); s[4 * r4 + 3] = e; p[4 * r4 + 3] = e * (p[4 * r4 + 3] - d4.w);
            }
            const bf16x8 pb0 = pack8(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7]);
            const bf16x8 pb1 = pack8(s[8], s[9], s[10], s[11], s[12], s[13], s[14], s[15]);
            const bf16x8 db0 = pack8(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]);
            const bf16x8 db1 = pack8(p[8], p[9], p[10], p[11], p[12], p[13], p[14], p[15]);
            const short *ot = Os[cur] + troff, *qt = Qs[cur] + troff;
            dv0 = AT_MFMA(AT_TFRAG(ot, 0, 0), pb0, dv0);
            dv1 = AT_MFMA(AT_TFRAG(ot, 0, 32), pb0, dv1);
            dk0 = AT_MFMA(AT_TFRAG(qt, 0, 0), db0, dk0);
            dk1 = AT_MFMA(AT_TFRAG(qt, 0, 32), db0, dk1);
            dv0 = AT_MFMA(AT_TFRAG(ot, 16, 0), pb1, dv0);
            dv1 = AT_MFMA(AT_TFRAG(ot, 16, 32), pb1, dv1);
            dk0 = AT_MFMA(AT_TFRAG(qt, 16, 0), db1, dk0);
            dk1 = AT_MFMA(AT_TFRAG(qt, 16, 32), db1, dk1);
        }
        if (t + 1 < ntiles) KV_STORE(cur ^ 1)
        __syncthreads();
    }
#undef KV_LOAD
#undef KV_STORE

    {
        uint2 k0[4], k1[4], v0[4], v1[4];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            k0[r4] = make_uint2(pack_bf16(dk0[4 * r4] * scale, dk0[4 * r4 + 1] * scale), pack_bf16(dk0[4 * r4 + 2] * scale, dk0[4 * r4 + 3] * scale));   // softmax scale, once
            k1[r4] = make_uint2(pack_bf16(dk1[4 * r4] * scale, dk1[4 * r4 + 1] * scale), pack_bf16(dk1[4 * r4 + 2] * scale, dk1[4 * r4 + 3] * scale));
            v0[r4] = make_uint2(pack_bf16(dv0[4 * r4], dv0[4 * r4 + 1]), pack_bf16(dv0[4 * r4 + 2], dv0[4 * r4 + 3]));
            v1[r4] = make_uint2(pack_bf16(dv1[4 * r4], dv1[4 * r4 + 1]), pack_bf16(dv1[4 * r4 + 2], dv1[4 * r4 + 3]));
        }
        const bool okk = kn < N;
        short *kp = dqkv + ((long)b * N + (okk ? kn : 0)) * RS + H * 64 + h * 64 + 8 * hh;
        short *vp = kp + H * 64;
#pragma unroll
        for (int r4 = 0; r4 < 4; r4 += 2) {      // 16-byte stores through the half-wave exchange (see attn_fwd_kernel): 8 instead of 16 instructions
            AT_STORE16_SWAPPED(kp + 8 * r4, k0[r4], k0[r4 + 1], okk);
            AT_STORE16_SWAPPED(kp + 32 + 8 * r4, k1[r4], k1[r4 + 1], okk);
            AT_STORE16_SWAPPED(vp + 8 * r4, v0[r4], v0[r4 + 1], okk);
            AT_STORE16_SWAPPED(vp + 32 + 8 * r4, v1[r4], v1[r4 + 1], okk);
        }
    }
}

static int attn_check(const char *fn, int B, int N, int H, int head_dim) {
    if (B < 0 || N < 1 || H < 1) return xq_set_error(XQ_EINVAL, "%s: bad sizes (N=%ld, H=%ld)", fn, (long)N, (long)H);
    if (head_dim != 64) return xq_set_error(XQ_EINVAL, "%s: head_dim must be 64 (got %ld)", fn, (long)head_dim);
    return XQ_OK;
}

extern "C" int xq_attn_forward(const void *qkv, int B, int N, int H, int head_dim, float scale, void *out, float *lse, xq_stream_t stream) {
    const char *fn = "xq_attn_forward";
    if (int rc = attn_check(fn, B, N, H, head_dim)) return rc;
    if (B == 0) return XQ_OK;
    if (!qkv || !out || !lse) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const int nqb = (N + 127) / 128, G8 = (B * H + 7) / 8 * 8;
    const int pslot = prof_begin(XQ_PROF_ATTN_FWD, 4.0 * B * H * (double)N * N * 64.0, (hipStream_t)stream);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)(G8 * nqb)), dim3(256), 0, (hipStream_t)stream, (const short *)qkv, B, N, H,
                       scale * 1.4426950408889634f, (short *)out, lse, nqb);
    prof_end(pslot, (hipStream_t)stream);
    return xq_check_launch(fn);
}

extern "C" int xq_attn_backward(const void *qkv, const void *out, const void *dout, const float *lse, int B, int N, int H, int head_dim,
                                float scale, void *dqkv, float *delta, xq_stream_t stream) {
    const char *fn = "xq_attn_backward";
    if (int rc = attn_check(fn, B, N, H, head_dim)) return rc;
    if (B == 0) return XQ_OK;
    if (!qkv || !out || !dout || !lse || !dqkv || !delta) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    hipStream_t s = (hipStream_t)stream;
    const int pslot = prof_begin(XQ_PROF_ATTN_BWD, 10.0 * B * H * (double)N * N * 64.0, s);
    const int nb = (N + 127) / 128, G8 = (B * H + 7) / 8 * 8;
    // dQ first: its prologue computes delta and writes it for the dK/dV kernel behind it
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)(G8 * nb)), dim3(256), 0, s, (const short *)qkv, (const short *)out, (const short *)dout, lse,
                       delta, B, N, H, scale, (short *)dqkv, nb);
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel, dim3((unsigned)(G8 * nb)), dim3(256), 0, s, (const short *)qkv, (const short *)dout, lse, delta, B, N,
                       H, scale, (short *)dqkv, nb);
    prof_end(pslot, s);
    return xq_check_launch(fn);
}
