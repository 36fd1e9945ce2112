// xq_gemm.hip — bf16 GEMMs of the ViT encoder / decoder blocks on gfx950 MFMA (fp32 accumulate), hand-written.
//
// Replaces the cuBLAS / hipBLASLt calls behind nn.Linear in the reference's transformer blocks
// (tokenizer/tokenizer_image/dino_enc/vision_transformer.py:145-197 Attention.qkv / proj, :295-339 Block -> timm Mlp.fc1 / fc2,
// patch embedding :684-692, dino_enc/to_pixel.py:70-86) in all three passes:
//     NT  y[M,N]   = x[M,K] . W[N,K]^T + b[N]          forward            (both operands K-major)
//     NN  gx[M,N]  = g[M,K] . W[K,N]                   data gradient      (A K-major, B K-strided)
//     TN  gW[P,Q]  = g[R,P]^T . x[R,Q]                 weight gradient    (both K-strided; split over R, fp32 slabs)
// One tile engine serves the three: 256 x BN x 64 block tile (BN = 256 or 128), 512 threads = 8 waves as 2 x 4, each wave
// 128 x BN/4 through v_mfma_f32_32x32x16_bf16.  Operand tiles come in by LDS-DMA (global_load_lds_dwordx4: no staging
// registers, no ds_write pass) in 16 KiB pieces laid out for conflict-free fragment reads (xq_gemm_map.hpp); K-strided operands
// are read with the gfx950 transpose read ds_read_b64_tr_b16, so no operand is ever transposed in memory.
//
// Three schedules:
//   gemm_simple_kernel : 2 LDS buffers, one vmcnt(0) + barrier per K tile; any BN; the reference schedule of the tests.
//   gemm_ring_kernel   : BN = 256.  8-slot piece ring, one piece (2 LDS-DMA instructions per wave) issued per phase, 6 pieces
//                        ahead of the reads, counted s_waitcnt vmcnt(8) (never 0 in the loop), 4 phases of 8 MFMAs per K tile
//                        and wave; the two wave rows run one barrier interval apart, so that on every SIMD one wave is in its
//                        MFMA segment while the other issues its LDS reads / DMA (cdna_hip_programming.md §5 "8-phase").
//   gemm_pring_kernel  : the product schedule.  The same ring kept streaming across a per-CU list of work items, two phases of 16
//                        MFMAs per K tile, operand tile pointers in scalar registers.
// Epilogue: accumulators (+ bias) -> bf16 through a wave-private LDS region -> full-row 16-byte stores; TN: fp32 float4 stores
// into the split's slab, summed by splitk_reduce_kernel (which also folds the < 64-row remainder of R).
#include "xq_common.hpp"
#include "xq_internal.hpp"
#include "xq_gemm_map.hpp"
#include "xq_act.hpp"
#include "../../include/xq_ops.h"

#include <hip/hip_bf16.h>
#include <cstdlib>


using namespace xq;

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bfv2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) void gbl_void;

constexpr int GT = 512;               // threads per block
enum { EPI_BF16 = 0, EPI_F32_SLAB = 1 };
// epilogue activation of the persistent kernel's bf16 outputs (template parameter ACT)
enum { ACT_NONE = 0, ACT_GELU_FWD = 1, ACT_GELU_BWD = 2 };

struct GemmArgs {
    const char *A, *B;      // bf16
    const float *bias;      // [N] or null (EPI_BF16)
    char *C;                // bf16 [M][ldc] or fp32 slabs [splits][M][ldc]
    long M, N;              // output rows, columns
    long lda, ldb, ldc;     // leading dimensions (elements)
    int ktiles;             // 64-deep K tiles per split (split s < kt_rem runs one more)
    int kt_rem;
    int tiles_m, tiles_n, splits;
    // batched products (simple schedule, blockIdx.y = batch index): element strides between consecutive matrices
    long batch_a, batch_b, batch_c;
    // persistent schedule (gemm_pring_kernel): items [0, main_items) are whole tiles with the bf16 epilogue; the remaining
    // tail_tiles tiles are cut into tail_splits K ranges each, every range writing an fp32 slab [item][256][256] in `slabs`
    long main_items;
    int tail_tiles, tail_splits, kt_full;
    float *slabs;
    // implicit-GEMM convolution (A operand = NHWC image gathered tap by tap; Stager<KMAJOR_CONV>): output pixel m = (b, oy, ox),
    // k = tap * Cin + c, tap = ky * 3 + kx.  forward: input pixel (oy * stride + ky - pad, ox * stride + kx - pad), divided by 2
    // when `up` (the conv sees the nearest-2x upsampled image); transposed (data gradient of a strided conv): input pixel
    // ((oy + ky - (2 - pad)) / stride, ...) where divisible.  Pixels outside the image read the zero page.
    int cv_Hi, cv_Wi, cv_Cin, cv_Ho, cv_Wo, cv_stride, cv_pad, cv_up, cv_transposed, relu;
    // fused GELU (persistent schedule).  ACT_GELU_FWD: C = h (pre-activation, + bias), C2 = gelu(h).  ACT_GELU_BWD: C = acc * gelu'(H),
    // colpart[2 * row_tile + wave_row][N] = column sums of C over the 128 rows of that wave row (fc1 bias gradient)
    char *C2;
    const char *H;
    float *colpart;
    int gelu_tanh;
    int nt_store;           // non-temporal bf16 output stores (default; XQ_GEMM_PLAIN_STORE turns them off): the 128 KiB a CU
                            // writes per tile do not displace the operand panels in L2 (qkv forward 839 -> 960 TF/s, others unchanged)
    int debug_no_store;     // XQ_GEMM_DEBUG_NO_STORE: timing experiments only (the result is NOT written)
    int tile_major_debug;   // XQ_GEMM_TILE_MAJOR: keep the weight gradient's items tile-major (A/B timing of the order below)
    unsigned long long *trace;   // XQ_GEMM_TRACE_SUMS (diagnostics): where workgroup trace_block writes its summed phase clocks,
    int trace_cap, trace_block;  // [8 waves][trace_cap] (layout: xq_gemm_trace_bind in include/xq_ops.h); null = off
    int step_r, step_c;          // grid / tiles_n, grid % tiles_n: how the (row, column) tile of a workgroup's next whole-tile item follows
                                 // from its current one (the persistent kernel walks its tiles with scalar adds instead of dividing)
    int split_major;        // order of the K-split items of the persistent schedule.  1 (weight gradient): split-major — the items
                            // of one reduction range sit next to each other, so an XCD (contiguous run of items, xcd_order) streams
                            // ONE range of g / x rows through its L2 for all of that range's output tiles; tile-major order (0, the
                            // tail tiles of NT / NN, whose splits share the operand panels of one tile) made every output tile of the
                            // weight gradient fetch its two full-height operand panels by itself: 1.2 - 1.6 GB per launch against
                            // 0.4 - 0.5 GB algorithmic (profiles/r02_kernel_hbm_traffic_shapes.json), i.e. fabric-bound at ~5 TB/s
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const f32x2 v = {a, b};
    const bfv2 r = __builtin_convertvector(v, bfv2);   // v_cvt_pk_bf16_f32 (RNE)
    return __builtin_bit_cast(unsigned, r);
}

__device__ __forceinline__ bf16x4 tr4(const char *p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3))) *)p);
}

// ---------------------------------------------------------------------------------------------------------------------
// per-thread staging state of one operand: byte offsets (relative to the operand's tile base) of this lane's 16-byte chunk
// in the two LDS-DMA instructions of each half piece
// ---------------------------------------------------------------------------------------------------------------------
// address arithmetic (tile base, per-lane offsets, scalar cursor): gm::StagerAddr in xq_gemm_map.hpp, shared with the CPU replay
template <int KIND, bool IS_A>
struct Stager : gm::StagerAddr<KIND, IS_A> {
    using gm::StagerAddr<KIND, IS_A>::off;
    using gm::StagerAddr<KIND, IS_A>::base;
    using gm::StagerAddr<KIND, IS_A>::adv;
    using gm::StagerAddr<KIND, IS_A>::cur;
    __device__ __forceinline__ void bind(const GemmArgs &) {}
    // issue the two LDS-DMA instructions of piece `half` of K tile `kt` into LDS `dst` (wave-uniform piece base)
    __device__ __forceinline__ void issue(int half, long kt, char *dst, int wave) const {
        const char *b = base + kt * adv;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void *)(b + off[half][i]), (lds_void *)(dst + (2 * wave + i) * 1024), 16, 0, 0);
    }
    // persistent schedule: the tile pointer of the K tile being staged lives in scalar registers (StagerAddr::cur) and moves by a scalar
    // add per K tile — no v_lshl_add_u64 + 2 x v_readfirstlane per piece in the load phase (profiles/r03_gemm_where_the_cycles_go.md)
    __device__ __forceinline__ void issue_cur(int half, char *dst, int wave) const {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void *)(cur + off[half][i]), (lds_void *)(dst + (2 * wave + i) * 1024), 16, 0, 0);
    }
};

__device__ __attribute__((aligned(64))) char xq_zero_page[64];      // zero-initialised: the source of out-of-image taps

// A operand of the implicit-GEMM convolution: same LDS image and lane map as Stager<KMAJOR, true>, the source address of every
// 16-byte chunk (8 channels of one input pixel) is computed per K tile from the tap the tile lies in (Cin % 64 == 0: a K tile
// never straddles two taps)
template <>
struct Stager<gm::KMAJOR_CONV, true> {
    const char *X;
    int Hi, Wi, Cin, stride, pad, up, transposed, Hl, Wl;
    long kt0;                     // first K tile of this work item
    int pix0[2][2];               // [half][i]: pixel index of (b, 0, 0)  (B * Hi * Wi < 2^31, checked by the launcher)
    int oyx[2][2];                // (oy << 16) | ox; -1: row beyond M
    unsigned kbyte[2];            // [i]: byte offset of the lane's chunk inside the 64-wide K slice (independent of the half)
    __device__ __forceinline__ void bind(const GemmArgs &g) {
        X = g.A; Hi = g.cv_Hi; Wi = g.cv_Wi; Cin = g.cv_Cin; stride = g.cv_stride; pad = g.cv_pad; up = g.cv_up; transposed = g.cv_transposed;
        Hl = up ? 2 * Hi : Hi; Wl = up ? 2 * Wi : Wi;
        ho_ = g.cv_Ho; wo_ = g.cv_Wo;
    }
    int ho_, wo_;
    __device__ __forceinline__ void retarget(const char *m, long ld, long rc0, long rc_count, long k0, int wave, int lane, int wtn) {
        init(m, ld, rc0, rc_count, k0, wave, lane, wtn, 2);
    }
    // persistent schedule: the cursor's tap / channel offset live in wave-uniform registers and advance by adds (round 4: the per-piece
    // form below divided a 64-bit K index by Cin four times per K tile — ~1400 scalar instructions per K tile in the load phase of the
    // conv-mode kernel against ~45 in the Linear kernels)
    int c_ky, c_kx, c_c0;
    __device__ __forceinline__ void make_scalar() {
        const unsigned kg = (unsigned)kt0 * gm::BKT;
        const unsigned tap = kg / (unsigned)Cin;
        c_c0 = __builtin_amdgcn_readfirstlane((int)(kg - tap * (unsigned)Cin));
        c_ky = __builtin_amdgcn_readfirstlane((int)(tap / 3u));
        c_kx = __builtin_amdgcn_readfirstlane((int)(tap - 3u * (tap / 3u)));
    }
    __device__ __forceinline__ void step() {
        c_c0 += gm::BKT;
        if (c_c0 >= Cin) {
            c_c0 = 0;
            if (++c_kx == 3) { c_kx = 0; ++c_ky; }
        }
    }
    __device__ __forceinline__ void issue_cur(int half, char *dst, int wave) const { issue_at(half, c_ky, c_kx, c_c0, dst, wave); }
    __device__ __forceinline__ void init(const char *, long, long rc0, long rc_count, long k0, int wave, int lane, int wtn, int) {
        kt0 = k0 / gm::BKT;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const gm::StageSrc s = gm::stage_src<gm::KMAJOR, true>(h, wave, i, lane, wtn);
                const long m = rc0 + s.rc;
                kbyte[i] = (unsigned)(s.k * 2);
                if (m < rc_count) {
                    const int b = (int)(m / ((long)ho_ * wo_));
                    const int rem = (int)(m - (long)b * ho_ * wo_);
                    const int y = rem / wo_;
                    oyx[h][i] = (y << 16) | (rem - y * wo_);
                    pix0[h][i] = b * Hi * Wi;
                } else {
                    oyx[h][i] = -1; pix0[h][i] = 0;
                }
            }
    }
    // simple / ring schedules: K tile kt of this work item (K = 9 Cin < 2^31: 32-bit arithmetic)
    __device__ __forceinline__ void issue(int half, long kt, char *dst, int wave) const {
        const unsigned kg = (unsigned)(kt0 + kt) * gm::BKT;
        const unsigned tap = kg / (unsigned)Cin;
        issue_at(half, (int)(tap / 3u), (int)(tap - 3u * (tap / 3u)), (int)(kg - tap * (unsigned)Cin), dst, wave);
    }
    __device__ __forceinline__ void issue_at(int half, int ky, int kx, int c0, char *dst, int wave) const {
        const int sh = stride >> 1, smask = stride - 1;      // stride is 1 or 2 (checked by the launcher)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int iy, ix;
            bool ok = oyx[half][i] >= 0;
            const int oy_ = oyx[half][i] >> 16, ox_ = oyx[half][i] & 0xffff;
            if (!transposed) {
                iy = oy_ * stride + ky - pad;
                ix = ox_ * stride + kx - pad;
                ok = ok && iy >= 0 && iy < Hl && ix >= 0 && ix < Wl;
                if (up) { iy >>= 1; ix >>= 1; }
            } else {
                iy = oy_ + ky - (2 - pad);
                ix = ox_ + kx - (2 - pad);
                ok = ok && iy >= 0 && ix >= 0 && (iy & smask) == 0 && (ix & smask) == 0;
                iy >>= sh; ix >>= sh;
                ok = ok && iy < Hi && ix < Wi;
            }
            const char *src = ok ? X + ((long)(pix0[half][i] + iy * Wi + ix) * Cin + c0) * 2 + kbyte[i] : xq_zero_page;
            __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(dst + (2 * wave + i) * 1024), 16, 0, 0);
        }
    }
};

// fragment of 8 reduction indices for row / column (lane & 31) out of a piece
// ASM (K-strided only): the two transpose reads by inline asm.  hipcc puts an s_waitcnt vmcnt(0) in front of the ds_read_b64_tr_b16 INTRINSIC
// whenever an LDS-DMA may be in flight (it cannot tell the ring slots apart): the whole prefetch queue drains once per K tile.  The asm reads are
// invisible to that pass; the CALLER must retire them with an explicit s_waitcnt lgkmcnt(0) + sched_barrier in front of the first MFMA that uses
// them (cdna_hip_programming.md 5.4 rule 18) — the duo schedules do.  The 256 x 256 schedules keep the intrinsic: A/B on the persistent kernel
// (profiles/r06_gemm_2wg_ab.txt): data gradients +-1..3 %, weight gradients -3..-7 % (both operands transposed: the pinned asm order costs more than
// the drained queue, whose last pieces were issued a whole phase earlier).
template <int KIND, bool IS_A, bool ASM = false>
__device__ __forceinline__ bf16x8 read_frag(const char *piece, int w, int f, int s, int lane) {
    if (KIND != gm::KSTRIDED) {
        return *reinterpret_cast<const bf16x8 *>(piece + gm::frag_off_kmajor<IS_A>(w, f, s, lane));
    } else if (ASM) {
        typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
        u32x2_ lo, hi;
        const unsigned a0 = (unsigned)(size_t)(piece + gm::frag_off_kstrided<IS_A>(w, f, s, 0, lane));
        const unsigned a1 = (unsigned)(size_t)(piece + gm::frag_off_kstrided<IS_A>(w, f, s, 1, lane));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1));
        return __builtin_shufflevector(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
    } else {
        const bf16x4 lo = tr4(piece + gm::frag_off_kstrided<IS_A>(w, f, s, 0, lane));
        const bf16x4 hi = tr4(piece + gm::frag_off_kstrided<IS_A>(w, f, s, 1, lane));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// epilogues
// ---------------------------------------------------------------------------------------------------------------------
using xq::keep_where_positive;      // out_mask of the convolutions (xq_common.hpp)
template <int NFJ>
__device__ __forceinline__ void epilogue_bf16(const f32x16 (&acc)[4][NFJ], const GemmArgs &g, char *region, long m0, long n0,
                                              int wr, int wc, int lane) {
    constexpr int WTN = 32 * NFJ;
    const int h = lane >> 5;
    const long ncol0 = n0 + (long)WTN * wc;
#pragma unroll
    for (int fj = 0; fj < NFJ; ++fj) {
        float4 bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            long n = ncol0 + 32 * fj + 8 * q + 4 * h;
            if (n > g.N - 4) n = g.N - 4;
            bv[q] = g.bias ? *reinterpret_cast<const float4 *>(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int fi = 0; fi < 4; ++fi)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint2 pk;
                float e0 = acc[fi][fj][4 * q + 0] + bv[q].x, e1 = acc[fi][fj][4 * q + 1] + bv[q].y;
                float e2 = acc[fi][fj][4 * q + 2] + bv[q].z, e3 = acc[fi][fj][4 * q + 3] + bv[q].w;
                if (g.relu) { e0 = fmaxf(e0, 0.f); e1 = fmaxf(e1, 0.f); e2 = fmaxf(e2, 0.f); e3 = fmaxf(e3, 0.f); }
                pk.x = pack_bf16(e0, e1);
                pk.y = pack_bf16(e2, e3);
                *reinterpret_cast<uint2 *>(region + gm::epi_write_off(fi, fj, q, lane, WTN)) = pk;
            }
    }
    // same wave wrote and reads the region: LDS operations of one wave complete in order
    constexpr int PASSES = 128 / (64 / (WTN / 8));
    __hip_bfloat16 *C = reinterpret_cast<__hip_bfloat16 *>(g.C);
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
        int row, c, off;
        gm::epi_read_map(it, lane, WTN, &row, &c, &off);
        uint4 v = *reinterpret_cast<const uint4 *>(region + off);
        const long gr = m0 + 128 * wr + row;
        const long gc = ncol0 + 8 * c;
        if (gr < g.M && gc + 8 <= g.N) {
            if (g.H) {      // out_mask (convolutions only; the fused-GELU kernels, the other users of H, have their own epilogue)
                const uint4 mk = *reinterpret_cast<const uint4 *>(reinterpret_cast<const __hip_bfloat16 *>(g.H) + gr * g.ldc + gc);
                v.x = keep_where_positive(v.x, mk.x); v.y = keep_where_positive(v.y, mk.y);
                v.z = keep_where_positive(v.z, mk.z); v.w = keep_where_positive(v.w, mk.w);
            }
            *reinterpret_cast<uint4 *>(C + gr * g.ldc + gc) = v;
        }
    }
}

template <int NFJ>
__device__ __forceinline__ void epilogue_f32_slab(const f32x16 (&acc)[4][NFJ], const GemmArgs &g, long split, long m0, long n0,
                                                  int wr, int wc, int lane) {
    constexpr int WTN = 32 * NFJ;
    float *C = reinterpret_cast<float *>(g.C) + (size_t)split * g.M * g.ldc;
    const int h = lane >> 5;
#pragma unroll
    for (int fi = 0; fi < 4; ++fi) {
        const long gr = m0 + 128 * wr + 32 * fi + (lane & 31);
#pragma unroll
        for (int fj = 0; fj < NFJ; ++fj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long gc = n0 + (long)WTN * wc + 32 * fj + 8 * q + 4 * h;
                if (gr < g.M && gc + 4 <= g.N)
                    *reinterpret_cast<float4 *>(C + gr * g.ldc + gc) =
                        make_float4(acc[fi][fj][4 * q + 0], acc[fi][fj][4 * q + 1], acc[fi][fj][4 * q + 2], acc[fi][fj][4 * q + 3]);
            }
    }
}

// tile -> (m0, n0, split) with the XCD-aware order; column tiles of one row tile (and the splits of one tile) back to back
__device__ __forceinline__ bool tile_of_block(const GemmArgs &g, int BN, long &m0, long &n0, long &split) {
    const long total = (long)g.tiles_m * g.tiles_n * g.splits;
    const long id = blockIdx.x;
    if (id >= total) return false;
    const long pos = gm::xcd_order(id, total);
    split = pos % g.splits;
    const long t = pos / g.splits;
    m0 = (t / g.tiles_n) * gm::BM;
    n0 = (t % g.tiles_n) * BN;
    return true;
}

// =====================================================================================================================
// simple schedule: stage tile t+1, compute tile t, vmcnt(0) + barrier
// =====================================================================================================================
template <int AK, int BK, int BN, int EPI>
__global__ __launch_bounds__(GT) void gemm_simple_kernel(const GemmArgs g0) {
    constexpr int NFJ = BN / 128;            // 32-wide column fragments per wave (2 or 1)
    constexpr int WTN = BN / 4;
    constexpr int NPB = 2 + NFJ;             // pieces of B... A-top, A-bottom, B-left (, B-right)
    constexpr int BUF = NPB * gm::PIECE_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    long m0, n0, split;
    if (!tile_of_block(g0, BN, m0, n0, split)) return;
    GemmArgs g = g0;
    if (gridDim.y > 1) {     // batched: matrix blockIdx.y
        g.A += (long)blockIdx.y * g0.batch_a * 2;
        g.B += (long)blockIdx.y * g0.batch_b * 2;
        g.C += (long)blockIdx.y * g0.batch_c * (EPI == EPI_BF16 ? 2 : 4);
    }
    const long k0 = (split * (long)g.ktiles + (split < g.kt_rem ? split : g.kt_rem)) * gm::BKT;
    const int KT = g.ktiles + (split < g.kt_rem ? 1 : 0);

    Stager<AK, true> sa;
    Stager<BK, false> sb;
    sa.bind(g);
    sa.init(g.A, g.lda, m0, g.M, k0, wave, lane, WTN, 2);
    sb.init(g.B, g.ldb, n0, g.N, k0, wave, lane, WTN, NFJ);

    f32x16 acc[4][NFJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NFJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    auto stage_tile = [&](long kt, int buf) {
        char *b = smem + buf * BUF;
        sa.issue(0, kt, b, wave);
        sa.issue(1, kt, b + gm::PIECE_BYTES, wave);
        sb.issue(0, kt, b + 2 * gm::PIECE_BYTES, wave);
        if (NFJ > 1) sb.issue(1, kt, b + 3 * gm::PIECE_BYTES, wave);
    };

    stage_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < KT; ++t) {
        const int cur = t & 1;
        if (t + 1 < KT) stage_tile(t + 1, cur ^ 1);
        const char *b = smem + cur * BUF;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 bf[NFJ];
#pragma unroll
            for (int fj = 0; fj < NFJ; ++fj) bf[fj] = read_frag<BK, false>(b + (2 + fj) * gm::PIECE_BYTES, wc, 0, s, lane);
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) {
                const bf16x8 af = read_frag<AK, true>(b + (fi >> 1) * gm::PIECE_BYTES, wr, fi & 1, s, lane);
#pragma unroll
                for (int fj = 0; fj < NFJ; ++fj) acc[fi][fj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[fj], af, acc[fi][fj], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (EPI == EPI_BF16) epilogue_bf16<NFJ>(acc, g, smem + wave * (2 * WTN * 128), m0, n0, wr, wc, lane);
    else epilogue_f32_slab<NFJ>(acc, g, split, m0, n0, wr, wc, lane);
}

// =====================================================================================================================
// ring schedule (BN = 256).  Piece x = 4 t + q of K tile t: q = 0 A-top, 1 B-left, 2 B-right, 3 A-bottom; LDS slot x & 7.
//   phase p of K tile t (phase index j = 4 t + p):  reads   p = 0: A-top + B-left, 1: B-right, 2: A-bottom, 3: -
//                                                   MFMAs   p = 0: top x left, 1: top x right, 2: bottom x right, 3: bottom x left
//                                                   stages  piece j + 6 (its slot held piece j - 2, last read in phase <= j - 2)
//                                                   waits   vmcnt(8): pieces <= j + 2 have landed (read in phase >= j + 1,
//                                                           i.e. after the barrier that follows every wave's wait)
//   each phase = [reads, DMA issue, vmcnt] barrier [8 MFMAs] barrier; wave row 1 runs one barrier behind wave row 0.
// =====================================================================================================================
#define GR_BARRIER()                                   \
    do {                                               \
        __builtin_amdgcn_sched_barrier(0);             \
        asm volatile("s_barrier" ::: "memory");        \
        __builtin_amdgcn_sched_barrier(0);             \
    } while (0)
#define GR_VMCNT(N)                                                \
    do {                                                           \
        asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");      \
    } while (0)

template <int AK, int BK, int EPI>
__global__ __launch_bounds__(GT) void gemm_ring_kernel(const GemmArgs g) {
    constexpr int BN = 256, WTN = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    long m0, n0, split;
    if (!tile_of_block(g, BN, m0, n0, split)) return;
    const long k0 = (split * (long)g.ktiles + (split < g.kt_rem ? split : g.kt_rem)) * gm::BKT;
    const int KT = g.ktiles + (split < g.kt_rem ? 1 : 0);     // >= 2 (checked by the launcher)

    Stager<AK, true> sa;
    Stager<BK, false> sb;
    sa.bind(g);
    sa.init(g.A, g.lda, m0, g.M, k0, wave, lane, WTN, 2);
    sb.init(g.B, g.ldb, n0, g.N, k0, wave, lane, WTN, 2);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // piece (kt, q) -> LDS slot
#define GR_SLOT(KT_, Q) (smem + ((((KT_) & 1) << 2) + (Q)) * gm::PIECE_BYTES)
#define GR_STAGE(KT_, Q)                                                      \
    do {                                                                      \
        if ((Q) == 0) sa.issue(0, (KT_), GR_SLOT(KT_, 0), wave);              \
        else if ((Q) == 1) sb.issue(0, (KT_), GR_SLOT(KT_, 1), wave);         \
        else if ((Q) == 2) sb.issue(1, (KT_), GR_SLOT(KT_, 2), wave);         \
        else sa.issue(1, (KT_), GR_SLOT(KT_, 3), wave);                       \
    } while (0)

    bf16x8 af[2][4], bl[4], br[4];
#define GR_READ_A(KT_, Q)                                                                                     \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                                        \
        af[0][s_] = read_frag<AK, true>(GR_SLOT(KT_, Q), wr, 0, s_, lane);                                    \
        af[1][s_] = read_frag<AK, true>(GR_SLOT(KT_, Q), wr, 1, s_, lane);                                    \
    }
#define GR_READ_B(DST, KT_, Q) \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) DST[s_] = read_frag<BK, false>(GR_SLOT(KT_, Q), wc, 0, s_, lane);
    // MFMAs are register-only: neither the "memory" clobber of the barrier nor sched_barrier keeps instruction selection from
    // moving them into another phase.  The empty volatile asm statements pin the two accumulators of the phase on both sides
    // (volatile asm statements, the barriers included, keep their program order).
#define GR_PIN(X) asm volatile("" : "+v"(X))
#define GR_MFMA(FI0, FJ, BFR)                                                                                          \
    do {                                                                                                               \
        GR_PIN(acc[FI0][FJ]);                                                                                          \
        GR_PIN(acc[FI0 + 1][FJ]);                                                                                      \
        __builtin_amdgcn_s_setprio(1);                                                                                 \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                                             \
            acc[FI0][FJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BFR[s_], af[0][s_], acc[FI0][FJ], 0, 0, 0);         \
            acc[FI0 + 1][FJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BFR[s_], af[1][s_], acc[FI0 + 1][FJ], 0, 0, 0); \
        }                                                                                                              \
        GR_PIN(acc[FI0][FJ]);                                                                                          \
        GR_PIN(acc[FI0 + 1][FJ]);                                                                                      \
        __builtin_amdgcn_s_setprio(0);                                                                                 \
    } while (0)

    // one K tile; ST = number of pieces still to be staged from this tile's phases (4 in the steady state, 2 / 0 at the end);
    // V0..V3 = vmcnt immediates of the four phases
#define GR_TILE(T_, ST, V0, V1, V2, V3)                                       \
    do {                                                                      \
        /* phase 0 */                                                         \
        GR_READ_B(bl, T_, 1)                                                  \
        GR_READ_A(T_, 0)                                                      \
        if ((ST) > 0) GR_STAGE((T_) + 1, 2);                                  \
        GR_VMCNT(V0);                                                         \
        GR_BARRIER();                                                         \
        GR_MFMA(0, 0, bl);                                                    \
        GR_BARRIER();                                                         \
        /* phase 1 */                                                         \
        GR_READ_B(br, T_, 2)                                                  \
        if ((ST) > 1) GR_STAGE((T_) + 1, 3);                                  \
        GR_VMCNT(V1);                                                         \
        GR_BARRIER();                                                         \
        GR_MFMA(0, 1, br);                                                    \
        GR_BARRIER();                                                         \
        /* phase 2 */                                                         \
        GR_READ_A(T_, 3)                                                      \
        if ((ST) > 2) GR_STAGE((T_) + 2, 0);                                  \
        GR_VMCNT(V2);                                                         \
        GR_BARRIER();                                                         \
        GR_MFMA(2, 1, br);                                                    \
        GR_BARRIER();                                                         \
        /* phase 3 */                                                         \
        if ((ST) > 3) GR_STAGE((T_) + 2, 1);                                  \
        GR_VMCNT(V3);                                                         \
        GR_BARRIER();                                                         \
        GR_MFMA(2, 0, bl);                                                    \
        GR_BARRIER();                                                         \
    } while (0)

    // prologue: pieces 0..5 (all of K tile 0, A-top + B-left of K tile 1); the first two must have landed
    GR_STAGE(0, 0);
    GR_STAGE(0, 1);
    GR_STAGE(0, 2);
    GR_STAGE(0, 3);
    GR_STAGE(1, 0);
    GR_STAGE(1, 1);
    GR_VMCNT(8);
    GR_BARRIER();
    if (wr == 1) GR_BARRIER();     // wave row 1 runs one barrier interval behind wave row 0

    int t = 0;
    for (; t < KT - 2; ++t) GR_TILE(t, 4, 8, 8, 8, 8);
    // K tile KT-2 stages the last two pieces (B-right, A-bottom of K tile KT-1); K tile KT-1 stages nothing
    GR_TILE(t, 2, 8, 8, 6, 4);
    ++t;
    GR_TILE(t, 0, 2, 0, 0, 0);
    if (wr == 0) GR_BARRIER();     // re-align: every wave has passed a barrier after the last LDS read of the other row

    if (EPI == EPI_BF16) epilogue_bf16<2>(acc, g, smem + wave * gm::PIECE_BYTES, m0, n0, wr, wc, lane);
    else epilogue_f32_slab<2>(acc, g, split, m0, n0, wr, wc, lane);
#undef GR_TILE
#undef GR_MFMA
#undef GR_PIN
#undef GR_READ_A
#undef GR_READ_B
#undef GR_STAGE
#undef GR_SLOT
}

// =====================================================================================================================
// persistent ring schedule: one workgroup per CU walks a list of work items; the piece ring keeps streaming ACROSS items
// (the pieces of the next item's first 1.5 K tiles are in flight while this item's epilogue runs), so the per-tile prologue
// latency and the workgroup relaunch disappear (measured on the non-persistent ring: ~6 us of fixed cost per 256 x 256 x 768
// tile, a quarter of its time).  Items are whole output tiles or, for the tiles beyond the last full round of CUs and for
// the weight gradient, K ranges of a tile that leave fp32 slabs for slab_reduce_kernel — so that a launch never ends with
// a round in which a handful of CUs run a full tile while the rest idle.
//   staging cursor (item, K tile) : runs 1.5 K tiles ahead of the compute position, retargets the stagers when it enters
//                                   the next item; ring slot parity = parity of the cursor's global K-tile count
//   per item                      : [wave row 1: +1 barrier] K tiles [wave row 0: +1 barrier] epilogue — both wave rows run
//                                   their epilogues side by side; the epilogue stages through 4 KiB per wave OUTSIDE the ring
//   tile order                    : row-major items, XCD k takes a contiguous run of every round (xcd_order).  Tried in round 3 and taken out
//                                   again (git history: "XCD-banded tile order"): column groups sized to the 4 MiB L2, one contiguous run of
//                                   the sequence per XCD — 5-14 % fewer fabric bytes, no speed-up (profiles/r03_gemm_fabric_traffic_by_order.txt,
//                                   r03_gemm_schedules_v3.txt), and its index arithmetic cost registers in a kernel that already spills;
//                                   likewise non-temporal LDS-DMA loads of the A operand (-1..-4 %)
//   vmcnt                         : stores of the epilogue may still be outstanding in the next item's first phases; they only
//                                   make the counted waits stricter (more operations pending than the count assumes)
// =====================================================================================================================
// work items of the persistent schedule: gm::Item / gm::decode_item / gm::next_item_walk in xq_gemm_map.hpp (host + device: the CPU test
// tests/test_gemm_map_cpu.py replays every workgroup's item list through them)
typedef gm::Item PItem;
using gm::decode_item;
using gm::next_item_walk;

__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// ---- GELU / GELU' of the fused MLP epilogues by table (round 6) -------------------------------------------------------------------------
// The activation acts on bf16 values: it is a function of 16 bits.  profiles/r06_gelu_epilogue_cost.txt: the erf arithmetic (one rcp, one exp2,
// ~10 packed-fp32 operations per value, A&S 7.1.26) costs 0.06 - 0.10 ms per fused launch with the matrix pipe idle — 25 k cycles per 256 x 256
// tile and SIMD.  Each workgroup of a fused kernel therefore evaluates csrc/xq_act.hpp's OWN functions once per input value at kernel start (under
// the first pieces' way from HBM) into LDS — forward: bf16 gelu(x) for |x| in [2^-20, 16), both signs, 12 KiB; backward: fp32 gelu'(x) for |x| in
// [2^-12, 16), 16 KiB — and the epilogue reads the result: bit-identical by construction.  Values outside the table (|x| >= 16, tiny or zero: 1 in
// ~5000 at the backward's range) are caught per wave instruction batch by a ballot and that batch is redone by the formula.
constexpr unsigned GT_LO_F = 107u << 7, GT_N_F = 24u * 128u;      // forward table: biased exponents 107 .. 130
constexpr unsigned GT_LO_B = 115u << 7, GT_N_B = 16u * 128u;      // backward table: biased exponents 115 .. 130
constexpr int GT_TABLE_OFF = 16384;                               // behind the 8 x 2 KiB staging regions of the fused kernels
__device__ __forceinline__ unsigned gelu_tab_pair(unsigned w, const char *tab, unsigned &bad) {
    const unsigned lo = w & 0xffffu, hi = w >> 16;
    unsigned il = (lo & 0x7fffu) - GT_LO_F, ih = (hi & 0x7fffu) - GT_LO_F;
    bad |= (unsigned)(il >= GT_N_F) | (unsigned)(ih >= GT_N_F);
    il = (il < GT_N_F ? il : GT_N_F - 1) + (lo >> 15) * GT_N_F;
    ih = (ih < GT_N_F ? ih : GT_N_F - 1) + (hi >> 15) * GT_N_F;
    const unsigned rl = *reinterpret_cast<const unsigned short *>(tab + 2 * il), rh = *reinterpret_cast<const unsigned short *>(tab + 2 * ih);
    return rl | (rh << 16);
}
__device__ __forceinline__ float gelu_tab_grad(unsigned u, const char *tab, unsigned &bad) {
    unsigned i = (u & 0x7fffu) - GT_LO_B;
    bad |= (unsigned)(i >= GT_N_B);
    i = (i < GT_N_B ? i : GT_N_B - 1) + (u >> 15) * GT_N_B;
    return *reinterpret_cast<const float *>(tab + 4 * i);
}

template <int AK, int BK, int ACT, bool SUMS = false>
__global__ __launch_bounds__(GT) void gemm_pring_kernel(const GemmArgs g) {
    constexpr int WTN = 64;
    // SUMS (diagnostics: XQ_GEMM_TRACE_SUMS + xq_gemm_trace_bind, tools/gemm_timeline.py): four shader-clock reads per phase (phase start,
    // arrival at the first barrier, first barrier passed, arrival at the second barrier), differenced and summed in SGPRs (scalar ALU only:
    // no VALU, no LDS, no extra s_waitcnt — the differences are taken right behind the phase's own lgkmcnt(0), one phase late) over every
    // phase of every item of the workgroup except each item's first; the traced workgroup writes the six numbers at the end of the kernel.
    // Outputs are bit-identical to the plain kernel (tests/test_gemm_gpu.py).
    unsigned long long q_s = 0, q_a = 0, q_p = 0, q_e = 0;
    const unsigned long long q_life0 = SUMS ? __builtin_amdgcn_s_memtime() : 0ull, q_real0 = SUMS ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned q_s_prev = 0, q_n = 0, q_phases = 0, q_items = 0, q_load = 0, q_bar1 = 0, q_mfma = 0, q_bar2 = 0;
#define PR_Q(X)                                                   \
    do {                                                          \
        if (SUMS) {                                               \
            __builtin_amdgcn_sched_barrier(0);                    \
            X = __builtin_amdgcn_s_memtime();                     \
            __builtin_amdgcn_sched_barrier(0);                    \
        }                                                         \
    } while (0)
#define PR_Q_ACC()                                                                       \
    do {                                                                                 \
        if (SUMS) {                                                                      \
            if (q_n) {                                                                   \
                q_load += (unsigned)q_a - q_s_prev;                                      \
                q_bar1 += (unsigned)q_p - (unsigned)q_a;                                 \
                q_mfma += (unsigned)q_e - (unsigned)q_p;                                 \
                q_bar2 += (unsigned)q_s - (unsigned)q_e;                                 \
                ++q_phases;                                                              \
            }                                                                            \
            q_s_prev = (unsigned)q_s;                                                    \
            ++q_n;                                                                       \
        }                                                                                \
    } while (0)
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 8 ring slots + 8 x 4 KiB epilogue staging
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const long G = gridDim.x;
    const long items = g.main_items + (long)g.tail_tiles * g.tail_splits;
    long cp = gm::xcd_order(blockIdx.x, G);          // compute position in the item list (stride G)
    if (cp >= items) return;
    PItem cit;
    decode_item(g, cp, cit);
    // tile coordinates of the compute / staging cursor's whole-tile item, in scalar registers: a workgroup's whole-tile items are G tiles
    // apart, so the next one follows by scalar adds (next_item_walk) instead of decode_item's 64-bit divisions
    int c_row = __builtin_amdgcn_readfirstlane((int)(cit.m0 / gm::BM));
    int c_col = __builtin_amdgcn_readfirstlane((int)(cit.n0 / 256));
    int s_row = c_row, s_col = c_col;

    // staging cursor: the tile pointers of the K tile being staged live in scalar registers and move by one scalar add per K tile;
    // between two interior tiles the per-lane offsets are kept (Stager::retarget) — round 4: +5..17 % on every ViT-B shape over the
    // per-piece v_lshl_add_u64 + v_readfirstlane form (profiles/r04_gemm_scalar_base.txt)
    Stager<AK, true> sa;
    Stager<BK, false> sb;
    sa.bind(g);
    sa.init(g.A, g.lda, cit.m0, g.M, cit.k0, wave, lane, WTN, 2);
    sb.init(g.B, g.ldb, cit.n0, g.N, cit.k0, wave, lane, WTN, 2);
    sa.make_scalar();
    sb.make_scalar();
    long sp = cp;
    int s_kt = 0, s_KT = __builtin_amdgcn_readfirstlane(cit.KT), s_par = 0, r_par = 0;
    // past the last item the cursor keeps issuing the SAME number of LDS-DMA instructions per phase (re-reading its last K
    // tile into this wave's own epilogue staging area), so that the counted vmcnt of the phases stays exact to the end
    // of the stream and one tile body serves every K tile
    int s_dummy = 0;
    constexpr bool TABLE = ACT != ACT_NONE;      // fused GELU kernels: 2 KiB of staging per wave + the activation table (see gelu_tab_pair)
    char *const region = smem + 8 * gm::PIECE_BYTES + wave * (TABLE ? 2048 : 4096);
    const char *const act_tab = smem + 8 * gm::PIECE_BYTES + GT_TABLE_OFF;

    f32x16 acc[4][2];
#define PR_ZERO()                                                    \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                 \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                 \
    _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) acc[i_][j_][r_] = 0.0f;
    PR_ZERO()

#define PR_SSLOT(Q) (smem + ((s_par << 2) + (Q)) * gm::PIECE_BYTES)
#define PR_RSLOT(Q) (smem + ((r_par << 2) + (Q)) * gm::PIECE_BYTES)
    // (dummy mode: destination = region - wave * 2048, so that issue_cur()'s (2 wave + i) * 1024 lands inside this wave's area)
#define PR_SDST(Q) (s_dummy ? region - wave * 2048 : PR_SSLOT(Q))
#define PR_STAGE(Q)                                               \
    do {                                                          \
        if ((Q) == 0) sa.issue_cur(0, PR_SDST(0), wave);          \
        else if ((Q) == 1) sb.issue_cur(0, PR_SDST(1), wave);     \
        else if ((Q) == 2) sb.issue_cur(1, PR_SDST(2), wave);     \
        else sa.issue_cur(1, PR_SDST(3), wave);                   \
    } while (0)
    // next K tile of the stream; entering the next item retargets the stagers (once per item)
#define PR_ADVANCE()                                                               \
    do {                                                                           \
        s_par ^= 1;                                                                \
        if (!s_dummy) {                                                            \
            if (++s_kt == s_KT) {                                                  \
                sp += G;                                                           \
                if (sp < items) {                                                  \
                    PItem nx_;                                                     \
                    next_item_walk(g, sp, s_row, s_col, nx_);                      \
                    sa.retarget(g.A, g.lda, nx_.m0, g.M, nx_.k0, wave, lane, WTN); \
                    sb.retarget(g.B, g.ldb, nx_.n0, g.N, nx_.k0, wave, lane, WTN); \
                    sa.make_scalar();                                              \
                    sb.make_scalar();                                              \
                    s_KT = __builtin_amdgcn_readfirstlane(nx_.KT);                 \
                    s_kt = 0;                                                      \
                } else {                                                           \
                    s_dummy = 1;      /* the cursor stays on the last K tile */    \
                    s_kt = s_KT - 1;                                               \
                }                                                                  \
            } else {                                                               \
                sa.step();                                                         \
                sb.step();                                                         \
            }                                                                      \
        }                                                                          \
    } while (0)

    bf16x8 af[2][4], bl[4], br[4];
#define PR_READ_A(Q)                                                                    \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                  \
        af[0][s_] = read_frag<AK, true>(PR_RSLOT(Q), wr, 0, s_, lane);                  \
        af[1][s_] = read_frag<AK, true>(PR_RSLOT(Q), wr, 1, s_, lane);                  \
    }
#define PR_READ_B(DST, Q) \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) DST[s_] = read_frag<BK, false>(PR_RSLOT(Q), wc, 0, s_, lane);
    // MFMAs are register-only: neither the "memory" clobber of the barrier nor sched_barrier keeps instruction selection from moving
    // them into another phase.  The empty volatile asm statements pin the two accumulators of the group on both sides (volatile asm
    // statements, the barriers included, keep their program order).
#define PR_PIN(X) asm volatile("" : "+v"(X))
#define PR_MFMA(FI0, FJ, BFR)                                                                                          \
    do {                                                                                                               \
        PR_PIN(acc[FI0][FJ]);                                                                                          \
        PR_PIN(acc[FI0 + 1][FJ]);                                                                                      \
        __builtin_amdgcn_s_setprio(1);                                                                                 \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                                             \
            acc[FI0][FJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BFR[s_], af[0][s_], acc[FI0][FJ], 0, 0, 0);         \
            acc[FI0 + 1][FJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BFR[s_], af[1][s_], acc[FI0 + 1][FJ], 0, 0, 0); \
        }                                                                                                              \
        PR_PIN(acc[FI0][FJ]);                                                                                          \
        PR_PIN(acc[FI0 + 1][FJ]);                                                                                      \
        __builtin_amdgcn_s_setprio(0);                                                                                 \
    } while (0)
    // One K tile = two phases of 16 MFMAs per wave, each [fragment reads, 4 LDS-DMA instructions, counted waits] barrier [16 MFMAs]
    // barrier; the two wave rows run one barrier apart, so that on every SIMD one wave loads while the other multiplies.  Piece
    // x = 4 t + q of K tile t: q = 0 A-top, 1 B-left, 2 B-right, 3 A-bottom; LDS slot x & 7.
    // Hazards (HW barrier numbering: wave row 1 runs one barrier interval behind row 0):
    //   RAW  phase A reads pieces (t,1) (t,2) (t,0): in flight after phase B(t-1) staged (t+1,0) (t+1,1) are, oldest first,
    //        (t,0..3) (t+1,0) (t+1,1) = 12 instructions -> vmcnt(6) there retires (t,0) (t,1) (t,2);  phase B reads (t,3): in flight
    //        after phase A(t) staged (t+1,2) (t+1,3) are (t,3) (t+1,0..3) = 10 -> vmcnt(8) retires (t,3);  prologue: vmcnt(6)
    //   WAR  slot of (t+1,3) = slot of (t-1,3), last read at the head of phase B(t-1) by wave row 0 and one barrier later by row 1:
    //        the explicit lgkmcnt(0) in front of every phase's first barrier retires a wave's reads before it signals, so a piece
    //        is restaged at least two barriers after its last read has RETURNED (cdna_hip_programming.md, 8-phase template rule)
    //   stores of an epilogue may still be outstanding in the next item's first phases; they only make the counted waits stricter
#define GR_LGKM0()                                                            \
    do {                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    \
    } while (0)
#define PR_TILE2()                                                            \
    do {                                                                      \
        PR_Q(q_s);                                                            \
        PR_READ_B(bl, 1)                                                      \
        PR_READ_B(br, 2)                                                      \
        PR_READ_A(0)                                                          \
        PR_STAGE(2);                                                          \
        PR_STAGE(3);                                                          \
        GR_LGKM0();                                                           \
        PR_Q_ACC();                                                           \
        GR_VMCNT(8);                                                          \
        PR_Q(q_a);                                                            \
        GR_BARRIER();                                                         \
        PR_Q(q_p);                                                            \
        PR_MFMA(0, 0, bl);                                                    \
        PR_MFMA(0, 1, br);                                                    \
        PR_Q(q_e);                                                            \
        GR_BARRIER();                                                         \
        PR_Q(q_s);                                                            \
        PR_READ_A(3)                                                          \
        PR_ADVANCE();                                                         \
        PR_STAGE(0);                                                          \
        PR_STAGE(1);                                                          \
        GR_LGKM0();                                                           \
        PR_Q_ACC();                                                           \
        GR_VMCNT(6);                                                          \
        PR_Q(q_a);                                                            \
        GR_BARRIER();                                                         \
        PR_Q(q_p);                                                            \
        PR_MFMA(2, 1, br);                                                    \
        PR_MFMA(2, 0, bl);                                                    \
        PR_Q(q_e);                                                            \
        GR_BARRIER();                                                         \
        r_par ^= 1;                                                           \
    } while (0)

    // prologue: K tile 0 of the first item completely, A-top + B-left of its K tile 1
    PR_STAGE(0);
    PR_STAGE(1);
    PR_STAGE(2);
    PR_STAGE(3);
    PR_ADVANCE();
    PR_STAGE(0);
    PR_STAGE(1);
    if (TABLE) {      // the activation table, built by xq_act.hpp's own functions while the first pieces are on their way
        char *tab = smem + 8 * gm::PIECE_BYTES + GT_TABLE_OFF;
        if (ACT == ACT_GELU_FWD) {
            for (unsigned i = tid; i < 2 * GT_N_F; i += GT) {
                const unsigned sg = i >= GT_N_F ? 1u : 0u, u = (sg << 15) | (GT_LO_F + (i - sg * GT_N_F));
                const float x = __uint_as_float(u << 16);
                const float y = g.gelu_tanh ? gelu_val<true>(x) : gelu_val<false>(x);
                reinterpret_cast<unsigned short *>(tab)[i] = (unsigned short)(pack_bf16(y, y) & 0xffffu);
            }
        } else {
            for (unsigned i = tid; i < 2 * GT_N_B; i += GT) {
                const unsigned sg = i >= GT_N_B ? 1u : 0u, u = (sg << 15) | (GT_LO_B + (i - sg * GT_N_B));
                const float x = __uint_as_float(u << 16);
                reinterpret_cast<float *>(tab)[i] = g.gelu_tanh ? gelu_grad<true>(x) : gelu_grad<false>(x);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // my table entries are written before I signal the barrier below
    }
    GR_VMCNT(6);
    GR_BARRIER();

    const int h = lane >> 5;
    for (;;) {
        const bool has_next = cp + G < items;
        if (wr == 1) GR_BARRIER();
        {      // trip count in a scalar register: no VALU compare + VCC branch per K tile
            const int kt_n = __builtin_amdgcn_readfirstlane(cit.KT);
            for (int kt = 0; kt < kt_n; ++kt) PR_TILE2();
        }
        if (wr == 0) GR_BARRIER();
        if (!has_next) GR_VMCNT(0);      // the dummy pieces target `region`
        // ---- epilogue of this item (the next item's first pieces are landing meanwhile) ----
        if (cit.slab) {
            float *S = g.slabs + (size_t)cit.slab_idx * 65536;
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) {
                const int row = 128 * wr + 32 * fi + (lane & 31);
#pragma unroll
                for (int fj = 0; fj < 2; ++fj)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4 *>(S + row * 256 + 64 * wc + 32 * fj + 8 * q + 4 * h) =
                            make_float4(acc[fi][fj][4 * q + 0], acc[fi][fj][4 * q + 1], acc[fi][fj][4 * q + 2], acc[fi][fj][4 * q + 3]);
            }
        } else {
            const long ncol0 = cit.n0 + 64 * wc;
            // bias and ReLU exist for the K-major B operand only (Linear forward, convolutions): the data gradients have neither, and
            // their epilogues keep the 32 registers
            constexpr bool HAS_BIAS = BK == gm::KMAJOR && ACT != ACT_GELU_BWD;
            constexpr bool OUT_MASK = AK == gm::KMAJOR_CONV && ACT == ACT_NONE;   // H = out_mask of a convolution's data gradient
            float4 bv[2][4];
#pragma unroll
            for (int fj = 0; fj < 2; ++fj)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    long n = ncol0 + 32 * fj + 8 * q + 4 * h;
                    if (n > g.N - 4) n = g.N - 4;
                    bv[fj][q] = (HAS_BIAS && g.bias) ? *reinterpret_cast<const float4 *>(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            __hip_bfloat16 *C = reinterpret_cast<__hip_bfloat16 *>(g.C);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // ACT_GELU_BWD: this lane's 8 columns over its 16 rows
            if constexpr (TABLE) {
                // fused GELU products: 16 rows x 64 columns per round trip through this wave's 2 KiB (the half of the lanes whose fragment row lies in
                // the 16 writes, all 64 read), the activation from the table
                // backward: the pre-activations of the NEXT 32-row block are requested before this block's stores are issued (two register sets).
                // Fetched at the head of each pass (round 5) every pass began with a wait for its own loads, which — vmcnt retires in order —
                // is also a wait for the previous pass's STORES to be acknowledged: a serial memory round trip per pass.  (All 16 loads of the item
                // up front: 64 registers more than the kernel has — measured, 272 bytes of scratch, 0.505 -> 0.572 ms.)
                u32x4 hvbuf[2][2][2];
#define PR_LOAD_H(FI_)                                                                                                                    \
    _Pragma("unroll") for (int u_ = 0; u_ < 2; ++u_)                                                                                      \
    _Pragma("unroll") for (int it_ = 0; it_ < 2; ++it_) {                                                                                 \
        int row_, c_, off_;                                                                                                               \
        gm::epi_read_map(it_, lane, WTN, &row_, &c_, &off_);                                                                              \
        long gr_ = cit.m0 + 128 * wr + 32 * (FI_) + 16 * u_ + row_, gc_ = ncol0 + 8 * c_;                                                 \
        if (gr_ > g.M - 1) gr_ = g.M - 1;                                                                                                 \
        if (gc_ > g.N - 8) gc_ = g.N - 8;                                                                                                 \
        hvbuf[(FI_) & 1][u_][it_] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const __hip_bfloat16 *>(g.H) + gr_ * g.ldc + gc_);  \
    }
                if (ACT == ACT_GELU_BWD) { PR_LOAD_H(0) }
#pragma unroll
                for (int fi = 0; fi < 4; ++fi)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (ACT == ACT_GELU_BWD && u == 0 && fi + 1 < 4) { PR_LOAD_H(fi + 1) }
                        const u32x4 (&hv)[2] = hvbuf[fi & 1][u];
                        if (((lane >> 4) & 1) == u) {
#pragma unroll
                            for (int fj = 0; fj < 2; ++fj)
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    uint2 pk;
                                    float e0 = acc[fi][fj][4 * q + 0], e1 = acc[fi][fj][4 * q + 1], e2 = acc[fi][fj][4 * q + 2], e3 = acc[fi][fj][4 * q + 3];
                                    if (HAS_BIAS) { e0 += bv[fj][q].x; e1 += bv[fj][q].y; e2 += bv[fj][q].z; e3 += bv[fj][q].w; }
                                    pk.x = pack_bf16(e0, e1);
                                    pk.y = pack_bf16(e2, e3);
                                    // rows 0 .. 15 of the region: lane (l & 15) + 32 (l >> 5) of the writing half plays lane (l & 31) of the 32-row map
                                    *reinterpret_cast<uint2 *>(region + gm::epi_write_off(0, fj, q, lane & 47, WTN)) = pk;
                                }
                        }
                        // same wave wrote and reads: LDS operations of one wave complete in order
#pragma unroll
                        for (int it = 0; it < 2; ++it) {
                            int row, c, off;
                            gm::epi_read_map(it, lane, WTN, &row, &c, &off);
                            const u32x4 v = *reinterpret_cast<const u32x4 *>(region + off);
                            const long gr = cit.m0 + 128 * wr + 32 * fi + 16 * u + row;
                            const long gc = ncol0 + 8 * c;
                            const bool ok = gr < g.M && gc + 8 <= g.N;
                            u32x4 o = v;
                            unsigned bad = 0;
                            if (ACT == ACT_GELU_FWD) {
                                // the activation acts on the bf16-ROUNDED pre-activation, as the unfused pair (Linear -> GELU) does
                                u32x4 a2;
#pragma unroll
                                for (int e = 0; e < 4; ++e) a2[e] = gelu_tab_pair(v[e], act_tab, bad);
                                if (__builtin_amdgcn_ballot_w64(bad != 0) != 0) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const float x0 = bf16_lo(v[e]), x1 = bf16_hi(v[e]);
                                        a2[e] = g.gelu_tanh ? pack_bf16(gelu_val<true>(x0), gelu_val<true>(x1)) : pack_bf16(gelu_val<false>(x0), gelu_val<false>(x1));
                                    }
                                }
                                if (ok) {
                                    u32x4 *p2 = reinterpret_cast<u32x4 *>(reinterpret_cast<__hip_bfloat16 *>(g.C2) + gr * g.ldc + gc);
                                    if (g.nt_store) __builtin_nontemporal_store(a2, p2); else *p2 = a2;
                                }
                            } else {
                                float d[8];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    d[2 * e] = gelu_tab_grad(hv[it][e] & 0xffffu, act_tab, bad);
                                    d[2 * e + 1] = gelu_tab_grad(hv[it][e] >> 16, act_tab, bad);
                                }
                                if (__builtin_amdgcn_ballot_w64(bad != 0) != 0) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        d[2 * e] = g.gelu_tanh ? gelu_grad<true>(bf16_lo(hv[it][e])) : gelu_grad<false>(bf16_lo(hv[it][e]));
                                        d[2 * e + 1] = g.gelu_tanh ? gelu_grad<true>(bf16_hi(hv[it][e])) : gelu_grad<false>(bf16_hi(hv[it][e]));
                                    }
                                }
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    o[e] = pack_bf16(bf16_lo(v[e]) * d[2 * e], bf16_hi(v[e]) * d[2 * e + 1]);
                                    if (gr < g.M) { csum[2 * e] += bf16_lo(o[e]); csum[2 * e + 1] += bf16_hi(o[e]); }
                                }
                            }
                            if (ok && (ACT != ACT_GELU_FWD || g.C != nullptr)) {      // (fused GELU forward without a gradient to come: h is not written)
                                u32x4 *p1 = reinterpret_cast<u32x4 *>(C + gr * g.ldc + gc);
                                if (g.nt_store) __builtin_nontemporal_store(o, p1); else *p1 = o;
                            }
                        }
                    }
#undef PR_LOAD_H
            } else
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) {
                u32x4 hv[4];
                if (ACT == ACT_GELU_BWD || (OUT_MASK && g.H)) {      // the pre-activations (GELU') / the out_mask values of the 4 store passes, fetched ahead of the LDS round trip
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        int row, c, off;
                        gm::epi_read_map(it, lane, WTN, &row, &c, &off);
                        long gr = cit.m0 + 128 * wr + 32 * fi + row, gc = ncol0 + 8 * c;
                        if (gr > g.M - 1) gr = g.M - 1;
                        if (gc > g.N - 8) gc = g.N - 8;
                        hv[it] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const __hip_bfloat16 *>(g.H) + gr * g.ldc + gc);
                    }
                }
#pragma unroll
                for (int fj = 0; fj < 2; ++fj)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint2 pk;
                        float e0 = acc[fi][fj][4 * q + 0], e1 = acc[fi][fj][4 * q + 1], e2 = acc[fi][fj][4 * q + 2], e3 = acc[fi][fj][4 * q + 3];
                        if (HAS_BIAS) {
                            e0 += bv[fj][q].x; e1 += bv[fj][q].y; e2 += bv[fj][q].z; e3 += bv[fj][q].w;
                            if (g.relu) { e0 = fmaxf(e0, 0.f); e1 = fmaxf(e1, 0.f); e2 = fmaxf(e2, 0.f); e3 = fmaxf(e3, 0.f); }
                        }
                        pk.x = pack_bf16(e0, e1);
                        pk.y = pack_bf16(e2, e3);
                        *reinterpret_cast<uint2 *>(region + gm::epi_write_off(0, fj, q, lane, WTN)) = pk;
                    }
                // same wave wrote and reads: LDS operations of one wave complete in order
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    int row, c, off;
                    gm::epi_read_map(it, lane, WTN, &row, &c, &off);
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(region + off);
                    const long gr = cit.m0 + 128 * wr + 32 * fi + row;
                    const long gc = ncol0 + 8 * c;
                    const bool ok = gr < g.M && gc + 8 <= g.N && !g.debug_no_store;
                    u32x4 o = v;
                    if (ACT == ACT_GELU_FWD) {
                        // the activation acts on the bf16-ROUNDED pre-activation, as the unfused pair (Linear -> GELU) does
                        u32x4 a2;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x0 = bf16_lo(v[e]), x1 = bf16_hi(v[e]);
                            a2[e] = g.gelu_tanh ? pack_bf16(gelu_val<true>(x0), gelu_val<true>(x1)) : pack_bf16(gelu_val<false>(x0), gelu_val<false>(x1));
                        }
                        if (ok) {
                            u32x4 *p2 = reinterpret_cast<u32x4 *>(reinterpret_cast<__hip_bfloat16 *>(g.C2) + gr * g.ldc + gc);
                            if (g.nt_store) __builtin_nontemporal_store(a2, p2); else *p2 = a2;
                        }
                    } else if (ACT == ACT_GELU_BWD) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float d0 = g.gelu_tanh ? gelu_grad<true>(bf16_lo(hv[it][e])) : gelu_grad<false>(bf16_lo(hv[it][e]));
                            const float d1 = g.gelu_tanh ? gelu_grad<true>(bf16_hi(hv[it][e])) : gelu_grad<false>(bf16_hi(hv[it][e]));
                            o[e] = pack_bf16(bf16_lo(v[e]) * d0, bf16_hi(v[e]) * d1);
                            if (gr < g.M) { csum[2 * e] += bf16_lo(o[e]); csum[2 * e + 1] += bf16_hi(o[e]); }
                        }
                    }
                    if (OUT_MASK && g.H) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = keep_where_positive(v[e], hv[it][e]);
                    }
                    if (ok && (ACT != ACT_GELU_FWD || g.C != nullptr)) {      // (fused GELU forward without a gradient to come: h is not written)
                        u32x4 *p1 = reinterpret_cast<u32x4 *>(C + gr * g.ldc + gc);
                        if (g.nt_store) __builtin_nontemporal_store(o, p1); else *p1 = o;
                    }
                }
            }
            if (ACT == ACT_GELU_BWD && g.colpart) {
                // lanes l, l + 8, ..., l + 56 hold the same 8 columns (c = l & 7) for different rows
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = csum[e];
                    t += __shfl_xor(t, 8);
                    t += __shfl_xor(t, 16);
                    t += __shfl_xor(t, 32);
                    csum[e] = t;
                }
                const long gc = ncol0 + 8 * (lane & 7);
                if (lane < 8 && gc + 8 <= g.N) {
                    float *cp = g.colpart + ((cit.m0 / 128) + wr) * g.N + gc;
                    *reinterpret_cast<float4 *>(cp) = make_float4(csum[0], csum[1], csum[2], csum[3]);
                    *reinterpret_cast<float4 *>(cp + 4) = make_float4(csum[4], csum[5], csum[6], csum[7]);
                }
            }
        }
        if (SUMS) { q_n = 0; ++q_items; }      // the first phase of the next item is not differenced against this item's last
        if (!has_next) break;
        PR_ZERO()
        cp += G;
        next_item_walk(g, cp, c_row, c_col, cit);
    }
    if (SUMS) {
        if (g.trace != nullptr && (int)blockIdx.x == g.trace_block && lane == 0) {
            unsigned long long *out = g.trace + (long)wave * g.trace_cap;
            out[0] = 0;
            out[8] = q_phases; out[9] = q_load; out[10] = q_bar1; out[11] = q_mfma; out[12] = q_bar2; out[13] = q_items;
            out[14] = __builtin_amdgcn_s_memtime() - q_life0;          // shader cycles of this workgroup's life ...
            out[15] = __builtin_amdgcn_s_memrealtime() - q_real0;      // ... and the 100 MHz reference ticks: their ratio is the clock the chip held
        }
    }
#undef PR_Q
#undef PR_Q_ACC
#undef PR_TILE2
#undef GR_LGKM0
#undef PR_MFMA
#undef PR_PIN
#undef PR_READ_A
#undef PR_READ_B
#undef PR_ADVANCE
#undef PR_STAGE
#undef PR_SSLOT
#undef PR_SDST
#undef PR_RSLOT
#undef PR_ZERO
}

// =====================================================================================================================
// duo schedule (round 6): 128 x 256 block tile, TWO workgroups per CU (8 waves of 64 x 64 each: 64 accumulator registers, <= 128
// VGPRs -> 4 waves per SIMD; 80 KiB of LDS per workgroup).  What the 256 x 256 persistent kernel cannot do (its register file and
// LDS are both full) the hardware does by itself here: while one workgroup of a CU stores its tile (or waits for its first pieces, or sits at
// a barrier) the other one's K loop owns the matrix pipe, so the store burst of a short-K item is no longer serial time.
//   pieces   per K tile: A (128 rows x 64 k, piece-row = tile row), B-left, B-right (128 columns each, as above) = 48 KiB
//   ring     A(t) in A slot t & 1; B piece y = 2 t + h in B slot y % 3  (2 + 3 slots of 16 KiB = 80 KiB)
//   phase    (t, h), one barrier each:  wait [own DMA of the pieces read now] - barrier - stage - fragment reads - 8 MFMAs
//            (t, 0): reads A(t) (kept in registers for both phases) + B-left(t), stages B-left(t + 1)  into the slot of B-right(t - 1)
//            (t, 1): reads B-right(t),                                            stages B-right(t + 1) into the slot of B-left(t),
//                                                                                        A(t + 2)       into the slot of A(t)
//   stream   A0 BL0 BR0 A1 | BL1 | BR1 A2 | BL2 | BR2 A3 | ...  (2 LDS-DMA instructions per piece and wave)
//   RAW      a piece is read after every wave's own counted vmcnt AND the barrier behind it: at (t, 0) the stream holds A(t) BL(t) BR(t)
//            A(t + 1) -> vmcnt(4); at (t, 1) BR(t) A(t + 1) BL(t + 1) -> vmcnt(4); the last K tile (nothing staged behind it): vmcnt(2), vmcnt(0)
//   WAR      a slot is restaged behind the barrier that follows the phase which read it; every wave retires its reads (lgkmcnt(0)) before
//            its MFMAs, i.e. before it can arrive at that barrier
// Same MFMA order per accumulator as every other schedule: outputs are bit-identical to gemm_simple_kernel's.
// =====================================================================================================================
// bf16 epilogue of one 128 x 256 duo tile: accumulators (+ bias / ReLU) -> bf16, 32 rows x 64 columns at a time through 4 KiB of this wave's own
// LDS region, full-row 16-byte stores; ACT as in the persistent kernel.  m_lo: first row that belongs to this tile row (the rows [m0, m_lo) of a
// moved-back edge tile are the tile above's: computed identically, not summed twice into the fc1-bias partials); trow: tile row.
template <int BK, int ACT>
__device__ __forceinline__ void duo_epilogue(const f32x16 (&acc)[2][2], const GemmArgs &g, char *region, long m0, long n0, long m_lo, long trow,
                                             int wr, int wc, int lane) {
    constexpr int WTN = 64;
        const int h = lane >> 5;
    const long ncol0 = n0 + 64 * wc;
    constexpr bool HAS_BIAS = BK == gm::KMAJOR && ACT != ACT_GELU_BWD;
    float4 bv[2][4];
#pragma unroll
    for (int fj = 0; fj < 2; ++fj)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            long n = ncol0 + 32 * fj + 8 * q + 4 * h;
            if (n > g.N - 4) n = g.N - 4;
            bv[fj][q] = (HAS_BIAS && g.bias) ? *reinterpret_cast<const float4 *>(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    __hip_bfloat16 *C = reinterpret_cast<__hip_bfloat16 *>(g.C);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int fi = 0; fi < 2; ++fi) {
        u32x4 hv[4];
        if (ACT == ACT_GELU_BWD) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                int row, c, off;
                gm::epi_read_map(it, lane, WTN, &row, &c, &off);
                long gr = m0 + 64 * wr + 32 * fi + row, gc = ncol0 + 8 * c;
                if (gr > g.M - 1) gr = g.M - 1;
                if (gc > g.N - 8) gc = g.N - 8;
                hv[it] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const __hip_bfloat16 *>(g.H) + gr * g.ldc + gc);
            }
        }
#pragma unroll
        for (int fj = 0; fj < 2; ++fj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint2 pk;
                float e0 = acc[fi][fj][4 * q + 0], e1 = acc[fi][fj][4 * q + 1], e2 = acc[fi][fj][4 * q + 2], e3 = acc[fi][fj][4 * q + 3];
                if (HAS_BIAS) {
                    e0 += bv[fj][q].x; e1 += bv[fj][q].y; e2 += bv[fj][q].z; e3 += bv[fj][q].w;
                    if (g.relu) { e0 = fmaxf(e0, 0.f); e1 = fmaxf(e1, 0.f); e2 = fmaxf(e2, 0.f); e3 = fmaxf(e3, 0.f); }
                }
                pk.x = pack_bf16(e0, e1);
                pk.y = pack_bf16(e2, e3);
                *reinterpret_cast<uint2 *>(region + gm::epi_write_off(0, fj, q, lane, WTN)) = pk;
            }
        // same wave wrote and reads: LDS operations of one wave complete in order
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            int row, c, off;
            gm::epi_read_map(it, lane, WTN, &row, &c, &off);
            const u32x4 v = *reinterpret_cast<const u32x4 *>(region + off);
            const long gr = m0 + 64 * wr + 32 * fi + row;
            const long gc = ncol0 + 8 * c;
            const bool ok = gr < g.M && gc + 8 <= g.N && !g.debug_no_store;
            u32x4 o = v;
            if (ACT == ACT_GELU_FWD) {
                u32x4 a2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = bf16_lo(v[e]), x1 = bf16_hi(v[e]);
                    a2[e] = g.gelu_tanh ? pack_bf16(gelu_val<true>(x0), gelu_val<true>(x1)) : pack_bf16(gelu_val<false>(x0), gelu_val<false>(x1));
                }
                if (ok) {
                    u32x4 *p2 = reinterpret_cast<u32x4 *>(reinterpret_cast<__hip_bfloat16 *>(g.C2) + gr * g.ldc + gc);
                    if (g.nt_store) __builtin_nontemporal_store(a2, p2); else *p2 = a2;
                }
            } else if (ACT == ACT_GELU_BWD) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d0 = g.gelu_tanh ? gelu_grad<true>(bf16_lo(hv[it][e])) : gelu_grad<false>(bf16_lo(hv[it][e]));
                    const float d1 = g.gelu_tanh ? gelu_grad<true>(bf16_hi(hv[it][e])) : gelu_grad<false>(bf16_hi(hv[it][e]));
                    o[e] = pack_bf16(bf16_lo(v[e]) * d0, bf16_hi(v[e]) * d1);
                    if (gr >= m_lo) { csum[2 * e] += bf16_lo(o[e]); csum[2 * e + 1] += bf16_hi(o[e]); }
                }
            }
            if (ACT == ACT_NONE && g.H && ok) {      // out_mask of a convolution's data gradient (gemm_w64x64_kernel; the duo products have no H)
                const u32x4 mk = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const __hip_bfloat16 *>(g.H) + gr * g.ldc + gc);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = keep_where_positive(v[e], mk[e]);
            }
            if (ok && (ACT != ACT_GELU_FWD || g.C != nullptr)) {
                u32x4 *p1 = reinterpret_cast<u32x4 *>(C + gr * g.ldc + gc);
                if (g.nt_store) __builtin_nontemporal_store(o, p1); else *p1 = o;
            }
        }
    }
    if (ACT == ACT_GELU_BWD && g.colpart) {
        // lanes l, l + 8, ..., l + 56 hold the same 8 columns (c = l & 7) for different rows; one colpart row per (tile row, wave row)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = csum[e];
            t += __shfl_xor(t, 8);
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            csum[e] = t;
        }
        const long gc = ncol0 + 8 * (lane & 7);
        if (lane < 8 && gc + 8 <= g.N) {
            float *cp = g.colpart + (2 * trow + wr) * g.N + gc;
            *reinterpret_cast<float4 *>(cp) = make_float4(csum[0], csum[1], csum[2], csum[3]);
            *reinterpret_cast<float4 *>(cp + 4) = make_float4(csum[4], csum[5], csum[6], csum[7]);
        }
    }
}

// fp32 slab of a K-split duo item: [128][256], this wave's 64 x 64 block
__device__ __forceinline__ void duo_epilogue_slab(const f32x16 (&acc)[2][2], float *S, int wr, int wc, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int fi = 0; fi < 2; ++fi) {
        const int row = 64 * wr + 32 * fi + (lane & 31);
#pragma unroll
        for (int fj = 0; fj < 2; ++fj)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4 *>(S + row * 256 + 64 * wc + 32 * fj + 8 * q + 4 * h) =
                    make_float4(acc[fi][fj][4 * q + 0], acc[fi][fj][4 * q + 1], acc[fi][fj][4 * q + 2], acc[fi][fj][4 * q + 3]);
    }
}

// tile pointer + wave-uniform deltas in scalar registers, ONE per-lane offset register per operand (xq_gemm_map.hpp)
template <int KIND, bool IS_A>
struct DuoStager : gm::DuoStagerAddr<KIND, IS_A> {
    using gm::DuoStagerAddr<KIND, IS_A>::voff;
    using gm::DuoStagerAddr<KIND, IS_A>::cur;
    using gm::DuoStagerAddr<KIND, IS_A>::adv;
    using gm::DuoStagerAddr<KIND, IS_A>::d_i;
    using gm::DuoStagerAddr<KIND, IS_A>::d_half;
    static __device__ __forceinline__ long sgpr64(long v) {
        const unsigned long long u = (unsigned long long)v;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return (long)(((unsigned long long)hi << 32) | lo);
    }
    __device__ __forceinline__ void make_scalar() {
        cur = (const char *)sgpr64((long)cur);
        adv = __builtin_amdgcn_readfirstlane(adv); d_i = __builtin_amdgcn_readfirstlane(d_i); d_half = __builtin_amdgcn_readfirstlane(d_half);
    }
    // the two LDS-DMA instructions of piece `half` of the K tile under the cursor; the tile pointer goes through an opaque scalar so that
    // loop strength reduction cannot turn (cursor + per-lane offset) into per-instruction 64-bit VECTOR induction variables (12 VGPRs + spills)
    __device__ __forceinline__ void issue_cur(int half, char *dst, int wave) const {
        unsigned long long b = (unsigned long long)(cur + (half ? d_half : 0u));
        asm volatile("" : "+s"(b));
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void *)((const char *)(b + (i ? (unsigned long long)d_i : 0ull)) + voff),
                                             (lds_void *)(dst + (8 * i + wave) * 1024), 16, 0, 0);
    }
    __device__ __forceinline__ void issue_one(int half, int i, char *dst, int wave) const {
        unsigned long long b = (unsigned long long)(cur + (half ? d_half : 0u) + (i ? d_i : 0u));
        asm volatile("" : "+s"(b));
        __builtin_amdgcn_global_load_lds((gbl_void *)((const char *)b + voff), (lds_void *)(dst + (8 * i + wave) * 1024), 16, 0, 0);
    }
    __device__ __forceinline__ void step() { cur += adv; }
};

// ---- the K loop of the duo schedules (gemm_duo_kernel, gemm_pduo_kernel): macros over the kernels' local names (smem, acc, af, bf, sa, sb, a_rd,
//      b_rd, wave, lane, wr, wc, BK, ASM_TR; DU_Q / DU_Q_ACC: the clock stamps of the SUMS twins) ------------------------------------------------
// Fragment reads are software-pipelined under the MFMAs with NO extra registers: the barrier of a phase certifies the pieces of the NEXT phase, so
// each fragment register is reloaded right behind the MFMA pair that consumed it and the read latency runs under the rest of the cluster and
// the barrier wait (tools/duo_timeline.py on the first form — reads in front of the cluster: the counted vmcnt waits ~20 cycles, the data is
// always there; a phase paid a 255 - 384-cycle read window with the matrix pipe idle for this workgroup).  The LDS-DMA instructions of a phase
// sit between its MFMA pairs as well: their issue time (~100 cycles each with the TA queue busy) hides in the matrix pipe's 32-cycle slots.
//   (t, 0)  [vmcnt: B-right(t) landed] [lgkmcnt(0): my reads of A(t), B-left(t) returned] barrier
//           MFMAs on A(t) x B-left(t); behind pair s: bf[s] <- B-right(t);  stages B-right(t + 1) -> slot of B-left(t), A(t + 2) -> slot of A(t)
//   (t, 1)  [vmcnt: A(t + 1), B-left(t + 1) landed] [lgkmcnt(0)] barrier
//           MFMAs on A(t) x B-right(t); behind pair s: bf[s] <- B-left(t + 1), af[.][s] <- A(t + 1);  stages B-left(t + 2) -> slot of B-right(t)
//   stream  A0 BL0 BR0 A1 BL1 | BR1 A2 | BL2 | BR2 A3 | BL3 | ...   (all five slots are filled before the first MFMA; 2 LDS-DMA instructions per
//           piece and wave)
//   RAW     a piece is read after every wave's own counted vmcnt AND the barrier behind it: vmcnt(4) in both phases (behind the pieces a phase
//           certifies the stream holds two more); tile n - 2: 4, 2; tile n - 1: 0, 0
//   WAR     a slot is restaged behind the barrier in front of which EVERY wave has seen its reads of that slot return (the lgkmcnt(0))
// b_rd: slot of the B piece whose fragments are in bf; the next piece sits in DUO_BNEXT(b_rd), piece + 3 is staged into b_rd itself.
#define DUO_ASLOT(I) (smem + (I) * gm::PIECE_BYTES)
#define DUO_BSLOT(I) (smem + (2 + (I)) * gm::PIECE_BYTES)
#define DUO_BNEXT(I) ((I) == 2 ? 0 : (I) + 1)
#define DUO_PIN(X) asm volatile("" : "+v"(X))
#define DUO_PAIR(FJ, S_)                                                                                       \
    do {                                                                                                       \
        acc[0][FJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[S_], af[0][S_], acc[0][FJ], 0, 0, 0);          \
        acc[1][FJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[S_], af[1][S_], acc[1][FJ], 0, 0, 0);          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    } while (0)
#define DUO_RD_B(S_) bf[S_] = read_frag<BK, false, ASM_TR>(DUO_BSLOT(DUO_BNEXT(b_rd)), wc, 0, S_, lane)
#define DUO_RD_A(S_)                                                                                           \
    do {                                                                                                       \
        af[0][S_] = read_frag<gm::KMAJOR, true>(DUO_ASLOT(a_rd ^ 1), wr, 0, S_, lane);                         \
        af[1][S_] = read_frag<gm::KMAJOR, true>(DUO_ASLOT(a_rd ^ 1), wr, 1, S_, lane);                         \
    } while (0)
#define DUO_SB() __builtin_amdgcn_sched_barrier(0)
#define DUO_HEAD(V_)                                                          \
    do {                                                                      \
        DU_Q(q_a);                                                            \
        GR_VMCNT(V_);                                                         \
        __builtin_amdgcn_sched_barrier(0);                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    \
        DU_Q(q_v);                                                            \
        GR_BARRIER();                                                         \
        DU_Q(q_b);                                                            \
    } while (0)
// one K tile.  ST_BR / ST_A / ST_BL: stage B-right(t + 1), A(t + 2), B-left(t + 2); RDN: tile t + 1 exists (its A, B-left are read in (t, 1))
#define DUO_TILE(ST_BR, ST_A, ST_BL, RDN, V0, V1)                                                                     \
    do {                                                                                                              \
        DUO_HEAD(V0);                                                                                                 \
        if (ST_BR) sb.issue_one(1, 0, DUO_BSLOT(b_rd), wave);                                                         \
        DUO_PIN(acc[0][0]); DUO_PIN(acc[1][0]);                                                                       \
        __builtin_amdgcn_s_setprio(1);                                                                                \
        DUO_PAIR(0, 0);                                                                                               \
        DUO_RD_B(0); if (ST_BR) { sb.issue_one(1, 1, DUO_BSLOT(b_rd), wave); sb.step(); } DUO_SB();                   \
        DUO_PAIR(0, 1);                                                                                               \
        DUO_RD_B(1); if (ST_A) sa.issue_one(0, 0, DUO_ASLOT(a_rd), wave); DUO_SB();                                   \
        DUO_PAIR(0, 2);                                                                                               \
        DUO_RD_B(2); if (ST_A) { sa.issue_one(0, 1, DUO_ASLOT(a_rd), wave); sa.step(); } DUO_SB();                    \
        DUO_PAIR(0, 3);                                                                                               \
        DUO_RD_B(3); DUO_SB();                                                                                        \
        DUO_PIN(acc[0][0]); DUO_PIN(acc[1][0]);                                                                       \
        __builtin_amdgcn_s_setprio(0);                                                                                \
        DU_Q(q_e);                                                                                                    \
        DU_Q_ACC();                                                                                                   \
        b_rd = DUO_BNEXT(b_rd);                                                                                       \
        DUO_HEAD(V1);                                                                                                 \
        if (ST_BL) sb.issue_one(0, 0, DUO_BSLOT(b_rd), wave);                                                         \
        DUO_PIN(acc[0][1]); DUO_PIN(acc[1][1]);                                                                       \
        __builtin_amdgcn_s_setprio(1);                                                                                \
        DUO_PAIR(1, 0);                                                                                               \
        if (RDN) { DUO_RD_B(0); DUO_RD_A(0); } if (ST_BL) sb.issue_one(0, 1, DUO_BSLOT(b_rd), wave); DUO_SB();        \
        DUO_PAIR(1, 1);                                                                                               \
        if (RDN) { DUO_RD_B(1); DUO_RD_A(1); } DUO_SB();                                                              \
        DUO_PAIR(1, 2);                                                                                               \
        if (RDN) { DUO_RD_B(2); DUO_RD_A(2); } DUO_SB();                                                              \
        DUO_PAIR(1, 3);                                                                                               \
        if (RDN) { DUO_RD_B(3); DUO_RD_A(3); } DUO_SB();                                                              \
        DUO_PIN(acc[0][1]); DUO_PIN(acc[1][1]);                                                                       \
        __builtin_amdgcn_s_setprio(0);                                                                                \
        DU_Q(q_e);                                                                                                    \
        DU_Q_ACC();                                                                                                   \
        b_rd = DUO_BNEXT(b_rd);                                                                                       \
        a_rd ^= 1;                                                                                                    \
    } while (0)

// one workgroup per tile (the form the schedule was developed and timed in; tools/duo_timeline.py reads its clock stamps)
template <int BK, int ACT, bool SUMS = false>
__global__ __attribute__((amdgpu_flat_work_group_size(GT, GT), amdgpu_waves_per_eu(4, 4))) void gemm_duo_kernel(const GemmArgs g) {
    // SUMS (diagnostics, XQ_GEMM_TRACE_SUMS + xq_gemm_trace_bind; tools/duo_timeline.py): shader-clock reads at the workgroup's start, K-loop
    // start / end and end, the CU it ran on, the 100 MHz reference ticks of its life, and per-phase segment sums (wait = counted vmcnt, bar =
    // barrier, mfma = the MFMA cluster with the phase's fragment reads and LDS-DMA instructions in it) of every wave — [workgroup][8 waves][16]
    // uint64.  Outputs are bit-identical to the plain kernel.
    unsigned long long q_t0 = 0, q_t1 = 0, q_t2 = 0, q_a = 0, q_v = 0, q_b = 0, q_e = 0;
    unsigned q_wait = 0, q_bar = 0, q_mfma = 0, q_ph = 0;
#define DU_Q(X)                                                   \
    do {                                                          \
        if (SUMS) {                                               \
            __builtin_amdgcn_sched_barrier(0);                    \
            X = __builtin_amdgcn_s_memtime();                     \
            __builtin_amdgcn_sched_barrier(0);                    \
        }                                                         \
    } while (0)
#define DU_Q_ACC()                                                \
    do {                                                          \
        if (SUMS) {                                               \
            q_wait += (unsigned)q_v - (unsigned)q_a;              \
            q_bar += (unsigned)q_b - (unsigned)q_v;               \
            q_mfma += (unsigned)q_e - (unsigned)q_b;              \
            ++q_ph;                                               \
        }                                                         \
    } while (0)
    DU_Q(q_t0);
    const unsigned long long q_real0 = SUMS ? __builtin_amdgcn_s_memrealtime() : 0ull;
    extern __shared__ __attribute__((aligned(16))) char smem[];     // A slots 0, 1 | B slots 0, 1, 2; the epilogue restages through the first 32 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const long total = (long)g.tiles_m * g.tiles_n;
    const long pos = gm::xcd_order(blockIdx.x, total);
    const long trow = pos / g.tiles_n;
    long m0 = trow * gm::BM_DUO, n0 = (pos - trow * g.tiles_n) * 256;
    // ragged edges: the last tile row / column moves back inside the matrix (the overlap is computed twice, identically)
    const long m_lo = m0;                 // rows below m_lo belong to the previous tile row (ACT_GELU_BWD: not summed twice)
    if (m0 > g.M - gm::BM_DUO) m0 = g.M - gm::BM_DUO;
    if (n0 > g.N - 256) n0 = g.N - 256;
    const int KT = __builtin_amdgcn_readfirstlane(g.kt_full);

    DuoStager<gm::KMAJOR, true> sa;
    DuoStager<BK, false> sb;
    sa.init(g.A, g.lda, m0, 0, wave, lane);
    sb.init(g.B, g.ldb, n0, 0, wave, lane);
    sa.make_scalar();
    sb.make_scalar();
    constexpr bool ASM_TR = BK == gm::KSTRIDED;      // K-strided fragments by inline asm (read_frag)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    int a_rd = 0, b_rd = 0;      // slot of the A piece / the B piece whose fragments are in registers (scalar)
    bf16x8 af[2][4], bf[4];
    DU_Q(q_t1);
    sa.issue_cur(0, DUO_ASLOT(0), wave);
    sa.step();
    sb.issue_cur(0, DUO_BSLOT(0), wave);
    sb.issue_cur(1, DUO_BSLOT(1), wave);
    sb.step();
    if (KT > 1) {
        sa.issue_cur(0, DUO_ASLOT(1), wave);
        sa.step();
        sb.issue_cur(0, DUO_BSLOT(2), wave);
        GR_VMCNT(6);
    } else {
        GR_VMCNT(2);
    }
    GR_BARRIER();
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
        bf[s_] = read_frag<BK, false, ASM_TR>(DUO_BSLOT(0), wc, 0, s_, lane);
        af[0][s_] = read_frag<gm::KMAJOR, true>(DUO_ASLOT(0), wr, 0, s_, lane);
        af[1][s_] = read_frag<gm::KMAJOR, true>(DUO_ASLOT(0), wr, 1, s_, lane);
    }
    if (KT > 1) {
        for (int t = 0; t < KT - 2; ++t) DUO_TILE(1, 1, 1, 1, 4, 4);
        DUO_TILE(1, 0, 0, 1, 4, 2);
    }
    DUO_TILE(0, 0, 0, 0, 0, 0);
    GR_BARRIER();      // (the head barrier of the last phase already freed the ring; this one keeps the epilogues of a workgroup together)
    DU_Q(q_t2);
    duo_epilogue<BK, ACT>(acc, g, smem + wave * 4096, m0, n0, m_lo, trow, wr, wc, lane);
    if (SUMS) {
        if (g.trace != nullptr && (long)blockIdx.x * 128 + 128 <= (long)g.trace_cap * 8 && lane == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have been acknowledged
            unsigned long long q_t3 = __builtin_amdgcn_s_memtime();
            unsigned long long *out = g.trace + (long)blockIdx.x * 128 + wave * 16;
            out[0] = q_t0; out[1] = q_t1; out[2] = q_t2; out[3] = q_t3;
            out[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
            out[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
            out[6] = q_ph; out[7] = q_wait; out[8] = q_bar; out[9] = 0; out[10] = q_mfma; out[11] = (unsigned long long)pos;
            out[12] = __builtin_amdgcn_s_getreg((31 << 11) | 6);      // HW_REG_LDS_ALLOC: which half of the CU's LDS this workgroup got
            out[13] = __builtin_amdgcn_s_memrealtime() - q_real0;      // 100 MHz reference ticks over (t0, t3): shader clock = (t3 - t0) / ticks * 100 MHz
        }
    }
#undef DU_Q_ACC
#undef DU_Q
}

// =====================================================================================================================
// persistent duo schedule (round 6): the duo kernel's pipelined K loop (fragment reads under the MFMAs) inside a per-workgroup item loop — grid =
// 2 workgroups per CU, each walks the item list with stride grid (whole 128 x 256 tiles, then the tail tiles beyond the last full round cut along
// K into fp32 slabs, as the 256 x 256 persistent kernel does).  What tools/duo_timeline.py measured on the one-workgroup-per-tile form: a workgroup
// spends 43 % of its life outside the phases of its K loop — launch + tile arithmetic (1.3 k cycles), the first pieces' way from HBM (~8 k), the
// epilogue INCLUDING the wait for its stores' acknowledgement (5 - 10 k) — against 27 k cycles of phases at K = 768.  Here the next item's
// pieces are requested the moment the epilogue's last store is issued, nothing waits for acknowledgements, and no workgroup is relaunched.
//   item boundary   [K loop] [epilogue: 32 KiB of the A slots] barrier [A0 BL0 BR0 A1 BL1 of the next item] vmcnt(6) barrier [fragment reads] ...
//                   (the stores are OLDER than the pieces: they never enter a counted wait's arithmetic, they only make it stricter)
// =====================================================================================================================
template <int BK, int ACT, bool SUMS = false>
__global__ __attribute__((amdgpu_flat_work_group_size(GT, GT), amdgpu_waves_per_eu(4, 4))) void gemm_pduo_kernel(const GemmArgs g0) {
    // The kernel runs at the SGPR limit (two 64-bit tile cursors with their deltas, slot arithmetic, loop state), and a spilled scalar is a scratch
    // load with s_waitcnt vmcnt(0) at the item boundary — a wait for the previous tile's store acknowledgements, the very thing this schedule
    // removes.  So the launch arguments are NOT kept in registers across the K loop: every item re-reads what it needs from the kernarg segment
    // (scalar loads through an opaque copy of the segment pointer, which the optimiser can neither hoist nor merge with earlier reads).
    typedef const __attribute__((address_space(4))) unsigned long long *KArgs;
    const KArgs kp = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    struct ArgWords { unsigned long long w[sizeof(GemmArgs) / 8]; };
    static_assert(sizeof(GemmArgs) % 8 == 0 && sizeof(ArgWords) == sizeof(GemmArgs), "GemmArgs is copied word by word from the kernarg segment");
#define PD_ARGS(NAME)                                                                                  \
    KArgs NAME##_p = kp;                                                                               \
    asm volatile("" : "+s"(NAME##_p));                                                                 \
    ArgWords NAME##_w;                                                                                 \
    _Pragma("unroll") for (unsigned i_ = 0; i_ < sizeof(GemmArgs) / 8; ++i_) NAME##_w.w[i_] = NAME##_p[i_]; \
    const GemmArgs NAME = __builtin_bit_cast(GemmArgs, NAME##_w);
    extern __shared__ __attribute__((aligned(16))) char smem[];     // A slots 0, 1 | B slots 0, 1, 2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const unsigned G = gridDim.x;
    unsigned items;
    unsigned cp = (unsigned)gm::xcd_order(blockIdx.x, G);
    gm::DuoItem cit;
    {
        PD_ARGS(g);
        items = (unsigned)(g.main_items + (long)g.tail_tiles * g.tail_splits);
        if (cp >= items) return;
        gm::decode_item_duo(g, cp, cit);
    }
    // SUMS (diagnostics, tools/duo_timeline.py --persistent): [workgroup][8 waves][16] uint64 — life of the workgroup in shader cycles and 100 MHz
    // reference ticks, items, phases, per-phase segment sums (vmcnt wait, barrier, MFMA cluster), cycles inside K loops / epilogues / boundaries
    const unsigned long long q_t0 = SUMS ? __builtin_amdgcn_s_memtime() : 0ull, q_real0 = SUMS ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned long long q_a = 0, q_v = 0, q_b = 0, q_e = 0, q_x = 0, q_y = 0;
    unsigned q_wait = 0, q_bar = 0, q_mfma = 0, q_ph = 0, q_items = 0, q_loop = 0, q_epi = 0, q_fill = 0;
#define DU_Q(X)                                                   \
    do {                                                          \
        if (SUMS) {                                               \
            __builtin_amdgcn_sched_barrier(0);                    \
            X = __builtin_amdgcn_s_memtime();                     \
            __builtin_amdgcn_sched_barrier(0);                    \
        }                                                         \
    } while (0)
#define DU_Q_ACC()                                                \
    do {                                                          \
        if (SUMS) {                                               \
            q_wait += (unsigned)q_v - (unsigned)q_a;              \
            q_bar += (unsigned)q_b - (unsigned)q_v;               \
            q_mfma += (unsigned)q_e - (unsigned)q_b;              \
            ++q_ph;                                               \
        }                                                         \
    } while (0)

    DuoStager<gm::KMAJOR, true> sa;
    DuoStager<BK, false> sb;
    constexpr bool ASM_TR = BK == gm::KSTRIDED;

    f32x16 acc[2][2];
#define PD_ZERO()                                                    \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                 \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                 \
    _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) acc[i_][j_][r_] = 0.0f;
    PD_ZERO()

    int a_rd = 0, b_rd = 0;      // slot of the A piece / B piece whose fragments are in registers (scalar)
    bf16x8 af[2][4], bf[4];

    for (;;) {
        // ---- the item's first pieces: A0 BL0 BR0 (A1 BL1); every slot is free (the barrier below / the kernel start) ----
        DU_Q(q_x);
        const int KT = __builtin_amdgcn_readfirstlane(cit.KT);
        {
            int lane_s = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));      // = lane (not a live register across the K loop)
            asm volatile("" : "+v"(lane_s));
            PD_ARGS(g);
            sa.init(g.A, g.lda, cit.m0, cit.k0, wave, lane_s);
            sb.init(g.B, g.ldb, cit.n0, cit.k0, wave, lane_s);
        }
        sa.make_scalar();
        sb.make_scalar();
        sa.issue_cur(0, DUO_ASLOT(a_rd), wave);
        sa.step();
        sb.issue_cur(0, DUO_BSLOT(b_rd), wave);
        sb.issue_cur(1, DUO_BSLOT(DUO_BNEXT(b_rd)), wave);
        sb.step();
        if (KT > 1) {
            sa.issue_cur(0, DUO_ASLOT(a_rd ^ 1), wave);
            sa.step();
            sb.issue_cur(0, DUO_BSLOT(DUO_BNEXT(DUO_BNEXT(b_rd))), wave);
            GR_VMCNT(6);
        } else {
            GR_VMCNT(2);
        }
        GR_BARRIER();
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            bf[s_] = read_frag<BK, false, ASM_TR>(DUO_BSLOT(b_rd), wc, 0, s_, lane);
            af[0][s_] = read_frag<gm::KMAJOR, true>(DUO_ASLOT(a_rd), wr, 0, s_, lane);
            af[1][s_] = read_frag<gm::KMAJOR, true>(DUO_ASLOT(a_rd), wr, 1, s_, lane);
        }
        DU_Q(q_y);
        if (SUMS) q_fill += (unsigned)q_y - (unsigned)q_x;
        if (KT > 1) {
            for (int t = 0; t < KT - 2; ++t) DUO_TILE(1, 1, 1, 1, 4, 4);
            DUO_TILE(1, 0, 0, 1, 4, 2);
        }
        DUO_TILE(0, 0, 0, 0, 0, 0);
        DU_Q(q_x);
        if (SUMS) q_loop += (unsigned)q_x - (unsigned)q_y;
        // ---- epilogue (behind the head barrier of the last phase every fragment has been read: the ring is free) ----
        {
            // the lane id goes through an opaque register once per item: everything the epilogue derives from it (LDS offsets, row / column of
            // every store) is then recomputed here instead of being hoisted out of the item loop — 30 registers that do not exist: they were
            // spilled, and every reload (a scratch load) waited with vmcnt(0) for the previous stores' acknowledgement
            int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));      // = lane, from the hardware
            asm volatile("" : "+v"(lane_e));
            PD_ARGS(g);
            if (cit.slab) duo_epilogue_slab(acc, g.slabs + (size_t)cit.slab_idx * (gm::BM_DUO * 256), wr, wc, lane_e);
            else duo_epilogue<BK, ACT>(acc, g, smem + wave * 4096, cit.m0, cit.n0, cit.m_lo, cit.trow, wr, wc, lane_e);
        }
        DU_Q(q_y);
        if (SUMS) { q_epi += (unsigned)q_y - (unsigned)q_x; ++q_items; }
        cp += G;
        if (cp >= items) break;
        {
            PD_ARGS(g);
            gm::decode_item_duo(g, cp, cit);
        }
        PD_ZERO()
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's epilogue reads of its LDS region have returned
        GR_BARRIER();                                            // ... every wave's: the A slots may be restaged
    }
    if (SUMS) {
        PD_ARGS(g);
        if (g.trace != nullptr && (long)blockIdx.x * 128 + 128 <= (long)g.trace_cap * 8 && lane == 0) {
            unsigned long long *out = g.trace + (long)blockIdx.x * 128 + wave * 16;
            out[0] = q_t0; out[1] = 0; out[2] = 0; out[3] = __builtin_amdgcn_s_memtime();
            out[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
            out[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
            out[6] = q_ph; out[7] = q_wait; out[8] = q_bar; out[9] = q_items; out[10] = q_mfma; out[11] = q_loop;
            out[12] = q_epi; out[13] = __builtin_amdgcn_s_memrealtime() - q_real0; out[14] = q_fill;
        }
    }
#undef PD_ZERO
#undef PD_ARGS
#undef DU_Q_ACC
#undef DU_Q
}

#undef DUO_TILE
#undef DUO_HEAD
#undef DUO_SB
#undef DUO_RD_A
#undef DUO_RD_B
#undef DUO_PAIR
#undef DUO_PIN
#undef DUO_BNEXT
#undef DUO_BSLOT
#undef DUO_ASLOT

// slabs [item = tail tile * nsplit + split][256][256] fp32 -> output: the sum over the splits of every tail tile,
//   bf16 (+ bias) into C (NT / NN tail tiles), or fp32 into out (TN; + the < 64 remainder rows of the reduction, folded in here)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float *__restrict__ slabs, int nsplit, long first_tile, int tiles_n,
                                                          long M, long N, long ldc, const float *__restrict__ bias,
                                                          __hip_bfloat16 *__restrict__ out16, float *__restrict__ out32,
                                                          const __hip_bfloat16 *__restrict__ G, const __hip_bfloat16 *__restrict__ X,
                                                          long r_begin, long r_end, __hip_bfloat16 *__restrict__ out16_act, int gelu_tanh,
                                                          int tile_rows, int move_back) {
    // tile_rows = 256 (persistent schedule; grid.y = 32) or 128 (persistent duo schedule; grid.y = 16, edge tiles moved back inside the matrix)
    const long t = blockIdx.x;
    const long tile = first_tile + t;
    long m0 = (tile / tiles_n) * tile_rows, n0 = (tile % tiles_n) * 256;
    if (move_back) {
        if (m0 > M - tile_rows) m0 = M - tile_rows;
        if (n0 > N - 256) n0 = N - 256;
    }
    const int rl = blockIdx.y * 8 + (threadIdx.x >> 5), cl = (threadIdx.x & 31) * 8;
    const long gr = m0 + rl, gc = n0 + cl;
    if (gr >= M || gc + 8 > N) return;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // four splits requested before any is added (the adds keep their ascending order: same bits): one split per trip made the walk a chain of
    // dependent memory round trips — ~300 blocks on 256 CUs, nothing else to hide them (9.3 us per launch, 313 launches per train step)
    const float *sp0 = slabs + ((size_t)(t * nsplit) * tile_rows + rl) * 256 + cl;
    const size_t sstride = (size_t)tile_rows * 256;
    int k = 0;
    for (; k + 4 <= nsplit; k += 4) {
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float *sp = sp0 + (size_t)(k + u) * sstride;
            a[u] = *reinterpret_cast<const float4 *>(sp);
            b[u] = *reinterpret_cast<const float4 *>(sp + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[0] += a[u].x; v[1] += a[u].y; v[2] += a[u].z; v[3] += a[u].w;
            v[4] += b[u].x; v[5] += b[u].y; v[6] += b[u].z; v[7] += b[u].w;
        }
    }
    for (; k < nsplit; ++k) {
        const float *sp = sp0 + (size_t)k * sstride;
        const float4 a = *reinterpret_cast<const float4 *>(sp), b = *reinterpret_cast<const float4 *>(sp + 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
        v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    if (out32) {
        for (long r = r_begin; r < r_end; ++r) {
            const float gv = __bfloat162float(G[r * M + gr]);
            const __hip_bfloat16 *x = X + r * N + gc;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(gv, __bfloat162float(x[e]), v[e]);
        }
        float *o = out32 + gr * ldc + gc;
        *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4 *>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        if (bias) {
            const float4 a = *reinterpret_cast<const float4 *>(bias + gc), b = *reinterpret_cast<const float4 *>(bias + gc + 4);
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
            v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
        }
        uint4 pk;
        pk.x = pack_bf16(v[0], v[1]); pk.y = pack_bf16(v[2], v[3]); pk.z = pack_bf16(v[4], v[5]); pk.w = pack_bf16(v[6], v[7]);
        if (out16) *reinterpret_cast<uint4 *>(out16 + gr * ldc + gc) = pk;
        if (out16_act) {      // ACT_GELU_FWD tail tiles: the activation of the bf16-ROUNDED pre-activation, as the whole-tile epilogue does
            const unsigned u[4] = {pk.x, pk.y, pk.z, pk.w};
            unsigned a[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const act_f2 x2 = {bf16_lo(u[e]), bf16_hi(u[e])};
                const act_f2 y2 = gelu_tanh ? gelu_val2<true>(x2) : gelu_val2<false>(x2);
                a[e] = pack_bf16(y2.x, y2.y);
            }
            *reinterpret_cast<uint4 *>(out16_act + gr * ldc + gc) = make_uint4(a[0], a[1], a[2], a[3]);
        }
    }
}

// ACT_GELU_BWD tail tiles: g_h = (sum of the splits' slabs, rounded to bf16) * gelu'(H), and the column sums of g_h over each 128-row
// half of the tile -> colpart[2 * row_tile + half][N], exactly the rows the whole-tile epilogue writes for its tiles.
// grid (tail tiles, 2 halves, 4 column quarters), 1024 threads: thread = (row of the half 0..127, 8-column group 0..7) — ONE row per thread.
// (Until round 6: grid (tail tiles, 2) with four rows per thread — the tail tiles of M = 65 664 are the half-empty last tile row, so 12 of the
// 24 blocks did all the work on 12 of 256 CUs: 18.7 us per launch.)  The column sums keep the order of that kernel — per 32-row lane the rows
// r, r + 32, r + 64, r + 96, then the 32 lanes ascending — so colpart is unchanged bit for bit.
__global__ __launch_bounds__(1024) void slab_reduce_gelu_bwd_kernel(const float *__restrict__ slabs, int nsplit, long first_tile, int tiles_n,
                                                                    long M, long N, long ldc, const __hip_bfloat16 *__restrict__ H,
                                                                    __hip_bfloat16 *__restrict__ out16, float *__restrict__ colpart, int gelu_tanh,
                                                                    int tile_rows, int move_back) {
    // tile_rows = 256: colpart row (m0 / 128) + half, 128 rows per half;  tile_rows = 128 (persistent duo schedule): colpart row 2 * tile row +
    // half, 64 rows per half, edge tiles moved back inside the matrix (their rows above the tile row's own first row are not summed again)
    __shared__ float red[128][65];
    const long t = blockIdx.x;
    const long tile = first_tile + t;
    const long trow = tile / tiles_n;
    long m0 = trow * tile_rows, n0 = (tile % tiles_n) * 256;
    const long m_lo = m0;
    if (move_back) {
        if (m0 > M - tile_rows) m0 = M - tile_rows;
        if (n0 > N - 256) n0 = N - 256;
    }
    const int half_rows = tile_rows / 2;
    const int half = blockIdx.y, quarter = blockIdx.z, rh = threadIdx.x >> 3, cq = (threadIdx.x & 7) * 8;
    const int rl = half_rows * half + rh, cl = 64 * quarter + cq;
    const long gr = m0 + rl, gc = n0 + cl;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (rh < half_rows && gr < M && gc + 8 <= N) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const uint4 hq = *reinterpret_cast<const uint4 *>(H + gr * ldc + gc);
        const float *sp0 = slabs + ((size_t)(t * nsplit) * tile_rows + rl) * 256 + cl;
        const size_t sstride = (size_t)tile_rows * 256;
        int k = 0;
        for (; k + 4 <= nsplit; k += 4) {       // four splits in flight, added in ascending order (see slab_reduce_kernel)
            float4 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float *sp = sp0 + (size_t)(k + u) * sstride;
                a[u] = *reinterpret_cast<const float4 *>(sp);
                b[u] = *reinterpret_cast<const float4 *>(sp + 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[0] += a[u].x; v[1] += a[u].y; v[2] += a[u].z; v[3] += a[u].w;
                v[4] += b[u].x; v[5] += b[u].y; v[6] += b[u].z; v[7] += b[u].w;
            }
        }
        for (; k < nsplit; ++k) {
            const float *sp = sp0 + (size_t)k * sstride;
            const float4 a = *reinterpret_cast<const float4 *>(sp), b = *reinterpret_cast<const float4 *>(sp + 4);
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
            v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
        }
        const unsigned hu[4] = {hq.x, hq.y, hq.z, hq.w};
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned vr = pack_bf16(v[2 * e], v[2 * e + 1]);      // the whole-tile epilogue multiplies the bf16-rounded product too
            const act_f2 h2 = {bf16_lo(hu[e]), bf16_hi(hu[e])};
            const act_f2 d2 = act_f2{bf16_lo(vr), bf16_hi(vr)} * (gelu_tanh ? gelu_grad2<true>(h2) : gelu_grad2<false>(h2));
            o[e] = pack_bf16(d2.x, d2.y);
            if (gr >= m_lo) {
                cs[2 * e] = bf16_lo(o[e]);
                cs[2 * e + 1] = bf16_hi(o[e]);
            }
        }
        *reinterpret_cast<uint4 *>(out16 + gr * ldc + gc) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    if (rh < 128) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rh][cq + e] = cs[e];
    }
    __syncthreads();
    if (colpart && threadIdx.x < 64) {
        const long c = n0 + 64 * quarter + threadIdx.x;
        if (c < N) {
            float tsum = 0.f;
            const int groups = half_rows / 32;       // 4 (256-row tiles) or 2 (128-row tiles): rows r, r + 32, ... of one 32-row lane first
            for (int r = 0; r < 32; ++r) {
                float lane_sum = 0.f;
                for (int i = 0; i < groups; ++i) lane_sum += red[32 * i + r][threadIdx.x];
                tsum += lane_sum;
            }
            colpart[(tile_rows == 256 ? (m0 / 128) + half : 2 * trow + half) * N + c] = tsum;
        }
    }
}

// sum of the split-K slabs (+ the < 64-row remainder of the reduction, folded in here so the tile kernels only see whole K tiles)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ slabs, int splits, long P, long Q,
                                                            const __hip_bfloat16 *__restrict__ G, const __hip_bfloat16 *__restrict__ X,
                                                            long r_begin, long r_end, float *__restrict__ out) {
    const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= P * Q) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < splits; ++k) {
        const float4 v = *reinterpret_cast<const float4 *>(slabs + (size_t)k * P * Q + e);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (r_begin < r_end) {
        const long p = e / Q, q = e - p * Q;
        for (long r = r_begin; r < r_end; ++r) {
            const float gv = __bfloat162float(G[r * P + p]);
            const __hip_bfloat16 *x = X + r * Q + q;
            s.x = __builtin_fmaf(gv, __bfloat162float(x[0]), s.x);
            s.y = __builtin_fmaf(gv, __bfloat162float(x[1]), s.y);
            s.z = __builtin_fmaf(gv, __bfloat162float(x[2]), s.z);
            s.w = __builtin_fmaf(gv, __bfloat162float(x[3]), s.w);
        }
    }
    *reinterpret_cast<float4 *>(out + e) = s;
}

// 256 x 128 tiles with the eight waves as 4 x 2, 64 x 64 outputs each (round 6).  gemm_simple_kernel<.., 128, ..> runs them as 2 x 4 waves of
// 128 rows x 32 columns: 4 A + 1 B fragment reads per 4 MFMAs = 160 KiB of LDS reads per K tile, 1274 LDS cycles against 1024 matrix-pipe cycles —
// the 128-channel convolutions (LPIPS conv2_x, the CNN tokenizer's 128-channel levels) were LDS-read-bound at 600-700 TF/s.  64 x 64 per wave reads
// 2 A + 2 B fragments per 4 MFMAs (128 KiB per K tile).  Same staging, same LDS pieces, same fragment maps (wave row wr -> A piece wr & 1, half wr >> 1;
// wave column wc -> B fragments 2 wc, 2 wc + 1 of the one B piece), same k order per output element: bit-identical to the simple schedule.
template <int AK, int BK>
__global__ __launch_bounds__(GT) void gemm_w64x64_kernel(const GemmArgs g0) {
    constexpr int BN = 128, WTN = 32, NPB = 3, BUF = NPB * gm::PIECE_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // THREE tile buffers of 48 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;      // 4 x 2
    long m0, n0, split;
    if (!tile_of_block(g0, BN, m0, n0, split)) return;
    const GemmArgs &g = g0;
    const int KT = g.ktiles;

    Stager<AK, true> sa;
    Stager<BK, false> sb;
    sa.bind(g);
    sa.init(g.A, g.lda, m0, g.M, 0, wave, lane, WTN, 2);
    sb.init(g.B, g.ldb, n0, g.N, 0, wave, lane, WTN, 1);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // 6 LDS-DMA instructions per wave and K tile (3 pieces x 2)
    auto stage_tile = [&](long kt, int buf) {
        char *b = smem + buf * BUF;
        sa.issue(0, kt, b, wave);
        sa.issue(1, kt, b + gm::PIECE_BYTES, wave);
        sb.issue(0, kt, b + 2 * gm::PIECE_BYTES, wave);
    };

    // K tiles t + 1 and t + 2 are in flight while tile t is multiplied: ONE barrier per K tile, behind a counted wait — vmcnt(6): everything but the
    // six instructions of tile t + 1 has landed (in-order retirement) — instead of the simple schedule's vmcnt(0) + barrier with nothing in flight.
    // The barrier also says every wave has finished reading tile t - 1, whose buffer tile t + 2 is then staged into.
    stage_tile(0, 0);
    if (KT > 1) stage_tile(1, 1);
    int cur = 0, nxt2 = 2;
    for (int t = 0; t < KT; ++t) {
        if (t + 1 < KT) GR_VMCNT(6); else GR_VMCNT(0);
        GR_BARRIER();
        if (t + 2 < KT) stage_tile(t + 2, nxt2);
        const char *b = smem + cur * BUF;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 bf[2], af[2];
#pragma unroll
            for (int fj = 0; fj < 2; ++fj) bf[fj] = read_frag<BK, false>(b + 2 * gm::PIECE_BYTES, 2 * wc + fj, 0, s, lane);
#pragma unroll
            for (int fi = 0; fi < 2; ++fi) af[fi] = read_frag<AK, true>(b + (wr & 1) * gm::PIECE_BYTES, wr >> 1, fi, s, lane);
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int fj = 0; fj < 2; ++fj) acc[fi][fj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[fj], af[fi], acc[fi][fj], 0, 0, 0);
        }
        cur = cur == 2 ? 0 : cur + 1;
        nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
    }
    GR_BARRIER();      // every wave is past its last fragment read: the buffers become the epilogue's staging regions
    duo_epilogue<BK, ACT_NONE>(acc, g, smem + wave * 4096, m0, n0, m0, 0, wr, wc, lane);
}

template <void (*KERNEL)(const GemmArgs)>
int set_lds(int bytes) {
    static unsigned long long devs = 0;   // one static per kernel instantiation; one bit per device
    if (first_call_on_this_device(&devs)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
            return -1;
    }
    return 0;
}

// 256-column tiles (ring / persistent schedules) from N = 256 on;
// impl bit 8 (XQ_GEMM_WIDE_TILES) forces them (tests, tuning)
int pick_bn(long N, int impl = 0) {
    if (impl & XQ_GEMM_WIDE_TILES) return 256;
    return N >= 256 ? 256 : 128;     // measured (profiles/r02_gemm_shapes_d384.txt): a ragged last 256-column tile on the
                                     // persistent ring beats 128-column tiles on the simple schedule at N = 384 / 1152
}

// splits of the reduction for the weight gradient: fill the chip once, at least two K tiles per split
int tn_splits(long kt_all, long tiles) {
    long s = num_cus() / (tiles > 0 ? tiles : 1);
    if (s > kt_all / 2) s = kt_all / 2;
    if (s < 1) s = 1;
    return (int)s;
}

// persistent schedule: how many whole tiles, how the tiles beyond the last full round of CUs are cut along K
struct PPlan {
    long main_items;
    int tail_tiles, tail_splits;
    size_t slab_bytes;
};
PPlan plan_persistent(long tiles, int kt_full, bool weight_grad) {
    PPlan p{tiles, 0, 1, 0};
    const long G = num_cus();
    if (weight_grad) {
        p.main_items = 0;
        p.tail_tiles = (int)tiles;
        p.tail_splits = tn_splits(kt_full, tiles);
    } else {
        const long rem = tiles % G;
        if (tiles > G && rem > 0 && rem <= G / 4 && kt_full >= 4) {
            long S = G / rem;
            if (S > kt_full / 2) S = kt_full / 2;
            if (S >= 2) { p.main_items = tiles - rem; p.tail_tiles = (int)rem; p.tail_splits = (int)S; }
        }
    }
    p.slab_bytes = (size_t)p.tail_tiles * p.tail_splits * 65536 * sizeof(float);
    return p;
}

// persistent duo schedule: 2 workgroups per CU; the tiles beyond the last full round of 2 x CUs are cut along K (fp32 slabs [item][128][256])
PPlan plan_duo(long tiles, int kt_full) {
    PPlan p{tiles, 0, 1, 0};
    const long G = 2L * num_cus();
    const long rem = tiles % G;
    if (tiles > G && rem > 0 && rem <= G / 4 && kt_full >= 4) {
        long S = G / rem;
        if (S > kt_full / 2) S = kt_full / 2;
        if (S >= 2) { p.main_items = tiles - rem; p.tail_tiles = (int)rem; p.tail_splits = (int)S; }
    }
    p.slab_bytes = (size_t)p.tail_tiles * p.tail_splits * (gm::BM_DUO * 256) * sizeof(float);
    return p;
}

// the persistent kernel, or (XQ_GEMM_TRACE_SUMS with a bound trace buffer; plain NT / NN / TN only) its clock-summing twin
template <int AK, int BK, int ACT>
int launch_pring(const GemmArgs &g, long grid, int lds, hipStream_t s) {
    if constexpr (ACT == ACT_NONE && AK != gm::KMAJOR_CONV) {
        if (g.trace) {
            if (set_lds<gemm_pring_kernel<AK, BK, ACT_NONE, true>>(lds)) return -1;
            hipLaunchKernelGGL((gemm_pring_kernel<AK, BK, ACT_NONE, true>), dim3((unsigned)grid), dim3(GT), lds, s, g);
            return 0;
        }
    }
    if (set_lds<gemm_pring_kernel<AK, BK, ACT>>(lds)) return -1;
    hipLaunchKernelGGL((gemm_pring_kernel<AK, BK, ACT>), dim3((unsigned)grid), dim3(GT), lds, s, g);
    return 0;
}

template <int AK, int BK, int EPI, int ACT = ACT_NONE>
int launch_gemm(GemmArgs g, int BN, int impl, void *ws, size_t ws_bytes, hipStream_t s, const char *fn, double flops, int prof_kind = XQ_PROF_GEMM) {
    const long tiles = (long)g.tiles_m * g.tiles_n;
    if (tiles <= 0) return XQ_OK;
    if (tiles * (g.splits > 0 ? g.splits : 1) > 0x7fffffffL) return xq_set_error(XQ_EINVAL, "%s: too many tiles", fn);
    const bool ring_ok = BN == 256 && g.kt_full >= 2 && (EPI != EPI_F32_SLAB || g.ktiles >= 2);
    if (impl == XQ_GEMM_AUTO) impl = ring_ok ? XQ_GEMM_PERSISTENT : XQ_GEMM_SIMPLE;
    if ((impl == XQ_GEMM_RING || impl == XQ_GEMM_PERSISTENT) && !ring_ok)
        return xq_set_error(XQ_EINVAL, "%s: the ring schedules need 256-column tiles and >= 2 K tiles per work item", fn);
    const int pslot = prof_begin(prof_kind, flops, s);
    if (impl == XQ_GEMM_PERSISTENT) {
        PPlan pl = plan_persistent(tiles, g.kt_full, EPI == EPI_F32_SLAB);
        if (pl.slab_bytes > ws_bytes || (pl.slab_bytes && !ws)) {
            if (EPI == EPI_F32_SLAB) return xq_set_error(XQ_ENOSPACE, "%s: workspace too small", fn);
            pl = PPlan{tiles, 0, 1, 0};      // no workspace: every tile whole
        }
        g.main_items = pl.main_items;
        g.tail_tiles = pl.tail_tiles;
        g.tail_splits = pl.tail_splits;
        g.split_major = (EPI == EPI_F32_SLAB && !g.tile_major_debug) ? 1 : 0;
        g.slabs = (float *)ws;
        const long items = pl.main_items + (long)pl.tail_tiles * pl.tail_splits;
        const long grid = items < num_cus() ? items : num_cus();
        g.step_r = (int)(grid / g.tiles_n);
        g.step_c = (int)(grid % g.tiles_n);
        const int lds = 8 * gm::PIECE_BYTES + 8 * 4096;
        if (launch_pring<AK, BK, ACT>(g, grid, lds, s)) return xq_set_error(XQ_ELAUNCH, "%s: hipFuncSetAttribute failed", fn);
        // tail tiles: sum of the K-range slabs (+ bias / + the fused activation of the whole-tile epilogue)
        if (EPI == EPI_BF16 && pl.tail_tiles) {
            if (ACT == ACT_GELU_BWD)
                hipLaunchKernelGGL(slab_reduce_gelu_bwd_kernel, dim3((unsigned)pl.tail_tiles, 2, 4), dim3(1024), 0, s, (const float *)ws, pl.tail_splits,
                                   pl.main_items, g.tiles_n, g.M, g.N, g.ldc, (const __hip_bfloat16 *)g.H, (__hip_bfloat16 *)g.C, g.colpart, g.gelu_tanh, 256, 0);
            else
                hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)pl.tail_tiles, 32), dim3(256), 0, s, (const float *)ws, pl.tail_splits,
                                   pl.main_items, g.tiles_n, g.M, g.N, g.ldc, g.bias, (__hip_bfloat16 *)g.C, (float *)nullptr,
                                   (const __hip_bfloat16 *)nullptr, (const __hip_bfloat16 *)nullptr, 0L, 0L,
                                   ACT == ACT_GELU_FWD ? (__hip_bfloat16 *)g.C2 : (__hip_bfloat16 *)nullptr, g.gelu_tanh, 256, 0);
        }
    } else if (impl == XQ_GEMM_PDUO) {
        if constexpr (AK == gm::KMAJOR && EPI == EPI_BF16) {
            if (BN != 256 || g.M < gm::BM_DUO || g.N < 256)
                return xq_set_error(XQ_EINVAL, "%s: the duo schedules need M >= 128 and N >= 256 (M=%ld N=%ld)", fn, g.M, g.N);
            g.tiles_m = (int)((g.M + gm::BM_DUO - 1) / gm::BM_DUO);
            const long dtiles = (long)g.tiles_m * g.tiles_n;
            PPlan pl = plan_duo(dtiles, g.kt_full);
            if (pl.slab_bytes > ws_bytes || (pl.slab_bytes && !ws)) pl = PPlan{dtiles, 0, 1, 0};      // no workspace: every tile whole
            g.main_items = pl.main_items;
            g.tail_tiles = pl.tail_tiles;
            g.tail_splits = pl.tail_splits;
            g.slabs = (float *)ws;
            const long items = pl.main_items + (long)pl.tail_tiles * pl.tail_splits;
            if (items > 0x7fffffffL) return xq_set_error(XQ_EINVAL, "%s: too many work items", fn);
            const long grid = items < 2L * num_cus() ? items : 2L * num_cus();
            const int lds = 5 * gm::PIECE_BYTES;
            if (ACT == ACT_NONE && g.trace) {
                if (set_lds<gemm_pduo_kernel<BK, ACT_NONE, true>>(lds)) return xq_set_error(XQ_ELAUNCH, "%s: hipFuncSetAttribute failed", fn);
                hipLaunchKernelGGL((gemm_pduo_kernel<BK, ACT_NONE, true>), dim3((unsigned)grid), dim3(GT), lds, s, g);
            } else {
                if (set_lds<gemm_pduo_kernel<BK, ACT>>(lds)) return xq_set_error(XQ_ELAUNCH, "%s: hipFuncSetAttribute failed", fn);
                hipLaunchKernelGGL((gemm_pduo_kernel<BK, ACT>), dim3((unsigned)grid), dim3(GT), lds, s, g);
            }
            if (pl.tail_tiles) {
                if (ACT == ACT_GELU_BWD)
                    hipLaunchKernelGGL(slab_reduce_gelu_bwd_kernel, dim3((unsigned)pl.tail_tiles, 2, 4), dim3(1024), 0, s, (const float *)ws, pl.tail_splits,
                                       pl.main_items, g.tiles_n, g.M, g.N, g.ldc, (const __hip_bfloat16 *)g.H, (__hip_bfloat16 *)g.C, g.colpart, g.gelu_tanh,
                                       gm::BM_DUO, 1);
                else
                    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)pl.tail_tiles, gm::BM_DUO / 8), dim3(256), 0, s, (const float *)ws, pl.tail_splits,
                                       pl.main_items, g.tiles_n, g.M, g.N, g.ldc, g.bias, (__hip_bfloat16 *)g.C, (float *)nullptr,
                                       (const __hip_bfloat16 *)nullptr, (const __hip_bfloat16 *)nullptr, 0L, 0L,
                                       ACT == ACT_GELU_FWD ? (__hip_bfloat16 *)g.C2 : (__hip_bfloat16 *)nullptr, g.gelu_tanh, gm::BM_DUO, 1);
            }
        } else {
            return xq_set_error(XQ_EINVAL, "%s: the duo schedules serve the K-major A operand with the bf16 epilogue only", fn);
        }
    } else if (impl == XQ_GEMM_DUO) {
        if constexpr (AK == gm::KMAJOR && EPI == EPI_BF16) {
            if (BN != 256 || g.M < gm::BM_DUO || g.N < 256)
                return xq_set_error(XQ_EINVAL, "%s: the duo schedule needs M >= 128 and N >= 256 (M=%ld N=%ld)", fn, g.M, g.N);
            g.tiles_m = (int)((g.M + gm::BM_DUO - 1) / gm::BM_DUO);
            const long total = (long)g.tiles_m * g.tiles_n;
            const int lds = 5 * gm::PIECE_BYTES;
            if (ACT == ACT_NONE && g.trace) {
                if (set_lds<gemm_duo_kernel<BK, ACT_NONE, true>>(lds)) return xq_set_error(XQ_ELAUNCH, "%s: hipFuncSetAttribute failed", fn);
                hipLaunchKernelGGL((gemm_duo_kernel<BK, ACT_NONE, true>), dim3((unsigned)total), dim3(GT), lds, s, g);
            } else {
                if (set_lds<gemm_duo_kernel<BK, ACT>>(lds)) return xq_set_error(XQ_ELAUNCH, "%s: hipFuncSetAttribute failed", fn);
                hipLaunchKernelGGL((gemm_duo_kernel<BK, ACT>), dim3((unsigned)total), dim3(GT), lds, s, g);
            }
        } else {
            return xq_set_error(XQ_EINVAL, "%s: the duo schedule serves the K-major A operand with the bf16 epilogue only", fn);
        }
    } else {
        if (ACT != ACT_NONE) return xq_set_error(XQ_EINVAL, "%s: the fused activation needs the persistent schedule", fn);
        const long total = tiles * g.splits;
        if (impl == XQ_GEMM_RING) {
            const int lds = 8 * gm::PIECE_BYTES;
            if (set_lds<gemm_ring_kernel<AK, BK, EPI>>(lds)) return xq_set_error(XQ_ELAUNCH, "%s: hipFuncSetAttribute failed", fn);
            hipLaunchKernelGGL((gemm_ring_kernel<AK, BK, EPI>), dim3((unsigned)total), dim3(GT), lds, s, g);
        } else if (BN == 256) {
            const int lds = 8 * gm::PIECE_BYTES;
            if (set_lds<gemm_simple_kernel<AK, BK, 256, EPI>>(lds)) return xq_set_error(XQ_ELAUNCH, "%s: hipFuncSetAttribute failed", fn);
            hipLaunchKernelGGL((gemm_simple_kernel<AK, BK, 256, EPI>), dim3((unsigned)total), dim3(GT), lds, s, g);
        } else {
            const int lds = 6 * gm::PIECE_BYTES, lds3 = 9 * gm::PIECE_BYTES;
            static const int w64 = [] { const char *e = getenv("XQ_GEMM_W64X64"); return e ? atoi(e) : 1; }();      // 0: the 2 x 4-wave form of the simple schedule
            bool done = false;
            if constexpr (EPI == EPI_BF16 && ACT == ACT_NONE) {
                // whole-K products of one matrix with the bf16 epilogue (every 128-column convolution and Linear): the 4 x 2-wave kernel
                if (w64 && g.splits == 1 && g.batch_a == 0 && g.batch_b == 0 && g.batch_c == 0) {
                    if (set_lds<gemm_w64x64_kernel<AK, BK>>(lds3)) return xq_set_error(XQ_ELAUNCH, "%s: hipFuncSetAttribute failed", fn);
                    hipLaunchKernelGGL((gemm_w64x64_kernel<AK, BK>), dim3((unsigned)total), dim3(GT), lds3, s, g);
                    done = true;
                }
            }
            if (!done) {
                if (set_lds<gemm_simple_kernel<AK, BK, 128, EPI>>(lds)) return xq_set_error(XQ_ELAUNCH, "%s: hipFuncSetAttribute failed", fn);
                hipLaunchKernelGGL((gemm_simple_kernel<AK, BK, 128, EPI>), dim3((unsigned)total), dim3(GT), lds, s, g);
            }
        }
    }
    prof_end(pslot, s);
    return xq_check_launch(fn);
}

// XQ_GEMM_TRACE_SUMS target (xq_gemm_trace_bind)
unsigned long long *g_trace_buf = nullptr;
int g_trace_cap = 0, g_trace_block = 0;
void bind_trace(GemmArgs &g, int impl) {
    if ((impl & XQ_GEMM_TRACE_SUMS) && g_trace_buf && g_trace_cap >= 16) { g.trace = g_trace_buf; g.trace_cap = g_trace_cap; g.trace_block = g_trace_block; }
}

int check_mnk(const char *fn, int64_t M, int64_t N, int64_t K) {
    if (M < 0 || N < 0 || K < 0) return xq_set_error(XQ_EINVAL, "%s: negative size", fn);
    if (K < 64 || K % 64 || N % 8 || N < 32)
        return xq_set_error(XQ_EINVAL, "%s: needs K %% 64 == 0, N %% 8 == 0, N >= 32 (K=%ld N=%ld)", fn, (long)K, (long)N);
    return XQ_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int xq_gemm_trace_bind(void *buf, int cap_per_wave, int workgroup) {
    if (buf && cap_per_wave < 16) return xq_set_error(XQ_EINVAL, "xq_gemm_trace_bind: cap_per_wave < 16");
    g_trace_buf = (unsigned long long *)buf;
    g_trace_cap = buf ? cap_per_wave : 0;
    g_trace_block = workgroup;
    return XQ_OK;
}

extern "C" size_t xq_gemm_bf16_workspace_bytes(int op, int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K < 0) return 0;
    // enough for either tile width (the XQ_GEMM_WIDE_TILES bit may force 256-column tiles)
    const long tiles256 = ((M + 255) / 256) * ((N + 255) / 256);
    const long tilesbn = ((M + 255) / 256) * ((N + pick_bn(N) - 1) / pick_bn(N));
    const int kt = (int)(K / 64);
    if (op == XQ_GEMM_OP_TN) {
        if (kt < 2) return 0;
        const size_t compact = plan_persistent(tiles256, kt, true).slab_bytes;
        const size_t flat = (size_t)tn_splits(kt, tilesbn) * M * N * sizeof(float);
        return compact > flat ? compact : flat;
    }
    const size_t a = plan_persistent(tiles256, kt, false).slab_bytes;
    const size_t b = (M >= gm::BM_DUO && N >= 256) ? plan_duo(((M + gm::BM_DUO - 1) / gm::BM_DUO) * ((N + 255) / 256), kt).slab_bytes : 0;
    return a > b ? a : b;
}

extern "C" int xq_gemm_bf16_nt(const void *x, const void *w, const float *bias, int64_t M, int64_t N, int64_t K, void *y,
                               void *ws, size_t ws_bytes, int impl, xq_stream_t stream) {
    const char *fn = "xq_gemm_bf16_nt";
    if (int rc = check_mnk(fn, M, N, K)) return rc;
    if (M == 0 || N == 0) return XQ_OK;
    if (!x || !w || !y) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const int BN = pick_bn(N, impl);
    GemmArgs g{};
    g.debug_no_store = (impl & XQ_GEMM_DEBUG_NO_STORE) ? 1 : 0;
    g.nt_store = (impl & XQ_GEMM_PLAIN_STORE) ? 0 : 1;
    bind_trace(g, impl);
    impl &= 0xff;
    g.A = (const char *)x; g.B = (const char *)w; g.bias = bias; g.C = (char *)y;
    g.M = M; g.N = N; g.lda = K; g.ldb = K; g.ldc = N;
    g.ktiles = g.kt_full = (int)(K / 64); g.kt_rem = 0; g.splits = 1;
    g.tiles_m = (int)((M + 255) / 256); g.tiles_n = (int)((N + BN - 1) / BN);
    return launch_gemm<gm::KMAJOR, gm::KMAJOR, EPI_BF16>(g, BN, impl, ws, ws_bytes, (hipStream_t)stream, fn, 2.0 * M * N * K);
}

extern "C" int xq_gemm_bf16_nn(const void *g_y, const void *w, int64_t M, int64_t N, int64_t K, void *g_x, void *ws, size_t ws_bytes,
                               int impl, xq_stream_t stream) {
    const char *fn = "xq_gemm_bf16_nn";
    if (int rc = check_mnk(fn, M, N, K)) return rc;
    if (M == 0 || N == 0) return XQ_OK;
    if (!g_y || !w || !g_x) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const int BN = pick_bn(N, impl);
    GemmArgs g{};
    g.nt_store = (impl & XQ_GEMM_PLAIN_STORE) ? 0 : 1;
    g.debug_no_store = (impl & XQ_GEMM_DEBUG_NO_STORE) ? 1 : 0;
    bind_trace(g, impl);
    impl &= 0xff;
    g.A = (const char *)g_y; g.B = (const char *)w; g.bias = nullptr; g.C = (char *)g_x;
    g.M = M; g.N = N; g.lda = K; g.ldb = N; g.ldc = N;
    g.ktiles = g.kt_full = (int)(K / 64); g.kt_rem = 0; g.splits = 1;
    g.tiles_m = (int)((M + 255) / 256); g.tiles_n = (int)((N + BN - 1) / BN);
    return launch_gemm<gm::KMAJOR, gm::KSTRIDED, EPI_BF16>(g, BN, impl, ws, ws_bytes, (hipStream_t)stream, fn, 2.0 * M * N * K);
}

extern "C" int xq_gemm_bf16_tn(const void *g_y, const void *x, int64_t R, int64_t P, int64_t Q, float *g_w, void *ws, size_t ws_bytes,
                               int impl, xq_stream_t stream) {
    const char *fn = "xq_gemm_bf16_tn";
    if (R < 0 || P < 0 || Q < 0) return xq_set_error(XQ_EINVAL, "%s: negative size", fn);
    if (P == 0 || Q == 0) return XQ_OK;
    if (!g_w || (R > 0 && (!g_y || !x))) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    if (P % 8 || Q % 8 || P < 32 || Q < 32) return xq_set_error(XQ_EINVAL, "%s: needs P, Q multiples of 8 and >= 32 (P=%ld Q=%ld)", fn, (long)P, (long)Q);
    hipStream_t s = (hipStream_t)stream;
    const int BN = pick_bn(Q, impl);
    const int tile_major_debug = (impl & XQ_GEMM_TILE_MAJOR) ? 1 : 0;
    const int impl_bits = impl;
    impl &= 0xff;
    const long kt_all = R / 64;
    GemmArgs g{};
    bind_trace(g, impl_bits);
    g.A = (const char *)g_y; g.B = (const char *)x; g.C = (char *)ws;
    g.M = P; g.N = Q; g.lda = P; g.ldb = Q; g.ldc = Q;
    g.tiles_m = (int)((P + 255) / 256); g.tiles_n = (int)((Q + BN - 1) / BN);
    const long tiles = (long)g.tiles_m * g.tiles_n;
    bool compact = false;
    int splits = 0;
    if (kt_all >= 2) {
        splits = tn_splits(kt_all, tiles);
        g.splits = splits; g.ktiles = (int)(kt_all / splits); g.kt_rem = (int)(kt_all % splits); g.kt_full = (int)kt_all;
        if (impl == XQ_GEMM_AUTO) impl = BN == 256 ? XQ_GEMM_PERSISTENT : XQ_GEMM_SIMPLE;
        compact = impl == XQ_GEMM_PERSISTENT;
        g.tile_major_debug = tile_major_debug;
        if (!compact && (ws_bytes < (size_t)splits * P * Q * 4 || !ws)) return xq_set_error(XQ_ENOSPACE, "%s: workspace too small", fn);
        const int rc = launch_gemm<gm::KSTRIDED, gm::KSTRIDED, EPI_F32_SLAB>(g, BN, impl, ws, ws_bytes, s, fn, 2.0 * P * Q * (double)(kt_all * 64));
        if (rc) return rc;
    }
    const long done = splits ? kt_all * 64 : 0;   // rows covered by whole K tiles
    if (compact) {
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)tiles, 32), dim3(256), 0, s, (const float *)ws, splits, 0L, g.tiles_n, (long)P,
                           (long)Q, (long)Q, (const float *)nullptr, (__hip_bfloat16 *)nullptr, g_w, (const __hip_bfloat16 *)g_y,
                           (const __hip_bfloat16 *)x, done, (long)R, (__hip_bfloat16 *)nullptr, 0, 256, 0);
    } else {
        const long quads = (P * Q) / 4;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, (const float *)ws, splits, (long)P,
                           (long)Q, (const __hip_bfloat16 *)g_y, (const __hip_bfloat16 *)x, done, (long)R, g_w);
    }
    return xq_check_launch(fn);
}

// ---- fused MLP GEMMs (persistent / duo schedules: N >= 256, K >= 128) ----------------------------------------------------
namespace {
int g_fused_impl = -1;      // -1: not yet read from XQ_GEMM_FUSED_SCHEDULE
int fused_impl() {
    if (g_fused_impl < 0) {
        const char *e = getenv("XQ_GEMM_FUSED_SCHEDULE");
        g_fused_impl = e ? (int)strtol(e, nullptr, 0) : XQ_GEMM_AUTO;
    }
    return g_fused_impl;
}
// schedule of a fused product: AUTO = the persistent 256 x 256 schedule; DUO / PDUO where the shape admits them
int pick_fused(int64_t M, int64_t N) {
    const int want = fused_impl();
    if ((want == XQ_GEMM_DUO || want == XQ_GEMM_PDUO) && M >= gm::BM_DUO && N >= 256) return want;
    return XQ_GEMM_PERSISTENT;
}
}  // namespace
extern "C" int xq_gemm_fused_schedule(int impl) {
    const int prev = fused_impl();
    g_fused_impl = impl < 0 ? XQ_GEMM_AUTO : impl;
    return prev;
}

extern "C" int xq_gemm_bf16_nt_gelu(const void *x, const void *w, const float *bias, int64_t M, int64_t N, int64_t K, void *h, void *h_act,
                                    int approximate_tanh, void *ws, size_t ws_bytes, xq_stream_t stream) {
    const char *fn = "xq_gemm_bf16_nt_gelu";
    if (int rc = check_mnk(fn, M, N, K)) return rc;
    if (M == 0 || N == 0) return XQ_OK;
    if (!x || !w || !h_act) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);      // h may be null: inference, the pre-activation is not kept
    if (N < 256 || K < 128) return xq_set_error(XQ_EINVAL, "%s: needs N >= 256 and K >= 128 (N=%ld K=%ld)", fn, (long)N, (long)K);
    GemmArgs g{};
    g.nt_store = 1;
    g.A = (const char *)x; g.B = (const char *)w; g.bias = bias; g.C = (char *)h; g.C2 = (char *)h_act; g.gelu_tanh = approximate_tanh;
    g.M = M; g.N = N; g.lda = K; g.ldb = K; g.ldc = N;
    g.ktiles = g.kt_full = (int)(K / 64); g.kt_rem = 0; g.splits = 1;
    g.tiles_m = (int)((M + 255) / 256); g.tiles_n = (int)((N + 255) / 256);
    return launch_gemm<gm::KMAJOR, gm::KMAJOR, EPI_BF16, ACT_GELU_FWD>(g, 256, pick_fused(M, N), ws, ws_bytes, (hipStream_t)stream, fn, 2.0 * M * N * K);
}

// Partial rows of the GELU' products.  xq_gemm_colpart_rows(M): what a caller ALLOCATES — enough for every schedule (one row per 128-row block
// of the 256-row schedule = 2 per tile; one per 64-row wave tile of the duo schedules = 2 per 128-row tile).  xq_gemm_colpart_rows_written(M, N):
// the leading rows the schedule now in force for that shape writes — what the caller sums (xq_colsum_partials); the rest stays untouched.
extern "C" size_t xq_gemm_colpart_rows(int64_t M) { return M > 0 ? (size_t)(2 * ((M + 127) / 128)) : 0; }
extern "C" size_t xq_gemm_colpart_rows_written(int64_t M, int64_t N) {
    if (M <= 0) return 0;
    return pick_fused(M, N) == XQ_GEMM_PERSISTENT ? (size_t)(2 * ((M + 255) / 256)) : (size_t)(2 * ((M + 127) / 128));
}

namespace {
// g_h = (g_y W2) * GELU'(h) with W2 as stored ([K][N], BK = KSTRIDED: transpose reads) or its transposed copy ([N][K], BK = KMAJOR)
template <int BK>
int gelu_bwd_product(const char *fn, const void *g_y, const void *w, const void *h, int64_t M, int64_t N, int64_t K, void *g_h, float *colpart,
                     int approximate_tanh, void *ws, size_t ws_bytes, xq_stream_t stream) {
    if (int rc = check_mnk(fn, M, N, K)) return rc;
    if (M == 0 || N == 0) return XQ_OK;
    if (!g_y || !w || !h || !g_h) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    if (N < 256 || K < 128) return xq_set_error(XQ_EINVAL, "%s: needs N >= 256 and K >= 128 (N=%ld K=%ld)", fn, (long)N, (long)K);
    GemmArgs g{};
    g.nt_store = 1;
    g.A = (const char *)g_y; g.B = (const char *)w; g.C = (char *)g_h; g.H = (const char *)h; g.colpart = colpart; g.gelu_tanh = approximate_tanh;
    g.M = M; g.N = N; g.lda = K; g.ldb = BK == gm::KMAJOR ? K : N; g.ldc = N;
    g.ktiles = g.kt_full = (int)(K / 64); g.kt_rem = 0; g.splits = 1;
    g.tiles_m = (int)((M + 255) / 256); g.tiles_n = (int)((N + 255) / 256);
    return launch_gemm<gm::KMAJOR, BK, EPI_BF16, ACT_GELU_BWD>(g, 256, pick_fused(M, N), ws, ws_bytes, (hipStream_t)stream, fn, 2.0 * M * N * K);
}
}  // namespace

extern "C" int xq_gemm_bf16_nn_gelu_bwd(const void *g_y, const void *w, const void *h, int64_t M, int64_t N, int64_t K, void *g_h,
                                        float *colpart, int approximate_tanh, void *ws, size_t ws_bytes, xq_stream_t stream) {
    return gelu_bwd_product<gm::KSTRIDED>("xq_gemm_bf16_nn_gelu_bwd", g_y, w, h, M, N, K, g_h, colpart, approximate_tanh, ws, ws_bytes, stream);
}

extern "C" int xq_gemm_bf16_nt_gelu_bwd(const void *g_y, const void *w_t, const void *h, int64_t M, int64_t N, int64_t K, void *g_h,
                                        float *colpart, int approximate_tanh, void *ws, size_t ws_bytes, xq_stream_t stream) {
    return gelu_bwd_product<gm::KMAJOR>("xq_gemm_bf16_nt_gelu_bwd", g_y, w_t, h, M, N, K, g_h, colpart, approximate_tanh, ws, ws_bytes, stream);
}

// ---- 3x3 convolution as an implicit GEMM on the tile engine (NHWC bf16; Cin % 64 == 0, Cout % 8 == 0, Cout >= 64) -------------
extern "C" int xq_conv3x3_gemm_bf16(const void *x, const void *w_packed, const float *bias, int B, int Hi, int Wi, int Cin, int Cout, int Ho,
                                    int Wo, int stride, int pad, int upsample2x, int transposed, int relu, const void *out_mask, void *y,
                                    int impl, xq_stream_t stream) {
    const char *fn = "xq_conv3x3_gemm_bf16";
    if (B < 0 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1) return xq_set_error(XQ_EINVAL, "%s: bad shape", fn);
    if (Cin % 64 || Cout % 8 || Cout < 64) return xq_set_error(XQ_EINVAL, "%s: needs Cin %% 64 == 0, Cout %% 8 == 0, Cout >= 64 (Cin=%ld Cout=%ld)", fn, (long)Cin, (long)Cout);
    if ((stride != 1 && stride != 2) || pad < 0 || pad > 2 || (upsample2x && (stride != 1 || transposed)))
        return xq_set_error(XQ_EINVAL, "%s: stride 1 / 2, pad 0..2, upsampling only with stride 1 forward", fn);
    if (B == 0) return XQ_OK;
    if (!x || !w_packed || !y) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    if ((long)B * Hi * Wi >= 0x7fffffffL || Ho > 32767 || Wo > 32767) return xq_set_error(XQ_EINVAL, "%s: image too large for 32-bit pixel indices", fn);
    const long M = (long)B * Ho * Wo, K = 9L * Cin;
    const int BN = pick_bn(Cout, impl);
    impl &= 0xff;
    GemmArgs g{};
    g.nt_store = 1;
    g.A = (const char *)x; g.B = (const char *)w_packed; g.bias = bias; g.C = (char *)y; g.relu = relu;
    g.H = (const char *)out_mask;
    g.M = M; g.N = Cout; g.lda = K; g.ldb = K; g.ldc = Cout;
    g.ktiles = g.kt_full = (int)(K / 64); g.kt_rem = 0; g.splits = 1;
    g.tiles_m = (int)((M + 255) / 256); g.tiles_n = (int)((Cout + BN - 1) / BN);
    g.cv_Hi = Hi; g.cv_Wi = Wi; g.cv_Cin = Cin; g.cv_Ho = Ho; g.cv_Wo = Wo; g.cv_stride = stride; g.cv_pad = pad; g.cv_up = upsample2x ? 1 : 0;
    g.cv_transposed = transposed ? 1 : 0;
    return launch_gemm<gm::KMAJOR_CONV, gm::KMAJOR, EPI_BF16>(g, BN, impl, nullptr, 0, (hipStream_t)stream, fn, 2.0 * (double)M * (double)K * Cout,
                                                              XQ_PROF_CONV3X3);
}

// ---- batched small products (simple schedule; one launch for `batch` independent matrices): the single-head spatial attention of
//      the CNN AttnBlock (xqgan_model.py:646-656) is five of them per direction -----------------------------------------------------
extern "C" int xq_gemm_bf16_batched(int op, const void *a, const void *b, int batch, int64_t M, int64_t N, int64_t K, int64_t stride_a,
                                    int64_t stride_b, int64_t stride_c, void *c, xq_stream_t stream) {
    const char *fn = "xq_gemm_bf16_batched";
    if (batch < 0 || M < 0 || N < 0 || K < 0) return xq_set_error(XQ_EINVAL, "%s: negative size", fn);
    if (batch == 0 || M == 0 || N == 0) return XQ_OK;
    if (!a || !b || !c) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    if (K < 64 || K % 64 || N % 8 || N < 32 || (op == XQ_GEMM_OP_TN && (M % 8 || M < 32)))
        return xq_set_error(XQ_EINVAL, "%s: needs K %% 64 == 0, N (and M for TN) %% 8 == 0 and >= 32 (N=%ld K=%ld)", fn, (long)N, (long)K);
    if (batch > 65535) return xq_set_error(XQ_EINVAL, "%s: batch > 65535", fn);
    const int BN = pick_bn(N);
    GemmArgs g{};
    g.nt_store = 1;
    g.A = (const char *)a; g.B = (const char *)b; g.C = (char *)c;
    g.M = M; g.N = N; g.ldc = N;
    g.ktiles = g.kt_full = (int)(K / 64); g.kt_rem = 0; g.splits = 1;
    g.tiles_m = (int)((M + 255) / 256); g.tiles_n = (int)((N + BN - 1) / BN);
    g.batch_a = stride_a; g.batch_b = stride_b; g.batch_c = stride_c;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(g.tiles_m * g.tiles_n), (unsigned)batch);
    const int lds = (BN == 256 ? 8 : 6) * gm::PIECE_BYTES;
    const int pslot = prof_begin(XQ_PROF_GEMM, 2.0 * batch * (double)M * (double)N * (double)K, s);
#define BATCHED(AKK, BKK, EPII)                                                                                                     \
    do {                                                                                                                             \
        if (BN == 256) {                                                                                                             \
            if (set_lds<gemm_simple_kernel<AKK, BKK, 256, EPII>>(lds)) return xq_set_error(XQ_ELAUNCH, "%s: hipFuncSetAttribute failed", fn); \
            hipLaunchKernelGGL((gemm_simple_kernel<AKK, BKK, 256, EPII>), grid, dim3(GT), lds, s, g);                                \
        } else {                                                                                                                     \
            if (set_lds<gemm_simple_kernel<AKK, BKK, 128, EPII>>(lds)) return xq_set_error(XQ_ELAUNCH, "%s: hipFuncSetAttribute failed", fn); \
            hipLaunchKernelGGL((gemm_simple_kernel<AKK, BKK, 128, EPII>), grid, dim3(GT), lds, s, g);                                \
        }                                                                                                                            \
    } while (0)
    if (op == XQ_GEMM_OP_NT) { g.lda = K; g.ldb = K; BATCHED(gm::KMAJOR, gm::KMAJOR, EPI_BF16); }
    else if (op == XQ_GEMM_OP_NN) { g.lda = K; g.ldb = N; BATCHED(gm::KMAJOR, gm::KSTRIDED, EPI_BF16); }
    else if (op == XQ_GEMM_OP_TN) { g.lda = M; g.ldb = N; BATCHED(gm::KSTRIDED, gm::KSTRIDED, EPI_F32_SLAB); }
    else return xq_set_error(XQ_EINVAL, "%s: unknown op", fn);
#undef BATCHED
    prof_end(pslot, s);
    return xq_check_launch(fn);
}
