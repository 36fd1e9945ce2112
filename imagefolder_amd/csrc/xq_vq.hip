// xq_vq.hip — single-scale vector quantizer on gfx950 (MI355X): fused l2-normalise + distance +
// argmin (fp32 MFMA), gather/straight-through/loss/histogram epilogue, hand-written backward.
//
// Replaces (reference file:line, lxa9867/ImageFolder):
//   VectorQuantizer.forward            tokenizer/tokenizer_image/xqgan_model.py:745-801
//   VectorQuantizer.f_to_idxBl_or_fhat tokenizer/tokenizer_image/xqgan_model.py:803-833
//   the distance/argmin of add_perturbation and VectorQuantizer2 (latent_perturbation.py:9-18,
//   quant.py:93-101) through xq_assign.
//
// Kernel plan (one forward = 3 launches + 1 memset, no N x V matrix ever reaches HBM):
//   K0 prep_codebook : E[V][C] -> ehat in MFMA B-fragment order + |ehat|^2 (once per forward; 1-2 MB, L2-resident)
//   K1 assign        : 512-thread blocks = 8 waves x 32 tokens; A fragments (zhat) stay in VGPRs for the
//                      whole kernel, codebook chunks stream through double-buffered LDS, one
//                      v_mfma_f32_32x32x2_f32 chain of C/2 instructions per 32x32 (token x code) tile,
//                      running (min, tile) per accumulator register, wave-shuffle reduction, 64-bit
//                      atomicMin of (ordered(d) << 32 | code) so the V axis can be split across blocks
//                      (fills 256 CUs even at N = 32k tokens) with exact lowest-index tie-breaking.
//   K2 vq_finish     : per token: gather E[idx], renormalise, straight-through form, NCHW store,
//                      loss partials, histogram atomics.   (HBM-bound: 8C+8 bytes per token.)
// Roofline: K1 is fp32-MFMA bound: 2*N*V*C flop; K2/backward are HBM bound.
#include "xq_common.hpp"
#include "xq_internal.hpp"
#include "../../include/xq_ops.h"

#include <stdio.h>
#include <string.h>

using namespace xq;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
thread_local char g_err[512] = "";
int xq_set_error(int code, const char *fmt, const char *a, long b, long c) {
    snprintf(g_err, sizeof(g_err), fmt, a, b, c);
    return code;
}
int xq_check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return XQ_OK;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return XQ_ELAUNCH;
}
bool first_call_on_this_device(unsigned long long *mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;      // unknown device: just do the call again (it is idempotent)
    const unsigned long long bit = 1ull << dev;
    const unsigned long long old = __atomic_fetch_or(mask, bit, __ATOMIC_RELAXED);
    return (old & bit) == 0;
}
extern "C" const char *xq_last_error(void) { return g_err; }
extern "C" int xq_abi_version(void) { return XQ_ABI_VERSION; }

// ------------------------------------------------------------------------------------------------
// measurement hooks: HIP events around the hand-written hot kernels on their launch stream (bench.py roofline leg).
// Every instrumented launch records (kind, algorithmic work) and a start/stop event pair.
// ------------------------------------------------------------------------------------------------
static bool g_prof_on = false;
static constexpr int PROF_MAX = 16384;
static hipEvent_t g_prof_ev[PROF_MAX][2];
static int g_prof_kind[PROF_MAX];
static double g_prof_work[PROF_MAX];
static int g_prof_n = 0, g_prof_created = 0;

extern "C" int xq_prof_enable(int on) {     // 0: off (recorded launches stay collectable), 1: reset + arm, 2: arm, keeping what is recorded
    g_prof_on = on != 0;
    if (on == 1) g_prof_n = 0;
    return XQ_OK;
}
extern "C" int xq_prof_collect_kind(int kind, double *ms_total, int *launches, double *work_total) {
    double tot = 0.0, work = 0.0;
    int n = 0;
    for (int i = 0; i < g_prof_n; ++i) {
        if (g_prof_kind[i] != kind) continue;
        float ms = 0.f;
        if (hipEventSynchronize(g_prof_ev[i][1]) != hipSuccess) return xq_set_error(XQ_ELAUNCH, "%s", "hipEventSynchronize failed");
        if (hipEventElapsedTime(&ms, g_prof_ev[i][0], g_prof_ev[i][1]) != hipSuccess) return xq_set_error(XQ_ELAUNCH, "%s", "hipEventElapsedTime failed");
        tot += ms;
        work += g_prof_work[i];
        ++n;
    }
    if (ms_total) *ms_total = tot;
    if (launches) *launches = n;
    if (work_total) *work_total = work;
    return XQ_OK;
}
// per-launch view of one kind: fills ms[i] / work[i] for up to `cap` recorded launches in launch order, returns how many there are
// (tools/prof_gemm_shapes.py groups them by their algorithmic work = by shape)
extern "C" int xq_prof_entries(int kind, double *ms_out, double *work_out, int cap) {
    int n = 0;
    for (int i = 0; i < g_prof_n; ++i) {
        if (g_prof_kind[i] != kind) continue;
        if (n < cap && ms_out && work_out) {
            float ms = 0.f;
            if (hipEventSynchronize(g_prof_ev[i][1]) != hipSuccess || hipEventElapsedTime(&ms, g_prof_ev[i][0], g_prof_ev[i][1]) != hipSuccess) {
                xq_set_error(XQ_ELAUNCH, "%s", "xq_prof_entries: event query failed");
                return -1;
            }
            ms_out[n] = ms;
            work_out[n] = g_prof_work[i];
        }
        ++n;
    }
    return n;
}
extern "C" int xq_prof_collect(double *assign_ms_total, int *assign_launches) {
    const int rc = xq_prof_collect_kind(XQ_PROF_ASSIGN, assign_ms_total, assign_launches, nullptr);
    g_prof_n = 0;
    return rc;
}
// correction of the algorithmic work of the most recent launch of `kind` (e.g. the conv1_1 data gradient runs with its 3 input
// channels zero-padded to 64: the padded flops are not algorithmic work)
extern "C" int xq_prof_add_work(int kind, double delta) {
    if (!g_prof_on) return XQ_OK;
    for (int i = g_prof_n - 1; i >= 0; --i)
        if (g_prof_kind[i] == kind) { g_prof_work[i] += delta; break; }
    return XQ_OK;
}
int xq::prof_begin(int kind, double work, hipStream_t s) {
    if (!g_prof_on || g_prof_n >= PROF_MAX) return -1;
    if (g_prof_n >= g_prof_created) {
        if (hipEventCreate(&g_prof_ev[g_prof_created][0]) != hipSuccess || hipEventCreate(&g_prof_ev[g_prof_created][1]) != hipSuccess) return -1;
        ++g_prof_created;
    }
    const int slot = g_prof_n++;
    g_prof_kind[slot] = kind;
    g_prof_work[slot] = work;
    (void)hipEventRecord(g_prof_ev[slot][0], s);
    return slot;
}
void xq::prof_end(int slot, hipStream_t s) {
    if (slot >= 0) (void)hipEventRecord(g_prof_ev[slot][1], s);
}
// section markers for kernel traces: an empty kernel whose grid size is the section id (tools/rocpd_sections.py)
__global__ void xq_marker_kernel() {}
extern "C" int xq_prof_marker(int id, xq_stream_t stream) {
    if (id < 1) return xq_set_error(XQ_EINVAL, "%s: id must be >= 1", "xq_prof_marker");
    hipLaunchKernelGGL(xq_marker_kernel, dim3((unsigned)id), dim3(64), 0, (hipStream_t)stream);
    return xq_check_launch("xq_prof_marker");
}

// ------------------------------------------------------------------------------------------------
// tiling constants
// ------------------------------------------------------------------------------------------------
template <int C> struct Tiling {
    static constexpr int CH = (C == 8) ? 256 : ((C <= 64) ? 128 : 64);  // codes per LDS stage
    static constexpr int TILES = CH / 32;            // 32-code MFMA tiles per stage
    static constexpr int KQ = C / 8;                 // float4 B-fragment reads per tile per lane (4 k-pairs each)
    static constexpr int STAGE_F4 = CH * C / 4;      // float4 per stage (codebook part)
    static constexpr int BUF_BYTES = CH * C * 4 + CH * 4;
};
#ifndef XQ_LDS_BPC
#define XQ_LDS_BPC 1
#endif
static constexpr int ASSIGN_THREADS = 512;                 // 8 waves = 2 anti-phase wave groups x 4 SIMDs
static constexpr int TOK_PER_BLOCK = ASSIGN_THREADS / 2;   // one 32-token MFMA row tile per wave


extern "C" size_t xq_assign_workspace_bytes(int64_t N, int C, int V) {
    if (N < 0 || V < 1 || C < 1) return 0;
    return assign_ws_layout(N, C, V, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------
// K0: codebook -> fragment-ordered (optionally l2-normalised) copy + squared norms
//   element (code j, channel k) lands at float offset ((T*KQ + q)*64 + lane)*4 + ti with
//   T = j/32, lane = (k&1)*32 + j%32, t = k/2, q = t/4, ti = t%4: lane l of MFMA step t then holds
//   B[k = 2t + (l>>5)][col = l&31] — the v_mfma_f32_32x32x2_f32 operand layout — and 4 consecutive
//   steps come from one ds_read_b128.  Padding codes (j >= V) get ehat = 0 and |e|^2 = +inf.
// ------------------------------------------------------------------------------------------------
template <int C, int MODE>
__global__ __launch_bounds__(256) void prep_codebook_kernel(const float *__restrict__ E, int V, int Vpad,
                                                            float *__restrict__ wb, float *__restrict__ ee) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= Vpad) return;
    float e[C], eh[C];
    float een;
    if (j < V) {
        const float4 *row = reinterpret_cast<const float4 *>(E + (size_t)j * C);
#pragma unroll
        for (int q = 0; q < C / 4; ++q) {
            float4 v = row[q];
            e[4 * q + 0] = v.x; e[4 * q + 1] = v.y; e[4 * q + 2] = v.z; e[4 * q + 3] = v.w;
        }
        if (MODE == XQ_MODE_L2_RAW) {
#pragma unroll
            for (int k = 0; k < C; ++k) eh[k] = e[k];
        } else {
            l2norm_row<C>(e, eh);
        }
        een = (MODE == XQ_MODE_COSINE) ? 0.0f : chain_sq<C>(eh);
    } else {
#pragma unroll
        for (int k = 0; k < C; ++k) eh[k] = 0.0f;
        een = __builtin_inff();
    }
    ee[j] = een;
    constexpr int KQ = C / 8;
    const int T = j >> 5, jj = j & 31;
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const int t = k >> 1, hh = k & 1, q = t >> 2, ti = t & 3;
        wb[(((size_t)T * KQ + q) * 64 + hh * 32 + jj) * 4 + ti] = eh[k];
    }
}

// ------------------------------------------------------------------------------------------------
// K1: fused normalise + distance + argmin
//   512-thread block = 8 waves = two wave groups (waves 0-3 / 4-7: one wave of each group per SIMD).
//   The groups run in anti-phase, separated by one s_barrier per phase:
//       phase p   : group X runs the v_mfma_f32_32x32x2_f32 chains of one LDS stage (TILES x C/2 MFMAs),
//                   group Y retires the accumulators of ITS previous stage (VALU: add, fma, cmp, 2 cndmask
//                   per accumulator register);
//       phase p+1 : roles swap.
//   Measured on MI355X (tools/ubench/mfma_valu_overlap.hip, profiles/r01_mfma_valu_overlap.txt): the f32-input
//   MFMA does NOT run beside fp32 VALU work on the same SIMD — an MFMA-only wave and a VALU-only wave sharing
//   a SIMD take ~the SUM of their solo times (64 -> 77 cycles per MFMA, 67 -> 272 cycles per 32 v_fma), and
//   interleaving VALU between dependent MFMAs of one wave is worse than running them back to back.  So the
//   ceiling of this kernel is MFMA/(MFMA+VALU) cycles: ~75 % (C=32) / ~86 % (C=64) of the 157 TFLOP/s fp32
//   matrix peak with the 5-op epilogue; the in-loop efficiency measured with s_memtime is 72 % / 83 %.
//   What the phase structure buys is order: each wave issues its MFMAs as one uninterrupted dependent chain
//   (hipcc otherwise interleaves VALU into the chain, the slow case above) and the epilogues run as dense VALU.
//   Stages (CH codes in fragment order + |e|^2) arrive by LDS-DMA (global_load_lds) into a 2-deep ring.
// ------------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void epi_reg(float accv, float zz, float e, int code0, float &best, int &bcode) {
#ifdef XQ_NOEPI
    asm volatile("" ::"v"(accv)); (void)zz; (void)e; (void)code0; (void)best; (void)bcode;
    return;
#endif
    float d;
    if (MODE == XQ_MODE_COSINE) d = e - accv;                    // e = 0 (real code) / +inf (padding)
    else d = __builtin_fmaf(-2.0f, accv, zz + e);                // (A3): 2*dot is exact
    const bool lt = d < best;                                    // strict: the earlier tile wins ties
    best = lt ? d : best;
    bcode = lt ? code0 : bcode;
}

// cross-lane reduction over the 32 code columns of each half, then one 64-bit atomicMin per token
__device__ __forceinline__ void publish_best(const float (&best)[16], const int (&bcode)[16], long tok0, long N, int lane,
                                             unsigned long long *__restrict__ keys) {
    const int h = lane >> 5, li = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        unsigned long long key = ((unsigned long long)f2ord(best[r]) << 32) | (unsigned)(bcode[r] + li);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(key, o);
            key = other < key ? other : key;
        }
        if (li == r) {
            const long tn = tok0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (tn < N) atomicMin(keys + tn, key);
        }
    }
}

#ifdef XQ_DEBUG_STAMPS
__device__ unsigned long long xq_dbg_stamps[4096 * 8];
#define XQ_STAMP(i) do { if (threadIdx.x == 0) { xq_dbg_stamps[((blockIdx.y * gridDim.x + blockIdx.x) & 4095) * 8 + (i)] = wall_clock64(); \
    if ((i) == 2) xq_dbg_stamps[((blockIdx.y * gridDim.x + blockIdx.x) & 4095) * 8 + 6] = __builtin_readcyclecounter(); \
    if ((i) == 3) xq_dbg_stamps[((blockIdx.y * gridDim.x + blockIdx.x) & 4095) * 8 + 7] = __builtin_readcyclecounter(); } } while (0)
extern "C" int xq_debug_read_stamps(unsigned long long *host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(xq_dbg_stamps), sizeof(unsigned long long) * 4096 * 8) == hipSuccess ? 0 : -1;
}
#else
#define XQ_STAMP(i)
#endif

template <int C, int MODE>
__global__ __launch_bounds__(ASSIGN_THREADS) void assign_kernel(const float *__restrict__ z, long N, int HW,
                                                                const float *__restrict__ wb,
                                                                const float *__restrict__ ee, int n_chunks,
                                                                int chunks_per_split,
                                                                unsigned long long *__restrict__ keys) {
    using TL = Tiling<C>;
    constexpr int T = TL::TILES, KQ = TL::KQ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;  // wave group: 0 = waves 0-3, 1 = waves 4-7
    const int li = lane & 31;

    XQ_STAMP(0);
    const long tok0 = (long)blockIdx.x * TOK_PER_BLOCK + wave * 32;
    float a[C / 2];
    float zzr[16];
    load_tokens<C, MODE>(z, N, HW, tok0, lane, a, zzr);
    XQ_STAMP(1);

    float best[16];
    int bcode[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { best[r] = __builtin_inff(); bcode[r] = 0; }

    const int c0 = blockIdx.y * chunks_per_split;
    int c1 = c0 + chunks_per_split;
    if (c1 > n_chunks) c1 = n_chunks;
    const int S = c1 - c0;  // stages of this block

    constexpr int NST = TL::STAGE_F4 / ASSIGN_THREADS;
    static_assert(TL::STAGE_F4 % ASSIGN_THREADS == 0, "stage size must be a multiple of the block");
    // LDS-DMA: destination = wave-uniform base + lane*16 — exactly the fragment order of the workspace.
    auto stage_dma = [&](int chunk, int bufi) {
        const float4 *src = reinterpret_cast<const float4 *>(wb) + (size_t)chunk * TL::STAGE_F4;
        float4 *dst = reinterpret_cast<float4 *>(smem + (size_t)bufi * TL::BUF_BYTES);
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int o = i * ASSIGN_THREADS + wave * 64;
            __builtin_amdgcn_global_load_lds((gbl_void *)(src + o + lane), (lds_void *)(dst + o), 16, 0, 0);
        }
        // |e|^2: CH floats; waves beyond CH/64 re-copy an existing 64-float segment (same bytes)
        const int eo = (wave * 64) & (TL::CH - 1);
        __builtin_amdgcn_global_load_lds((gbl_void *)(ee + (size_t)chunk * TL::CH + eo + lane),
                                         (lds_void *)(reinterpret_cast<float *>(dst + TL::STAGE_F4) + eo), 4, 0, 0);
    };

    if (S > 0) stage_dma(c0, 0);
    __syncthreads();  // (vmcnt is drained before the barrier: the DMA has landed)
    XQ_STAMP(2);

    f32x16 acc[T];
    float pe[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        pe[t] = __builtin_inff();
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    }

    // phases 0 .. 2S: group g computes stage s in phase 2s+g and retires it in phase 2s+g+1
    for (int p = 0; p <= 2 * S; ++p) {
        const int q = p - grp;
        if ((p & 1) == 0) {  // even phase: everybody helps to prefetch stage p/2+1 (its buffer was released in phase p-1)
            const int s_next = (p >> 1) + 1;
            if (s_next < S) stage_dma(c0 + s_next, s_next & 1);
        }
        if (q >= 0 && (q & 1) == 0 && (q >> 1) < S) {
            // ---- MFMA phase ----
            const int s_cur = q >> 1;
            const float4 *bl = reinterpret_cast<const float4 *>(smem + (size_t)(s_cur & 1) * TL::BUF_BYTES) + lane;
            const float *el = reinterpret_cast<const float *>(bl - lane + TL::STAGE_F4) + li;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                pe[t] = el[t * 32];
#pragma unroll
                for (int g = 0; g < KQ; ++g) {
                    const float4 bv = bl[(t * KQ + g) * 64];
                    if (g == 0) {
                        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], bv.x, zero, 0, 0, 0);
                    } else {
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * g + 0], bv.x, acc[t], 0, 0, 0);
                    }
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * g + 1], bv.y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * g + 2], bv.z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * g + 3], bv.w, acc[t], 0, 0, 0);
                }
            }
        } else if (q >= 1 && (q & 1) == 1) {
            // ---- epilogue phase: retire stage (q-1)/2 ----
            const int code_base = (c0 + ((q - 1) >> 1)) * TL::CH;
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) epi_reg<MODE>(acc[t][r], zzr[r], pe[t], code_base + t * 32, best[r], bcode[r]);
            }
        }
        __syncthreads();
    }
    XQ_STAMP(3);
    XQ_STAMP(4);
    publish_best(best, bcode, tok0, N, lane, keys);
    XQ_STAMP(5);
}

// keys -> idx (+ optional winning score)
__global__ __launch_bounds__(256) void keys_to_idx_kernel_(const unsigned long long *__restrict__ keys, long N,
                                                          int64_t *__restrict__ idx, float *__restrict__ best) {
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const unsigned long long k = keys[n];
    idx[n] = (int64_t)(k & 0xffffffffull);
    if (best) best[n] = ord2f((uint32_t)(k >> 32));
}

// ------------------------------------------------------------------------------------------------
// K2: per-token epilogue of VectorQuantizer.forward (xqgan_model.py:769-799)
// ------------------------------------------------------------------------------------------------
template <int C, bool NORMED>
__global__ __launch_bounds__(256) void vq_finish_kernel(const float *__restrict__ z, long N, int HW,
                                                        const float *__restrict__ E,
                                                        const unsigned long long *__restrict__ keys, int ste,
                                                        float *__restrict__ zq, int64_t *__restrict__ idx_out,
                                                        float *__restrict__ hist, float *__restrict__ partials) {
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    float lsum = 0.0f;
    if (n < N) {
        const unsigned idx = (unsigned)(keys[n] & 0xffffffffull);
        idx_out[n] = (int64_t)idx;
        const long b = n / HW;
        const int hw = (int)(n - b * HW);
        const size_t off = (size_t)b * C * HW + hw;
        float x[C], zh[C], e[C], eh[C];
#pragma unroll
        for (int k = 0; k < C; ++k) x[k] = z[off + (size_t)k * HW];
        const float4 *row = reinterpret_cast<const float4 *>(E + (size_t)idx * C);
#pragma unroll
        for (int q = 0; q < C / 4; ++q) {
            float4 v = row[q];
            e[4 * q + 0] = v.x; e[4 * q + 1] = v.y; e[4 * q + 2] = v.z; e[4 * q + 3] = v.w;
        }
        if (NORMED) {
            l2norm_row<C>(x, zh);
            l2norm_row<C>(e, eh);
        } else {
#pragma unroll
            for (int k = 0; k < C; ++k) { zh[k] = x[k]; eh[k] = e[k]; }
        }
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const float diff = eh[k] - zh[k];
            lsum = __builtin_fmaf(diff, diff, lsum);
            if (zq) zq[off + (size_t)k * HW] = ste ? (zh[k] + diff) : eh[k];
        }
        if (hist) atomicAdd(hist + idx, 1.0f);
    }
    if (partials) {
        __shared__ float red[4];
        lsum = wave_sum(lsum);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
        __syncthreads();
        if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// fixed-order sum of per-block partials (double) -> out[0]
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float *__restrict__ partials, int n,
                                                              float *__restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)red[0];
}

// ------------------------------------------------------------------------------------------------
// backward of VectorQuantizer.forward (SURVEY §8a "Backward structure")
//
// g_z is token-parallel.  The codebook gradient g_E[v] = sum over the tokens that chose code v of their row ge[n] is a
// scatter-reduce; it is done WITHOUT floating-point atomics so that the result is deterministic (bit-identical from run to run):
//   pass 1 (this kernel, one block = one chunk of 256 consecutive tokens): the rows ge[n][C] are staged in LDS; every token
//           finds the next token of the chunk with the same code (an LDS broadcast scan) and whether it is the first of its
//           code; each first token heads a chain that is summed in ascending token order, C lanes per chain (one per
//           channel), into partial[chunk][slot][C]; first[chunk][code] = slot + 1 (table zeroed by a memset node);
//   pass 2 (vq_codebook_grad_kernel): 4 lanes per code walk the chunks in ascending order (a quarter each) and add the
//           chunk partials; the four sums meet in a fixed order.
// A collapsed codebook (all tokens on a few codes) costs 256 sequential adds per channel lane in pass 1 and <= nchunks rows
// per code in pass 2 — no contention, unlike atomics.
// ------------------------------------------------------------------------------------------------
static constexpr int BWD_CHUNK = 256;

template <int C, bool NORMED>
__global__ __launch_bounds__(256) void vq_backward_kernel(const float *__restrict__ z, long N, int HW,
                                                          const float *__restrict__ E, int V,
                                                          const int64_t *__restrict__ idx_in,
                                                          const float *__restrict__ g_out,
                                                          const float *__restrict__ g_vq,
                                                          const float *__restrict__ g_commit, float beta,
                                                          float *__restrict__ g_z, float *__restrict__ partial,
                                                          unsigned short *__restrict__ first) {
    extern __shared__ __attribute__((aligned(16))) char bwd_smem[];
    int *sidx = reinterpret_cast<int *>(bwd_smem);                       // [256] code of each token of the chunk
    short *snxt = reinterpret_cast<short *>(sidx + BWD_CHUNK);           // [256] next token with the same code, -1 = none
    short *heads = snxt + BWD_CHUNK;                                     // [256] first token of every code present
    int *nheads = reinterpret_cast<int *>(heads + BWD_CHUNK);            // [1] (+3 pad)
    float *sge = reinterpret_cast<float *>(nheads + 4);                  // [256][C + 1]
    const int t = threadIdx.x;
    const long n = (long)blockIdx.x * BWD_CHUNK + t;
    const bool valid = n < N;
    const bool scatter = partial != nullptr;
    const int code = valid ? (int)idx_in[n] : -1 - t;       // padding tokens: unique negative codes, never heads
    if (scatter) {
        sidx[t] = code;
        if (t == 0) nheads[0] = 0;
    }
    float ge[C];
    if (valid) {
        const float inv = 1.0f / ((float)N * (float)C);
        const float cc = (g_commit ? g_commit[0] : 0.0f) * beta * 2.0f * inv;  // commit: d/dzhat of beta*mean((sg(eh)-zh)^2)
        const float cv = (g_vq ? g_vq[0] : 0.0f) * 2.0f * inv;                 // vq:     d/dehat of mean((eh-sg(zh))^2)
        const long b = n / HW;
        const int hw = (int)(n - b * HW);
        const size_t off = (size_t)b * C * HW + hw;
        float x[C], zh[C], e[C], eh[C];
#pragma unroll
        for (int k = 0; k < C; ++k) x[k] = z[off + (size_t)k * HW];
        const float4 *row = reinterpret_cast<const float4 *>(E + (size_t)code * C);
#pragma unroll
        for (int q = 0; q < C / 4; ++q) {
            float4 v = row[q];
            e[4 * q + 0] = v.x; e[4 * q + 1] = v.y; e[4 * q + 2] = v.z; e[4 * q + 3] = v.w;
        }
        float nz = 1.0f, ne = 1.0f;
        if (NORMED) {
            nz = l2norm_row<C>(x, zh);
            ne = l2norm_row<C>(e, eh);
        } else {
#pragma unroll
            for (int k = 0; k < C; ++k) { zh[k] = x[k]; eh[k] = e[k]; }
        }
        float gzh[C], geh[C];
        float dz = 0.0f, de = 0.0f;
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const float go = g_out ? g_out[off + (size_t)k * HW] : 0.0f;
            const float diff = zh[k] - eh[k];
            gzh[k] = __builtin_fmaf(cc, diff, go);
            geh[k] = -cv * diff;
            dz = __builtin_fmaf(gzh[k], zh[k], dz);
            de = __builtin_fmaf(geh[k], eh[k], de);
        }
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const float gz = NORMED ? (gzh[k] - zh[k] * dz) / nz : gzh[k];
            ge[k] = NORMED ? (geh[k] - eh[k] * de) / ne : geh[k];
            g_z[off + (size_t)k * HW] = gz;
        }
    } else {
#pragma unroll
        for (int k = 0; k < C; ++k) ge[k] = 0.0f;
    }
    if (!scatter) return;
#pragma unroll
    for (int k = 0; k < C; ++k) sge[t * (C + 1) + k] = ge[k];      // row pitch C + 1: lane t, word t (C + 1) + k -> bank (t + k) mod 32
    __syncthreads();
    bool has_prev = false;
    int next = -1;
#pragma unroll 8
    for (int u = 0; u < BWD_CHUNK; ++u) {          // every lane reads the same word: an LDS broadcast
        const bool eq = sidx[u] == code;
        has_prev |= eq && u < t;
        if (eq && u > t && next < 0) next = u;
    }
    snxt[t] = (short)next;
    if (valid && !has_prev) {
        const int slot = atomicAdd(nheads, 1);     // slot numbering is arbitrary; every chain's sum is not
        heads[slot] = (short)t;
        first[(size_t)blockIdx.x * V + code] = (unsigned short)(slot + 1);
    }
    __syncthreads();
    const int nh = nheads[0];
    const int k = t % C;
    for (int e = t / C; e < nh; e += BWD_CHUNK / C) {
        int u = heads[e];
        float acc = 0.0f;
        do {
            acc += sge[u * (C + 1) + k];
            u = snxt[u];
        } while (u >= 0);
        partial[((size_t)blockIdx.x * BWD_CHUNK + e) * C + k] = acc;
    }
}

// g_E[v][:] = sum over the chunks (ascending inside each quarter of the chunk range) of the chunk partial of code v
template <int C>
__global__ __launch_bounds__(256) void vq_codebook_grad_kernel(const float *__restrict__ partial, const unsigned short *__restrict__ first,
                                                               int V, int nchunks, float *__restrict__ g_E) {
    const int v = blockIdx.x * 64 + (threadIdx.x >> 2), q = threadIdx.x & 3;
    const int vc = v < V ? v : V - 1;
    const int per = (nchunks + 3) / 4;
    const int c0 = q * per, c1 = min(nchunks, c0 + per);
    float acc[C];
#pragma unroll
    for (int k = 0; k < C; ++k) acc[k] = 0.0f;
    for (int cb = c0; cb < c1; cb += 8) {
        unsigned short s8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s8[j] = cb + j < c1 ? first[(size_t)(cb + j) * V + vc] : (unsigned short)0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (s8[j]) {
                const float4 *row = reinterpret_cast<const float4 *>(partial + ((size_t)(cb + j) * BWD_CHUNK + (s8[j] - 1)) * C);
#pragma unroll
                for (int r = 0; r < C / 4; ++r) {
                    const float4 w = row[r];
                    acc[4 * r] += w.x; acc[4 * r + 1] += w.y; acc[4 * r + 2] += w.z; acc[4 * r + 3] += w.w;
                }
            }
        }
    }
    // quarters meet in a fixed order: ((q0 + q1) + q2) + q3, computed on the q = 0 lane
    const int l0 = threadIdx.x & 60;
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const float a1 = __shfl(acc[k], l0 + 1), a2 = __shfl(acc[k], l0 + 2), a3 = __shfl(acc[k], l0 + 3);
        acc[k] = ((acc[k] + a1) + a2) + a3;
    }
    if (q == 0 && v < V) {
        float4 *dst = reinterpret_cast<float4 *>(g_E + (size_t)v * C);
#pragma unroll
        for (int r = 0; r < C / 4; ++r) dst[r] = make_float4(acc[4 * r], acc[4 * r + 1], acc[4 * r + 2], acc[4 * r + 3]);
    }
}

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
static int g_num_cus = 0;
int num_cus() {
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) g_num_cus = p.multiProcessorCount;
        if (g_num_cus <= 0) g_num_cus = 256;
    }
    return g_num_cus;
}

template <int C, int MODE>
static int launch_assign_t(const float *z, long N, int HW, const float *E, int V, const AssignWs &ws, hipStream_t s, int what) {
    using TL = Tiling<C>;
    const int Vpad = ws.Vpad;
    if (what & XQI_PREP)
        hipLaunchKernelGGL((prep_codebook_kernel<C, MODE>), dim3((Vpad + 255) / 256), dim3(256), 0, s, E, V, Vpad, ws.wb, ws.ee);
    if (!(what & XQI_SEARCH)) return xq_check_launch("prep_codebook_kernel");
    if (hipMemsetAsync(ws.keys, 0xFF, (size_t)N * 8, s) != hipSuccess) return xq_set_error(XQ_ELAUNCH, "%s", "hipMemsetAsync(keys) failed");
    {
        const int n_chunks = Vpad / TL::CH;
        const int tok_blocks = (int)((N + TOK_PER_BLOCK - 1) / TOK_PER_BLOCK);
        // split the code axis so that the grid covers the chip once (1 block of 8 waves per CU)
        int want = (XQ_LDS_BPC * num_cus() + tok_blocks - 1) / tok_blocks;
        if (want < 1) want = 1;
        if (want > n_chunks) want = n_chunks;
        const int cps = (n_chunks + want - 1) / want;
        const int splits = (n_chunks + cps - 1) / cps;
        const size_t lds = 2 * (size_t)TL::BUF_BYTES;
        static unsigned long long attr_devs = 0;      // per device and per instantiation
        if (first_call_on_this_device(&attr_devs))
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&assign_kernel<C, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const int pslot = prof_begin(XQ_PROF_ASSIGN, 2.0 * (double)N * Vpad * C, s);
        hipLaunchKernelGGL((assign_kernel<C, MODE>), dim3(tok_blocks, splits), dim3(ASSIGN_THREADS), lds, s, z, N, HW, ws.wb, ws.ee,
                           n_chunks, cps, ws.keys);
        prof_end(pslot, s);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_err, sizeof(g_err), "assign_kernel<C=%d>: %s", C, hipGetErrorString(e)); return XQ_ELAUNCH; }
    return XQ_OK;
}

template <int C>
static int launch_assign_c(int mode, const float *z, long N, int HW, const float *E, int V, const AssignWs &ws, hipStream_t s, int what) {
    switch (mode) {
        case XQ_MODE_L2_NORMED: return launch_assign_t<C, XQ_MODE_L2_NORMED>(z, N, HW, E, V, ws, s, what);
        case XQ_MODE_L2_RAW: return launch_assign_t<C, XQ_MODE_L2_RAW>(z, N, HW, E, V, ws, s, what);
        case XQ_MODE_COSINE: return launch_assign_t<C, XQ_MODE_COSINE>(z, N, HW, E, V, ws, s, what);
    }
    return xq_set_error(XQ_EINVAL, "%s: unknown mode %ld", "xq_assign", mode);
}

int launch_assign(int mode, int C, const float *z, long N, int HW, const float *E, int V, const AssignWs &ws, hipStream_t s, int what) {
    switch (C) {
        case 8: return launch_assign_c<8>(mode, z, N, HW, E, V, ws, s, what);
        case 16: return launch_assign_c<16>(mode, z, N, HW, E, V, ws, s, what);
        case 32: return launch_assign_c<32>(mode, z, N, HW, E, V, ws, s, what);
        case 64: return launch_assign_c<64>(mode, z, N, HW, E, V, ws, s, what);
    }
    return xq_set_error(XQ_EINVAL, "%s: unsupported channel count C=%ld (supported: 8,16,32,64)", "xq_assign", C);
}

int check_common(const char *fn, const void *z, int B, int C, int HW, const void *E, int V) {
    if (B == 0) return XQ_OK; /* empty batch: nothing to validate against, callers return early */
    if (!z || !E) return xq_set_error(XQ_EINVAL, "%s: null input pointer", fn);
    if (B < 0 || HW < 1 || V < 1) return xq_set_error(XQ_EINVAL, "%s: bad shape (B=%ld, HW=%ld)", fn, B, HW);
    if (C != 8 && C != 16 && C != 32 && C != 64)
        return xq_set_error(XQ_EINVAL, "%s: unsupported channel count C=%ld (supported: 8,16,32,64)", fn, C);
    if ((long)V > 0x7fffff00L) return xq_set_error(XQ_EINVAL, "%s: V=%ld too large", fn, V);
    return XQ_OK;
}

extern "C" int xq_assign(const float *z, int B, int C, int HW, const float *E, int V, int mode, int64_t *idx, float *best,
                         void *workspace, size_t workspace_bytes, xq_stream_t stream) {
    int rc = check_common("xq_assign", z, B, C, HW, E, V);
    if (rc) return rc;
    if (!idx) return xq_set_error(XQ_EINVAL, "%s: idx is null", "xq_assign");
    const long N = (long)B * HW;
    if (N == 0) return XQ_OK;
    AssignWs ws;
    const size_t need = assign_ws_layout(N, C, V, (char *)workspace, &ws);
    if (!workspace || workspace_bytes < need)
        return xq_set_error(XQ_ENOSPACE, "%s: workspace %ld < %ld bytes", "xq_assign", (long)workspace_bytes, (long)need);
    hipStream_t s = (hipStream_t)stream;
    rc = launch_assign(mode, C, z, N, HW, E, V, ws, s);
    if (rc) return rc;
    hipLaunchKernelGGL(keys_to_idx_kernel_, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, ws.keys, N, idx, best);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_err, sizeof(g_err), "keys_to_idx_kernel: %s", hipGetErrorString(e)); return XQ_ELAUNCH; }
    return XQ_OK;
}

template <int C>
static void launch_finish(bool normed, const float *z, long N, int HW, const float *E, const unsigned long long *keys, int ste,
                          float *zq, int64_t *idx, float *hist, float *partials, int blocks, hipStream_t s) {
    if (normed)
        hipLaunchKernelGGL((vq_finish_kernel<C, true>), dim3(blocks), dim3(256), 0, s, z, N, HW, E, keys, ste, zq, idx, hist, partials);
    else
        hipLaunchKernelGGL((vq_finish_kernel<C, false>), dim3(blocks), dim3(256), 0, s, z, N, HW, E, keys, ste, zq, idx, hist, partials);
}

extern "C" int xq_vq_forward(const float *z, int B, int C, int HW, const float *E, int V, int codebook_norm, int ste,
                             float *zq, int64_t *idx, float *hist, float *loss_sq, void *workspace, size_t workspace_bytes,
                             xq_stream_t stream) {
    int rc = check_common("xq_vq_forward", z, B, C, HW, E, V);
    if (rc) return rc;
    if (!idx) return xq_set_error(XQ_EINVAL, "%s: idx is null", "xq_vq_forward");
    const long N = (long)B * HW;
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        if (loss_sq) (void)hipMemsetAsync(loss_sq, 0, 4, s);
        return XQ_OK;
    }
    AssignWs ws;
    const size_t need = assign_ws_layout(N, C, V, (char *)workspace, &ws);
    if (!workspace || workspace_bytes < need)
        return xq_set_error(XQ_ENOSPACE, "%s: workspace %ld < %ld bytes", "xq_vq_forward", (long)workspace_bytes, (long)need);
    const int blocks = (int)((N + 255) / 256);
    if (loss_sq && blocks > MAX_PARTIALS)
        return xq_set_error(XQ_EINVAL, "%s: N=%ld exceeds the loss-partials capacity (%ld tokens)", "xq_vq_forward", N, (long)MAX_PARTIALS * 256);
    rc = launch_assign(codebook_norm ? XQ_MODE_L2_NORMED : XQ_MODE_L2_RAW, C, z, N, HW, E, V, ws, s);
    if (rc) return rc;
    float *partials = loss_sq ? ws.partials : nullptr;
    // vq_finish: z read (4C) + z_q written (4C) + the gathered codebook rows (4C, L2-resident) + idx (8) per token
    const int pslot_f = xq::prof_begin(XQ_PROF_VQ_ELEM, (double)N * (8.0 * C + 8.0) + (double)V * C * 4.0, s);
    switch (C) {
        case 8: launch_finish<8>(codebook_norm != 0, z, N, HW, E, ws.keys, ste, zq, idx, hist, partials, blocks, s); break;
        case 16: launch_finish<16>(codebook_norm != 0, z, N, HW, E, ws.keys, ste, zq, idx, hist, partials, blocks, s); break;
        case 32: launch_finish<32>(codebook_norm != 0, z, N, HW, E, ws.keys, ste, zq, idx, hist, partials, blocks, s); break;
        case 64: launch_finish<64>(codebook_norm != 0, z, N, HW, E, ws.keys, ste, zq, idx, hist, partials, blocks, s); break;
    }
    if (loss_sq) hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, s, ws.partials, blocks, loss_sq);
    xq::prof_end(pslot_f, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_err, sizeof(g_err), "vq_finish_kernel: %s", hipGetErrorString(e)); return XQ_ELAUNCH; }
    return XQ_OK;
}

struct BwdWs {
    float *partial;           // [nchunks][256][C] chunk partial sums, one row per code present in the chunk
    unsigned short *first;    // [nchunks][V]      slot + 1 of the code's row in the chunk, 0 = absent
    size_t first_bytes;
};
static size_t bwd_ws_layout(long N, int C, int V, char *base, BwdWs *ws) {
    const long nchunks = (N + BWD_CHUNK - 1) / BWD_CHUNK;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = xq::align_up(off + bytes, 256); return o; };
    const size_t o_part = take((size_t)nchunks * BWD_CHUNK * C * 4), o_first = take((size_t)nchunks * V * 2);
    if (ws) {
        ws->partial = (float *)(base + o_part);
        ws->first = (unsigned short *)(base + o_first);
        ws->first_bytes = (size_t)nchunks * V * 2;
    }
    return off;
}
extern "C" size_t xq_vq_backward_workspace_bytes(int64_t N, int C, int V) {
    if (N < 0 || V < 1 || C < 1) return 0;
    return bwd_ws_layout((long)N, C, V, nullptr, nullptr);
}

template <int C>
static void launch_bwd(bool normed, const float *z, long N, int HW, const float *E, int V, const int64_t *idx, const float *g_out,
                       const float *g_vq, const float *g_commit, float beta, float *g_z, float *g_E, const BwdWs *ws, hipStream_t s) {
    const int blocks = (int)((N + BWD_CHUNK - 1) / BWD_CHUNK);
    float *partial = ws ? ws->partial : nullptr;
    unsigned short *first = ws ? ws->first : nullptr;
    const size_t lds = ws ? (size_t)BWD_CHUNK * 4 + BWD_CHUNK * 2 * 2 + 16 + (size_t)BWD_CHUNK * (C + 1) * 4 : 0;
    if (ws) {
        (void)hipMemsetAsync(ws->first, 0, ws->first_bytes, s);
        static unsigned long long attr_devs = 0;       // C = 64 needs 68 KiB of dynamic LDS; per device
        if (first_call_on_this_device(&attr_devs)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&vq_backward_kernel<C, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&vq_backward_kernel<C, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
    }
    if (normed)
        hipLaunchKernelGGL((vq_backward_kernel<C, true>), dim3(blocks), dim3(256), lds, s, z, N, HW, E, V, idx, g_out, g_vq, g_commit, beta, g_z,
                           partial, first);
    else
        hipLaunchKernelGGL((vq_backward_kernel<C, false>), dim3(blocks), dim3(256), lds, s, z, N, HW, E, V, idx, g_out, g_vq, g_commit, beta, g_z,
                           partial, first);
    if (ws)
        hipLaunchKernelGGL((vq_codebook_grad_kernel<C>), dim3((V + 63) / 64), dim3(256), 0, s, partial, first, V, blocks, g_E);
    else
        (void)hipMemsetAsync(g_E, 0, (size_t)V * C * 4, s);
}

extern "C" int xq_vq_backward(const float *z, int B, int C, int HW, const float *E, int V, int codebook_norm,
                              const int64_t *idx, const float *g_out, const float *g_vq, const float *g_commit, float beta,
                              float *g_z, float *g_E, void *workspace, size_t workspace_bytes, xq_stream_t stream) {
    int rc = check_common("xq_vq_backward", z, B, C, HW, E, V);
    if (rc) return rc;
    if (!idx || !g_z || !g_E) return xq_set_error(XQ_EINVAL, "%s: null idx/g_z/g_E", "xq_vq_backward");
    const long N = (long)B * HW;
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        (void)hipMemsetAsync(g_E, 0, (size_t)V * C * 4, s);
        return XQ_OK;
    }
    BwdWs ws;
    const bool scatter = g_vq != nullptr;       // without a vq-loss gradient the codebook receives none
    if (scatter) {
        const size_t need = bwd_ws_layout(N, C, V, nullptr, nullptr);
        if (!workspace || workspace_bytes < need)
            return xq_set_error(XQ_EINVAL, "%s: workspace too small (need %ld bytes, got %ld)", "xq_vq_backward", (long)need, (long)workspace_bytes);
        bwd_ws_layout(N, C, V, (char *)workspace, &ws);
    }
    const BwdWs *w = scatter ? &ws : nullptr;
    // vq_backward + codebook gradient: z, g_out read, g_z written (12C B/token) + idx + the V x C gradient written
    const int pslot_b = xq::prof_begin(XQ_PROF_VQ_ELEM, (double)N * (12.0 * C + 8.0) + (double)V * C * 4.0, s);
    switch (C) {
        case 8: launch_bwd<8>(codebook_norm != 0, z, N, HW, E, V, idx, g_out, g_vq, g_commit, beta, g_z, g_E, w, s); break;
        case 16: launch_bwd<16>(codebook_norm != 0, z, N, HW, E, V, idx, g_out, g_vq, g_commit, beta, g_z, g_E, w, s); break;
        case 32: launch_bwd<32>(codebook_norm != 0, z, N, HW, E, V, idx, g_out, g_vq, g_commit, beta, g_z, g_E, w, s); break;
        case 64: launch_bwd<64>(codebook_norm != 0, z, N, HW, E, V, idx, g_out, g_vq, g_commit, beta, g_z, g_E, w, s); break;
    }
    xq::prof_end(pslot_b, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_err, sizeof(g_err), "vq_backward_kernel: %s", hipGetErrorString(e)); return XQ_ELAUNCH; }
    return XQ_OK;
}
