// xq_gemm_map.hpp — index maps of the bf16 GEMM kernels (csrc/xq_gemm.hip): which 16 bytes every LDS-DMA lane fetches, where
// they land in LDS, and which LDS bytes every MFMA fragment read touches.  Host + device: tests/gemm_map_emulator.cpp replays
// one K tile through exactly these functions on the CPU (LDS-DMA = lane-linear image, ds_read_b128, ds_read_b64_tr_b16 and the
// v_mfma_f32_32x32x16_bf16 lane layouts as measured on gfx950, profiles/r01_ds_read_tr_probe.txt) and compares with a plain
// matrix product, so that an indexing mistake is caught without a GPU.
//
// Block tile 256 x BN x 64 (BN = 256 or 128), 8 waves as 2 (rows) x 4 (columns): wave (wr, wc) owns rows [128 wr, +128) and
// columns [WTN wc, +WTN), WTN = BN / 4 (64 or 32).  A K tile of one operand is cut into 16 KiB PIECES by the order in which
// the waves consume them (a wave works through its 128 x WTN tile in 64 x 32 quadrants):
//     A-top   : rows {128 wr + 0..63}          A-bottom : rows {128 wr + 64..127}        (both wave rows in one piece)
//     B-left  : columns {WTN wc + 0..31}       B-right  : columns {WTN wc + 32..63}      (all four wave columns; BN = 256 only)
// An operand is K-MAJOR when its reduction index is the fast (contiguous) axis in memory (x[M][K], W[N][K]) and K-STRIDED
// when the reduction index is the slow axis (W[K][N] in the data gradient, both operands of the weight gradient):
//   K-major   piece = [128 rows][64 k]  bf16, 128-byte LDS rows; fragment = one ds_read_b128 (8 consecutive k of one row);
//             16-byte chunk c of LDS row R holds source chunk c ^ ((R >> 1) & 7)            (conflict-free b128 lane groups)
//   K-strided piece = [64 k][128 cols] bf16, 256-byte LDS rows; fragment = two ds_read_b64_tr_b16 (4 consecutive k each);
//             16-byte chunk c of LDS row kr holds source chunk c ^ (4 * (kr & 3))           (conflict-free transpose reads)
// The XOR is applied to the SOURCE address of the LDS-DMA (its destination is lane-linear by construction) and again on the read.
#pragma once

#if defined(__HIPCC__)
#define GM_HD __host__ __device__ __forceinline__
#else
#define GM_HD inline
#endif

namespace gm {

enum { KMAJOR = 0, KSTRIDED = 1, KMAJOR_CONV = 2 };   // KMAJOR_CONV: K-major A operand gathered from an NHWC image (implicit GEMM)
constexpr int PIECE_BYTES = 16384;
constexpr int BM = 256, BKT = 64;   // block rows, reduction depth of one K tile

// tile-local row of the A operand / column of the B operand held by piece-row R (0..127); half = 0 (top/left), 1 (bottom/right)
GM_HD int a_rc(int half, int R) { return (R >> 6) * 128 + half * 64 + (R & 63); }
GM_HD int b_rc(int half, int R, int wtn) { return (R >> 5) * wtn + half * 32 + (R & 31); }

// One LDS-DMA wave instruction moves 64 lanes x 16 B = 1 KiB to piece offset (2 * wave + i) * 1024 + lane * 16 (i = 0, 1).
struct StageSrc {
    int rc;   // K-major: the row (A) / column (B) of the 8 elements;  K-strided: the first of 8 consecutive rows / columns
    int k;    // K-major: the first of 8 consecutive k;                K-strided: the k of the 8 elements
};
template <int KIND, bool IS_A>
GM_HD StageSrc stage_src(int half, int wave, int i, int lane, int wtn) {
    StageSrc s;
    if (KIND == KMAJOR) {
        const int R = 16 * wave + 8 * i + (lane >> 3);
        const int c = (lane & 7) ^ ((R >> 1) & 7);
        s.k = 8 * c;
        s.rc = IS_A ? a_rc(half, R) : b_rc(half, R, wtn);
    } else {
        const int kr = 4 * (2 * wave + i) + (lane >> 4);
        const int c = (lane & 15) ^ (4 * (kr & 3));
        s.k = kr;
        s.rc = IS_A ? ((c >> 3) * 128 + half * 64 + 8 * (c & 7)) : ((c >> 2) * wtn + half * 32 + 8 * (c & 3));
    }
    return s;
}
GM_HD int stage_dst(int wave, int i, int lane) { return (2 * wave + i) * 1024 + lane * 16; }

// Fragment reads.  w = wave row (A) or wave column (B), f = 32-wide fragment inside the 64-row half (A: 0, 1; B: 0), s = 16-deep
// k slice (0..3).  The lane ends up with the 8 reduction indices k = 16 s + 8 (lane >> 5) + 0..7 of row / column (lane & 31).
template <bool IS_A>
GM_HD int frag_off_kmajor(int w, int f, int s, int lane) {
    const int R = (IS_A ? 64 * w + 32 * f : 32 * w) + (lane & 31);
    const int chunk = 2 * s + (lane >> 5);
    return R * 128 + ((chunk ^ ((R >> 1) & 7)) << 4);
}
// u = 0, 1: the two 8-byte transpose reads (k = 16 s + 8 (lane >> 5) + 4 u + 0..3)
template <bool IS_A>
GM_HD int frag_off_kstrided(int w, int f, int s, int u, int lane) {
    const int kr = 16 * s + 8 * (lane >> 5) + 4 * u + ((lane & 15) >> 2);
    const int col = (IS_A ? 64 * w + 32 * f : 32 * w) + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    return kr * 256 + (((col >> 3) ^ (4 * (kr & 3))) << 4) + ((col & 7) << 1);
}

// Accumulators: acc = mfma(B fragment, A fragment, acc) ("swapped": the MFMA's row index is the output COLUMN), so lane l,
// register r of a 32 x 32 fragment hold output row (l & 31), output column (r & 3) + 8 (r >> 2) + 4 (l >> 5): four consecutive
// columns per register quad -> 8-byte bf16 / 16-byte fp32 stores.
GM_HD int acc_col(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// bf16 epilogue: every wave stages its 128 x WTN tile in a private LDS region, [128 rows][WTN] bf16 (2 WTN-byte rows; chunks
// XOR-ed with (row >> 1) & 7 resp. & 3), then streams it out in full rows.  Offsets are bytes inside the region.
GM_HD int epi_write_off(int fi, int fj, int g, int lane, int wtn) {   // 8 bytes: columns 32 fj + 8 g + 4 (lane >> 5) + 0..3
    const int row = 32 * fi + (lane & 31);
    const int chunk = 4 * fj + g;
    const int cmask = wtn / 8 - 1;
    return row * (2 * wtn) + ((chunk ^ ((row >> 1) & cmask)) << 4) + 8 * (lane >> 5);
}
// pass `it`: lane reads 16 bytes = columns 8 c .. 8 c + 7 of row `row`
GM_HD void epi_read_map(int it, int lane, int wtn, int *row, int *c, int *off) {
    const int cpr = wtn / 8;            // 16-byte chunks per row (8 or 4)
    *row = it * (64 / cpr) + lane / cpr;
    *c = lane % cpr;
    *off = *row * (2 * wtn) + (((*c) ^ ((*row >> 1) & (cpr - 1))) << 4);
}

// XCD-aware, bijective tile order: workgroup id -> position in the tile sequence (XCD k = id % 8 walks a contiguous range)
GM_HD long xcd_order(long id, long total) {
    const long q = total / 8, r = total % 8, x = id % 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + id / 8;
}

// ---- staging addresses ------------------------------------------------------------------------------------------------------------------
// Per-thread staging state of one operand: byte offsets (relative to the operand's tile base) of this lane's 16-byte chunk in the two
// LDS-DMA instructions of each half piece, the tile base and its advance per K tile.  csrc/xq_gemm.hip derives its Stager (which adds the
// LDS-DMA issue) from it; tests/gemm_map_emulator.cpp replays whole item streams through it.
#if defined(__HIP_DEVICE_COMPILE__)
#define GM_RFL(x) __builtin_amdgcn_readfirstlane(x)
#else
#define GM_RFL(x) (x)
#endif
template <int KIND, bool IS_A>
struct StagerAddr {
    unsigned off[2][2];     // [half][i]
    const char *base;       // tile base at K tile 0 (wave-uniform)
    long adv;               // bytes per K tile
    bool interior;          // the tile `off` was computed for has all 256 rows / columns inside the matrix: no lane was clamped
    // persistent schedule: next item.  Between two interior tiles the per-lane offsets do not change — only the tile base moves
    GM_HD void retarget(const char *mat, long ld, long rc0, long rc_count, long k0, int wave, int lane, int wtn) {
        if (interior && rc_count - rc0 >= 256) {
            if (KIND == KMAJOR) base = mat + (rc0 * ld + k0) * 2;
            else base = mat + (k0 * ld + rc0) * 2;
        } else {
            init(mat, ld, rc0, rc_count, k0, wave, lane, wtn, 2);
        }
    }
    // rows/cols beyond `limit` (elements of the non-reduction axis inside this tile) are clamped (their outputs are never stored)
    GM_HD void init(const char *mat, long ld, long rc0, long rc_count, long k0, int wave, int lane, int wtn, int halves) {
        const long avail = rc_count - rc0;          // valid rows / columns from the tile origin
        interior = avail >= 256 && halves == 2;
        if (KIND == KMAJOR) {
            base = mat + (rc0 * ld + k0) * 2;
            adv = BKT * 2;
        } else {
            base = mat + (k0 * ld + rc0) * 2;
            adv = (long)BKT * ld * 2;
        }
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int h = 0; h < 2; ++h)
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int i = 0; i < 2; ++i) {
                if (h >= halves) { off[h][i] = 0; continue; }
                const StageSrc s = stage_src<KIND, IS_A>(h, wave, i, lane, wtn);
                if (KIND == KMAJOR) {
                    long rc = s.rc;
                    if (rc > avail - 1) rc = avail - 1;
                    off[h][i] = (unsigned)((rc * ld + s.k) * 2);
                } else {
                    long rc = s.rc;
                    if (rc > avail - 8) rc = avail - 8;     // 8 consecutive columns (dimension is a multiple of 8)
                    off[h][i] = (unsigned)(((long)s.k * ld + rc) * 2);
                }
            }
    }
    // persistent schedule: the tile pointer of the K tile being staged, wave-uniform by construction (kept in scalar registers), moved by a
    // scalar add per K tile
    const char *cur;
    GM_HD void make_scalar() {
        const unsigned long long b = (unsigned long long)base, a = (unsigned long long)adv;
        const unsigned blo = GM_RFL((unsigned)b), bhi = GM_RFL((unsigned)(b >> 32));
        const unsigned alo = GM_RFL((unsigned)a), ahi = GM_RFL((unsigned)(a >> 32));
        cur = (const char *)(((unsigned long long)bhi << 32) | blo);
        adv = (long)(((unsigned long long)ahi << 32) | alo);
    }
    GM_HD void step() { cur += adv; }
};

// ---- 128-row tiles (gemm_duo_kernel: 128 x 256 block tile, two workgroups per CU) ----------------------------------------------------
// Block tile 128 x 256 x 64, 8 waves as 2 x 4, wave (wr, wc) owns rows [64 wr, +64) and columns [64 wc, +64).  Pieces per K tile: A (piece-row R
// = tile row R: frag_off_kmajor<true>(wr, f, s, lane) with R = 64 wr + 32 f + lane % 32 reads it unchanged), B-left, B-right (b_rc at WTN = 64,
// read by frag_off_kmajor<false> / frag_off_kstrided<false> as above).  The LDS images are the ones above; what differs is WHICH wave instruction
// fills which KiB: instruction (wave, i) writes piece bytes [(8 i + wave) * 1024, +1024), so that the source addresses of a lane's two (A) /
// four (B: 2 halves x 2) instructions differ by wave-uniform constants — one 32-bit per-lane offset register per operand, everything else is
// scalar arithmetic on the tile pointer (the 128-VGPR budget of two workgroups per CU has no room for six offset registers):
//     K-major    piece-row R = 64 i + 8 wave + lane / 8, 16-byte chunk lane % 8 <- source chunk (lane % 8) ^ ((R >> 1) & 7)
//                i -> +64 rows (A) / +128 columns (B), half -> +32 columns        ((R >> 1) & 7 does not depend on i)
//     K-strided  k row kr = 32 i + 4 wave + lane / 16, chunk lane % 16 <- source chunk (lane % 16) ^ (4 (kr & 3))
//                i -> +32 k rows, half -> +32 columns                              (kr & 3 does not depend on i)
// Ragged edges are handled by moving the last tile row / column BACK inside the matrix (m0 = M - 128, n0 = N - 256): the overlap is
// computed twice with identical values (same operands, same order), no lane is ever clamped.
constexpr int BM_DUO = 128;
GM_HD int duo_stage_dst(int wave, int i, int lane) { return (8 * i + wave) * 1024 + lane * 16; }
template <int KIND, bool IS_A>
GM_HD StageSrc duo_stage_src(int half, int wave, int i, int lane) {
    StageSrc s;
    if (KIND == KMAJOR) {
        const int R = 64 * i + 8 * wave + (lane >> 3);
        s.k = 8 * ((lane & 7) ^ ((R >> 1) & 7));
        s.rc = IS_A ? R : b_rc(half, R, 64);
    } else {
        const int kr = 32 * i + 4 * wave + (lane >> 4);
        const int c = (lane & 15) ^ (4 * (kr & 3));
        s.k = kr;
        s.rc = (c >> 2) * 64 + half * 32 + 8 * (c & 3);
    }
    return s;
}
// byte offset of the lane's chunk for (half 0, i 0) relative to the tile pointer, and the wave-uniform deltas
template <int KIND, bool IS_A>
struct DuoStagerAddr {
    unsigned voff;          // per lane
    unsigned d_i, d_half;   // bytes: instruction i = 1, half = 1 (B only)   (32-bit: 128 rows of a matrix row < 2^31 bytes, checked by the launcher)
    const char *cur;        // tile pointer of the K tile being staged (wave-uniform)
    unsigned adv;           // bytes per K tile
    GM_HD void init(const char *mat, long ld, long rc0, long k0, int wave, int lane) {
        const StageSrc s = duo_stage_src<KIND, IS_A>(0, wave, 0, lane);
        if (KIND == KMAJOR) {
            cur = mat + (rc0 * ld + k0) * 2;
            adv = BKT * 2;
            voff = (unsigned)(((long)s.rc * ld + s.k) * 2);
            d_i = (unsigned)((IS_A ? 64 : 128) * ld * 2);
            d_half = (unsigned)(32 * ld * 2);
        } else {
            cur = mat + (k0 * ld + rc0) * 2;
            adv = (unsigned)((long)BKT * ld * 2);
            voff = (unsigned)(((long)s.k * ld + s.rc) * 2);
            d_i = (unsigned)(32 * ld * 2);
            d_half = 64;
        }
    }
    GM_HD const char *src(int half, int i) const { return cur + (half ? d_half : 0) + (i ? d_i : 0) + voff; }
};

// ---- work items of the persistent duo schedule (gemm_pduo_kernel; 128 x 256 tiles, two workgroups per CU, grid = 2 x CUs) ------------
// Same list as decode_item's (whole tiles, then the tail tiles cut along K, tile-major), on 128-row tiles, with the moved-back edge tiles.
// G: main_items, tail_tiles, tail_splits, tiles_n, kt_full, M, N.  32-bit arithmetic (the launcher checks tiles * splits < 2^31).
struct DuoItem {
    int m0, n0, k0;      // output row / column of the tile (moved back inside the matrix at the ragged edges), first reduction index (all < 2^31)
    int m_lo;            // first row that belongs to THIS tile row (rows [m0, m_lo) are the overlap with the tile row above)
    int KT;              // K tiles of the item
    int slab;            // 0: bf16 epilogue into C; 1: fp32 slab [slab_idx][128][256]
    int slab_idx;
    int trow;            // tile row (index of the fc1-bias column partials: 2 trow + wave row)
};
template <class G>
GM_HD void decode_item_duo(const G &g, unsigned p, DuoItem &it) {
    unsigned tile, split, nsplit;
    if ((long)p < g.main_items) { tile = p; split = 0; nsplit = 1; it.slab = 0; it.slab_idx = 0; }
    else {
        const unsigned q = p - (unsigned)g.main_items;
        nsplit = (unsigned)g.tail_splits;
        const unsigned tl = q / nsplit;
        split = q - tl * nsplit;
        tile = (unsigned)g.main_items + tl;
        it.slab = 1;
        it.slab_idx = (int)q;
    }
    const unsigned trow = tile / (unsigned)g.tiles_n, tcol = tile - trow * (unsigned)g.tiles_n;
    it.trow = (int)trow;
    it.m_lo = (int)trow * BM_DUO;
    it.m0 = it.m_lo > (int)g.M - BM_DUO ? (int)g.M - BM_DUO : it.m_lo;
    it.n0 = (int)tcol * 256 > (int)g.N - 256 ? (int)g.N - 256 : (int)tcol * 256;
    const unsigned base = (unsigned)g.kt_full / nsplit, rem = (unsigned)g.kt_full % nsplit;
    it.k0 = (int)((split * base + (split < rem ? split : rem)) * BKT);
    it.KT = (int)(base + (split < rem ? 1 : 0));
}

// ---- work items of the persistent schedule (gemm_pring_kernel) ----------------------------------------------------------------------
// Position p of the item list: p < main_items is the whole output tile p (bf16 epilogue); the tail_tiles tiles behind them are cut into
// tail_splits K ranges each, one item per range (fp32 slab [tile][split]), executed tile-major or split-major.  G = any struct with the
// fields main_items, tail_tiles, tail_splits, split_major, tiles_n, kt_full (csrc/xq_gemm.hip GemmArgs; the CPU test's own struct).
struct Item {
    long m0, n0, k0;     // output row / column of the tile, first reduction index of the item's K range
    int KT;              // K tiles of the item
    int slab;            // 0: bf16 epilogue into C; 1: fp32 slab
    long slab_idx;
};

template <class G>
GM_HD void decode_item(const G &g, long p, Item &it) {
    long tile, split, nsplit;
    if (p < g.main_items) { tile = p; split = 0; nsplit = 1; it.slab = 0; it.slab_idx = 0; }
    else {
        const long q = p - g.main_items;
        nsplit = g.tail_splits;
        long tl;
        if (g.split_major) { split = q / g.tail_tiles; tl = q - split * g.tail_tiles; }
        else { tl = q / nsplit; split = q - tl * nsplit; }
        tile = g.main_items + tl;
        it.slab = 1;
        it.slab_idx = tl * nsplit + split;       // slab layout [tile][split] whatever the execution order (slab_reduce_kernel)
    }
    it.m0 = (tile / g.tiles_n) * BM;
    it.n0 = (tile % g.tiles_n) * 256;
    const long base = g.kt_full / nsplit, rem = g.kt_full % nsplit;
    it.k0 = (split * base + (split < rem ? split : rem)) * BKT;
    it.KT = (int)(base + (split < rem ? 1 : 0));
}

// persistent schedule: the whole-tile items of one workgroup are `grid` tiles apart; (row, col) of the next one by scalar adds
// (G additionally has step_r = grid / tiles_n, step_c = grid % tiles_n).  decode_item's 64-bit divisions are ~700 scalar instructions =
// the 1.6 - 2.3 k-cycle stall of the load phase once per item (profiles/r03_gemm_where_the_cycles_go.md).
// Precondition: the item before (p - grid) was a whole-tile item whose tile coordinates are (row, col).
template <class G>
GM_HD void next_item_walk(const G &g, long p, int &row, int &col, Item &it) {
    if (p < g.main_items) {
        col += g.step_c;
        row += g.step_r;
        if (col >= g.tiles_n) { col -= g.tiles_n; ++row; }
        it.m0 = (long)row * BM;
        it.n0 = (long)col * 256;
        it.k0 = 0;
        it.KT = g.kt_full;
        it.slab = 0;
        it.slab_idx = 0;
    } else {
        decode_item(g, p, it);
    }
}

}  // namespace gm
