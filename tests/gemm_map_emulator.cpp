// gemm_map_emulator.cpp — CPU replay of one K tile of csrc/xq_gemm.hip through the index maps of csrc/xq_gemm_map.hpp.
// Test infrastructure (built and run by tests/test_gemm_map_cpu.py with g++; no GPU).  Emulated hardware semantics:
//   * LDS-DMA (global_load_lds_dwordx4): lane l of a wave instruction copies 16 source bytes to LDS base + 16 l;
//   * ds_read_b128: 16 bytes at the lane's address; ds_read_b64_tr_b16: within a 16-lane group, result lane L receives
//     element (L % 4) of the 4-element rows addressed by lanes (L / 4) + 4 j, j = 0..3 (profiles/r01_ds_read_tr_probe.txt);
//   * v_mfma_f32_32x32x16_bf16 D = A.B + C: operand lane l holds row/column (l & 31), k = 8 (l >> 5) + 0..7; D register r of
//     lane l is D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31]  (cdna_hip_programming.md §3).
// It also counts LDS bank conflicts of every fragment read with the gfx950 lane grouping (MI355X_MICROARCH.md §LDS), and replays the
// item lists of the persistent schedule (check_item_lists below).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#include "../imagefolder_amd/csrc/xq_gemm_map.hpp"

using namespace gm;

static int g_fail = 0;
static long g_conf_b128 = 0, g_conf_tr = 0;

struct Piece {
    int16_t e[PIECE_BYTES / 2];
};

// src(rc, k) value tables: small integers (exact in bf16 and in fp32 sums)
struct Operand {
    int rows;                       // 256 (A) or BN (B)
    std::vector<int16_t> v;         // [rows][64]
    int16_t at(int rc, int k) const { return v[rc * 64 + k]; }
};

template <int KIND, bool IS_A>
static void stage_piece(const Operand &op, int half, int wtn, Piece &p) {
    std::vector<int> written(PIECE_BYTES / 16, 0);
    for (int wave = 0; wave < 8; ++wave)
        for (int i = 0; i < 2; ++i)
            for (int lane = 0; lane < 64; ++lane) {
                const StageSrc s = stage_src<KIND, IS_A>(half, wave, i, lane, wtn);
                const int dst = stage_dst(wave, i, lane);
                written[dst / 16]++;
                for (int j = 0; j < 8; ++j)
                    p.e[dst / 2 + j] = (KIND == KMAJOR) ? op.at(s.rc, s.k + j) : op.at(s.rc + j, s.k);
            }
    for (int c : written)
        if (c != 1) { g_fail++; std::printf("stage: a 16-byte LDS chunk written %d times\n", c); return; }
}

// bank conflicts: extra cycles = (max distinct addresses on one bank) - 1 per service group
static long conflicts(const std::vector<int> &addr, const std::vector<std::vector<int>> &groups, int bytes, int nbanks) {
    long extra = 0;
    for (const auto &grp : groups) {
        std::vector<std::set<int>> per(nbanks);
        for (int l : grp)
            for (int b = 0; b < bytes; b += 4) per[((addr[l] + b) / 4) % nbanks].insert((addr[l] + b) / 4);
        size_t worst = 1;
        for (auto &s : per) worst = s.size() > worst ? s.size() : worst;
        extra += (long)worst - 1;
    }
    return extra;
}
static std::vector<std::vector<int>> groups_b128() {
    return {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
            {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
            {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
            {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
}
static std::vector<std::vector<int>> groups_2x32() {
    std::vector<std::vector<int>> g(2);
    for (int l = 0; l < 64; ++l) g[l / 32].push_back(l);
    return g;
}

// fragment values: frag[lane][j] = operand value for (row/col base + (lane & 31), k = 16 s + 8 (lane >> 5) + j)
template <int KIND, bool IS_A>
static void read_frag(const Piece &p, int w, int f, int s, int16_t frag[64][8]) {
    if (KIND == KMAJOR) {
        std::vector<int> addr(64);
        for (int lane = 0; lane < 64; ++lane) {
            addr[lane] = frag_off_kmajor<IS_A>(w, f, s, lane);
            for (int j = 0; j < 8; ++j) frag[lane][j] = p.e[addr[lane] / 2 + j];
        }
        g_conf_b128 += conflicts(addr, groups_b128(), 16, 64);
    } else {
        for (int u = 0; u < 2; ++u) {
            std::vector<int> addr(64);
            for (int lane = 0; lane < 64; ++lane) addr[lane] = frag_off_kstrided<IS_A>(w, f, s, u, lane);
            g_conf_tr += conflicts(addr, groups_2x32(), 8, 64);
            for (int lane = 0; lane < 64; ++lane) {
                const int g0 = lane & ~15, L = lane & 15;
                for (int j = 0; j < 4; ++j) frag[lane][4 * u + j] = p.e[addr[g0 + (L / 4) + 4 * j] / 2 + (L % 4)];
            }
        }
    }
}

template <int AK, int BK>
static void run(int BN, unsigned seed) {
    const int wtn = BN / 4, nfj = BN / 128;
    Operand A{256, std::vector<int16_t>(256 * 64)}, B{BN, std::vector<int16_t>((size_t)BN * 64)};
    srand(seed);
    for (auto &x : A.v) x = (int16_t)(rand() % 7 - 3);
    for (auto &x : B.v) x = (int16_t)(rand() % 5 - 2);
    std::vector<Piece> pa(2), pb(2);
    stage_piece<AK, true>(A, 0, wtn, pa[0]);
    stage_piece<AK, true>(A, 1, wtn, pa[1]);
    for (int h = 0; h < nfj; ++h) stage_piece<BK, false>(B, h, wtn, pb[h]);

    std::vector<int> C((size_t)256 * BN, 0x7fffffff);
    for (int wave = 0; wave < 8; ++wave) {
        const int wr = wave >> 2, wc = wave & 3;
        // accumulators acc[fi][fj][lane][r]
        std::vector<int> acc((size_t)4 * nfj * 64 * 16, 0);
        for (int s = 0; s < 4; ++s)
            for (int fi = 0; fi < 4; ++fi)
                for (int fj = 0; fj < nfj; ++fj) {
                    int16_t af[64][8], bf[64][8];
                    read_frag<AK, true>(pa[fi >> 1], wr, fi & 1, s, af);
                    read_frag<BK, false>(pb[fj], wc, 0, s, bf);
                    // acc = mfma(bf, af, acc): MFMA-A = bf (rows = output columns), MFMA-B = af (columns = output rows)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int r = 0; r < 16; ++r) {
                            const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), j = lane & 31;
                            int sum = 0;
                            for (int k = 0; k < 16; ++k) sum += (int)bf[i + 32 * (k / 8)][k % 8] * (int)af[j + 32 * (k / 8)][k % 8];
                            acc[(((size_t)fi * nfj + fj) * 64 + lane) * 16 + r] += sum;
                        }
                }
        // (1) the accumulator -> (row, column) map used by the fp32 epilogue
        // (2) the bf16 epilogue through the wave-private LDS region
        std::vector<int16_t> region((size_t)128 * wtn, -32768);
        for (int fi = 0; fi < 4; ++fi)
            for (int fj = 0; fj < nfj; ++fj)
                for (int lane = 0; lane < 64; ++lane) {
                    for (int r = 0; r < 16; ++r) {
                        const int row = 128 * wr + 32 * fi + (lane & 31);
                        const int col = wtn * wc + 32 * fj + acc_col(r, lane);
                        const int want = [&] { int t = 0; for (int k = 0; k < 64; ++k) t += (int)A.at(row, k) * (int)B.at(col, k); return t; }();
                        const int got = acc[(((size_t)fi * nfj + fj) * 64 + lane) * 16 + r];
                        if (got != want && g_fail < 10) { g_fail++; std::printf("acc map: wave %d fi %d fj %d lane %d r %d: got %d want %d\n", wave, fi, fj, lane, r, got, want); }
                    }
                    for (int q = 0; q < 4; ++q) {
                        const int off = epi_write_off(fi, fj, q, lane, wtn);
                        for (int e = 0; e < 4; ++e) region[off / 2 + e] = (int16_t)acc[(((size_t)fi * nfj + fj) * 64 + lane) * 16 + 4 * q + e];
                    }
                }
        const int passes = 128 / (64 / (wtn / 8));
        for (int it = 0; it < passes; ++it)
            for (int lane = 0; lane < 64; ++lane) {
                int row, c, off;
                epi_read_map(it, lane, wtn, &row, &c, &off);
                for (int e = 0; e < 8; ++e) C[(size_t)(128 * wr + row) * BN + wtn * wc + 8 * c + e] = region[off / 2 + e];
            }
    }
    for (int m = 0; m < 256; ++m)
        for (int n = 0; n < BN; ++n) {
            int want = 0;
            for (int k = 0; k < 64; ++k) want += (int)A.at(m, k) * (int)B.at(n, k);
            if (C[(size_t)m * BN + n] != want && g_fail < 10) { g_fail++; std::printf("C[%d][%d] = %d, want %d (AK %d BK %d BN %d)\n", m, n, C[(size_t)m * BN + n], want, AK, BK, BN); }
        }
    std::printf("AK=%d BK=%d BN=%d: %s; bank-conflict cycles: b128 %ld, tr %ld\n", AK, BK, BN, g_fail ? "FAIL" : "ok", g_conf_b128, g_conf_tr);
}

// ---- duo schedule (128 x 256 tiles): the staging goes through gm::DuoStagerAddr's own address arithmetic (tile pointer + wave-uniform
// deltas + ONE per-lane offset) on a real matrix in memory, ragged edges by moving the last tile back inside the matrix; then fragment reads,
// MFMA lane layout and the accumulator -> (row, column) map of the epilogue (epi_write_off / epi_read_map at WTN = 64, two 32-row passes)
template <int BK>
static void run_duo(long M, long N, long K, long trow, long tcol, unsigned seed) {
    const long lda = K, ldb = (BK == KMAJOR) ? K : N;
    std::vector<int16_t> A((size_t)M * K), B((size_t)N * K);
    srand(seed);
    for (auto &x : A) x = (int16_t)(rand() % 7 - 3);
    for (auto &x : B) x = (int16_t)(rand() % 5 - 2);
    auto bat = [&](long n, long k) { return BK == KMAJOR ? B[(size_t)n * K + k] : B[(size_t)k * N + n]; };
    long m0 = trow * BM_DUO, n0 = tcol * 256;
    if (m0 > M - BM_DUO) m0 = M - BM_DUO;
    if (n0 > N - 256) n0 = N - 256;
    std::vector<int> acc((size_t)8 * 2 * 2 * 64 * 16, 0);
    const int fail0 = g_fail;
    for (long kt = 0; kt < K / 64; ++kt) {
        Piece pa, pb[2];
        std::vector<int> wa(PIECE_BYTES / 16, 0), wb0(PIECE_BYTES / 16, 0), wb1(PIECE_BYTES / 16, 0);
        for (int wave = 0; wave < 8; ++wave)
            for (int lane = 0; lane < 64; ++lane) {
                DuoStagerAddr<KMAJOR, true> sa;
                DuoStagerAddr<BK, false> sb;
                sa.init((const char *)A.data(), lda, m0, 0, wave, lane);
                sb.init((const char *)B.data(), ldb, n0, 0, wave, lane);
                for (long t = 0; t < kt; ++t) { sa.cur += sa.adv; sb.cur += sb.adv; }
                for (int i = 0; i < 2; ++i) {
                    const int dst = duo_stage_dst(wave, i, lane);
                    std::memcpy(&pa.e[dst / 2], sa.src(0, i), 16);
                    wa[dst / 16]++;
                    for (int h = 0; h < 2; ++h) {
                        std::memcpy(&pb[h].e[dst / 2], sb.src(h, i), 16);
                        (h ? wb1 : wb0)[dst / 16]++;
                    }
                    // the map the addresses implement
                    const StageSrc s = duo_stage_src<KMAJOR, true>(0, wave, i, lane);
                    if (sa.src(0, i) != (const char *)&A[(size_t)(m0 + s.rc) * K + kt * 64 + s.k]) { if (g_fail++ < 5) std::printf("duo A address != map\n"); }
                }
            }
        for (size_t c = 0; c < wa.size(); ++c)
            if (wa[c] != 1 || wb0[c] != 1 || wb1[c] != 1) { g_fail++; std::printf("duo stage: chunk written %d/%d/%d times\n", wa[c], wb0[c], wb1[c]); break; }
        for (int wave = 0; wave < 8; ++wave) {
            const int wr = wave >> 2, wc = wave & 3;
            for (int s = 0; s < 4; ++s)
                for (int fi = 0; fi < 2; ++fi)
                    for (int fj = 0; fj < 2; ++fj) {
                        int16_t af[64][8], bf[64][8];
                        read_frag<KMAJOR, true>(pa, wr, fi, s, af);
                        read_frag<BK, false>(pb[fj], wc, 0, s, bf);
                        for (int lane = 0; lane < 64; ++lane)
                            for (int r = 0; r < 16; ++r) {
                                const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), j = lane & 31;
                                int sum = 0;
                                for (int k = 0; k < 16; ++k) sum += (int)bf[i + 32 * (k / 8)][k % 8] * (int)af[j + 32 * (k / 8)][k % 8];
                                acc[((((size_t)wave * 2 + fi) * 2 + fj) * 64 + lane) * 16 + r] += sum;
                            }
                    }
        }
    }
    // epilogue: 32 rows x 64 columns per pass through 4 KiB
    std::vector<long> C((size_t)BM_DUO * 256, 0x7fffffff);
    for (int wave = 0; wave < 8; ++wave) {
        const int wr = wave >> 2, wc = wave & 3;
        for (int fi = 0; fi < 2; ++fi) {
            std::vector<int> region(2048, 0x7fffffff);      // 4 KiB as 16-bit slots holding ints
            for (int fj = 0; fj < 2; ++fj)
                for (int lane = 0; lane < 64; ++lane)
                    for (int q = 0; q < 4; ++q) {
                        const int off = epi_write_off(0, fj, q, lane, 64);
                        for (int e = 0; e < 4; ++e) region[off / 2 + e] = acc[((((size_t)wave * 2 + fi) * 2 + fj) * 64 + lane) * 16 + 4 * q + e];
                    }
            for (int it = 0; it < 4; ++it)
                for (int lane = 0; lane < 64; ++lane) {
                    int row, c, off;
                    epi_read_map(it, lane, 64, &row, &c, &off);
                    for (int e = 0; e < 8; ++e) C[(size_t)(64 * wr + 32 * fi + row) * 256 + 64 * wc + 8 * c + e] = region[off / 2 + e];
                }
        }
    }
    for (int m = 0; m < BM_DUO; ++m)
        for (int n = 0; n < 256; ++n) {
            long want = 0;
            for (long k = 0; k < K; ++k) want += (long)A[(size_t)(m0 + m) * K + k] * (long)bat(n0 + n, k);
            if (C[(size_t)m * 256 + n] != want && g_fail < 10) { g_fail++; std::printf("duo C[%d][%d] = %ld, want %ld (BK %d)\n", m, n, C[(size_t)m * 256 + n], want, BK); }
        }
    std::printf("duo BK=%d M=%ld N=%ld K=%ld tile (%ld, %ld): %s; bank-conflict cycles: b128 %ld, tr %ld\n", BK, M, N, K, trow, tcol, g_fail == fail0 ? "duo-ok" : "FAIL",
                g_conf_b128, g_conf_tr);
}

// ---- work items of the persistent schedule --------------------------------------------------------------------------------------------
// Replays every workgroup's item list (positions cp, cp + G, ... with cp = xcd_order(block, G)) for the plans csrc/xq_gemm.hip makes
// (plan_persistent is restated here: whole tiles for the full rounds of CUs, the remainder cut along K; weight gradient: every tile cut),
// and checks (1) every (tile, K tile) of the product is covered exactly once, slabs are distinct; (2) the scalar tile walk of the
// persistent kernel (next_item_walk) yields field for field what decode_item yields, for the compute AND the staging cursor.
struct Plan {
    long main_items;
    int tail_tiles, tail_splits, split_major, tiles_n, kt_full, step_r, step_c;
};

static void check_plan(long tiles_m, int tiles_n, int kt_full, long cus, bool weight_grad) {
    const long tiles = tiles_m * tiles_n;
    Plan g{tiles, 0, 1, 0, tiles_n, kt_full, 0, 0};
    if (weight_grad) {
        long s = cus / tiles; if (s > kt_full / 2) s = kt_full / 2; if (s < 1) s = 1;
        g.main_items = 0; g.tail_tiles = (int)tiles; g.tail_splits = (int)s; g.split_major = 1;
    } else {
        const long rem = tiles % cus;
        if (tiles > cus && rem > 0 && rem <= cus / 4 && kt_full >= 4) {
            long S = cus / rem; if (S > kt_full / 2) S = kt_full / 2;
            if (S >= 2) { g.main_items = tiles - rem; g.tail_tiles = (int)rem; g.tail_splits = (int)S; }
        }
    }
    const long items = g.main_items + (long)g.tail_tiles * g.tail_splits;
    const long G = items < cus ? items : cus;
    g.step_r = (int)(G / tiles_n);
    g.step_c = (int)(G % tiles_n);
    std::vector<int> cover((size_t)tiles * kt_full, 0);
    std::set<long> slabs;
    int bad = 0;
    for (long b = 0; b < G; ++b) {
        long cp = xcd_order(b, G);
        if (cp >= items) continue;
        Item ref, walk;
        decode_item(g, cp, ref);
        int row = (int)(ref.m0 / BM), col = (int)(ref.n0 / 256);
        walk = ref;
        for (;;) {
            if (walk.m0 != ref.m0 || walk.n0 != ref.n0 || walk.k0 != ref.k0 || walk.KT != ref.KT || walk.slab != ref.slab || walk.slab_idx != ref.slab_idx) {
                if (bad++ < 3) std::printf("item walk differs at p=%ld (tiles %ld x %d, kt %d, G %ld): m0 %ld/%ld n0 %ld/%ld k0 %ld/%ld KT %d/%d\n", cp, tiles_m, tiles_n,
                                           kt_full, G, walk.m0, ref.m0, walk.n0, ref.n0, walk.k0, ref.k0, walk.KT, ref.KT);
            }
            const long tile = (ref.m0 / BM) * tiles_n + ref.n0 / 256;
            if (ref.KT < (ref.slab ? 1 : 2) || ref.k0 % BKT || tile < 0 || tile >= tiles) { bad++; std::printf("bad item at p=%ld\n", cp); break; }
            for (int t = 0; t < ref.KT; ++t) cover[(size_t)tile * kt_full + ref.k0 / BKT + t]++;
            if (ref.slab && !slabs.insert(ref.slab_idx).second) { bad++; std::printf("slab %ld used twice\n", ref.slab_idx); }
            cp += G;
            if (cp >= items) break;
            decode_item(g, cp, ref);
            next_item_walk(g, cp, row, col, walk);
        }
    }
    for (int c : cover) if (c != 1) { bad++; break; }
    if (bad) { g_fail++; std::printf("item lists FAIL: tiles %ld x %d, kt %d, cus %ld, weight_grad %d (%d problems)\n", tiles_m, tiles_n, kt_full, cus, (int)weight_grad, bad); }
}

// ---- staging address streams ------------------------------------------------------------------------------------------------------------
// The persistent kernel (scalar tile cursor moved by adds, tile walk, per-lane offsets kept between interior tiles) must stage
// exactly the bytes the default kernel stages (base + kt * adv + off, decode_item + init per item — the path validated on the GPU): for
// every workgroup of a plan, for a sample of lanes, the two address streams are compared K tile by K tile, both operands, all four
// LDS-DMA instructions of a tile.  The control flow restates PR_ADVANCE of csrc/xq_gemm.hip; the arithmetic is gm::StagerAddr itself.
template <int AK, int BK>
static void check_streams(long M, long N, long Kred, long cus, bool weight_grad) {
    const long tiles_m = (M + 255) / 256, tiles = tiles_m * ((N + 255) / 256);
    const int tiles_n = (int)((N + 255) / 256), kt_full = (int)(Kred / 64);
    Plan g{tiles, 0, 1, 0, tiles_n, kt_full, 0, 0};
    if (weight_grad) {
        long sp = cus / tiles; if (sp > kt_full / 2) sp = kt_full / 2; if (sp < 1) sp = 1;
        g.main_items = 0; g.tail_tiles = (int)tiles; g.tail_splits = (int)sp; g.split_major = 1;
    } else {
        const long rem = tiles % cus;
        if (tiles > cus && rem > 0 && rem <= cus / 4 && kt_full >= 4) {
            long S = cus / rem; if (S > kt_full / 2) S = kt_full / 2;
            if (S >= 2) { g.main_items = tiles - rem; g.tail_tiles = (int)rem; g.tail_splits = (int)S; }
        }
    }
    const long items = g.main_items + (long)g.tail_tiles * g.tail_splits;
    const long G = items < cus ? items : cus;
    g.step_r = (int)(G / tiles_n);
    g.step_c = (int)(G % tiles_n);
    // operands: A [M rows][lda] or [Kred][lda = M] ; addresses only, never dereferenced
    const char *A = (const char *)0x100000000ULL, *B = (const char *)0x900000000ULL;
    const long lda = (AK == KMAJOR) ? Kred : M, ldb = (BK == KMAJOR) ? Kred : N;
    long compared = 0;
    int bad = 0;
    for (long b = 0; b < G; b += (G > 40 ? 7 : 1)) {
        const long cp0 = xcd_order(b, G);
        if (cp0 >= items) continue;
        for (int wave : {0, 3, 5, 7})
            for (int lane : {0, 9, 31, 32, 47, 63}) {
                // default kernel
                std::vector<unsigned long long> ref;
                for (long p = cp0; p < items; p += G) {
                    Item it;
                    decode_item(g, p, it);
                    StagerAddr<AK, true> sa;
                    StagerAddr<BK, false> sb;
                    sa.init(A, lda, it.m0, M, it.k0, wave, lane, 64, 2);
                    sb.init(B, ldb, it.n0, N, it.k0, wave, lane, 64, 2);
                    for (int kt = 0; kt < it.KT; ++kt)
                        for (int h = 0; h < 2; ++h)
                            for (int i = 0; i < 2; ++i) {
                                ref.push_back((unsigned long long)(sa.base + kt * sa.adv + sa.off[h][i]));
                                ref.push_back((unsigned long long)(sb.base + kt * sb.adv + sb.off[h][i]));
                            }
                }
                // persistent kernel: cursor + walk + retarget
                std::vector<unsigned long long> got;
                {
                    Item it;
                    long sp = cp0;
                    decode_item(g, sp, it);
                    int row = (int)(it.m0 / BM), col = (int)(it.n0 / 256);
                    StagerAddr<AK, true> sa;
                    StagerAddr<BK, false> sb;
                    sa.init(A, lda, it.m0, M, it.k0, wave, lane, 64, 2);
                    sb.init(B, ldb, it.n0, N, it.k0, wave, lane, 64, 2);
                    sa.make_scalar();
                    sb.make_scalar();
                    int s_kt = 0, s_KT = it.KT, s_dummy = 0;
                    while (!s_dummy) {
                        for (int h = 0; h < 2; ++h)
                            for (int i = 0; i < 2; ++i) {
                                got.push_back((unsigned long long)(sa.cur + sa.off[h][i]));
                                got.push_back((unsigned long long)(sb.cur + sb.off[h][i]));
                            }
                        // PR_ADVANCE, SB branch
                        if (++s_kt == s_KT) {
                            sp += G;
                            if (sp < items) {
                                Item nx;
                                next_item_walk(g, sp, row, col, nx);
                                sa.retarget(A, lda, nx.m0, M, nx.k0, wave, lane, 64);
                                sb.retarget(B, ldb, nx.n0, N, nx.k0, wave, lane, 64);
                                sa.make_scalar();
                                sb.make_scalar();
                                s_KT = nx.KT;
                                s_kt = 0;
                            } else {
                                s_dummy = 1;
                            }
                        } else {
                            sa.step();
                            sb.step();
                        }
                    }
                }
                if (got != ref) {
                    if (bad++ < 3) {
                        size_t k = 0;
                        while (k < got.size() && k < ref.size() && got[k] == ref[k]) ++k;
                        std::printf("address streams differ: block %ld wave %d lane %d at entry %zu of %zu / %zu (AK %d BK %d M %ld N %ld K %ld)\n", b, wave, lane, k,
                                    got.size(), ref.size(), AK, BK, M, N, Kred);
                    }
                }
                compared += (long)ref.size();
            }
    }
    if (bad) g_fail++;
    std::printf("staging streams AK=%d BK=%d M=%ld N=%ld K=%ld%s: %ld addresses compared, %s\n", AK, BK, M, N, Kred, weight_grad ? " (weight gradient)" : "", compared,
                bad ? "FAIL" : "ok");
}

static void check_item_lists() {
    int n = 0;
    for (long cus : {256L, 304L, 8L})
        for (long tiles_m : {1L, 2L, 3L, 29L, 88L, 129L, 257L})
            for (int tiles_n : {1, 2, 3, 9, 12})
                for (int kt : {2, 3, 4, 12, 36, 48}) {
                    check_plan(tiles_m, tiles_n, kt, cus, false);
                    ++n;
                }
    for (int tiles_n : {1, 3, 12})            // weight gradients: P x Q tiles, R / 64 K tiles
        for (long tiles_m : {3L, 9L, 12L})
            for (int kt : {2, 5, 401, 1026}) { check_plan(tiles_m, tiles_n, kt, 256, true); ++n; }
    std::printf("item lists: %d plans replayed (coverage exactly once; scalar tile walk == decode_item)%s\n", n, g_fail ? " — FAIL" : "");
}

int main() {
    for (int BN : {256, 128}) {
        run<KMAJOR, KMAJOR>(BN, 1);
        run<KMAJOR, KSTRIDED>(BN, 2);
        run<KSTRIDED, KSTRIDED>(BN, 3);
    }
    // duo schedule: interior tile, ragged last tile row (M = 300: rows 172..299), ragged last tile column (N = 384: columns 128..383)
    run_duo<KMAJOR>(256, 512, 128, 1, 1, 4);
    run_duo<KSTRIDED>(256, 512, 128, 0, 1, 5);
    run_duo<KMAJOR>(300, 384, 192, 2, 1, 6);
    run_duo<KSTRIDED>(300, 384, 192, 2, 1, 7);
    // tile order is a bijection for awkward totals
    for (long total : {1L, 7L, 8L, 9L, 771L, 2313L, 3084L, 252L}) {
        std::vector<int> seen(total, 0);
        for (long id = 0; id < total; ++id) {
            const long p = xcd_order(id, total);
            if (p < 0 || p >= total) { g_fail++; std::printf("xcd_order out of range\n"); break; }
            seen[p]++;
        }
        for (int c : seen) if (c != 1) { g_fail++; std::printf("xcd_order(total=%ld) is not a bijection\n", total); break; }
    }
    check_item_lists();
    // bench shapes (ragged last row tile: 65 664 = 256.5 x 256), a ragged column tile (N = 1152), K-split tails (22 300 x 768: 88 x 3 = 264 tiles)
    check_streams<KMAJOR, KMAJOR>(65664, 2304, 768, 256, false);
    check_streams<KMAJOR, KMAJOR>(65664, 768, 3072, 256, false);
    check_streams<KMAJOR, KMAJOR>(22300, 768, 768, 256, false);
    check_streams<KMAJOR, KMAJOR>(788, 1152, 384, 256, false);
    check_streams<KMAJOR, KSTRIDED>(65664, 768, 2304, 256, false);
    check_streams<KMAJOR, KSTRIDED>(22300, 768, 768, 256, false);
    check_streams<KMAJOR, KSTRIDED>(2052, 3072, 768, 256, false);
    check_streams<KSTRIDED, KSTRIDED>(2304, 768, 65664, 256, true);
    check_streams<KSTRIDED, KSTRIDED>(768, 3072, 65664, 256, true);
    check_streams<KSTRIDED, KSTRIDED>(1152, 384, 788 / 64 * 64, 256, true);
    std::printf(g_fail ? "FAILED\n" : "ALL OK\n");
    return g_fail ? 1 : 0;
}
