// Levels 1 AND 2 of the float32 2-D forward transform in one tile program: the level-1
// lowpass plane (LoLo1, as large as the image) never leaves the chip.
//
// One launch per level moves 30.5 B/px for a forward transform (X 4 -> LoLo1 4 + Yh[0] 12,
// LoLo1 4 -> LoLo2 1 + Yh[1] 3, ...) against 20 B/px compulsory; two thirds of the excess is
// the round trip of LoLo1.  Here a workgroup owns a tile of T2R x T2C LEVEL-2 lowpass
// outputs, i.e. a core of 2*T2R x 2*T2C level-1 samples, and
//
//   1. column-filters X straight from global memory: Lo over the core plus the level-2 window
//      halo of M-2 samples on every side, Hi over the core only.  A thread owns FOUR adjacent
//      columns of a strip of rows and requests its whole window as 16-byte loads before the
//      first use: one memory latency per tile (the first version, one column per thread in
//      four rounds, spent four);
//   2. row-filters from the LDS planes: a thread owns 2 rows x 4 columns of the core (two
//      quads -> two Yh[0] records, bounced through a wave-private slab and written as 1 KiB
//      runs) plus a share of the halo ring, where only LoLo1 is needed -- recomputed here
//      instead of being read back from HBM.  The LoLo1 values stay in registers until every
//      thread has finished reading the Lo plane, then they are written OVER it (LDS = max,
//      not sum);
//   2c (image-border tiles) fills the part of the LoLo1 window that lies outside the image by
//      the symmetric reflection the level-2 filter applies to ITS input (reflecting LoLo1, not
//      X: coldfilt's extension, dtcwt/numpy/lowlevel.py:82-154 via utils.py:136-153);
//   3. runs the level-2 column pass (coldfilt pairs) from the LoLo1 plane in LDS;
//   4. the level-2 row pass + q2c of fused2d_tiles_v2.hpp, writing LoLo2 and Yh[1].
//
// Level 2 must not need edge padding (the even-extended image is a multiple of 4 in both
// directions, transform2d.py:134-140), otherwise the plan keeps one launch per level.
// Reference semantics: dtcwt/numpy/transform2d.py:112-160.
#pragma once
#include "fused2d_tiles_v2.hpp"

namespace dt2d {

template <int T2R_, int T2C_, int RS_, int PS_, int M0_, int M1_, int M_, int NT_ = 256>
struct Fwd12Cfg {
    static constexpr int T2R = T2R_, T2C = T2C_, RS = RS_, PS = PS_, M0 = M0_, M1 = M1_, M = M_, NT = NT_;
    static constexpr int H0 = M0 / 2, H1 = M1 / 2, HH = cmax(H0, H1);
    static constexpr int HB = (H0 + 3) & ~3, HC4 = (HH + 3) & ~3;      // halos rounded up to whole 16-byte pieces
    static constexpr int HALO = M - 2;                            // level-2 window beyond the core, per side
    static constexpr int CR = 2 * T2R, CC = 2 * T2C;              // core, level-1 samples
    static constexpr int NR1 = CR + 2 * HALO, NC1 = CC + 2 * HALO;   // LoLo1 window
    static constexpr int PO = cmax(HALO + HB, HC4);               // Lo plane: columns core - PO .. core + CC + PO
    static constexpr int W = CC + 2 * PO;                         // plane width
    // columns actually loaded and column-filtered: core - PL .. core + CC + PL (what the LoLo1 window needs),
    // in pairs; they sit at plane columns XOFF ..
    static constexpr int PL = cmax(HALO + ((H0 + 1) & ~1), (HH + 1) & ~1);
    static constexpr int WX = CC + 2 * PL, NP = WX / 2, XOFF = PO - PL;
    static constexpr int WH = CC + 2 * HC4;                       // Hi plane: columns core - HC4 .. (core rows only)
    static constexpr int OH = PO - HC4;                           // Hi plane origin in Lo plane columns
    static constexpr int OL1 = PO - HALO;                         // LoLo1 window origin in Lo plane columns
    static constexpr int SLO = NR1 * W, SHI = CR * WH;
    static constexpr int S2 = T2R * NC1;                          // one level-2 plane (they alias the Hi plane)
    static constexpr int SB = cmax(SHI, 2 * S2);
    static constexpr int NWAVE = NT / 64;
    static constexpr int MIN_WAVES = 4;                           // __launch_bounds__: waves per SIMD (4 workgroups of 256 per CU, <= 128 VGPRs)
    static constexpr int LDS_FLOATS = SLO + SB + NWAVE * STAGE_FLOATS_PER_WAVE;
    static constexpr int TI = T2R / 2, TJ = T2C / 2;
    static constexpr int NS2 = TI / PS;
    static constexpr int WN2 = 4 * PS + 2 * M - 4;
    // phase 2 work: core tasks of 2 rows x 4 columns, halo tasks of 1 row x 4 columns
    static constexpr int NU = CR / 2, NV4 = CC / 4, NCORE = NU * NV4;
    static constexpr int NCR = (NCORE + NT - 1) / NT;             // core rounds per thread
    static constexpr bool CORE_EXACT = NCORE % NT == 0;           // every thread has a core task in every round
    static constexpr int HG = NC1 / 4, NTOP = HALO * HG;          // groups per full row, top band (= bottom band)
    static constexpr int HS = HALO / 4;                           // groups per side band row, each side
    static constexpr int NHALO = 2 * NTOP + CR * 2 * HS;
    static constexpr int NHR = (NHALO + NT - 1) / NT;             // halo rounds per thread
    static_assert(M0 % 2 == 1 && M1 % 2 == 1 && M % 2 == 0, "odd biort / even q-shift lengths");
    static_assert(HALO % 4 == 0, "q-shift length must be 2 mod 4 + ... (M - 2 a multiple of 4): 10, 14, 18");
    static_assert(HALO % RS == 0 && CR % RS == 0, "column-pass strips must not straddle the core edge");
    static_assert(TI % PS == 0 && T2R % 2 == 0 && T2C % 2 == 0 && NT % 64 == 0, "tile shape");
    // view with the member names the shared level-2 row pass expects
    struct L2View {
        static constexpr bool BP = false;
        static constexpr int TR = T2R, TC = T2C, M = M_, TI = T2R / 2, TJ = T2C / 2, NCI = NC1;
    };
};

// A wave's ds_read_b128 is served in four groups of 16 lanes -- {0-3,12-15,20-27}, {4-11,16-19,28-31} and
// the same again + 32 (MI355X_MICROARCH.md, LDS) -- conflict-free when the 16 lanes of a group read 256
// consecutive bytes.  Tasks are dealt in row-major order, 16 (or 20, 32) to a row of the tile, so giving
// lane l the task of index perm(l) makes every hardware group a run of 16 consecutive tasks.  In isolation
// (tools/kbench/lds_probe, profiles/r02/lds_probe.txt) such reads at row strides of 72-96 floats cost 3.38 ns
// per wave-instruction in lane order and 1.82 ns -- the conflict-free rate -- permuted.  The kernel's time and
// its SQ_LDS_BANK_CONFLICT count do not move (that counter also charges the 13-cycle transfer of every wide
// ds_write: 5.7 ns per ds_write_b128 against 1.9 ns per ds_read_b128), LDS is not what bounds it; the
// permutation is kept because it is free and the staged record order depends on it.
DT_HD int lds128_perm(int tid) {
    const int l = tid & 31;
    const int p = l < 4 ? l : l < 12 ? l + 12 : l < 16 ? l - 8 : l < 20 ? l + 8 : l < 28 ? l - 12 : l;
    return (tid & ~31) | p;
}

// per-thread values that cross the barrier between reading the Lo plane and overwriting it
template <class C>
struct Fwd12State {
    float ll[C::NCR][2][4];       // LoLo1 of the thread's core tasks
    float halo[C::NHR][4];        // LoLo1 of its halo tasks
};

// ---- phase 1: level-1 column pass from global memory, two columns per thread ------------
// Split in two so that a persistent workgroup can request the NEXT tile's window before it runs
// phases 2-4 of the current one (the loads then overlap that arithmetic and its stores):
//   fwd12_cols_load:    the thread's window (RS + 2 HH rows x 2 columns, 8-byte loads) -> registers
//   fwd12_cols_compute: registers -> Lo / Hi planes in LDS
// (r1, c1): core origin in level-1 coordinates.  One task per thread (NSTRIP * NP <= NT): with two
// columns per task the whole workgroup takes part (252 of 256 threads for the 16 x 32 tile) and the
// window is 28 registers per thread -- four columns per task left half the waves idle in this phase
// and cost the busy ones 56 registers, too many to keep a prefetched window alive through phase 2.
template <class C>
struct Fwd12Win {
    static constexpr int WN = C::RS + 2 * C::HH;
    float w[WN][2];
};

template <class C>
DT_HD void fwd12_cols_load(const Fwd1Params &p, int tid, int b, int r1, int c1, Fwd12Win<C> &win) {
    const float *Xb = p.X + (int64_t)b * p.inR * p.inC;
    constexpr int NSTRIP = C::NR1 / C::RS;
    constexpr int WN = C::RS + 2 * C::HH;
    static_assert(NSTRIP * C::NP <= C::NT, "one column-pass task per thread");
    if (tid >= NSTRIP * C::NP) return;
    const int rw = r1 - C::HALO, cw = c1 - C::PL;
    const bool interior = rw - C::HH >= 0 && rw + C::NR1 + C::HH <= p.inR && cw >= 0 && cw + C::WX <= p.inC &&
                          (p.inC & 1) == 0;
    const int strip = tid / C::NP, j2 = tid - strip * C::NP;
    const int row0 = strip * C::RS, cc = 2 * j2;
    if (interior) {
        const float *src = Xb + (int64_t)(rw + row0 - C::HH) * p.inC + (cw + cc);
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            // a vector-typed load: scalar loads here get merged with the border path's and stay scalar
            const dt_v2f v = *reinterpret_cast<const dt_v2f *>(src + (int64_t)j * p.inC);
            win.w[j][0] = v.x; win.w[j][1] = v.y;
        }
    } else {
        int gc[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) { gc[e] = reflect_i(cw + cc + e, p.LC); if (gc[e] > p.inC - 1) gc[e] = p.inC - 1; }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            int gr = reflect_i(rw + row0 - C::HH + j, p.LR); if (gr > p.inR - 1) gr = p.inR - 1;
            const float *row = Xb + (int64_t)gr * p.inC;
#pragma unroll
            for (int e = 0; e < 2; ++e) win.w[j][e] = row[gc[e]];
        }
    }
}

template <class C>
DT_HD void fwd12_cols_compute(const Fwd1Params &p, float *sLo, float *sHi, int tid, const Fwd12Win<C> &win) {
    constexpr int NSTRIP = C::NR1 / C::RS;
    if (tid >= NSTRIP * C::NP) return;
    const int strip = tid / C::NP, j2 = tid - strip * C::NP;
    const int row0 = strip * C::RS, cc = C::XOFF + 2 * j2;          // plane column of the pair
    const bool hi = row0 >= C::HALO && row0 < C::HALO + C::CR && cc >= C::OH && cc < C::OH + C::WH;
#pragma unroll
    for (int q = 0; q < C::RS; ++q) {
        float lo[2] = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < C::M0; ++k)
#pragma unroll
            for (int e = 0; e < 2; ++e) lo[e] += p.h0[k] * win.w[q + C::HH + C::H0 - k][e];
        *reinterpret_cast<f2 *>(sLo + (row0 + q) * C::W + cc) = f2{lo[0], lo[1]};
    }
    if (hi) {
#pragma unroll
        for (int q = 0; q < C::RS; ++q) {
            float h[2] = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < C::M1; ++k)
#pragma unroll
                for (int e = 0; e < 2; ++e) h[e] += p.h1[k] * win.w[q + C::HH + C::H1 - k][e];
            *reinterpret_cast<f2 *>(sHi + (row0 - C::HALO + q) * C::WH + (cc - C::OH)) = f2{h[0], h[1]};
        }
    }
}

template <class C>
DT_HD void fwd12_cols(const Fwd1Params &p, float *sLo, float *sHi, int tid, int b, int r1, int c1) {
    Fwd12Win<C> win;
    fwd12_cols_load<C>(p, tid, b, r1, c1, win);
    fwd12_cols_compute<C>(p, sLo, sHi, tid, win);
}

// ---- phase 2a: level-1 row pass over the core, 2 rows x 4 columns per task ----------------
// LoLo1 -> st.ll[round]; the two Yh[0] records of the task -> rec[2][12] (q2c applied)
template <class C>
DT_HD void fwd12_core_compute(const Fwd1Params &p, const float *sLo, const float *sHi, int tid, int round,
                              Fwd12State<C> &st, float (&rec)[2][12]) {
    constexpr int WL = 4 + 2 * C::HC4;
    const int task = round * C::NT + tid;
    if (task >= C::NCORE) return;
    const int u = task / C::NV4, v4 = task - u * C::NV4;
    float hl[2][4], lh[2][4], hh[2][4];
#pragma unroll
    for (int er = 0; er < 2; ++er) {
        float wl[WL], wh[WL];
        const f4 *pl = reinterpret_cast<const f4 *>(sLo + (C::HALO + 2 * u + er) * C::W + C::OH + 4 * v4);
        const f4 *ph = reinterpret_cast<const f4 *>(sHi + (2 * u + er) * C::WH + 4 * v4);
#pragma unroll
        for (int j = 0; j < WL / 4; ++j) {
            const f4 a = pl[j], c = ph[j];
            wl[4 * j] = a.x; wl[4 * j + 1] = a.y; wl[4 * j + 2] = a.z; wl[4 * j + 3] = a.w;
            wh[4 * j] = c.x; wh[4 * j + 1] = c.y; wh[4 * j + 2] = c.z; wh[4 * j + 3] = c.w;
        }
#pragma unroll
        for (int ec = 0; ec < 4; ++ec) {
            float s_ll = 0.f, s_hl = 0.f, s_lh = 0.f, s_hh = 0.f;
#pragma unroll
            for (int k = 0; k < C::M0; ++k) {
                s_ll += p.h0[k] * wl[ec + C::HC4 + C::H0 - k];
                s_hl += p.h0[k] * wh[ec + C::HC4 + C::H0 - k];
            }
#pragma unroll
            for (int k = 0; k < C::M1; ++k) {
                s_lh += p.h1[k] * wl[ec + C::HC4 + C::H1 - k];
                s_hh += p.h1[k] * wh[ec + C::HC4 + C::H1 - k];
            }
            st.ll[round][er][ec] = s_ll; hl[er][ec] = s_hl; lh[er][ec] = s_lh; hh[er][ec] = s_hh;
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float a[2][2] = {{hl[0][2 * q], hl[0][2 * q + 1]}, {hl[1][2 * q], hl[1][2 * q + 1]}};
        const float c[2][2] = {{lh[0][2 * q], lh[0][2 * q + 1]}, {lh[1][2 * q], lh[1][2 * q + 1]}};
        const float d[2][2] = {{hh[0][2 * q], hh[0][2 * q + 1]}, {hh[1][2 * q], hh[1][2 * q + 1]}};
        store_record(rec[q], a, c, d);
    }
}

// A wave's 64 core tasks hold 128 records, consecutive in memory; they leave through the
// wave's 64-record slab in two halves: lanes [32 half, 32 half + 32) deposit, all 64 lanes flush.
template <class C, bool FULL = false>
DT_HD void fwd12_core_deposit(float *stage, int tid, int round, int half, const float (&rec)[2][12]) {
    const int lane = tid & 63, wave = tid >> 6;
    if ((lane >> 5) != half || (!FULL && round * C::NT + tid >= C::NCORE)) return;
    f4 *slab = reinterpret_cast<f4 *>(stage + wave * STAGE_FLOATS_PER_WAVE + (lane & 31) * 24);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int k = 0; k < 3; ++k) slab[3 * q + k] = f4{rec[q][4 * k], rec[q][4 * k + 1], rec[q][4 * k + 2], rec[q][4 * k + 3]};
}

// FULL: the tile's core lies inside the image and NCORE fills every round: no bounds checks, so every lane
// issues every store (the persistent kernel counts on that, see k_fwd12p)
template <class C, bool FULL = false>
DT_HD void fwd12_core_flush(const Fwd1Params &p, const float *stage, int tid, int round, int half, int b, int r1,
                            int c1) {
    const int HR = p.LR / 2, HCc = p.LC / 2;
    const int lane = tid & 63, wave = tid >> 6;
    const f4 *slab = reinterpret_cast<const f4 *>(stage + wave * STAGE_FLOATS_PER_WAVE);
    if (32 % C::NV4 == 0) {         // the slab holds whole rows of 2 NV4 records: see flush_record_rows
        const int t0 = DT_WAVE_UNIFORM_I(round * C::NT + wave * 64 + half * 32);
        const int u0 = t0 / C::NV4;
        const int rows = (p.LR - r1) / 2 - u0, recs = (p.LC - c1) / 2, rows_tile = C::NU - u0;
        int rows_ok = rows < 32 / C::NV4 ? rows : 32 / C::NV4;
        if (rows_tile < rows_ok) rows_ok = rows_tile;
        float *row0 = p.Yh + (((int64_t)b * HR + r1 / 2 + u0) * HCc + c1 / 2) * 12;
        flush_record_rows<2 * C::NV4, true>(row0, HCc * 12, rows_ok, recs < 2 * C::NV4 ? recs : 2 * C::NV4, slab, lane);
        return;
    }
    const int task0 = round * C::NT + wave * 64 + half * 32;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int j = lane + 64 * k;            // 16-byte piece of the 64 records in the slab
        const int rr = j / 3, part = j - 3 * rr;
        const int task = task0 + (rr >> 1);
        const int u = task / C::NV4, v4 = task - u * C::NV4;
        const int R = r1 + 2 * u, Cc = c1 + 4 * v4 + 2 * (rr & 1);
        if (FULL || (task < C::NCORE && R < p.LR && Cc < p.LC)) {
            float *rec = p.Yh + (((int64_t)b * HR + R / 2) * HCc + Cc / 2) * 12;
            DT_STREAM_STORE_F4(reinterpret_cast<f4 *>(rec) + part, slab[j]);
        }
    }
}

// ---- phase 2b: LoLo1 of the halo ring, 1 row x 4 columns per task -------------------------
template <class C>
DT_HD void fwd12_halo_pos(int t, int &row, int &col) {
    if (t < 2 * C::NTOP) {
        const int bot = t >= C::NTOP, tt = t - bot * C::NTOP;
        const int rr = tt / C::HG;
        row = rr + (bot ? C::HALO + C::CR : 0);
        col = 4 * (tt - rr * C::HG);
    } else {
        const int tt = t - 2 * C::NTOP;
        const int rr = tt / (2 * C::HS), k = tt - rr * (2 * C::HS);
        row = C::HALO + rr;
        col = k < C::HS ? 4 * k : C::HALO + C::CC + 4 * (k - C::HS);
    }
}

template <class C>
DT_HD void fwd12_halo_compute(const Fwd1Params &p, const float *sLo, int tid, Fwd12State<C> &st) {
    constexpr int WL = 4 + 2 * C::HB;
#pragma unroll
    for (int r = 0; r < C::NHR; ++r) {
        const int t = r * C::NT + tid;
        if (t >= C::NHALO) break;
        int row, col;
        fwd12_halo_pos<C>(t, row, col);
        float wl[WL];
        const f4 *pl = reinterpret_cast<const f4 *>(sLo + row * C::W + (C::OL1 - C::HB) + col);
#pragma unroll
        for (int j = 0; j < WL / 4; ++j) {
            const f4 a = pl[j];
            wl[4 * j] = a.x; wl[4 * j + 1] = a.y; wl[4 * j + 2] = a.z; wl[4 * j + 3] = a.w;
        }
#pragma unroll
        for (int ec = 0; ec < 4; ++ec) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < C::M0; ++k) s += p.h0[k] * wl[ec + C::HB + C::H0 - k];
            st.halo[r][ec] = s;
        }
    }
}

// ---- phase 2 write-back (after a barrier): LoLo1 over the Lo plane (same row stride W, window
// column 0 at plane column OL1), and to HBM when the level-1 scale is an output -----------------
template <class C>
DT_HD void fwd12_writeback(const Fwd1Params &p, float *sLo, int tid, int b, int r1, int c1, const Fwd12State<C> &st) {
#pragma unroll
    for (int round = 0; round < C::NCR; ++round) {
        const int task = round * C::NT + tid;
        if (task >= C::NCORE) break;
        const int u = task / C::NV4, v4 = task - u * C::NV4;
#pragma unroll
        for (int er = 0; er < 2; ++er) {
            const f4 v = f4{st.ll[round][er][0], st.ll[round][er][1], st.ll[round][er][2], st.ll[round][er][3]};
            *reinterpret_cast<f4 *>(sLo + (C::HALO + 2 * u + er) * C::W + C::PO + 4 * v4) = v;
            const int R = r1 + 2 * u + er, Cc = c1 + 4 * v4;
            if (p.LoLo && R < p.LR && Cc < p.LC)        // include_scale (LC is a multiple of 4 on this path)
                *reinterpret_cast<f4 *>(p.LoLo + ((int64_t)b * p.LR + R) * p.LC + Cc) = v;
        }
    }
#pragma unroll
    for (int r = 0; r < C::NHR; ++r) {
        const int t = r * C::NT + tid;
        if (t >= C::NHALO) break;
        int row, col;
        fwd12_halo_pos<C>(t, row, col);
        *reinterpret_cast<f4 *>(sLo + row * C::W + C::OL1 + col) = f4{st.halo[r][0], st.halo[r][1], st.halo[r][2], st.halo[r][3]};
    }
}

// true when part of the tile's LoLo1 window lies outside the (even-extended) image
template <class C>
DT_HD bool fwd12_needs_fix(const Fwd1Params &p, int r1, int c1) {
    return r1 - C::HALO < 0 || r1 + C::CR + C::HALO > p.LR || c1 - C::HALO < 0 || c1 + C::CC + C::HALO > p.LC;
}

// ---- phase 2c (border tiles): window positions outside the image <- their mirror images ----
template <class C>
DT_HD void fwd12_fix(const Fwd1Params &p, float *sLo, int tid, int r1, int c1) {
    const int rw = r1 - C::HALO, cw = c1 - C::HALO;
    for (int t = tid; t < C::NR1 * C::NC1; t += C::NT) {
        const int wr = t / C::NC1, wc = t - wr * C::NC1;
        const int lr = rw + wr, lc = cw + wc;
        if (lr >= 0 && lr < p.LR && lc >= 0 && lc < p.LC) continue;
        const int mr = clamp_i(reflect_i(lr, p.LR) - rw, 0, C::NR1 - 1);
        const int mc = clamp_i(reflect_i(lc, p.LC) - cw, 0, C::NC1 - 1);
        sLo[wr * C::W + C::OL1 + wc] = sLo[mr * C::W + C::OL1 + mc];
    }
}

// ---- phase 3: level-2 column pass (coldfilt pairs) from the LoLo1 plane in LDS -----------
template <class C>
DT_HD void fwd12_cols2(const Fwd2Params &p, const float *sLo, float *sLo2, float *sHi2, int tid) {
    for (int task = tid; task < C::NS2 * C::NC1; task += C::NT) {
        const int strip = task / C::NC1, cc = task - strip * C::NC1;
        float w[C::WN2];
        const float *src = sLo + (4 * C::PS * strip) * C::W + C::OL1 + cc;
#pragma unroll
        for (int j = 0; j < C::WN2; ++j) w[j] = src[j * C::W];
#pragma unroll
        for (int q = 0; q < C::PS; ++q) {
            float A, Bv;
            const int row = 2 * (strip * C::PS + q);
            dfilt_pair<C::M>(w + 4 * q, p.l_a, p.l_b, A, Bv);
            sLo2[row * C::NC1 + cc] = p.lo_a_first ? A : Bv;
            sLo2[(row + 1) * C::NC1 + cc] = p.lo_a_first ? Bv : A;
            dfilt_pair<C::M>(w + 4 * q, p.h_a, p.h_b, A, Bv);
            sHi2[row * C::NC1 + cc] = p.hi_a_first ? A : Bv;
            sHi2[(row + 1) * C::NC1 + cc] = p.hi_a_first ? Bv : A;
        }
    }
}

}  // namespace dt2d
