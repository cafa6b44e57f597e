// The tile programs the fused 2-D kernels (fused2d.hip) and the per-slice passes of the 3-D
// kernels (fused3d.hip) are built from; parameter blocks and shared arithmetic (dfilt_pair,
// ifilt4, store_record, inv2_rows) live in fused2d_tiles.hpp.
//
// Column passes read their sliding window straight from global memory (coalesced rows, many
// independent loads in flight per lane) instead of staging the input window in LDS first:
// one LDS plane and one workgroup barrier less per tile.  Earlier generations and the
// variants that were measured and dropped are not kept in the tree (their measurements are in
// profiles/r01 and profiles/r02; the code is in the history up to round 2, tools/kbench/tile_variants.hpp);
// tools/kbench/ko_bench.hip times the kernels below with their loads and stores knocked out.
#pragma once
#include "fused2d_tiles.hpp"

// 16-byte store of write-once data (subband records) with the non-temporal hint, so that the
// lowpass plane the next level reads back is what stays in L2 / the Infinity Cache
typedef float dt_v4f __attribute__((ext_vector_type(4)));
typedef float dt_v2f __attribute__((ext_vector_type(2)));
#define DT_STREAM_STORE_F4(dst_, val_) \
    __builtin_nontemporal_store(*reinterpret_cast<const dt_v4f *>(&(val_)), reinterpret_cast<dt_v4f *>(dst_))
// ... and 16-byte load of read-once data (the records on the way back)
#define DT_STREAM_LOAD_F4(src_) __builtin_nontemporal_load(reinterpret_cast<const dt_v4f *>(src_))

namespace dt2d {

// ======================================================================================
// Level 1 forward, direct column pass.
// ======================================================================================
// M2_ > 0: band-pass variant (6-vector biort): a third plane Ba = colfilter(X, h2) feeds the
// diagonal subbands through h2 along the rows as well (transform2d.py:116-129).
template <int TR_, int TC_, int RS_, int M0_, int M1_, int M2_ = 0>
struct Fwd1DCfg {
    static constexpr int TR = TR_, TC = TC_, RS = RS_, M0 = M0_, M1 = M1_, M2 = M2_;
    static constexpr bool BP = M2 > 0;
    static constexpr int H0 = M0 / 2, H1 = M1 / 2, H2 = M2 / 2, HH = cmax(cmax(H0, H1), H2);
    static constexpr int HC = (HH + 1) & ~1;
    static constexpr int W = TC + 2 * HC;
    static constexpr int NS = TR / RS;
    static constexpr int WN = RS + 2 * HH;            // register window per task
    static constexpr int SL = TR * W;
    static constexpr int LDS_FLOATS = (BP ? 3 : 2) * SL;
    static_assert(TR % RS == 0 && TR % 2 == 0 && TC % 2 == 0, "tile shape");
    static_assert(M0 % 2 == 1 && M1 % 2 == 1 && (!BP || M2 % 2 == 1), "biort filters must have odd length");
};

// sBa: third LDS plane of the band-pass variant (ignored otherwise)
template <class C>
DT_HD void fwd1d_cols(const Fwd1Params &p, float *sLo, float *sHi, int tid, int b, int r0, int c0,
                      float *sBa = nullptr) {
    const float *Xb = p.X + (int64_t)b * p.inR * p.inC;
    const int ro = r0 - C::HH, co = c0 - C::HC;
    const bool rows_in = ro >= 0 && ro + C::TR + 2 * C::HH <= p.inR;
    const bool interior = rows_in && co >= 0 && co + C::W <= p.inC;
    for (int task = tid; task < C::NS * C::W; task += DT_NT) {
        int strip = task / C::W, cc = task - strip * C::W;
        float w[C::WN];
        if (interior) {
            const float *src = Xb + (int64_t)(ro + strip * C::RS) * p.inC + (co + cc);
#pragma unroll
            for (int j = 0; j < C::WN; ++j) w[j] = src[(int64_t)j * p.inC];
        } else if (rows_in) {       // left / right edge tiles: the column is reflected once, the rows run straight
            int gc = reflect_i(co + cc, p.LC); if (gc > p.inC - 1) gc = p.inC - 1;
            const float *src = Xb + (int64_t)(ro + strip * C::RS) * p.inC + gc;
#pragma unroll
            for (int j = 0; j < C::WN; ++j) w[j] = src[(int64_t)j * p.inC];
        } else {
            int gc = reflect_i(co + cc, p.LC); if (gc > p.inC - 1) gc = p.inC - 1;
#pragma unroll
            for (int j = 0; j < C::WN; ++j) {
                int gr = reflect_i(ro + strip * C::RS + j, p.LR); if (gr > p.inR - 1) gr = p.inR - 1;
                w[j] = Xb[(int64_t)gr * p.inC + gc];
            }
        }
        const dt_pk2 *cp = reinterpret_cast<const dt_pk2 *>(p.c01);
#pragma unroll
        for (int q = 0; q < C::RS; ++q) {
            dt_pk2 a = {0.f, 0.f};                  // (lowpass, highpass) of row q: one packed chain
#pragma unroll
            for (int d = 0; d < 2 * C::HH + 1; ++d) a += cp[d] * w[q + d];
            sLo[(strip * C::RS + q) * C::W + cc] = a.x;
            sHi[(strip * C::RS + q) * C::W + cc] = a.y;
            if (C::BP) {
                float ba = 0.f;
#pragma unroll
                for (int k = 0; k < C::M2; ++k) ba += p.h2[k] * w[q + C::HH + C::H2 - k];
                sBa[(strip * C::RS + q) * C::W + cc] = ba;
            }
        }
    }
}

// ---- staged record stores ------------------------------------------------------------
// A lane-owned 48-byte record written as three 16-byte stores at a 48-byte lane stride
// reaches only ~4.3 TB/s on MI355X; the same bytes as consecutive 16-byte stores reach
// ~6.8 TB/s (tools/kbench).  So each wavefront bounces the records of its 64 tasks through
// a private 3 KiB LDS slab and writes them out as 3 x 1 KiB consecutive runs.
// The two halves are separate functions because the host emulator has to run them as two
// passes over the threads of a wave; on the device they are called back to back (LDS
// operations of one wavefront execute in order, no barrier needed).
constexpr int STAGE_FLOATS_PER_WAVE = 64 * 12;

template <class C>
DT_HD void fwd1s_rows_compute(const Fwd1Params &p, const float *sLo, const float *sHi, float *stage,
                              int tid, int base, int b, int r0, int c0, const float *sBa = nullptr) {
    constexpr int NV = C::TC / 2, NU = C::TR / 2;
    constexpr int WL = 2 * C::HC + 2;
    const int lane = tid & 63, wave = tid >> 6;
    float *slab = stage + wave * STAGE_FLOATS_PER_WAVE + lane * 12;
    const int task = base + tid;
    const int u = task / NV, v = task - u * NV;
    const int R = r0 + 2 * u, Cc = c0 + 2 * v;
    if (task >= NU * NV || R >= p.LR || Cc >= p.LC) return;      // slab entry unused
    float ll[2][2], hl[2][2], lh[2][2], hh[2][2];
#pragma unroll
    for (int er = 0; er < 2; ++er) {
        float wl[WL], wh[WL];
        const f2 *pl = reinterpret_cast<const f2 *>(sLo + (2 * u + er) * C::W + 2 * v);
        const f2 *ph = reinterpret_cast<const f2 *>(sHi + (2 * u + er) * C::W + 2 * v);
#pragma unroll
        for (int j = 0; j < WL / 2; ++j) {
            f2 a = pl[j], c = ph[j];
            wl[2 * j] = a.x; wl[2 * j + 1] = a.y;
            wh[2 * j] = c.x; wh[2 * j + 1] = c.y;
        }
        const dt_pk2 *cp = reinterpret_cast<const dt_pk2 *>(p.c01);
#pragma unroll
        for (int ec = 0; ec < 2; ++ec) {
            dt_pk2 al = {0.f, 0.f}, ah = {0.f, 0.f};       // (h0, h1) over the Lo row: (ll, lh); over the Hi row: (hl, hh)
#pragma unroll
            for (int d = 0; d < 2 * C::HH + 1; ++d) {
                al += cp[d] * wl[ec + C::HC - C::HH + d];
                ah += cp[d] * wh[ec + C::HC - C::HH + d];
            }
            ll[er][ec] = al.x; lh[er][ec] = al.y; hl[er][ec] = ah.x; hh[er][ec] = ah.y;   // hh: h2 below when BP
        }
        if (C::BP) {            // diagonal subbands: Ba rows through h2
            float wb[WL];
            const f2 *pb = reinterpret_cast<const f2 *>(sBa + (2 * u + er) * C::W + 2 * v);
#pragma unroll
            for (int j = 0; j < WL / 2; ++j) { f2 a = pb[j]; wb[2 * j] = a.x; wb[2 * j + 1] = a.y; }
#pragma unroll
            for (int ec = 0; ec < 2; ++ec) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < C::M2; ++k) s += p.h2[k] * wb[ec + C::HC + C::H2 - k];
                hh[er][ec] = s;
            }
        }
    }
    float *L = p.LoLo + ((int64_t)b * p.LR + R) * p.LC + Cc;
    *reinterpret_cast<f2 *>(L) = f2{ll[0][0], ll[0][1]};
    *reinterpret_cast<f2 *>(L + p.LC) = f2{ll[1][0], ll[1][1]};
    store_record(slab, hl, lh, hh);
}

// The 64 records a wavefront has staged are consecutive in tile order.  When the tile holds NVREC records per
// row and NVREC divides 64 they are 64 / NVREC whole rows, each NVREC * 48 contiguous bytes of Yh, so the
// address of 16-byte piece j is (row base, uniform per wavefront: scalar arithmetic) + 16 (j mod 3 NVREC):
// a handful of vector instructions per piece instead of the ~20 of the general index algebra below (piece ->
// record -> task -> (u, v) -> 64-bit address), which was a seventh of all vector instructions of k_fwd1.
//   row0: first record of the wavefront's first row; row_stride: floats between record rows;
//   rows_ok / recs_ok: rows / records per row of this wavefront that lie inside the image
template <int NVREC, bool STREAM>
DT_HD void flush_record_rows(float *row0, int row_stride, int rows_ok, int recs_ok, const f4 *slab, int lane) {
    constexpr int PPR = 3 * NVREC;              // 16-byte pieces per record row
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int j = lane + 64 * k;
        const int ul = j / PPR, off = j - ul * PPR;
        if (ul < rows_ok && off < 3 * recs_ok) {
            float *dst = row0 + ul * row_stride + 4 * off;
            if (STREAM) DT_STREAM_STORE_F4(dst, slab[j]);
            else *reinterpret_cast<f4 *>(dst) = slab[j];
        }
    }
}

#if defined(__HIP_DEVICE_COMPILE__)
#define DT_WAVE_UNIFORM_I(x) __builtin_amdgcn_readfirstlane(x)
#else
#define DT_WAVE_UNIFORM_I(x) (x)
#endif

template <class C>
DT_HD void fwd1s_rows_flush(const Fwd1Params &p, const float *stage, int tid, int base, int b, int r0,
                            int c0) {
    constexpr int NV = C::TC / 2, NU = C::TR / 2;
    const int HR = p.LR / 2, HCc = p.LC / 2;
    const int lane = tid & 63, wave = tid >> 6;
    const f4 *slab = reinterpret_cast<const f4 *>(stage + wave * STAGE_FLOATS_PER_WAVE);
    if (64 % NV == 0 && (int64_t)p.B * HR * HCc * 12 < ((int64_t)1 << 31)) {       // whole record rows per wavefront
        const int task0 = DT_WAVE_UNIFORM_I(base + wave * 64);
        const int u0 = task0 / NV;                           // first record row of this wavefront in the tile
        const int rows = (p.LR - r0) / 2 - u0, recs = (p.LC - c0) / 2;
        const int rows_ok = rows < 64 / NV ? rows : 64 / NV, rows_tile = NU - u0;
        float *row0 = p.Yh + (((int64_t)b * HR + r0 / 2 + u0) * HCc + c0 / 2) * 12;
        flush_record_rows<NV, true>(row0, HCc * 12, rows_ok < rows_tile ? rows_ok : rows_tile, recs < NV ? recs : NV, slab, lane);
        return;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int j = lane + 64 * k;              // 16-byte piece of the wave's 64 records
        int rr = j / 3, part = j - 3 * rr;
        int task = base + wave * 64 + rr;
        int u = task / NV, v = task - u * NV;
        int R = r0 + 2 * u, Cc = c0 + 2 * v;
        if (task < NU * NV && R < p.LR && Cc < p.LC) {
            float *rec = p.Yh + (((int64_t)b * HR + R / 2) * HCc + Cc / 2) * 12;
            DT_STREAM_STORE_F4(reinterpret_cast<f4 *>(rec) + part, slab[j]);
        }
    }
}

// ======================================================================================
// Level >= 2 forward, direct column pass + staged record stores.
// ======================================================================================
// BP_: band-pass variant (12-vector q-shift): third plane Ba = coldfilt(X, h2b, h2a) for the
// diagonal subbands (transform2d.py:145-155)
template <int TR_, int TC_, int PS_, int M_, bool BP_ = false>
struct Fwd2DCfg {
    static constexpr bool BP = BP_;
    static constexpr int TR = TR_, TC = TC_, PS = PS_, M = M_;     // TR x TC outputs of LoLo'
    static constexpr int TI = TR / 2, TJ = TC / 2;                 // (A,B) pairs per axis
    static constexpr int NS = TI / PS;                             // strips of PS pairs
    static constexpr int WN = 4 * PS + 2 * M - 4;                  // register window per task
    static constexpr int NCI = 2 * TC + 2 * M - 4;                 // input window cols (% 4 == 0)
    static constexpr int SL = TR * NCI;
    static constexpr int LDS_FLOATS = (BP ? 3 : 2) * SL;
    static_assert(M % 2 == 0 && TR % 2 == 0 && TC % 2 == 0 && TI % PS == 0, "even taps / tile");
};

template <class C>
DT_HD void fwd2d_cols(const Fwd2Params &p, float *sLo, float *sHi, int tid, int b, int r0, int c0,
                      float *sBa = nullptr) {
    const float *Xb = p.X + (int64_t)b * p.inR * p.inC;
    const int ro = 2 * r0 - C::M + 2, co = 2 * c0 - C::M + 2;      // logical origin of the window
    const int NRI = 2 * C::TR + 2 * C::M - 4;
    const bool rows_in = ro - p.padR >= 0 && ro + NRI - p.padR <= p.inR;
    const bool interior = rows_in && co - p.padC >= 0 && co + C::NCI - p.padC <= p.inC;
    for (int task = tid; task < C::NS * C::NCI; task += DT_NT) {
        int strip = task / C::NCI, cc = task - strip * C::NCI;
        float w[C::WN];
        const int rs = ro + 4 * C::PS * strip;
        if (interior) {
            const float *src = Xb + (int64_t)(rs - p.padR) * p.inC + (co + cc - p.padC);
#pragma unroll
            for (int j = 0; j < C::WN; ++j) w[j] = src[(int64_t)j * p.inC];
        } else if (rows_in) {       // left / right edge tiles: the column is reflected once, the rows run straight
            const int gc = clamp_i(reflect_i(co + cc, p.LC) - p.padC, 0, p.inC - 1);
            const float *src = Xb + (int64_t)(rs - p.padR) * p.inC + gc;
#pragma unroll
            for (int j = 0; j < C::WN; ++j) w[j] = src[(int64_t)j * p.inC];
        } else {
            int gc = clamp_i(reflect_i(co + cc, p.LC) - p.padC, 0, p.inC - 1);
#pragma unroll
            for (int j = 0; j < C::WN; ++j) {
                int gr = clamp_i(reflect_i(rs + j, p.LR) - p.padR, 0, p.inR - 1);
                w[j] = Xb[(int64_t)gr * p.inC + gc];
            }
        }
#pragma unroll
        for (int q = 0; q < C::PS; ++q) {
            float A, Bv, Ah, Bh;
            int row = 2 * (strip * C::PS + q);
            dfilt_pair2<C::M>(w + 4 * q, p.lh_a, p.lh_b, A, Bv, Ah, Bh);
            sLo[row * C::NCI + cc] = p.lo_a_first ? A : Bv;
            sLo[(row + 1) * C::NCI + cc] = p.lo_a_first ? Bv : A;
            sHi[row * C::NCI + cc] = p.hi_a_first ? Ah : Bh;
            sHi[(row + 1) * C::NCI + cc] = p.hi_a_first ? Bh : Ah;
            if (C::BP) {
                dfilt_pair<C::M>(w + 4 * q, p.b_a, p.b_b, A, Bv);
                sBa[row * C::NCI + cc] = p.bp_a_first ? A : Bv;
                sBa[(row + 1) * C::NCI + cc] = p.bp_a_first ? Bv : A;
            }
        }
    }
}

template <class C>
DT_HD void fwd2s_rows_compute(const Fwd2Params &p, const float *sLo, const float *sHi, float *stage,
                              int tid, int base, int b, int r0, int c0, const float *sBa = nullptr) {
    const int OR = p.LR / 2, OC = p.LC / 2;
    const int lane = tid & 63, wave = tid >> 6;
    float *slab = stage + wave * STAGE_FLOATS_PER_WAVE + lane * 12;
    const int task = base + tid;
    const int il = task / C::TJ, jl = task - il * C::TJ;
    const int R = r0 + 2 * il, Cc = c0 + 2 * jl;
    if (task >= C::TI * C::TJ || R >= OR || Cc >= OC) return;
    float ll[2][2], hl[2][2], lh[2][2], hh[2][2];
#pragma unroll
    for (int er = 0; er < 2; ++er) {
        float wl[2 * C::M], wh[2 * C::M];
        const f4 *pl = reinterpret_cast<const f4 *>(sLo + (2 * il + er) * C::NCI + 4 * jl);
        const f4 *ph = reinterpret_cast<const f4 *>(sHi + (2 * il + er) * C::NCI + 4 * jl);
#pragma unroll
        for (int j = 0; j < C::M / 2; ++j) {
            f4 a = pl[j], c = ph[j];
            wl[4 * j] = a.x; wl[4 * j + 1] = a.y; wl[4 * j + 2] = a.z; wl[4 * j + 3] = a.w;
            wh[4 * j] = c.x; wh[4 * j + 1] = c.y; wh[4 * j + 2] = c.z; wh[4 * j + 3] = c.w;
        }
        float A, Bv, Ah, Bh;
        dfilt_pair2<C::M>(wl, p.lh_a, p.lh_b, A, Bv, Ah, Bh);
        ll[er][0] = p.lo_a_first ? A : Bv; ll[er][1] = p.lo_a_first ? Bv : A;
        lh[er][0] = p.hi_a_first ? Ah : Bh; lh[er][1] = p.hi_a_first ? Bh : Ah;
        if (C::BP) {
            dfilt_pair<C::M>(wh, p.l_a, p.l_b, A, Bv);
            hl[er][0] = p.lo_a_first ? A : Bv; hl[er][1] = p.lo_a_first ? Bv : A;            // diagonal subbands: Ba rows through the band-pass pair
            const f4 *pb = reinterpret_cast<const f4 *>(sBa + (2 * il + er) * C::NCI + 4 * jl);
#pragma unroll
            for (int j = 0; j < C::M / 2; ++j) {
                f4 a = pb[j];
                wh[4 * j] = a.x; wh[4 * j + 1] = a.y; wh[4 * j + 2] = a.z; wh[4 * j + 3] = a.w;
            }
            dfilt_pair<C::M>(wh, p.b_a, p.b_b, A, Bv);
            hh[er][0] = p.bp_a_first ? A : Bv; hh[er][1] = p.bp_a_first ? Bv : A;
        } else {
            dfilt_pair2<C::M>(wh, p.lh_a, p.lh_b, A, Bv, Ah, Bh);
            hl[er][0] = p.lo_a_first ? A : Bv; hl[er][1] = p.lo_a_first ? Bv : A;
            hh[er][0] = p.hi_a_first ? Ah : Bh; hh[er][1] = p.hi_a_first ? Bh : Ah;
        }
    }
    float *L = p.LoLo + ((int64_t)b * OR + R) * OC + Cc;
    *reinterpret_cast<f2 *>(L) = f2{ll[0][0], ll[0][1]};
    *reinterpret_cast<f2 *>(L + OC) = f2{ll[1][0], ll[1][1]};
    store_record(slab, hl, lh, hh);
}

template <class C>
DT_HD void fwd2s_rows_flush(const Fwd2Params &p, const float *stage, int tid, int base, int b, int r0,
                            int c0) {
    const int OR = p.LR / 2, OC = p.LC / 2;
    const int HR = OR / 2, HCc = OC / 2;
    const int lane = tid & 63, wave = tid >> 6;
    const f4 *slab = reinterpret_cast<const f4 *>(stage + wave * STAGE_FLOATS_PER_WAVE);
    if (64 % C::TJ == 0) {          // whole record rows per wavefront: see flush_record_rows
        const int task0 = DT_WAVE_UNIFORM_I(base + wave * 64);
        const int i0 = task0 / C::TJ;
        const int rows = (OR - r0) / 2 - i0, recs = (OC - c0) / 2, rows_tile = C::TI - i0;
        int rows_ok = rows < 64 / C::TJ ? rows : 64 / C::TJ;
        if (rows_tile < rows_ok) rows_ok = rows_tile;
        float *row0 = p.Yh + (((int64_t)b * HR + r0 / 2 + i0) * HCc + c0 / 2) * 12;
        if (p.stream_records) flush_record_rows<C::TJ, true>(row0, HCc * 12, rows_ok, recs < C::TJ ? recs : C::TJ, slab, lane);
        else flush_record_rows<C::TJ, false>(row0, HCc * 12, rows_ok, recs < C::TJ ? recs : C::TJ, slab, lane);
        return;
    }
    // record rows of TJ records, 64 records of the wavefront spread over several of them: row and column of a
    // record relative to the wavefront's first one, 32-bit offsets from a base that is uniform per wavefront
    const int task0 = DT_WAVE_UNIFORM_I(base + wave * 64);
    const int i0 = task0 / C::TJ, j0 = task0 - i0 * C::TJ;
    const int rows = (OR - r0) / 2 - i0, recs = (OC - c0) / 2;
    float *row0 = p.Yh + (((int64_t)b * HR + r0 / 2 + i0) * HCc + c0 / 2) * 12;
    const int rstride = HCc * 12;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int j = lane + 64 * k;
        const int rr = j / 3, part = j - 3 * rr;
        const int t = j0 + rr, il = t / C::TJ, jl = t - il * C::TJ;
        if (task0 + rr < C::TI * C::TJ && il < rows && jl < recs) {
            float *dst = row0 + il * rstride + jl * 12 + part * 4;
            if (p.stream_records) DT_STREAM_STORE_F4(dst, slab[j]);
            else *reinterpret_cast<f4 *>(dst) = slab[j];
        }
    }
}

// ======================================================================================
// Inverse
// ======================================================================================
// level-1 inverse row pass: y1 (*) g0 + y2 (*) g1 along the columns, 16-byte stores
template <class C>
DT_HD void inv1d_rows(const Inv1Params &p, const float *y1, const float *y2, int tid, int b, int r0,
                      int c0, const float *y3 = nullptr) {
    constexpr int NQ = C::TC / 4;
    constexpr int WL = 4 + 2 * C::HE;
    static_assert(NQ <= 32 && WL % 4 == 0 && C::NC % 4 == 0 && C::TR % (DT_NT / 32) == 0, "row-pass task grid");
    // 32 task slots to a row (NQ of them used): row and column group of a task are a shift and a mask of the
    // thread index instead of a division by NQ = 30, 31, 27; the windows are read 16 bytes at a time
    const int q = tid & 31;
#pragma unroll
    for (int round = 0; round < C::TR / (DT_NT / 32); ++round) {
        const int r = (tid >> 5) + round * (DT_NT / 32);
        int R = r0 + r, Cc = c0 + 4 * q;
        if (q >= NQ || R >= p.R || Cc >= p.C) continue;
        // y1 and y2 are interleaved in LDS ((y1, y2) pairs, written by inv1r_fir): a 16-byte read is two window
        // positions ready as packed operands, and (g0 over y1, g1 over y2) of an output is one packed chain
        dt_pk2 wab[WL];
        const f4 *pa = reinterpret_cast<const f4 *>(y1 + 2 * (r * C::NC + 4 * q));
#pragma unroll
        for (int j = 0; j < WL / 2; ++j) {
            f4 a = pa[j];
            wab[2 * j] = dt_pk2{a.x, a.y}; wab[2 * j + 1] = dt_pk2{a.z, a.w};
        }
        const dt_pk2 *gp = reinterpret_cast<const dt_pk2 *>(p.g01);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dt_pk2 s2 = {0.f, 0.f};
#pragma unroll
            for (int d = 0; d < 2 * C::HH + 1; ++d) s2 += gp[d] * wab[e + C::HE - C::HH + d];
            o[e] = s2.x + s2.y;
        }
        if (C::BP) {
            float wa[WL];
            const f4 *pc = reinterpret_cast<const f4 *>(y3 + r * C::NC + 4 * q);
#pragma unroll
            for (int j = 0; j < WL / 4; ++j) { f4 a = pc[j]; wa[4 * j] = a.x; wa[4 * j + 1] = a.y; wa[4 * j + 2] = a.z; wa[4 * j + 3] = a.w; }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < C::M2; ++k) s += p.g2[k] * wa[e + C::HE + C::H2 - k];
                o[e] += s;
            }
        }
        float *X = p.X + ((int64_t)b * p.R + R) * p.C + Cc;
        if (Cc + 3 < p.C && (p.C & 3) == 0) {
            const f4 ov = f4{o[0], o[1], o[2], o[3]};      // the reconstruction: written once, not read again here
            DT_STREAM_STORE_F4(X, ov);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (Cc + e < p.C) X[e] = o[e];
        }
    }
}

// ======================================================================================
// Inverse, third generation: raw records in LDS, c2q folded into the column pass.
// ======================================================================================
// PMC profiles (profiles/ round 1) show the inverse kernels VALU-bound, half of the
// instructions in the record -> quad-plane expansion (index math per quad).  Here the
// window's records are copied verbatim into LDS (coalesced 16-byte pieces, no arithmetic),
// and each column-pass task computes the samples it needs from them: a task owns one
// column (parity e) and a strip of rows, so per record row it needs exactly the "top" and
// "bottom" sample of each of the three planes = 2 multiplies + 2 FMAs per plane.  Tasks
// are dealt so that the column parity is uniform per wavefront (no divergence, no
// selects); border tiles, where reflection can flip a quad, take a generic path.
// M2_ > 0: band-pass variant: the diagonal plane goes through g2 into a third plane y3, which
// the row pass filters with g2 as well (transform2d.py:283-291).
template <int TR_, int TC_, int RS_, int M0_, int M1_, int M2_ = 0>
struct Inv1RCfg {
    static constexpr int TR = TR_, TC = TC_, RS = RS_, M0 = M0_, M1 = M1_, M2 = M2_;
    static constexpr bool BP = M2 > 0;
    static constexpr int NY = BP ? 3 : 2;             // column-pass planes
    static constexpr int H0 = M0 / 2, H1 = M1 / 2, H2 = M2 / 2, HH = cmax(cmax(H0, H1), H2);
    static constexpr int HE = (HH + 1) & ~1;
    static constexpr int NR = TR + 2 * HE, NC = TC + 2 * HE;
    static constexpr int QR = NR / 2, QC = NC / 2, NREC = QR * QC;
    static constexpr int NS = TR / RS;
    static constexpr int WN = RS + 2 * HE;
    static constexpr int SREC = NREC * 12;
    static constexpr int SY = TR * NC;
    static constexpr int LDS_FLOATS = SREC + NY * SY;
    static constexpr int LDS_ALIASED = SREC > NY * SY ? SREC : NY * SY;   // y planes over the records
    static_assert(TR % RS == 0 && RS % 2 == 0 && TC % 4 == 0, "tile shape");
    static_assert(NS * QC <= 128, "column-pass tasks: two wavefronts per column parity");
    static_assert(M0 % 2 == 1 && M1 % 2 == 1 && (!BP || M2 % 2 == 1), "biort filters must have odd length");
};

// stage the window's records verbatim: srec[uw][vw][12], source record reflected.  All of a
// thread's 16-byte pieces are requested before the first one is written to LDS (one memory
// latency per tile, not one per piece).
template <int QR, int QC, bool STREAM = false>
DT_HD void inv_rec_stage(const float *Yhb, int zr, int zc, float *srec, int ro, int co, int tid) {
    constexpr int NPIECE = 3 * QR * QC, NP = (NPIECE + DT_NT - 1) / DT_NT;
    const int hc = zc / 2;
    const bool interior = ro >= 0 && ro + 2 * QR <= zr && co >= 0 && co + 2 * QC <= zc;
    float px[NP], py[NP], pz[NP], pw[NP];       // scalar arrays: f4 arrays end up in scratch
#ifdef DT_INV_STAGE_GENERIC                     /* A/B builds only: the general index algebra on every tile */
    constexpr bool fast_rows = false;
#else
    constexpr bool fast_rows = true;
#endif
    if (interior && fast_rows) {
        // interior tiles: a row of the window is 3 QC consecutive 16-byte pieces of Yh, so piece -> address is
        // (row, offset in row) on a base that is uniform per workgroup -- the record / part / (uw, vw) / 64-bit
        // index algebra of the general path below cost ~18 vector instructions per piece, 9 pieces per thread
        constexpr int PPR = 3 * QC;
        const float *base = Yhb + ((int64_t)(ro >> 1) * hc + (co >> 1)) * 12;       // ro, co are even
        const int rstride = hc * 12;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            int piece = tid + k * DT_NT;
            if (NP * DT_NT > NPIECE && piece >= NPIECE) piece = NPIECE - 1;
            const int uw = piece / PPR, off = piece - uw * PPR;
            const f4 *src = reinterpret_cast<const f4 *>(base + uw * rstride + 4 * off);
            if (STREAM) {
                const dt_v4f t = __builtin_nontemporal_load(reinterpret_cast<const dt_v4f *>(src));
                px[k] = t.x; py[k] = t.y; pz[k] = t.z; pw[k] = t.w;
            } else {
                const dt_v4f t = *reinterpret_cast<const dt_v4f *>(src);     // vector-typed: stays ONE 16-byte load
                px[k] = t.x; py[k] = t.y; pz[k] = t.z; pw[k] = t.w;
            }
        }
    } else if (ro >= 0 && ro + 2 * QR <= zr && fast_rows) {
        // left / right edge tiles: rows run straight, only the record column is reflected
        constexpr int PPR = 3 * QC;
        const float *base = Yhb + (int64_t)(ro >> 1) * hc * 12;
        const int rstride = hc * 12;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            int piece = tid + k * DT_NT;
            if (NP * DT_NT > NPIECE && piece >= NPIECE) piece = NPIECE - 1;
            const int uw = piece / PPR, off = piece - uw * PPR;
            const int vw = off / 3, part = off - 3 * vw;
            const int vc = reflect_i(co + 2 * vw, zc);
            const f4 *src = reinterpret_cast<const f4 *>(base + uw * rstride + (vc >> 1) * 12 + 4 * part);
            const dt_v4f t = *reinterpret_cast<const dt_v4f *>(src);
            px[k] = t.x; py[k] = t.y; pz[k] = t.z; pw[k] = t.w;
        }
    } else
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        int piece = tid + k * DT_NT;
        if (NP * DT_NT > NPIECE && piece >= NPIECE) piece = NPIECE - 1;      // clamp: no branch
        int rec = piece / 3, part = piece - 3 * rec;
        int uw = rec / QC, vw = rec - uw * QC;
        int ur = reflect_i(ro + 2 * uw, zr), vc = reflect_i(co + 2 * vw, zc);
        const f4 *src = reinterpret_cast<const f4 *>(Yhb + ((int64_t)(ur >> 1) * hc + (vc >> 1)) * 12) + part;
        if (STREAM) {
            const dt_v4f t = __builtin_nontemporal_load(reinterpret_cast<const dt_v4f *>(src));
            px[k] = t.x; py[k] = t.y; pz[k] = t.z; pw[k] = t.w;
        } else {
            const f4 t = *src;
            px[k] = t.x; py[k] = t.y; pz[k] = t.z; pw[k] = t.w;
        }
    }
    f4 *dst = reinterpret_cast<f4 *>(srec);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        int piece = tid + k * DT_NT;
        if (NP * DT_NT == NPIECE || piece < NPIECE) dst[piece] = f4{px[k], py[k], pz[k], pw[k]};
    }
}

// column-pass task of a thread: parity uniform per wavefront
struct ColTask { int strip, e, i, valid; };
template <class C>
DT_HD ColTask inv_col_task(int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    ColTask t;
    t.e = wave & 1;
    int k = (wave >> 1) * 64 + lane;            // slot among the tasks of this parity
    t.strip = k / C::QC;
    t.i = k - t.strip * C::QC;
    t.valid = k < C::NS * C::QC;
    return t;
}

// Linear variant for plane inputs (3-D pass B): no records, so no reason to split the
// columns by parity -- consecutive lanes take consecutive columns (coalesced plane reads).
template <class C>
DT_HD ColTask inv_col_task_lin(int tid) {
    ColTask t;
    t.strip = tid / C::NC;
    const int cc = tid - t.strip * C::NC;
    t.e = cc & 1; t.i = cc >> 1;
    t.valid = tid < C::NS * C::NC;
    return t;
}

// Samples (top, bottom) of the three planes at column parity e from one raw record
// (A.4 with gains folded in): plane "lh" = subbands (0,5), "hl" = (2,3), "hh" = (1,4).
template <int E>
DT_HD void rec_samples_t(const float *rec, const float *g, float (&top)[3], float (&bot)[3]) {
    const f4 r0 = reinterpret_cast<const f4 *>(rec)[0];
    const f4 r1 = reinterpret_cast<const f4 *>(rec)[1];
    const f4 r2 = reinterpret_cast<const f4 *>(rec)[2];
    if (E == 0) {        // a = Re(g0 w0 + g1 w1), c = Im(g0 w0 - g1 w1)
        top[0] = r0.x * g[0] + r2.z * g[5]; bot[0] = r0.y * g[0] - r2.w * g[5];
        top[1] = r1.x * g[2] + r1.z * g[3]; bot[1] = r1.y * g[2] - r1.w * g[3];
        top[2] = r0.z * g[1] + r2.x * g[4]; bot[2] = r0.w * g[1] - r2.y * g[4];
    } else {             // b = Im(g0 w0 + g1 w1), d = -Re(g0 w0 - g1 w1)
        top[0] = r0.y * g[0] + r2.w * g[5]; bot[0] = r2.z * g[5] - r0.x * g[0];
        top[1] = r1.y * g[2] + r1.w * g[3]; bot[1] = r1.z * g[3] - r1.x * g[2];
        top[2] = r0.w * g[1] + r2.y * g[4]; bot[2] = r2.x * g[4] - r0.z * g[1];
    }
}
// runtime-parity form (border tiles: reflection can flip the column parity per lane);
// branch-free: both parities are cheap enough to compute and select
DT_HD void rec_samples(const float *rec, const float *g, int e, float (&top)[3], float (&bot)[3]) {
    float t0[3], b0[3], t1[3], b1[3];
    rec_samples_t<0>(rec, g, t0, b0);
    rec_samples_t<1>(rec, g, t1, b1);
#pragma unroll
    for (int k = 0; k < 3; ++k) { top[k] = e ? t1[k] : t0[k]; bot[k] = e ? b1[k] : b0[k]; }
}

// The column parity of a thread's task is uniform per wavefront (inv_col_task): tell the
// compiler (scalar branch, straight-line bodies, batched LDS reads).
#if defined(__HIP_DEVICE_COMPILE__)
#define DT_WAVE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#else
#define DT_WAVE_UNIFORM(x) (x)
#endif

template <class C, bool LIN = false>
DT_HD void inv1r_fetch_from(const Inv1Params &p, const float *Z, float (&w0)[C::WN], int tid, int b, int r0,
                            int c0) {
    const ColTask t = LIN ? inv_col_task_lin<C>(tid) : inv_col_task<C>(tid);
    if (!t.valid) return;
    const float *Zb = Z + (int64_t)b * p.R * p.C;
    const int ro = r0 - C::HE, co = c0 - C::HE;
    const bool rows_in = ro >= 0 && ro + C::NR <= p.R;
    const bool interior = rows_in && co >= 0 && co + C::NC <= p.C;
    const int cc = 2 * t.i + t.e;
    if (interior) {
        const float *src = Zb + (int64_t)(ro + t.strip * C::RS) * p.C + (co + cc);
#pragma unroll
        for (int j = 0; j < C::WN; ++j) w0[j] = src[(int64_t)j * p.C];
    } else if (rows_in) {           // left / right edge tiles: the column is reflected once, the rows run straight
        const float *src = Zb + (int64_t)(ro + t.strip * C::RS) * p.C + reflect_i(co + cc, p.C);
#pragma unroll
        for (int j = 0; j < C::WN; ++j) w0[j] = src[(int64_t)j * p.C];
    } else {
        int gc = reflect_i(co + cc, p.C);
#pragma unroll
        for (int j = 0; j < C::WN; ++j)
            w0[j] = Zb[(int64_t)reflect_i(ro + t.strip * C::RS + j, p.R) * p.C + gc];
    }
}

template <class C>
DT_HD void inv1r_fetch(const Inv1Params &p, float (&w0)[C::WN], int tid, int b, int r0, int c0) {
    inv1r_fetch_from<C>(p, p.Z, w0, tid, b, r0, c0);
}

// Two-step form of the column pass: (1) gather the three quad-plane windows from the raw
// records into registers, (2) after a workgroup barrier, filter and write y1/y2 -- which may
// then ALIAS the record buffer (LDS per workgroup = max(records, y planes): 8 workgroups
// per CU instead of 6).
template <class C, int E>
DT_HD void inv1r_gather_e(const Inv1Params &p, const float *srec, float (&w1)[C::WN], float (&w2)[C::WN],
                          float (&w3)[C::WN], int tid, int r0, int c0) {
    const ColTask t = inv_col_task<C>(tid);
    if (!t.valid) return;
    const int ro = r0 - C::HE, co = c0 - C::HE;
    const bool interior = ro >= 0 && ro + C::NR <= p.R && co >= 0 && co + C::NC <= p.C;
    const float *rbase = srec + ((t.strip * C::RS / 2) * C::QC + t.i) * 12;
    if (interior) {
#pragma unroll
        for (int ru = 0; ru < C::WN / 2; ++ru) {
            float top[3], bot[3];
            rec_samples_t<E>(rbase + ru * C::QC * 12, p.g, top, bot);
            w1[2 * ru] = top[0]; w1[2 * ru + 1] = bot[0];
            w2[2 * ru] = top[1]; w2[2 * ru + 1] = bot[1];
            w3[2 * ru] = top[2]; w3[2 * ru + 1] = bot[2];
        }
    } else {
        const int fc = reflect_i(co + 2 * t.i, p.C) & 1;
#pragma unroll
        for (int ru = 0; ru < C::WN / 2; ++ru) {
            const int fr = reflect_i(ro + t.strip * C::RS + 2 * ru, p.R) & 1;
            float top[3], bot[3];
            rec_samples(rbase + ru * C::QC * 12, p.g, t.e ^ fc, top, bot);
            w1[2 * ru] = fr ? bot[0] : top[0]; w1[2 * ru + 1] = fr ? top[0] : bot[0];
            w2[2 * ru] = fr ? bot[1] : top[1]; w2[2 * ru + 1] = fr ? top[1] : bot[1];
            w3[2 * ru] = fr ? bot[2] : top[2]; w3[2 * ru + 1] = fr ? top[2] : bot[2];
        }
    }
}
template <class C>
DT_HD void inv1r_gather(const Inv1Params &p, const float *srec, float (&w1)[C::WN], float (&w2)[C::WN],
                        float (&w3)[C::WN], int tid, int r0, int c0) {
    if (DT_WAVE_UNIFORM((tid >> 6) & 1)) inv1r_gather_e<C, 1>(p, srec, w1, w2, w3, tid, r0, c0);
    else inv1r_gather_e<C, 0>(p, srec, w1, w2, w3, tid, r0, c0);
}
template <class C, bool LIN = false>
DT_HD void inv1r_fir(const Inv1Params &p, const float (&w0)[C::WN], const float (&w1)[C::WN],
                     const float (&w2)[C::WN], const float (&w3)[C::WN], float *y1, float *y2, int tid,
                     float *y3 = nullptr) {
    const ColTask t = LIN ? inv_col_task_lin<C>(tid) : inv_col_task<C>(tid);
    if (!t.valid) return;
    const int cc = 2 * t.i + t.e;
    // (y1, y2) of a row as one packed chain: g0 over (w0, w2), g1 over (w1, w3); the pair goes to LDS interleaved
    // (one 8-byte write; y2 is not a plane of its own any more: y1 holds [TR][NC] pairs = the same 2 SY floats)
    dt_pk2 u[C::WN], v[C::WN];
#pragma unroll
    for (int j = 0; j < C::WN; ++j) { u[j] = dt_pk2{w0[j], w2[j]}; v[j] = dt_pk2{w1[j], C::BP ? 0.f : w3[j]}; }
    dt_pk2 *y12 = reinterpret_cast<dt_pk2 *>(y1);
#pragma unroll
    for (int q = 0; q < C::RS; ++q) {
        dt_pk2 a = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < C::M0; ++k) a += p.g0[k] * u[q + C::HE + C::H0 - k];
#pragma unroll
        for (int k = 0; k < C::M1; ++k) a += p.g1[k] * v[q + C::HE + C::H1 - k];
        y12[(t.strip * C::RS + q) * C::NC + cc] = a;
        if (C::BP) {
            float c = 0.f;
#pragma unroll
            for (int k = 0; k < C::M2; ++k) c += p.g2[k] * w3[q + C::HE + C::H2 - k];
            y3[(t.strip * C::RS + q) * C::NC + cc] = c;
        }
    }
}

// ---- level >= 2 inverse with raw records (c2q folded into the column pass) --------------
template <int TR_, int TC_, int JS_, int M_, bool BP_ = false>
struct Inv2RCfg {
    static constexpr bool BP = BP_;                                // band-pass q-shift (12 vectors)
    static constexpr int NY = BP ? 3 : 2;
    static constexpr int TR = TR_, TC = TC_, JS = JS_, M = M_;     // TR x TC INPUT samples per tile
    static constexpr int M2 = M / 2;
    static constexpr bool ODD = (M2 % 2) == 1;
    static constexpr int WN = ODD ? M : M + 2;
    static constexpr int ORG = ODD ? 1 - M2 : -M2;
    static constexpr int NR = TR + WN - 2, NC = TC + WN - 2;
    static constexpr int QR = NR / 2, QC = NC / 2, NREC = QR * QC;
    static constexpr int NJ = TR / 2;
    static constexpr int NS = NJ / JS;
    static constexpr int WS = 2 * JS + WN - 2;
    static constexpr int RS = 2 * JS;                              // window rows advanced per strip
    static constexpr int SREC = NREC * 12;
    static constexpr int SY = 2 * TR * NC;
    static constexpr int LDS_FLOATS = SREC + NY * SY;
    static constexpr int LDS_ALIASED = SREC > NY * SY ? SREC : NY * SY;
    static_assert(M % 2 == 0 && TR % 2 == 0 && TC % 2 == 0 && NJ % JS == 0, "even taps / tile");
    static_assert(NS * QC <= 128, "column-pass tasks: two wavefronts per column parity");
};

template <class C, bool LIN = false>
DT_HD void inv2r_fetch_from(const Inv2Params &p, const float *Z, float (&w0)[C::WS], int tid, int b, int r0,
                            int c0) {
    const ColTask t = LIN ? inv_col_task_lin<C>(tid) : inv_col_task<C>(tid);
    if (!t.valid) return;
    const float *Zb = Z + (int64_t)b * p.zr * p.zc;
    const int ro = r0 + C::ORG, co = c0 + C::ORG;
    const bool rows_in = ro >= 0 && ro + C::NR <= p.zr;
    const bool interior = rows_in && co >= 0 && co + C::NC <= p.zc;
    const int cc = 2 * t.i + t.e, rs = C::RS * t.strip;
    if (interior) {
        const float *src = Zb + (int64_t)(ro + rs) * p.zc + (co + cc);
#pragma unroll
        for (int j = 0; j < C::WS; ++j) w0[j] = src[(int64_t)j * p.zc];
    } else if (rows_in) {           // left / right edge tiles: the column is reflected once, the rows run straight
        const float *src = Zb + (int64_t)(ro + rs) * p.zc + reflect_i(co + cc, p.zc);
#pragma unroll
        for (int j = 0; j < C::WS; ++j) w0[j] = src[(int64_t)j * p.zc];
    } else {
        int gc = reflect_i(co + cc, p.zc);
#pragma unroll
        for (int j = 0; j < C::WS; ++j) w0[j] = Zb[(int64_t)reflect_i(ro + rs + j, p.zr) * p.zc + gc];
    }
}

template <class C>
DT_HD void inv2r_fetch(const Inv2Params &p, float (&w0)[C::WS], int tid, int b, int r0, int c0) {
    inv2r_fetch_from<C>(p, p.Z, w0, tid, b, r0, c0);
}

// two-step (gather / barrier / filter) form of the level >= 2 inverse column pass, so that
// y1/y2 may alias the record buffer (see inv1r_gather)
template <class C, int E>
DT_HD void inv2r_gather_e(const Inv2Params &p, const float *srec, float (&w1)[C::WS], float (&w2)[C::WS],
                          float (&w3)[C::WS], int tid, int r0, int c0) {
    const ColTask t = inv_col_task<C>(tid);
    if (!t.valid) return;
    const int ro = r0 + C::ORG, co = c0 + C::ORG;
    const bool interior = ro >= 0 && ro + C::NR <= p.zr && co >= 0 && co + C::NC <= p.zc;
    const int rs = C::RS * t.strip;
    const float *rbase = srec + ((rs / 2) * C::QC + t.i) * 12;
    if (interior) {
#pragma unroll
        for (int ru = 0; ru < C::WS / 2; ++ru) {
            float top[3], bot[3];
            rec_samples_t<E>(rbase + ru * C::QC * 12, p.g, top, bot);
            w1[2 * ru] = top[0]; w1[2 * ru + 1] = bot[0];
            w2[2 * ru] = top[1]; w2[2 * ru + 1] = bot[1];
            w3[2 * ru] = top[2]; w3[2 * ru + 1] = bot[2];
        }
    } else {
        const int fc = reflect_i(co + 2 * t.i, p.zc) & 1;
#pragma unroll
        for (int ru = 0; ru < C::WS / 2; ++ru) {
            const int fr = reflect_i(ro + rs + 2 * ru, p.zr) & 1;
            float top[3], bot[3];
            rec_samples(rbase + ru * C::QC * 12, p.g, t.e ^ fc, top, bot);
            w1[2 * ru] = fr ? bot[0] : top[0]; w1[2 * ru + 1] = fr ? top[0] : bot[0];
            w2[2 * ru] = fr ? bot[1] : top[1]; w2[2 * ru + 1] = fr ? top[1] : bot[1];
            w3[2 * ru] = fr ? bot[2] : top[2]; w3[2 * ru + 1] = fr ? top[2] : bot[2];
        }
    }
}
template <class C>
DT_HD void inv2r_gather(const Inv2Params &p, const float *srec, float (&w1)[C::WS], float (&w2)[C::WS],
                        float (&w3)[C::WS], int tid, int r0, int c0) {
    if (DT_WAVE_UNIFORM((tid >> 6) & 1)) inv2r_gather_e<C, 1>(p, srec, w1, w2, w3, tid, r0, c0);
    else inv2r_gather_e<C, 0>(p, srec, w1, w2, w3, tid, r0, c0);
}
template <class C, bool LIN = false, bool STD = false>
DT_HD void inv2r_fir(const Inv2Params &p, const float (&w0)[C::WS], const float (&w1)[C::WS],
                     const float (&w2)[C::WS], const float (&w3)[C::WS], float *y1, float *y2, int tid,
                     float *y3 = nullptr) {
    const ColTask t = LIN ? inv_col_task_lin<C>(tid) : inv_col_task<C>(tid);
    if (!t.valid) return;
    constexpr int LP = STD ? 1 : -1, HP = STD ? 0 : -1;
    const int cc = 2 * t.i + t.e;
#pragma unroll
    for (int q = 0; q < C::JS; ++q) {
        float *o1 = y1 + 4 * (t.strip * C::JS + q) * C::NC + cc, *o2 = y2 + 4 * (t.strip * C::JS + q) * C::NC + cc;
        dt_pk2 P = {0.f, 0.f}, Q = {0.f, 0.f};
        ifilt4_acc<C, LP>(w0 + 2 * q, p.l_a, p.l_b, p.lo_pos, P, Q);
        ifilt4_acc<C, HP>(w1 + 2 * q, p.h_a, p.h_b, p.hi_pos, P, Q);
        o1[0] = P.x; o1[C::NC] = Q.x; o1[2 * C::NC] = P.y; o1[3 * C::NC] = Q.y;
        P = dt_pk2{0.f, 0.f}; Q = dt_pk2{0.f, 0.f};
        ifilt4_acc<C, LP>(w2 + 2 * q, p.l_a, p.l_b, p.lo_pos, P, Q);
        if (C::BP) {
            o2[0] = P.x; o2[C::NC] = Q.x; o2[2 * C::NC] = P.y; o2[3 * C::NC] = Q.y;
            float *o3 = y3 + 4 * (t.strip * C::JS + q) * C::NC + cc;
            P = dt_pk2{0.f, 0.f}; Q = dt_pk2{0.f, 0.f};
            ifilt4_acc<C, HP>(w3 + 2 * q, p.b_a, p.b_b, p.bp_pos, P, Q);
            o3[0] = P.x; o3[C::NC] = Q.x; o3[2 * C::NC] = P.y; o3[3 * C::NC] = Q.y;
        } else {
            ifilt4_acc<C, HP>(w3 + 2 * q, p.h_a, p.h_b, p.hi_pos, P, Q);
            o2[0] = P.x; o2[C::NC] = Q.x; o2[2 * C::NC] = P.y; o2[3 * C::NC] = Q.y;
        }
    }
}

// ---- level >= 2 inverse, column phase in two halves (k_inv2s) -------------------------------------------------------
// y1 = colifilt(Z) + colifilt(lh) and y2 = colifilt(hl) + colifilt(hh) need their four windows one pair at a time.
// Gathering lh and hh first (both live in pieces 0 and 2 of a record), filtering y1 into a plane of its own, THEN
// gathering hl (piece 1) and filtering y2 over the record buffer keeps 36 instead of 48 window registers alive and lets
// the compiler stop hoarding LDS reads: 72 instead of 112 VGPRs at the packed-FMA instruction count, and with records +
// one y plane = 26 KB of LDS, six workgroups per CU instead of four.  (39 % of a k_inv2 workgroup's life is the wait for
// its window, and four workgroups do not cover it: profiles/r03/inv2_phase_stamps.txt.)  Same arithmetic in the same
// order as inv2r_gather + inv2r_fir: bit-identical results.
template <int E>
DT_HD void rec_samples13_t(const float *rec, const float *g, float &t1, float &b1, float &t3, float &b3) {
    const f4 r0 = reinterpret_cast<const f4 *>(rec)[0];
    const f4 r2 = reinterpret_cast<const f4 *>(rec)[2];
    if (E == 0) {
        t1 = r0.x * g[0] + r2.z * g[5]; b1 = r0.y * g[0] - r2.w * g[5];
        t3 = r0.z * g[1] + r2.x * g[4]; b3 = r0.w * g[1] - r2.y * g[4];
    } else {
        t1 = r0.y * g[0] + r2.w * g[5]; b1 = r2.z * g[5] - r0.x * g[0];
        t3 = r0.w * g[1] + r2.y * g[4]; b3 = r2.x * g[4] - r0.z * g[1];
    }
}
template <int E>
DT_HD void rec_samples2_t(const float *rec, const float *g, float &t2, float &b2) {
    const f4 r1 = reinterpret_cast<const f4 *>(rec)[1];
    if (E == 0) { t2 = r1.x * g[2] + r1.z * g[3]; b2 = r1.y * g[2] - r1.w * g[3]; }
    else { t2 = r1.y * g[2] + r1.w * g[3]; b2 = r1.z * g[3] - r1.x * g[2]; }
}

// WHICH 0: planes lh -> wa, hh -> wb;  WHICH 1: plane hl -> wa (wb untouched)
template <class C, int E, int WHICH>
DT_HD void inv2r_gather_half_e(const Inv2Params &p, const float *srec, float (&wa)[C::WS], float (&wb)[C::WS], int tid, int r0, int c0) {
    const ColTask t = inv_col_task<C>(tid);
    if (!t.valid) return;
    const int ro = r0 + C::ORG, co = c0 + C::ORG;
    const bool interior = ro >= 0 && ro + C::NR <= p.zr && co >= 0 && co + C::NC <= p.zc;
    const int rs = C::RS * t.strip;
    const float *rbase = srec + ((rs / 2) * C::QC + t.i) * 12;
    if (interior) {
#pragma unroll
        for (int ru = 0; ru < C::WS / 2; ++ru) {
            if (WHICH == 0) rec_samples13_t<E>(rbase + ru * C::QC * 12, p.g, wa[2 * ru], wa[2 * ru + 1], wb[2 * ru], wb[2 * ru + 1]);
            else rec_samples2_t<E>(rbase + ru * C::QC * 12, p.g, wa[2 * ru], wa[2 * ru + 1]);
        }
    } else {
        const int fc = reflect_i(co + 2 * t.i, p.zc) & 1;
        const bool e1 = (E ^ fc) != 0;              // reflection may flip the column parity per lane
#pragma unroll
        for (int ru = 0; ru < C::WS / 2; ++ru) {
            const int fr = reflect_i(ro + rs + 2 * ru, p.zr) & 1;
            const float *rec = rbase + ru * C::QC * 12;
            if (WHICH == 0) {
                float ta0, ba0, tb0, bb0, ta1, ba1, tb1, bb1;
                rec_samples13_t<0>(rec, p.g, ta0, ba0, tb0, bb0);
                rec_samples13_t<1>(rec, p.g, ta1, ba1, tb1, bb1);
                const float ta = e1 ? ta1 : ta0, ba = e1 ? ba1 : ba0, tb = e1 ? tb1 : tb0, bb = e1 ? bb1 : bb0;
                wa[2 * ru] = fr ? ba : ta; wa[2 * ru + 1] = fr ? ta : ba;
                wb[2 * ru] = fr ? bb : tb; wb[2 * ru + 1] = fr ? tb : bb;
            } else {
                float t0, b0, t1, b1;
                rec_samples2_t<0>(rec, p.g, t0, b0);
                rec_samples2_t<1>(rec, p.g, t1, b1);
                const float tt = e1 ? t1 : t0, bb = e1 ? b1 : b0;
                wa[2 * ru] = fr ? bb : tt; wa[2 * ru + 1] = fr ? tt : bb;
            }
        }
    }
}
template <class C, int WHICH>
DT_HD void inv2r_gather_half(const Inv2Params &p, const float *srec, float (&wa)[C::WS], float (&wb)[C::WS], int tid, int r0, int c0) {
    if (DT_WAVE_UNIFORM((tid >> 6) & 1)) inv2r_gather_half_e<C, 1, WHICH>(p, srec, wa, wb, tid, r0, c0);
    else inv2r_gather_half_e<C, 0, WHICH>(p, srec, wa, wb, tid, r0, c0);
}

// one y plane: colifilt(wa; lowpass pair) + colifilt(wb; highpass pair)
template <class C, bool STD>
DT_HD void inv2r_fir_plane(const Inv2Params &p, const float (&wa)[C::WS], const float (&wb)[C::WS], float *y, int tid) {
    const ColTask t = inv_col_task<C>(tid);
    if (!t.valid) return;
    constexpr int LP = STD ? 1 : -1, HP = STD ? 0 : -1;
    const int cc = 2 * t.i + t.e;
#pragma unroll
    for (int q = 0; q < C::JS; ++q) {
        float *o = y + 4 * (t.strip * C::JS + q) * C::NC + cc;
        dt_pk2 P = {0.f, 0.f}, Q = {0.f, 0.f};
        ifilt4_acc<C, LP>(wa + 2 * q, p.l_a, p.l_b, p.lo_pos, P, Q);
        ifilt4_acc<C, HP>(wb + 2 * q, p.h_a, p.h_b, p.hi_pos, P, Q);
        o[0] = P.x; o[C::NC] = Q.x; o[2 * C::NC] = P.y; o[3 * C::NC] = Q.y;
    }
}

}  // namespace dt2d
