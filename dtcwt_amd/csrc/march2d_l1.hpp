// Marching wavefront programs for LEVEL 1 ALONE of the float32 2-D DT-CWT (gfx950), for the biort sets whose filters are
// too long for the fused launches of march2d.hpp: near_sym_b (13 / 19 taps) and antonini (9 / 7).
//
// k_fwd12m / k_inv21m keep a window of 2 x 3 + 2 rows and the level-2 pending sums in one wavefront's registers; a 19-tap
// filter needs a window of 20 rows, which leaves no room for level 2.  Until round 5 those sets ran level 1 on the tile
// programs (k_fwd1 / k_inv1), whose 16- and 32-row tiles re-filter 18 halo rows per tile: 85 / 135 us per 4096^2 image
// against 65 / 67 for near_sym_a.  Here level 1 is a march of its own -- the same structure as march2d.hpp (one wavefront =
// a strip of 64 lanes x 4 columns marching down a band of rows, row filters through DPP lane shifts, records through a
// wave-private LDS slab, stores on every step dropped by a zero-byte descriptor outside the band, one loop exit) with
//   * a register ring of 2 HH + 2 rows (HH = 9: 80 registers), the march loop unrolled over its period of HH + 1 steps;
//   * HL = ceil(HH / 4) halo lanes either side of a strip, reached by chains of up to HL wave shifts;
//   * the level-1 lowpass written out (4 B/px more than the fused launches move): levels >= 2 stay with the tile programs.
// Traffic per pixel: forward X 4 -> LoLo1 4 + Yh[0] 12, inverse Z1 4 + Yh[0] 12 -> X 4 (20 B/px each, plus the warm-up rows
// of a band: HH input rows above and below for the forward, HH + 1 for the inverse, of which 3/4 are records).
//
// Reference: dtcwt/numpy/transform2d.py:112-130 (forward level 1), :275-293 (inverse level 1); colfilter
// dtcwt/numpy/lowlevel.py:47-80; q2c / c2q transform2d.py:301-350.
#pragma once
#include "march2d.hpp"

namespace dtm {

constexpr int MAXH1 = 9;        // longest half length of a level-1 filter: 19 taps (near_sym_b)

struct Fwd1mParams {
    const float *X;       // [B][R][C]
    float *LoLo;          // [B][R][C]; PLANES build: the first of four plane volumes [4][B][R][C], pstride elements apart
    float *Yh0;           // [B][R/2][C/2][12]; PLANES build: unused
    int64_t pstride;
    int B, R, C;          // R even, C % 4 == 0
    MarchJobs jb;
    // taps by distance d from the centre as (h0, h1) pairs, the shorter filter zero beyond its half length: the column
    // pass; and with the 1/sqrt2 of q2c folded in for the row pass: over the Lo plane (h0, h1 / sqrt2), over the Hi plane
    // (h0, h1) / sqrt2 (march2d.hpp: row_lohi_s)
    float hp[2 * (MAXH1 + 1)] __attribute__((aligned(8)));
    float hpl[2 * (MAXH1 + 1)] __attribute__((aligned(8))), hph[2 * (MAXH1 + 1)] __attribute__((aligned(8)));
};
// h0 / h1: the (odd-length, symmetric) level-1 analysis filters in double precision
inline void dtm_pack_fwd1m(Fwd1mParams &p, int m0, int m1, const double *h0, const double *h1) {
    const double rs = 0.70710678118654752440;
    for (int d = 0; d <= MAXH1; ++d) {
        const double a = d <= m0 / 2 ? h0[m0 / 2 - d] : 0.0, b = d <= m1 / 2 ? h1[m1 / 2 - d] : 0.0;
        p.hp[2 * d] = (float)a; p.hp[2 * d + 1] = (float)b;
        p.hpl[2 * d] = (float)a; p.hpl[2 * d + 1] = (float)(b * rs);
        p.hph[2 * d] = (float)(a * rs); p.hph[2 * d + 1] = (float)(b * rs);
    }
}

// PLANES build: the row taps unscaled
inline void dtm_pack_fwd1m_planes(Fwd1mParams &p, int m0, int m1, const double *h0, const double *h1) {
    dtm_pack_fwd1m(p, m0, m1, h0, h1);
    for (int d = 0; d < 2 * (MAXH1 + 1); ++d) { p.hpl[d] = p.hp[d]; p.hph[d] = p.hp[d]; }
}

template <int M0, int M1>
struct Fwd1m {
    static constexpr int H0 = M0 / 2, H1 = M1 / 2, HH = H0 > H1 ? H0 : H1;
    static constexpr int HL = (HH + 3) / 4;       // halo lanes either side
    static constexpr int VL = 64 - 2 * HL;        // lanes that own columns
    static constexpr int WR = 2 * HH + 2;         // window rows of a step
    static constexpr int PER = WR / 2;            // steps after which the ring is back where it was
    static_assert(M0 % 2 == 1 && M1 % 2 == 1 && HH <= MAXH1, "odd-length level-1 filters of at most 19 taps");
};

struct Inv1mParams {
    const float *Z;       // [B][R][C]          the level-1 lowpass (output of the level-2 inverse, or Yl); PLANES build: the
                          //                    first of four plane volumes [4][B][R][C], pstride elements apart
    const float *Yh0;     // [B][R/2][C/2][12]  (PLANES build: unused)
    float *X;             // [B][R][C]
    int64_t pstride;
    int B, R, C;          // R even, C % 4 == 0
    MarchJobs jb;
    float g1[6];          // gain x sqrt(1/2) per subband
    // synthesis taps by distance d from the centre, each twice (g, g): the row filters run on (plane, plane) pairs with
    // the tap common to both halves (march2d.hpp: a broadcast from half a scalar pair is not free); the transposed column pass
    // on pairs of neighbouring columns likewise
    float gd0[2 * (MAXH1 + 1)] __attribute__((aligned(8))), gd1[2 * (MAXH1 + 1)] __attribute__((aligned(8)));
};
inline void dtm_pack_inv1m(Inv1mParams &p, int m0, int m1, const double *g0, const double *g1) {
    for (int d = 0; d <= MAXH1; ++d) {
        p.gd0[2 * d] = p.gd0[2 * d + 1] = d <= m0 / 2 ? (float)g0[m0 / 2 - d] : 0.f;
        p.gd1[2 * d] = p.gd1[2 * d + 1] = d <= m1 / 2 ? (float)g1[m1 / 2 - d] : 0.f;
    }
}

template <int M0, int M1>
struct Inv1m {
    static constexpr int H0 = M0 / 2, H1 = M1 / 2, HM = H0 > H1 ? H0 : H1;
    static constexpr int HL = (HM + 3) / 4;
    static constexpr int VL = 64 - 2 * HL;
    static constexpr int NPX = 2 * HM + 2;        // pending rows of X
    static constexpr int WARM = (HM + 1) / 2;     // record rows a band reads above and below its own
    static_assert(M0 % 2 == 1 && M1 % 2 == 1 && HM <= MAXH1, "odd-length level-1 filters of at most 19 taps");
};

#if defined(__HIP_DEVICE_COMPILE__)

// The HH columns either side of the lane's four, from up to ceil(HH / 4) lanes away: W[HH .. HH + 3] are the lane's own
// (pairs of two planes), W[HH - d] is column -d, W[HH + 3 + d] column 3 + d.  A chain of wave shifts per component; only
// the components that a further lane still has to hand on are shifted again: 2 x HH shifts per component in all.
template <int HH>
__device__ __forceinline__ void halo_pairs(pk2 (&W)[4 + 2 * HH]) {
    constexpr int HL = (HH + 3) / 4;
    pk2 L[4] = {W[HH], W[HH + 1], W[HH + 2], W[HH + 3]}, Rr[4] = {W[HH], W[HH + 1], W[HH + 2], W[HH + 3]};
#pragma unroll
    for (int s = 1; s <= HL; ++s) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // left: after s shifts L[e] is column e of lane l - s = my column e - 4 s; wanted while 4 s' - e <= HH for some s' >= s
            if (4 * s - e <= HH || (s < HL && 4 * HL - e <= HH)) L[e] = pk2{dpp_from_left(L[e].x), dpp_from_left(L[e].y)};
            if (4 * s - e <= HH) W[HH - (4 * s - e)] = L[e];
            // right: column e of lane l + s = my column e + 4 s = 3 + d with d = 4 s + e - 3
            if (4 * s + e - 3 <= HH || (s < HL && 4 * HL + e - 3 <= HH)) Rr[e] = pk2{dpp_from_right(Rr[e].x), dpp_from_right(Rr[e].y)};
            if (4 * s + e - 3 <= HH) W[HH + 3 + (4 * s + e - 3)] = Rr[e];
        }
    }
}

// EDGE builds (a row of at most 256 columns is ONE strip, every lane owns columns: no halo lanes): what halo_pairs() took
// from beyond the image -- lanes that do not exist at the left face, idle lanes at the right one -- is replaced by the
// mirror columns, which are in the window already.  W[H + r] is column 4 lane + r; at the left face of lane l column
// 4 l - d < 0 mirrors to d - 4 l - 1, i.e. r = d - 8 l - 1; at the right face of lane nl - 1 - l column 3 + d mirrors to
// r = 4 + 8 l - d; both for d > 4 l.  (15 + 15 selects of pairs at H = 9.)
template <int H>
__device__ __forceinline__ void edge_mirror(pk2 (&W)[4 + 2 * H], int lane, int nl) {
    constexpr int HLn = (H + 3) / 4;
    pk2 T[4 + 2 * H];
#pragma unroll
    for (int i = 0; i < 4 + 2 * H; ++i) T[i] = W[i];
#pragma unroll
    for (int l = 0; l < HLn; ++l) {
        const bool isl = lane == l, isr = lane == nl - 1 - l;
#pragma unroll
        for (int d = 4 * l + 1; d <= H; ++d) {
            const pk2 ml = T[H + d - 8 * l - 1], mr = T[H + 4 + 8 * l - d];
            W[H - d] = pk2{isl ? ml.x : W[H - d].x, isl ? ml.y : W[H - d].y};
            W[H + 3 + d] = pk2{isr ? mr.x : W[H + 3 + d].x, isr ? mr.y : W[H + 3 + d].y};
        }
    }
}

// Symmetric row filters on (plane, plane) pairs with a tap common to both halves: out = g[0] w[0] + sum_d g[d] (w[-d] + w[d])
template <int H>
__device__ __forceinline__ pk2 sym_gg(const pk2 *wc_, const pk2 *gd) {
    pk2 a = gd[0] * wc_[0];
#pragma unroll
    for (int d = 1; d <= H; ++d) a += gd[d] * (wc_[-d] + wc_[d]);
    return a;
}

#endif  // __HIP_DEVICE_COMPILE__

// ======================================================================================================================
// Level 1 of the forward transform as a march: X -> LoLo1, Yh[0].
// A step = two rows.  Its window (rows r - HH .. r + HH + 1) sits in a register ring, the rows of the next P steps are
// on their way; the column pass makes (lo, hi) pairs of the lane's four columns for both rows, the row pass takes the
// HH columns either side from the neighbouring lanes; q2c is lane-local; the records leave through the slab.
// ======================================================================================================================
// PLANES (the in-slice half of the 3-D level 1 for long filters, fused3d_long.hpp): the four row-filtered planes (a1, a2) =
// (Lo|Hi down the columns, lo|hi along the rows) leave as they are -- plane 2 a1 + a2 of p.LoLo, no q2c, no records -- and the
// row taps p.hpl / p.hph come without the 1 / sqrt2 (dtm_pack_fwd1m_planes).
// EDGE: rows of at most 256 columns as one strip without halo lanes (edge_mirror above).
template <int M0, int M1, int P, bool PLANES = false, bool EDGE = false>
__global__ void __launch_bounds__(64, 2) k_fwd1m(const Fwd1mParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = Fwd1m<M0, M1>;
    constexpr int HH = G::HH, WR = G::WR, HL = EDGE ? 0 : G::HL, VL = EDGE ? 64 : G::VL, PER = G::PER;
    static_assert(!EDGE || PLANES, "the one-strip build writes planes");
    static_assert(PER % P == 0, "prefetch depth divides the ring period");
    __shared__ __attribute__((aligned(16))) f4 slab[64 * 6 + 6 * G::HL + 8];
    const int lane = threadIdx.x;
    int strip, band, b;
    if (!dtm_job(p.jb, blockIdx.x, strip, band, b)) return;
    const int R = p.R, C = p.C;

    const int c0 = strip * (4 * VL) - 4 * HL + 4 * lane;
    const bool rev = c0 < 0 || c0 >= C;
    int lc = c0 < 0 ? -c0 - 4 : (c0 >= C ? 2 * C - 4 - c0 : c0);
    lc = lc < 0 ? 0 : (lc > C - 4 ? C - 4 : lc);
    const bool edge_strip = !EDGE && (strip == 0 || (strip + 1) * (4 * VL) + 4 * HL >= C);
    const int nv = (C - strip * (4 * VL)) / 4 < VL ? (C - strip * (4 * VL)) / 4 : VL;     // owning lanes

    const int64_t img = (int64_t)b * R * C;
    const DtBuf bx = dt_buf2g(p.X + img);
    float *const Lb = p.LoLo + img + strip * (4 * VL);
    float *const Y0b = p.Yh0 + img * 3 + (int64_t)strip * (VL * 24);
    const unsigned pitch = (unsigned)C * 4u;

    const int rb = band * p.jb.band_rows;
    const int nrow = R - rb < p.jb.band_rows ? R - rb : p.jb.band_rows;
    // whole periods of the ring (one loop exit: march2d.hpp); the surplus steps re-read the last row and store nothing
    const int nst = (nrow / 2 + PER - 1) / PER * PER;
    const int last_row = rb + nrow - 1 + HH;

    auto ldrow = [&](int u) -> f4 {
        u = u > last_row ? last_row : u;
        u = u < 0 ? -1 - u : u;
        u = u >= R ? 2 * R - 1 - u : u;
        return dt2d::dt_buf_ld4(bx, (unsigned)lc * 4u, (unsigned)u * pitch);
    };
    auto fix = [&](f4 &v) { if (edge_strip) v = rev ? rev4(v) : v; };

    f4 ring[WR], pre[2 * P];
#pragma unroll
    for (int i = 0; i < WR; ++i) ring[i] = ldrow(rb - HH + i);
#pragma unroll
    for (int i = 0; i < 2 * P; ++i) pre[i] = ldrow(rb - HH + WR + i);
#pragma unroll
    for (int i = 0; i < WR; ++i) asm volatile("" : "+v"(ring[i].x), "+v"(ring[i].y), "+v"(ring[i].z), "+v"(ring[i].w) : : "memory");
#pragma unroll
    for (int i = 0; i < 2 * P; ++i) asm volatile("" : "+v"(pre[i].x), "+v"(pre[i].y), "+v"(pre[i].z), "+v"(pre[i].w) : : "memory");
#pragma unroll
    for (int i = 0; i < WR; ++i) fix(ring[i]);

    const unsigned yv = 16u * (unsigned)lane;
    const unsigned lv = 16u * (unsigned)(lane - HL);        // halo lanes: out of range either side
    const pk2 *hpp = reinterpret_cast<const pk2 *>(p.hp);
    const pk2 *hpl = reinterpret_cast<const pk2 *>(p.hpl), *hph = reinterpret_cast<const pk2 *>(p.hph);

    for (int t0 = 0; t0 < nst; t0 += PER) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int r = rb + 2 * (t0 + k);
            const f4 in0 = pre[(2 * k) % (2 * P)], in1 = pre[(2 * k + 1) % (2 * P)];
            pre[(2 * k) % (2 * P)] = ldrow(r - HH + WR + 2 * P);
            pre[(2 * k + 1) % (2 * P)] = ldrow(r - HH + WR + 2 * P + 1);
            const bool in_band = r < rb + nrow;             // uniform
            // the window as column pairs: wp[0][j] = columns (0, 1) of window row j, wp[1][j] = columns (2, 3)
            pk2 wp[2][WR];
#pragma unroll
            for (int j = 0; j < WR; ++j) {
                const f4 &w = ring[(2 * k + j) % WR];
                wp[0][j] = pk2{w.x, w.y}; wp[1][j] = pk2{w.z, w.w};
            }
            f4 ll[2], lh[2], hl[2], hh[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                pk2 W[4 + 2 * HH];
                col_lohi2<HH>(&wp[0][q + HH], hpp, W[HH], W[HH + 1]);
                col_lohi2<HH>(&wp[1][q + HH], hpp, W[HH + 2], W[HH + 3]);
                halo_pairs<HH>(W);
                if constexpr (EDGE) edge_mirror<HH>(W, lane, nv);
                pk2 ol[4], oh[4];           // (ll, lh) and (hl, hh) of the four columns, the highpass ones over sqrt2
#pragma unroll
                for (int c = 0; c < 4; ++c) row_lohi_s<HH>(&W[c + HH], hpl, hph, ol[c], oh[c]);
                ll[q] = f4{ol[0].x, ol[1].x, ol[2].x, ol[3].x}; lh[q] = f4{ol[0].y, ol[1].y, ol[2].y, ol[3].y};
                hl[q] = f4{oh[0].x, oh[1].x, oh[2].x, oh[3].x}; hh[q] = f4{oh[0].y, oh[1].y, oh[2].y, oh[3].y};
            }
            if constexpr (PLANES) {
                const int ro = in_band ? r : rb;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f4 pv[4] = {ll[q], lh[q], hl[q], hh[q]};
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        dt2d::dt_buf_st4<false>(dt_buf_n(Lb + v * p.pstride + (int64_t)(ro + q) * C, in_band ? 16u * nv : 0u), lv, 0u, pv[v]);
                }
            } else {
            {
                const Zq a0 = q2c_p(hl[0].x, hl[0].y, hl[1].x, hl[1].y), a1 = q2c_p(hl[0].z, hl[0].w, hl[1].z, hl[1].w);
                const Zq b0 = q2c_p(hh[0].x, hh[0].y, hh[1].x, hh[1].y), b1 = q2c_p(hh[0].z, hh[0].w, hh[1].z, hh[1].w);
                const Zq c0q = q2c_p(lh[0].x, lh[0].y, lh[1].x, lh[1].y), c1q = q2c_p(lh[0].z, lh[0].w, lh[1].z, lh[1].w);
                f4 *o = slab + lane * 6;
                o[0] = f4{a0.z0r, a0.z0i, b0.z0r, b0.z0i};
                o[1] = f4{c0q.z0r, c0q.z0i, c0q.z1r, c0q.z1i};
                o[2] = f4{b0.z1r, b0.z1i, a0.z1r, a0.z1i};
                o[3] = f4{a1.z0r, a1.z0i, b1.z0r, b1.z0i};
                o[4] = f4{c1q.z0r, c1q.z0i, c1q.z1r, c1q.z1i};
                o[5] = f4{b1.z1r, b1.z1i, a1.z1r, a1.z1i};
            }
            // stores on every step, dropped whole by a zero-byte descriptor beyond the band (march2d.hpp)
            const int ro = in_band ? r : rb;
            dt2d::dt_buf_st4<false>(dt_buf_n(Lb + (int64_t)ro * C, in_band ? 16u * nv : 0u), lv, 0u, ll[0]);
            dt2d::dt_buf_st4<false>(dt_buf_n(Lb + (int64_t)(ro + 1) * C, in_band ? 16u * nv : 0u), lv, 0u, ll[1]);
            DT_WAVE_LDS_SYNC();
            const DtBuf by = dt_buf_n(Y0b + (int64_t)(ro >> 1) * C * 6, in_band ? 96u * nv : 0u);
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                const f4 v = slab[6 * HL + lane + 64 * m];
                dt2d::dt_buf_st4<true>(by, yv + 1024u * m, 0u, v);
            }
            DT_WAVE_LDS_SYNC();
            }
            f4 e0 = in0, e1 = in1;
            fix(e0); fix(e1);
            ring[(2 * k) % WR] = e0;
            ring[(2 * k + 1) % WR] = e1;
        }
    }
#endif
}

// ======================================================================================================================
// Level 1 of the inverse transform as a march: Z1, Yh[0] -> X.
// A step takes ONE row of records (two rows of Z1, two records per lane) and completes two rows of X: c2q with the gains,
// the row filters (v0 = g0o * Z1 + g1o * q23, v1 = g0o * q05 + g1o * q14 -- the pairs (Z1, q05) and (q23, q14) share their
// taps) with a DPP halo of HM columns, then the column filters in transposed form into the pending rows of X (row i
// continues what row i + 2 held: the shift costs nothing), of which the two oldest leave as 16-byte stores.
// Rows are requested one step ahead.  Symmetric extension: reflected record rows swap the rows of their quads, mirrored
// lanes take the mirror lane's records in reverse (march2d.hpp: k_inv21m).
// ======================================================================================================================
// PLANES (the in-slice half of the 3-D level-1 inverse for long filters, fused3d_long.hpp): the lowpass and the three quad
// planes arrive as the four plane volumes (a1, a2) = plane 2 a1 + a2 of p.Z -- Z1 = plane 0, q23 = plane 1 (Lo down the columns,
// hi along the rows), q05 = plane 2, q14 = plane 3 -- instead of c2q of the records; no gains.
template <int M0, int M1, bool PLANES = false, bool EDGE = false>
__global__ void __launch_bounds__(64, 2) k_inv1m(const Inv1mParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = Inv1m<M0, M1>;
    constexpr int H0 = G::H0, H1 = G::H1, HM = G::HM, HL = EDGE ? 0 : G::HL, VL = EDGE ? 64 : G::VL, NPX = G::NPX, WARM = G::WARM;
    static_assert(!EDGE || PLANES, "the one-strip build reads planes");
    __shared__ __attribute__((aligned(16))) f4 slab[64 * 6 + 6 * G::HL + 8];
    const int lane = threadIdx.x;
    int strip, band, b;
    if (!dtm_job(p.jb, blockIdx.x, strip, band, b)) return;
    const int R = p.R, C = p.C;
    const int cb = strip * (4 * VL) - 4 * HL;             // column of lane 0
    const int c0 = cb + 4 * lane;
    const bool mir = c0 < 0 || c0 >= C;
    const bool edge_strip = !EDGE && (cb < 0 || cb + 256 > C);      // uniform: some lane is mirrored
    const int nv = (C - strip * (4 * VL)) / 4 < VL ? (C - strip * (4 * VL)) / 4 : VL;
    // where the lane's lowpass samples come from (mirrored lanes: the mirror block, reversed on arrival) and which slab
    // lane holds its records
    int lc = c0 < 0 ? -c0 - 4 : (c0 >= C ? 2 * C - 4 - c0 : c0);
    lc = lc < 0 ? 0 : (lc > C - 4 ? C - 4 : lc);
    int sl = c0 < 0 ? (-c0 - 4 - cb) / 4 : (c0 >= C ? (2 * C - 4 - c0 - cb) / 4 : lane);
    sl = sl < 0 ? 0 : (sl > 63 ? 63 : sl);
    const int lmin = cb < 0 ? -cb / 4 : 0, lmax = (C - cb) / 4 - 1 < 63 ? (C - cb) / 4 - 1 : 63;     // lanes inside the image

    const int64_t img = (int64_t)b * R * C;
    const DtBuf bz = dt_buf2g(p.Z + img);
    const float *const Y0b = p.Yh0 + img * 3 + (int64_t)(cb + 4 * lmin) * 6;       // record of lane lmin in row 0
    float *const Xb = p.X + img + strip * (4 * VL);
    const unsigned pitch = (unsigned)C * 4u;
    const unsigned r1bytes = (unsigned)(lmax - lmin + 1) * 96u;

    const int rb = band * p.jb.band_rows;
    const int nrow = R - rb < p.jb.band_rows ? R - rb : p.jb.band_rows;
    const int rr0 = rb / 2 - WARM, nst = nrow / 2 + 2 * WARM;      // record rows rr0 .. rr0 + nst - 1

    auto zrow = [&](int u) { u = u < 0 ? -1 - u : u; u = u >= R ? 2 * R - 1 - u : u; return u < 0 ? 0 : (u > R - 1 ? R - 1 : u); };
    auto rec_row = [&](int rr, bool &sw) { sw = rr < 0 || rr >= R / 2; rr = rr < 0 ? -1 - rr : rr; rr = rr >= R / 2 ? R - 1 - rr : rr; return rr < 0 ? 0 : (rr > R / 2 - 1 ? R / 2 - 1 : rr); };

    f4 zp[2], r1p[6];
    const DtBuf bq1 = dt_buf2g(p.Z + img + (PLANES ? p.pstride : 0)), bq2 = dt_buf2g(p.Z + img + (PLANES ? 2 * p.pstride : 0)),
                bq3 = dt_buf2g(p.Z + img + (PLANES ? 3 * p.pstride : 0));
    auto request = [&](int rr) {
        bool sw;
        zp[0] = dt2d::dt_buf_ld4(bz, (unsigned)lc * 4u, (unsigned)zrow(2 * rr) * pitch);
        zp[1] = dt2d::dt_buf_ld4(bz, (unsigned)lc * 4u, (unsigned)zrow(2 * rr + 1) * pitch);
        if constexpr (PLANES) {           // r1p[2 (plane - 1) + e]: rows 2 rr + e of planes 1, 2, 3
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const unsigned ro = (unsigned)zrow(2 * rr + e) * pitch;
                r1p[e] = dt2d::dt_buf_ld4(bq1, (unsigned)lc * 4u, ro);
                r1p[2 + e] = dt2d::dt_buf_ld4(bq2, (unsigned)lc * 4u, ro);
                r1p[4 + e] = dt2d::dt_buf_ld4(bq3, (unsigned)lc * 4u, ro);
            }
            return;
        }
        const DtBuf br = dt_buf_n(Y0b + (int64_t)rec_row(rr, sw) * C * 6, r1bytes);
#pragma unroll
        for (int m = 0; m < 6; ++m) r1p[m] = dt2d::dt_buf_ld4(br, 16u * (unsigned)lane + 1024u * m, 0u);
    };
    request(rr0);
    asm volatile("" : "+v"(zp[0].x), "+v"(zp[0].y), "+v"(zp[0].z), "+v"(zp[0].w), "+v"(zp[1].x), "+v"(zp[1].y), "+v"(zp[1].z), "+v"(zp[1].w) : : "memory");
#pragma unroll
    for (int m = 0; m < 6; ++m) asm volatile("" : "+v"(r1p[m].x), "+v"(r1p[m].y), "+v"(r1p[m].z), "+v"(r1p[m].w) : : "memory");

    pk2 PX[NPX][2];
#pragma unroll
    for (int i = 0; i < NPX; ++i) { PX[i][0] = pk2{0.f, 0.f}; PX[i][1] = pk2{0.f, 0.f}; }
    const unsigned xv = 16u * (unsigned)(lane - HL);
    const pk2 *gd0 = reinterpret_cast<const pk2 *>(p.gd0), *gd1 = reinterpret_cast<const pk2 *>(p.gd1);

    for (int st = 0; st < nst; ++st) {
        const int rr = rr0 + st, rho = 2 * rr;
        bool sw;
        (void)rec_row(rr, sw);
        // ---- what was requested a step ago: the records to the slab, the lowpass rows in place
        f4 qq[6];
        if constexpr (PLANES) {
#pragma unroll
            for (int m = 0; m < 6; ++m) { qq[m] = r1p[m]; if (edge_strip) qq[m] = mir ? rev4(qq[m]) : qq[m]; }
        } else {
#pragma unroll
        for (int m = 0; m < 6; ++m) slab[6 * lmin + lane + 64 * m] = r1p[m];
        }
        f4 zz[2] = {zp[0], zp[1]};
        // (rows 2 rr, 2 rr + 1 reflect to the two rows of the reflected record row in the other order: zrow() did that)
        if (edge_strip) { zz[0] = mir ? rev4(zz[0]) : zz[0]; zz[1] = mir ? rev4(zz[1]) : zz[1]; }
        request(rr + 1);
        DT_WAVE_LDS_SYNC();
        float q05[2][4], q23[2][4], q14[2][4];
        if constexpr (PLANES) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                q23[e][0] = qq[e].x; q23[e][1] = qq[e].y; q23[e][2] = qq[e].z; q23[e][3] = qq[e].w;
                q05[e][0] = qq[2 + e].x; q05[e][1] = qq[2 + e].y; q05[e][2] = qq[2 + e].z; q05[e][3] = qq[2 + e].w;
                q14[e][0] = qq[4 + e].x; q14[e][1] = qq[4 + e].y; q14[e][2] = qq[4 + e].z; q14[e][3] = qq[4 + e].w;
            }
        } else {
            const f4 *sp = slab + 6 * sl;
            f4 s_[6];
#pragma unroll
            for (int m = 0; m < 6; ++m) s_[m] = sp[m];
            float A05[2][2], A23[2][2], A14[2][2], B05[2][2], B23[2][2], B14[2][2];
            c2q_quad(s_[0].x, s_[0].y, s_[2].z, s_[2].w, p.g1[0], p.g1[5], A05);
            c2q_quad(s_[1].x, s_[1].y, s_[1].z, s_[1].w, p.g1[2], p.g1[3], A23);
            c2q_quad(s_[0].z, s_[0].w, s_[2].x, s_[2].y, p.g1[1], p.g1[4], A14);
            c2q_quad(s_[3].x, s_[3].y, s_[5].z, s_[5].w, p.g1[0], p.g1[5], B05);
            c2q_quad(s_[4].x, s_[4].y, s_[4].z, s_[4].w, p.g1[2], p.g1[3], B23);
            c2q_quad(s_[3].z, s_[3].w, s_[5].x, s_[5].y, p.g1[1], p.g1[4], B14);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    q05[e][f] = A05[e][f]; q05[e][2 + f] = B05[e][f];
                    q23[e][f] = A23[e][f]; q23[e][2 + f] = B23[e][f];
                    q14[e][f] = A14[e][f]; q14[e][2 + f] = B14[e][f];
                }
            }
            if (sw) {               // a reflected record row (image top / bottom): its quads upside down
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float t_;
                    t_ = q05[0][c]; q05[0][c] = q05[1][c]; q05[1][c] = t_;
                    t_ = q23[0][c]; q23[0][c] = q23[1][c]; q23[1][c] = t_;
                    t_ = q14[0][c]; q14[0][c] = q14[1][c]; q14[1][c] = t_;
                }
            }
            if (edge_strip) {       // mirrored lanes: the mirror lane's four columns in reverse
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float t_;
#define DTM_REV4(x_) t_ = x_[e][0]; x_[e][0] = mir ? x_[e][3] : t_; x_[e][3] = mir ? t_ : x_[e][3]; \
                     t_ = x_[e][1]; x_[e][1] = mir ? x_[e][2] : t_; x_[e][2] = mir ? t_ : x_[e][2];
                    DTM_REV4(q05) DTM_REV4(q23) DTM_REV4(q14)
#undef DTM_REV4
                }
            }
        }
        DT_WAVE_LDS_SYNC();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float zv[4] = {zz[e].x, zz[e].y, zz[e].z, zz[e].w};
            pk2 Wa[4 + 2 * H0], Wb[4 + 2 * H1], V[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                Wa[H0 + c] = pk2{zv[c], q05[e][c]};
                Wb[H1 + c] = pk2{q23[e][c], q14[e][c]};
            }
            halo_pairs<H0>(Wa);
            halo_pairs<H1>(Wb);
            if constexpr (EDGE) { edge_mirror<H0>(Wa, lane, nv); edge_mirror<H1>(Wb, lane, nv); }
#pragma unroll
            for (int c = 0; c < 4; ++c) V[c] = sym_gg<H0>(&Wa[H0 + c], gd0) + sym_gg<H1>(&Wb[H1 + c], gd1);
            // columns, transposed, on pairs of neighbouring columns: PX[i][h] = columns (2h, 2h + 1) of row rho - HM + i;
            // v0 through g0o reaches rows e + (HM - H0) + k, v1 through g1o rows e + (HM - H1) + k, the taps as (g, g) pairs.
            // The first row of a step (e == 0) also moves the pending rows up by the two that left at the end of the
            // previous step: row i continues what row i + 2 held (v_pk_fma_f32 has a destination of its own).
            const pk2 v0p[2] = {pk2{V[0].x, V[1].x}, pk2{V[2].x, V[3].x}}, v1p[2] = {pk2{V[0].y, V[1].y}, pk2{V[2].y, V[3].y}};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int i = 0; i < NPX; ++i) {
                    constexpr int o0 = HM - H0, o1 = HM - H1;
                    const int k0 = i - e - o0, k1 = i - e - o1;            // tap indices that reach row i
                    const bool t0 = k0 >= 0 && k0 < M0, t1 = k1 >= 0 && k1 < M1;
                    const int d0 = t0 ? (k0 < H0 ? H0 - k0 : k0 - H0) : 0, d1 = t1 ? (k1 < H1 ? H1 - k1 : k1 - H1) : 0;
                    if (e == 0) {
                        pk2 acc = i + 2 < NPX ? PX[i + 2][h] : pk2{0.f, 0.f};
                        if (t0) acc = gd0[d0] * v0p[h] + acc;
                        if (t1) acc = gd1[d1] * v1p[h] + acc;
                        PX[i][h] = acc;
                    } else {
                        if (t0) PX[i][h] += gd0[d0] * v0p[h];
                        if (t1) PX[i][h] += gd1[d1] * v1p[h];
                    }
                }
            }
        }
        // rows rho - HM, rho - HM + 1 are complete
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int x = rho - HM + e;
            const bool ok = x >= rb && x < rb + nrow;
            const int xo = ok ? x : 0;
            const DtBuf bo = dt_buf_n(Xb + (int64_t)xo * C, ok ? 16u * nv : 0u);
            dt2d::dt_buf_st4<true>(bo, xv, 0u, f4{PX[e][0].x, PX[e][0].y, PX[e][1].x, PX[e][1].y});
        }
    }
#endif
}

}  // namespace dtm
