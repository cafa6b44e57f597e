// Levels 1 + 2 of the float32 2-D forward DT-CWT as a marching PAIR of wavefronts (gfx950): k_fwd12p -- for the 14- / 18-tap
// q-shift sets (qshift_b, qshift_d: 112 / 144 registers of pending sums) with the short level-1 filters (near_sym_a 5 / 7).
//
// k_fwd12m (march2d.hpp) keeps the level-1 window AND the level-2 pending sums in one wavefront: 8 rows + 80 registers for
// near_sym_a / qshift_a, no more.  Here a workgroup is two wavefronts on the same strip and band:
//   wavefront 0   level 1, the march of k_fwd1m (march2d_l1.hpp): ring of 2 HH + 2 rows, (h0, h1) pairs, DPP halo, q2c,
//                 records through its slab.  The two LoLo1 rows of a step do not go to memory: they go into a double-buffered
//                 LDS exchange (2 x 2 x 1 KiB), one LDS-only s_barrier per step.
//   wavefront 1   level 2, the second half of k_fwd12m: the 2M-sample row windows are read straight from the exchange (the
//                 lane's four columns + HL2 lanes either side: no DPP chain), then the transposed column filter into the
//                 M / 2 pending row pairs, q2c, Yh[1] records through its slab, LoLo2.
// The exchange is double-buffered by step parity, so wavefront 0 may run one step ahead.  The level-1 lowpass is never written:
// 20 B/px like k_fwd12m, instead of the 28 B/px of a level-1 launch + a level-2 launch.  The template also builds for near_sym_b
// (13 / 19 taps; parity-green) but loses there -- its level-1 wavefront carries 573 instructions per step against the partner's
// 130, and a wavefront issues one vector instruction per four cycles -- so only the (5, 7) sets are instantiated (march2d.hip).
//
// Reference: dtcwt/numpy/transform2d.py:112-160; coldfilt dtcwt/numpy/lowlevel.py:82-154.
#pragma once
#include "march2d_l1.hpp"

namespace dtm {

template <int M0, int M1, int M>
struct Fwd12p {
    // the ring period (HH + 1 steps) must be even: legall's window (HH = 2) is run as a seven-row one whose outer taps are zero
    static constexpr int H0 = M0 / 2, H1 = M1 / 2, HHx = H0 > H1 ? H0 : H1, HH = HHx % 2 ? HHx : HHx + 1;
    static constexpr int HL1 = (HH + 3) / 4, HL2 = (M - 2) / 4, HL = HL1 + HL2;
    static constexpr int VL = 64 - 2 * HL;
    static constexpr int WR = 2 * HH + 2, PER = WR / 2;
    static constexpr int NP2 = M / 2, PRE = M - 2;
    static_assert(M0 % 2 == 1 && M1 % 2 == 1 && HH <= MAXH1 && (M - 2) % 4 == 0 && M <= MAXT2 && PER % 2 == 0,
                  "odd level-1 filters of at most 19 taps with an even ring period, q-shift filters of 10, 14 or 18 taps");
};

struct Fwd12pParams {
    const float *X;       // [B][R][C]
    float *Yh0;           // [B][R/2][C/2][12]
    float *Yh1;           // [B][R/4][C/4][12]
    float *LoLo2;         // [B][R/2][C/2]
    int B, R, C;          // R % 4 == 0, C % 4 == 0
    MarchJobs jb;
    // level 1 (march2d_l1.hpp: dtm_pack_fwd1m)
    float hp[2 * (MAXH1 + 1)] __attribute__((aligned(8)));
    float hpl[2 * (MAXH1 + 1)] __attribute__((aligned(8))), hph[2 * (MAXH1 + 1)] __attribute__((aligned(8)));
    // level 2 by window offset as (lowpass, highpass) pairs (march2d.hpp: dtm_pack_qshift); ta_lo .. tb_hi are scratch of the packer
    float ta_lo[MAXT2], tb_lo[MAXT2], ta_hi[MAXT2], tb_hi[MAXT2];
    float ta2[2 * MAXT2] __attribute__((aligned(8))), tb2[2 * MAXT2] __attribute__((aligned(8)));
};

#if defined(__HIP_DEVICE_COMPILE__)
#define DTM_PAIR_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// the H columns either side of the lane's four (one plane): the scalar form of halo_pairs
template <int H>
__device__ __forceinline__ void halo_scalar(float (&w)[4 + 2 * H]) {
    constexpr int HLn = (H + 3) / 4;
    float L[4] = {w[H], w[H + 1], w[H + 2], w[H + 3]}, Rr[4] = {w[H], w[H + 1], w[H + 2], w[H + 3]};
#pragma unroll
    for (int s = 1; s <= HLn; ++s) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (4 * s - e <= H || (s < HLn && 4 * HLn - e <= H)) L[e] = dpp_from_left(L[e]);
            if (4 * s - e <= H) w[H - (4 * s - e)] = L[e];
            if (4 * s + e - 3 <= H || (s < HLn && 4 * HLn + e - 3 <= H)) Rr[e] = dpp_from_right(Rr[e]);
            if (4 * s + e - 3 <= H) w[H + 3 + (4 * s + e - 3)] = Rr[e];
        }
    }
}
// the symmetric lowpass alone (the warm-up rows, where level 1 only has to feed level 2): taps = the low halves of hp[d]
template <int H>
__device__ __forceinline__ float sym_lo(const float *xc, const pk2 *hp) {
    float r = hp[0].x * xc[0];
#pragma unroll
    for (int d = 1; d <= H; ++d) r += hp[d].x * (xc[-d] + xc[d]);
    return r;
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// The level-2 wavefront: the second half of k_fwd12m for any M with (M - 2) % 4 == 0 on the strip whose lane 0 sits HL lanes left of
// its first owned column.  Step t brings the LoLo1 rows rbase + 2t, + 1 as xbuf[t & 1][0 / 1][HL2 + lane] (the lane's four columns;
// the 2M-sample windows are read across the lanes from there); stepsync(t, k) returns when they are in place.
template <int M, int HL, class Sync>
__device__ __forceinline__ void fwd2_wave(const Fwd12pParams &p, int lane, int strip, int64_t img, int nv, int rb, int nrow, int rbase, int nst,
                                          f4 (*xbuf)[2][64 + 2 * ((M - 2) / 4)], f4 *slab2, Sync &&stepsync) {
    constexpr int HL2 = (M - 2) / 4, NP2 = M / 2, VL = 64 - 2 * HL;
    const int C = p.C;
    const unsigned yv = 16u * (unsigned)lane;
    float *const Y1b = p.Yh1 + (img / 4) * 3 + (int64_t)strip * (VL * 12);
    float *const L2b = p.LoLo2 + img / 4 + strip * (VL * 2);
    const unsigned l2v = 8u * (unsigned)(lane - HL);
    const float sq = 0.70710678118654752440f;
    const pk2 *ta2 = reinterpret_cast<const pk2 *>(p.ta2), *tb2 = reinterpret_cast<const pk2 *>(p.tb2);
    pk2 S2[NP2][2][4];
#pragma unroll
    for (int a = 0; a < NP2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) S2[a][c][v] = pk2{0.f, 0.f};
    for (int t0 = 0; t0 < nst; t0 += 2) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int t = t0 + k, r = rbase + 2 * t;
            stepsync(t, k);                          // the rows of step t are in xbuf[k] (t0 is even: k = t & 1)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                // the 2M-sample window as M pairs of neighbouring samples: lanes l - HL2 .. l + HL2
                pk2 w2[M];
                const f4 *src = &xbuf[k][q][lane];
#pragma unroll
                for (int d = 0; d < 2 * HL2 + 1; ++d) { const f4 x = src[d]; w2[2 * d] = pk2{x.x, x.y}; w2[2 * d + 1] = pk2{x.z, x.w}; }
                pk2 rA = {0.f, 0.f}, rB = {0.f, 0.f};          // (L_A, H_A), (L_B, H_B)
#pragma unroll
                for (int tt = 0; tt < M; ++tt) { rA += ta2[tt] * DTM_BX(w2[tt]); rB += tb2[tt] * DTM_BY(w2[tt]); }
                const int phi = 2 * k + q;                 // row 4n + phi of its group (rbase % 4 == 0, t0 even)
#pragma unroll
                for (int a = 0; a < NP2; ++a) {
                    const int tt = (phi + 8 * HL2 - 4 * a) >> 1;
                    const pk2 cc = (phi & 1) ? tb2[tt] : ta2[tt];
                    if (phi < 2) {      // the first rows to touch the A / B halves after a pair left: slot a continues slot a + 1
                        if (a + 1 < NP2) {
                            S2[a][phi & 1][0] = cc * DTM_BX(rA) + S2[a + 1][phi & 1][0]; S2[a][phi & 1][1] = cc * DTM_BX(rB) + S2[a + 1][phi & 1][1];
                            S2[a][phi & 1][2] = cc * DTM_BY(rA) + S2[a + 1][phi & 1][2]; S2[a][phi & 1][3] = cc * DTM_BY(rB) + S2[a + 1][phi & 1][3];
                        } else {
                            S2[a][phi & 1][0] = cc * DTM_BX(rA); S2[a][phi & 1][1] = cc * DTM_BX(rB);
                            S2[a][phi & 1][2] = cc * DTM_BY(rA); S2[a][phi & 1][3] = cc * DTM_BY(rB);
                        }
                    } else {
                        S2[a][phi & 1][0] += cc * DTM_BX(rA); S2[a][phi & 1][1] += cc * DTM_BX(rB);
                        S2[a][phi & 1][2] += cc * DTM_BY(rA); S2[a][phi & 1][3] += cc * DTM_BY(rB);
                    }
                }
            }
            if (k & 1) {
                // rows 4n + 2, 4n + 3 are in: pair i = n - HL2 is complete (sum(ha hb) > 0 for the lowpass pair, < 0 for the
                // highpass pair: every shipped set; the launcher checks)
                const int i2 = (r - 2) / 4 - HL2;
                const bool pair_ok = 4 * i2 >= rb && 4 * i2 < rb + nrow;        // uniform; otherwise the stores are dropped
                constexpr bool la = true, ha = false;
                float pl[2][4], ph[2][4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    pl[0][v] = la ? S2[0][0][v].x : S2[0][1][v].x; pl[1][v] = la ? S2[0][1][v].x : S2[0][0][v].x;
                    ph[0][v] = ha ? S2[0][0][v].y : S2[0][1][v].y; ph[1][v] = ha ? S2[0][1][v].y : S2[0][0][v].y;
                }
                float llo[2][2], lh2[2][2], hl2[2][2], hh2[2][2];
#pragma unroll
                for (int er = 0; er < 2; ++er) {
                    llo[er][0] = la ? pl[er][0] : pl[er][1]; llo[er][1] = la ? pl[er][1] : pl[er][0];
                    lh2[er][0] = ha ? pl[er][2] : pl[er][3]; lh2[er][1] = ha ? pl[er][3] : pl[er][2];
                    hl2[er][0] = la ? ph[er][0] : ph[er][1]; hl2[er][1] = la ? ph[er][1] : ph[er][0];
                    hh2[er][0] = ha ? ph[er][2] : ph[er][3]; hh2[er][1] = ha ? ph[er][3] : ph[er][2];
                }
                const int io = pair_ok ? i2 : 0;
                const DtBuf bl0 = dt_buf_n(L2b + (int64_t)(2 * io) * (C / 2), pair_ok ? 8u * nv : 0u);
                const DtBuf bl1 = dt_buf_n(L2b + (int64_t)(2 * io + 1) * (C / 2), pair_ok ? 8u * nv : 0u);
                dt2d::dt_buf_st2<false>(bl0, l2v, 0u, dt2d::f2{llo[0][0], llo[0][1]});
                dt2d::dt_buf_st2<false>(bl1, l2v, 0u, dt2d::f2{llo[1][0], llo[1][1]});
                const Zq a = q2c_s(hl2[0][0], hl2[0][1], hl2[1][0], hl2[1][1]);
                const Zq bq = q2c_s(hh2[0][0], hh2[0][1], hh2[1][0], hh2[1][1]);
                const Zq c = q2c_s(lh2[0][0], lh2[0][1], lh2[1][0], lh2[1][1]);
                f4 *o = slab2 + lane * 3;
                o[0] = f4{sq * a.z0r, sq * a.z0i, sq * bq.z0r, sq * bq.z0i};
                o[1] = f4{sq * c.z0r, sq * c.z0i, sq * c.z1r, sq * c.z1i};
                o[2] = f4{sq * bq.z1r, sq * bq.z1i, sq * a.z1r, sq * a.z1i};
                DT_WAVE_LDS_SYNC();
                const DtBuf by1 = dt_buf_n(Y1b + (int64_t)io * (C / 4) * 12, pair_ok ? 48u * nv : 0u);
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const f4 v = slab2[3 * HL + lane + 64 * m];
                    dt2d::dt_buf_st4<false>(by1, yv + 1024u * m, 0u, v);
                }
                DT_WAVE_LDS_SYNC();
            }
        }
    }
}
#endif

// WPS: wavefronts per SIMD the registers are allocated for (146 VGPRs at M = 14 when asked: three fit, i.e. six pairs per CU)
template <int M0, int M1, int M, int P, int WPS = 2>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, WPS))) k_fwd12p(const Fwd12pParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = Fwd12p<M0, M1, M>;
    constexpr int HH = G::HH, H0_ = G::H0, WR = G::WR, HL = G::HL, HL2 = G::HL2, VL = G::VL, PER = G::PER;
    static_assert(PER % P == 0, "prefetch depth divides the ring period");
    __shared__ __attribute__((aligned(16))) f4 slab[64 * 6 + 6 * G::HL + 8];
    __shared__ __attribute__((aligned(16))) f4 slab2[64 * 3 + 3 * G::HL + 8];
    __shared__ __attribute__((aligned(16))) f4 xbuf[2][2][64 + 2 * G::HL2];      // [step parity][row][HL2 + lane]
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int strip, band, b;
    if (!dtm_job(p.jb, blockIdx.x, strip, band, b)) return;
    const int R = p.R, C = p.C;
    const int nv = (C - strip * (4 * VL)) / 4 < VL ? (C - strip * (4 * VL)) / 4 : VL;     // owning lanes
    const int64_t img = (int64_t)b * R * C;
    const int rb = band * p.jb.band_rows;
    const int nrow = R - rb < p.jb.band_rows ? R - rb : p.jb.band_rows;
    const int rbase = rb - G::PRE;                     // first LoLo1 row the band's level-2 windows want (% 4 == 0)
    const int nst = (nrow / 2 + G::PRE + PER - 1) / PER * PER;         // whole ring periods, an even count
    const unsigned yv = 16u * (unsigned)lane;

    if (role == 0) {
        // ------------------------------------------------------------------ level 1 (k_fwd1m without its LoLo1 stores)
        const int c0 = strip * (4 * VL) - 4 * HL + 4 * lane;
        const bool rev = c0 < 0 || c0 >= C;
        int lc = c0 < 0 ? -c0 - 4 : (c0 >= C ? 2 * C - 4 - c0 : c0);
        lc = lc < 0 ? 0 : (lc > C - 4 ? C - 4 : lc);
        const bool edge_strip = strip == 0 || (strip + 1) * (4 * VL) + 4 * HL >= C;
        const DtBuf bx = dt_buf2g(p.X + img);
        float *const Y0b = p.Yh0 + img * 3 + (int64_t)strip * (VL * 24);
        const unsigned pitch = (unsigned)C * 4u;
        const int last_row = rbase + 2 * (nrow / 2 + G::PRE) - 1 + HH;
        auto ldrow = [&](int u) -> f4 {
            u = u > last_row ? last_row : u;
            u = u < 0 ? -1 - u : u;
            u = u >= R ? 2 * R - 1 - u : u;
            return dt2d::dt_buf_ld4(bx, (unsigned)lc * 4u, (unsigned)u * pitch);
        };
        auto fix = [&](f4 &v) { if (edge_strip) v = rev ? rev4(v) : v; };

        f4 ring[WR], pre[2 * P];
#pragma unroll
        for (int i = 0; i < WR; ++i) ring[i] = ldrow(rbase - HH + i);
#pragma unroll
        for (int i = 0; i < 2 * P; ++i) pre[i] = ldrow(rbase - HH + WR + i);
#pragma unroll
        for (int i = 0; i < WR; ++i) asm volatile("" : "+v"(ring[i].x), "+v"(ring[i].y), "+v"(ring[i].z), "+v"(ring[i].w) : : "memory");
#pragma unroll
        for (int i = 0; i < 2 * P; ++i) asm volatile("" : "+v"(pre[i].x), "+v"(pre[i].y), "+v"(pre[i].z), "+v"(pre[i].w) : : "memory");
#pragma unroll
        for (int i = 0; i < WR; ++i) fix(ring[i]);
        const pk2 *hpp = reinterpret_cast<const pk2 *>(p.hp);
        const pk2 *hpl = reinterpret_cast<const pk2 *>(p.hpl), *hph = reinterpret_cast<const pk2 *>(p.hph);

        for (int t0 = 0; t0 < nst; t0 += PER) {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int t = t0 + k, r = rbase + 2 * t;
                const f4 in0 = pre[(2 * k) % (2 * P)], in1 = pre[(2 * k + 1) % (2 * P)];
                pre[(2 * k) % (2 * P)] = ldrow(r - HH + WR + 2 * P);
                pre[(2 * k + 1) % (2 * P)] = ldrow(r - HH + WR + 2 * P + 1);
                const bool in_band = r >= rb && r < rb + nrow;          // uniform
                f4 ll[2];
                if (in_band) {
                    pk2 wp[2][WR];
#pragma unroll
                    for (int j = 0; j < WR; ++j) {
                        const f4 &w = ring[(2 * k + j) % WR];
                        wp[0][j] = pk2{w.x, w.y}; wp[1][j] = pk2{w.z, w.w};
                    }
                    f4 lh[2], hl[2], hh[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        pk2 W[4 + 2 * HH];
                        col_lohi2<HH>(&wp[0][q + HH], hpp, W[HH], W[HH + 1]);
                        col_lohi2<HH>(&wp[1][q + HH], hpp, W[HH + 2], W[HH + 3]);
                        halo_pairs<HH>(W);
                        pk2 ol[4], oh[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) row_lohi_s<HH>(&W[c + HH], hpl, hph, ol[c], oh[c]);
                        ll[q] = f4{ol[0].x, ol[1].x, ol[2].x, ol[3].x}; lh[q] = f4{ol[0].y, ol[1].y, ol[2].y, ol[3].y};
                        hl[q] = f4{oh[0].x, oh[1].x, oh[2].x, oh[3].x}; hh[q] = f4{oh[0].y, oh[1].y, oh[2].y, oh[3].y};
                    }
                    xbuf[t & 1][0][HL2 + lane] = ll[0];
                    xbuf[t & 1][1][HL2 + lane] = ll[1];
                    DTM_PAIR_BARRIER();                      // the step's LoLo1 rows are wavefront 1's now
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
                } else {
                    // warm-up rows above and below the band: the lowpass alone
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float wl[4 + 2 * H0_];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float xs[2 * H0_ + 1];
#pragma unroll
                            for (int j = 0; j < 2 * H0_ + 1; ++j) {
                                const f4 &x = ring[(2 * k + q + HH - H0_ + j) % WR];
                                xs[j] = c == 0 ? x.x : (c == 1 ? x.y : (c == 2 ? x.z : x.w));
                            }
                            wl[H0_ + c] = sym_lo<H0_>(&xs[H0_], hpp);
                        }
                        halo_scalar<H0_>(wl);
                        float a_[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) a_[c] = sym_lo<H0_>(&wl[c + H0_], hpp);
                        ll[q] = f4{a_[0], a_[1], a_[2], a_[3]};
                    }
                    xbuf[t & 1][0][HL2 + lane] = ll[0];
                    xbuf[t & 1][1][HL2 + lane] = ll[1];
                    DTM_PAIR_BARRIER();
                }
                {   // record stores on every step, dropped whole outside the band (march2d.hpp on vmcnt)
                    const int ro = in_band ? r : rb;
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
    } else {
        // ------------------------------------------------------------------ level 2: its rows arrive through the exchange
        fwd2_wave<M, HL>(p, lane, strip, img, nv, rb, nrow, rbase, nst, xbuf, slab2, [&](int, int) { DTM_PAIR_BARRIER(); });
    }
#endif
}

// ======================================================================================================================
// Level 2 of the forward ALONE as a march (transform2d.py:132-160 at the second level): the level-2 wavefront of k_fwd12p fed from
// memory instead of the exchange -- for the sets whose level 1 no pair takes (near_sym_b: k_fwd1m, march2d_l1.hpp).  LoLo1 = p.X
// [B][R][C] -> Yh[1], LoLo2; no level-1 halo lane: HL = HL2.  The wavefront puts the two rows of a step into its own (wave-private)
// copy of the exchange, so the windows are read across the lanes exactly as in the pair; rows are requested two steps ahead.
// ======================================================================================================================
template <int M>
struct Fwd2m {
    static constexpr int HL2 = (M - 2) / 4, HL = HL2, VL = 64 - 2 * HL, PRE = M - 2;
    static_assert((M - 2) % 4 == 0 && M >= 10 && M <= MAXT2, "q-shift lengths the level-2 march is built for");
};

template <int M, int WPS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, WPS))) k_fwd2m(const Fwd12pParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = Fwd2m<M>;
    constexpr int HL = G::HL, HL2 = G::HL2, VL = G::VL;
    __shared__ __attribute__((aligned(16))) f4 slab2[64 * 3 + 3 * G::HL + 8];
    __shared__ __attribute__((aligned(16))) f4 xbuf[2][2][64 + 2 * G::HL2];
    const int lane = threadIdx.x;
    int strip, band, b;
    if (!dtm_job(p.jb, blockIdx.x, strip, band, b)) return;
    const int R = p.R, C = p.C;
    const int nv = (C - strip * (4 * VL)) / 4 < VL ? (C - strip * (4 * VL)) / 4 : VL;     // owning lanes
    const int64_t img = (int64_t)b * R * C;
    const int rb = band * p.jb.band_rows;
    const int nrow = R - rb < p.jb.band_rows ? R - rb : p.jb.band_rows;
    const int rbase = rb - G::PRE;                     // first LoLo1 row the band's windows want (% 4 == 0)
    const int nst = (nrow / 2 + G::PRE + 1) / 2 * 2;   // an even count: the step loop runs in pairs
    const int c0 = strip * (4 * VL) - 4 * HL + 4 * lane;
    const bool rev = c0 < 0 || c0 >= C;
    int lc = c0 < 0 ? -c0 - 4 : (c0 >= C ? 2 * C - 4 - c0 : c0);
    lc = lc < 0 ? 0 : (lc > C - 4 ? C - 4 : lc);
    const bool edge_strip = strip == 0 || (strip + 1) * (4 * VL) + 4 * HL >= C;
    const DtBuf bx = dt_buf2g(p.X + img);
    const unsigned pitch = (unsigned)C * 4u;
    const int last_row = rbase + 2 * (nrow / 2 + G::PRE) - 1;
    auto ldrow = [&](int u) -> f4 {
        u = u > last_row ? last_row : u;
        u = u < 0 ? -1 - u : u;
        u = u >= R ? 2 * R - 1 - u : u;
        return dt2d::dt_buf_ld4(bx, (unsigned)lc * 4u, (unsigned)u * pitch);
    };
    f4 pre[4];          // the rows of steps t and t + 1: [2 (t & 1) + q]
#pragma unroll
    for (int i = 0; i < 4; ++i) pre[i] = ldrow(rbase + i);
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(pre[i].x), "+v"(pre[i].y), "+v"(pre[i].z), "+v"(pre[i].w) : : "memory");
    fwd2_wave<M, HL>(p, lane, strip, img, nv, rb, nrow, rbase, nst, xbuf, slab2, [&](int t, int k) {
        f4 e0 = pre[2 * k], e1 = pre[2 * k + 1];
        pre[2 * k] = ldrow(rbase + 2 * t + 4);
        pre[2 * k + 1] = ldrow(rbase + 2 * t + 5);
        if (edge_strip) { e0 = rev ? rev4(e0) : e0; e1 = rev ? rev4(e1) : e1; }       // mirrored lanes: the mirror block in reverse
        xbuf[k][0][HL2 + lane] = e0;
        xbuf[k][1][HL2 + lane] = e1;
        DT_WAVE_LDS_SYNC();
    });
#endif
}

}  // namespace dtm
