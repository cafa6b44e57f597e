// Levels 2 + 1 of the float32 2-D inverse DT-CWT as a marching PAIR of wavefronts (gfx950): k_inv21p -- for the 14- / 18-tap
// q-shift sets (qshift_b, qshift_d) with the short level-1 synthesis filters (near_sym_a 7 / 5, legall 3 / 5).
//
// k_inv21m (march2d.hpp) keeps the level-2 pending groups (M / 2 x 16 registers: 80 for qshift_a) AND the level-1 pending rows,
// record rows in flight and slab traffic in one wavefront; with 112 / 144 registers of pending groups that does not fit two
// wavefronts to a SIMD, and one to a SIMD costs more in flight than the fusion saves (DESIGN section 4 "Round 5" item 2).  Here,
// as in the forward's k_fwd12p (march2d_pair.hpp), a workgroup is two wavefronts on the same strip and band:
//   wavefront 0   level 2, the first half of a k_inv21m macro-step for any M with (M - 2) % 4 == 0: c2q of the lane's Yh[1] record,
//                 the row interpolation over the lane's two columns + (M - 2) / 4 lanes either side (DPP chains), the column
//                 interpolation in transposed form into M / 2 pending groups of four Z1 rows.  The completed group does not go to
//                 memory: it goes into a double-buffered LDS exchange (2 x 4 x 1 KiB), one LDS-only s_barrier per macro-step.
//   wavefront 1   level 1, the second half of the macro-step unchanged: two Yh[0] record rows per half-step through the slab,
//                 c2q with gains, row filters with a one-lane DPP halo, transposed column filters into the pending rows of X,
//                 16-byte stores.  Its Z1 rows come from the exchange.
// The level-2 lowpass Z1 is never written: 20 B/px like k_inv21m, instead of the 28 B/px of a level-2 launch + a level-1 launch.
// Only the standard phases (sum(g0a g0b) > 0 > sum(g1a g1b): every shipped q-shift set), as k_inv21m.
//
// Reference: dtcwt/numpy/transform2d.py:242-293; colifilt dtcwt/numpy/lowlevel.py:156-260; c2q transform2d.py:324-350.
#pragma once
#include "march2d_pair.hpp"

namespace dtm {

template <int M0, int M1, int M>
struct Inv21p {
    static constexpr int H0 = M0 / 2, H1 = M1 / 2;
    static constexpr int HL1 = 1, HL2 = (M - 2) / 4, HL = HL1 + HL2;
    static constexpr int VL = 64 - 2 * HL;
    static constexpr int NG = M / 2;              // pending groups of four Z1 rows = lane pairs of a row window
    static constexpr int NPX = M0 > M1 + 2 ? M0 + 1 : M1 + 3;     // pending rows of X
    static_assert((M - 2) % 4 == 0 && M >= 10 && M <= MAXT2 && H0 <= 4 && H1 <= 4 && M0 % 2 == 1 && M1 % 2 == 1, "filters the marching inverse pair is built for");
    static_assert(M0 == M1 + 2, "pending-row bookkeeping assumes len(g0o) = len(g1o) + 2");
};

#if defined(__HIP_DEVICE_COMPILE__)
// the window of a half-resolution plane row as lane pairs: W[i] = the (even, odd) samples of lane l - HL2 + i
template <int HL2>
__device__ __forceinline__ void win_pairs(float v0, float v1, pk2 (&W)[2 * HL2 + 1]) {
    W[HL2] = pk2{v0, v1};
    float l0 = v0, l1 = v1, r0 = v0, r1 = v1;
#pragma unroll
    for (int s = 1; s <= HL2; ++s) {
        l0 = dpp_from_left(l0); l1 = dpp_from_left(l1); r0 = dpp_from_right(r0); r1 = dpp_from_right(r1);
        W[HL2 - s] = pk2{l0, l1}; W[HL2 + s] = pk2{r0, r1};
    }
}
// ifilt_row2 of march2d.hpp for a window of NW lane pairs
template <bool POS, int NW>
__device__ __forceinline__ void ifilt_row_n(const pk2 (&W)[NW], const pk2 *ha2, const pk2 *hb2, pk2 &E, pk2 &O) {
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        if (POS) { E += ha2[k] * DTM_BX(W[NW - 1 - k]); O += hb2[k] * DTM_BY(W[NW - 1 - k]); }
        else     { E += ha2[k] * DTM_BY(W[NW - 1 - k]); O += hb2[k] * DTM_BX(W[NW - 1 - k]); }
    }
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// The level-2 wavefront: the first half of a k_inv21m macro-step for any M with (M - 2) % 4 == 0, on the strip whose lane 0 sits
// HL lanes left of its first owned column; macro-steps nfirst .. nfirst + nms - 1 (one pair of Z2 rows + one Yh[1] record row each).
// After macro-step ms (pair n) the group n - HL2 of four Z1 rows is complete: sink(ms, n, E, O) receives it -- rows (0, 2) of column c
// in E[c], rows (1, 3) in O[c] -- and the groups move up by one at the first touch of the next macro-step.
template <int M, int HL, class Sink>
__device__ __forceinline__ void inv2_wave(const Inv21mParams &p, int lane, int strip, int b, int nfirst, int nms, Sink &&sink) {
    constexpr int HL2 = (M - 2) / 4, NG = M / 2, VL = 64 - 2 * HL;
    const int R = p.R, C = p.C;
    const int cb = strip * (4 * VL) - 4 * HL;             // column of lane 0
    const int c0 = cb + 4 * lane;
    const bool mir = c0 < 0 || c0 >= C;
    const bool edge_strip = cb < 0 || cb + 256 > C;      // uniform: some lane is mirrored
    const int64_t img = (int64_t)b * R * C;
    int zl = c0 < 0 ? -2 - c0 / 2 : (c0 >= C ? C - 2 - c0 / 2 : c0 / 2);
    zl = zl < 0 ? 0 : (zl > C / 2 - 2 ? C / 2 - 2 : zl);
    int ql = c0 < 0 ? -1 - c0 / 4 : (c0 >= C ? C / 2 - 1 - c0 / 4 : c0 / 4);
    ql = ql < 0 ? 0 : (ql > C / 4 - 1 ? C / 4 - 1 : ql);
    const DtBuf bz = dt_buf2g(p.Z2 + img / 4);
    const DtBuf b2 = dt_buf2g(p.Yh1 + (img / 4) * 3);
    const unsigned zpitch = (unsigned)C * 2u, r2pitch = (unsigned)C * 12u;          // bytes per Z2 row, per Yh1 record row
    auto zrow = [&](int u) { u = u < 0 ? -1 - u : u; u = u >= R / 2 ? R - 1 - u : u; return u < 0 ? 0 : (u > R / 2 - 1 ? R / 2 - 1 : u); };
    auto pair_row = [&](int n, bool &sw) { sw = n < 0 || n >= R / 4; n = n < 0 ? -1 - n : n; n = n >= R / 4 ? R / 2 - 1 - n : n; return n < 0 ? 0 : (n > R / 4 - 1 ? R / 4 - 1 : n); };
    dt2d::f2 z2p[2];
    f4 r2p[3];
    auto request = [&](int n) {
        bool sw;
        z2p[0] = dt2d::dt_buf_ld2(bz, (unsigned)zl * 4u, (unsigned)zrow(2 * n) * zpitch);
        z2p[1] = dt2d::dt_buf_ld2(bz, (unsigned)zl * 4u, (unsigned)zrow(2 * n + 1) * zpitch);
        const unsigned ro2 = (unsigned)pair_row(n, sw) * r2pitch;
#pragma unroll
        for (int m = 0; m < 3; ++m) r2p[m] = dt2d::dt_buf_ld4(b2, (unsigned)ql * 48u + 16u * m, ro2);
    };
    request(nfirst);
    asm volatile("" : "+v"(z2p[0].x), "+v"(z2p[0].y), "+v"(z2p[1].x), "+v"(z2p[1].y) : : "memory");
#pragma unroll
    for (int m = 0; m < 3; ++m) asm volatile("" : "+v"(r2p[m].x), "+v"(r2p[m].y), "+v"(r2p[m].z), "+v"(r2p[m].w) : : "memory");

    // pending Z1 groups: PzE[a][c] = rows (0, 2), PzO[a][c] = rows (1, 3) of group slot a, column c
    pk2 PzE[NG][4], PzO[NG][4];
#pragma unroll
    for (int a = 0; a < NG; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) { PzE[a][c] = pk2{0.f, 0.f}; PzO[a][c] = pk2{0.f, 0.f}; }
    const pk2 *la2 = reinterpret_cast<const pk2 *>(p.l_a), *lb2 = reinterpret_cast<const pk2 *>(p.l_b);
    const pk2 *ha2 = reinterpret_cast<const pk2 *>(p.h_a), *hb2 = reinterpret_cast<const pk2 *>(p.h_b);

    for (int ms = 0; ms < nms; ++ms) {
        const int n = nfirst + ms;
        bool sw2;
        (void)pair_row(n, sw2);
        float z[2][2], p05[2][2], p23[2][2], p14[2][2];
        {
            const f4 ra = r2p[0], rc = r2p[1], re = r2p[2];
            c2q_quad(ra.x, ra.y, re.z, re.w, p.g2[0], p.g2[5], p05);
            c2q_quad(rc.x, rc.y, rc.z, rc.w, p.g2[2], p.g2[3], p23);
            c2q_quad(ra.z, ra.w, re.x, re.y, p.g2[1], p.g2[4], p14);
            z[0][0] = z2p[0].x; z[0][1] = z2p[0].y; z[1][0] = z2p[1].x; z[1][1] = z2p[1].y;
            if (sw2) {              // a reflected record row: its quads upside down
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    float t_;
                    t_ = p05[0][f]; p05[0][f] = p05[1][f]; p05[1][f] = t_;
                    t_ = p23[0][f]; p23[0][f] = p23[1][f]; p23[1][f] = t_;
                    t_ = p14[0][f]; p14[0][f] = p14[1][f]; p14[1][f] = t_;
                }
            }
            if (edge_strip) {       // mirrored lanes: the mirror lane's two columns in reverse
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float t_;
                    t_ = z[e][0]; z[e][0] = mir ? z[e][1] : t_; z[e][1] = mir ? t_ : z[e][1];
                    t_ = p05[e][0]; p05[e][0] = mir ? p05[e][1] : t_; p05[e][1] = mir ? t_ : p05[e][1];
                    t_ = p23[e][0]; p23[e][0] = mir ? p23[e][1] : t_; p23[e][1] = mir ? t_ : p23[e][1];
                    t_ = p14[e][0]; p14[e][0] = mir ? p14[e][1] : t_; p14[e][1] = mir ? t_ : p14[e][1];
                }
            }
        }
        request(n + 1);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            // u0 = R0 z + R1 p23, u1 = R0 p05 + R1 p14 as (u[0], u[2]) and (u[1], u[3])
            pk2 U0E = {0.f, 0.f}, U0O = {0.f, 0.f}, U1E = {0.f, 0.f}, U1O = {0.f, 0.f}, W[NG];
            win_pairs<HL2>(z[e][0], z[e][1], W);     ifilt_row_n<true, NG>(W, la2, lb2, U0E, U0O);
            win_pairs<HL2>(p23[e][0], p23[e][1], W); ifilt_row_n<false, NG>(W, ha2, hb2, U0E, U0O);
            win_pairs<HL2>(p05[e][0], p05[e][1], W); ifilt_row_n<true, NG>(W, la2, lb2, U1E, U1O);
            win_pairs<HL2>(p14[e][0], p14[e][1], W); ifilt_row_n<false, NG>(W, ha2, hb2, U1E, U1O);
#pragma unroll
            for (int a = 0; a < NG; ++a) {          // slot a = group n - HL2 + a, tap pair k = a
                if (e == 0) {       // the even row of the pair; the first touch of a slot also moves the groups up by one
                    if (a + 1 < NG) {
                        PzE[a][0] = la2[a] * DTM_BX(U0E) + PzE[a + 1][0]; PzE[a][2] = la2[a] * DTM_BY(U0E) + PzE[a + 1][2];
                        PzE[a][1] = la2[a] * DTM_BX(U0O) + PzE[a + 1][1]; PzE[a][3] = la2[a] * DTM_BY(U0O) + PzE[a + 1][3];
                        PzO[a][0] = hb2[a] * DTM_BX(U1E) + PzO[a + 1][0]; PzO[a][2] = hb2[a] * DTM_BY(U1E) + PzO[a + 1][2];
                        PzO[a][1] = hb2[a] * DTM_BX(U1O) + PzO[a + 1][1]; PzO[a][3] = hb2[a] * DTM_BY(U1O) + PzO[a + 1][3];
                    } else {
                        PzE[a][0] = la2[a] * DTM_BX(U0E); PzE[a][2] = la2[a] * DTM_BY(U0E);
                        PzE[a][1] = la2[a] * DTM_BX(U0O); PzE[a][3] = la2[a] * DTM_BY(U0O);
                        PzO[a][0] = hb2[a] * DTM_BX(U1E); PzO[a][2] = hb2[a] * DTM_BY(U1E);
                        PzO[a][1] = hb2[a] * DTM_BX(U1O); PzO[a][3] = hb2[a] * DTM_BY(U1O);
                        // the last slot starts from a PRODUCT, and the odd row adds another product to it: pinned as a rounded product here, so that
                        // the sum is fma(odd-row product's factors, this) in every kernel that inlines these lines (left alone, WHICH of the two
                        // products is rounded first differed between k_inv21m and the marching pair: their outputs disagreed in the last bit)
#ifndef DTM_NO_PIN      /* (A/B builds only: tools/build_variant.sh) */
#pragma unroll
                        for (int c = 0; c < 4; ++c) { asm("" : "+v"(PzE[a][c])); asm("" : "+v"(PzO[a][c])); }
#endif
                    }
                } else {
                    PzO[a][0] += lb2[a] * DTM_BX(U0E); PzO[a][2] += lb2[a] * DTM_BY(U0E);
                    PzO[a][1] += lb2[a] * DTM_BX(U0O); PzO[a][3] += lb2[a] * DTM_BY(U0O);
                    PzE[a][0] += ha2[a] * DTM_BX(U1E); PzE[a][2] += ha2[a] * DTM_BY(U1E);
                    PzE[a][1] += ha2[a] * DTM_BX(U1O); PzE[a][3] += ha2[a] * DTM_BY(U1O);
                }
            }
        }
        // group n - HL2 (Z1 rows 4j .. 4j + 3) is complete in slot 0
        sink(ms, n, PzE[0], PzO[0]);
    }
}
#endif

// WPS: wavefronts per SIMD the registers are allocated for (162 VGPRs at M = 14: three fit, i.e. six pairs per CU)
template <int M0, int M1, int M, int WPS = 2>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, WPS))) k_inv21p(const Inv21mParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = Inv21p<M0, M1, M>;
    constexpr int H0 = G::H0, H1 = G::H1, HL = G::HL, HL2 = G::HL2, VL = G::VL, NPX = G::NPX;
    __shared__ __attribute__((aligned(16))) f4 slab[2][64 * 6 + 6 * G::HL + 8];
    __shared__ __attribute__((aligned(16))) f4 xb[2][4][64];          // [macro-step parity][row of the group][lane]
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int strip, band, b;
    if (!dtm_job(p.jb, blockIdx.x, strip, band, b)) return;
    const int R = p.R, C = p.C;
    const int cb = strip * (4 * VL) - 4 * HL;             // column of lane 0
    const int c0 = cb + 4 * lane;
    const bool mir = c0 < 0 || c0 >= C;
    const bool edge_strip = cb < 0 || cb + 256 > C;      // uniform: some lane is mirrored
    const int nv = (C - strip * (4 * VL)) / 4 < VL ? (C - strip * (4 * VL)) / 4 : VL;
    const int64_t img = (int64_t)b * R * C;
    const int rb = band * p.jb.band_rows;
    const int nrow = R - rb < p.jb.band_rows ? R - rb : p.jb.band_rows;
    const int j0 = rb / 4 - 1, j1 = (rb + nrow) / 4;      // groups of Z1 rows level 1 reads
    const int nfirst = j0 - HL2, nms = j1 - j0 + 2 * HL2 + 1;       // level-2 pairs j0 - HL2 .. j1 + HL2

    if (role == 0) {
        // ------------------------------------------------------------------ level 2: the completed group goes into the exchange
        inv2_wave<M, HL>(p, lane, strip, b, nfirst, nms, [&](int ms, int, const pk2 (&E)[4], const pk2 (&O)[4]) {
            f4 (*xo)[64] = xb[ms & 1];
            xo[0][lane] = f4{E[0].x, E[1].x, E[2].x, E[3].x};
            xo[1][lane] = f4{O[0].x, O[1].x, O[2].x, O[3].x};
            xo[2][lane] = f4{E[0].y, E[1].y, E[2].y, E[3].y};
            xo[3][lane] = f4{O[0].y, O[1].y, O[2].y, O[3].y};
            DTM_PAIR_BARRIER();                      // the group is wavefront 1's now
        });
    } else {
        // ------------------------------------------------------------------ level 1 (the second half of a k_inv21m macro-step)
        int sl = c0 < 0 ? (-c0 - 4 - cb) / 4 : (c0 >= C ? (2 * C - 4 - c0 - cb) / 4 : lane);
        sl = sl < 0 ? 0 : (sl > 63 ? 63 : sl);
        // the level-1 record pieces of a row this wavefront fetches: lanes lmin .. lmax are inside the image
        const int lmin = cb < 0 ? -cb / 4 : 0, lmax = (C - cb) / 4 - 1 < 63 ? (C - cb) / 4 - 1 : 63;
        const float *const Y0b = p.Yh0 + img * 3 + (int64_t)(cb + 4 * lmin) * 6;       // record of lane lmin in row 0
        float *const Xb = p.X + img + strip * (4 * VL);
        const unsigned r1bytes = (unsigned)(lmax - lmin + 1) * 96u;
        auto rec_row = [&](int rr, bool &sw) { sw = rr < 0 || rr >= R / 2; rr = rr < 0 ? -1 - rr : rr; rr = rr >= R / 2 ? R - 1 - rr : rr; return rr < 0 ? 0 : (rr > R / 2 - 1 ? R / 2 - 1 : rr); };
        float g0o[M0], g1o[M1];
#pragma unroll
        for (int k = 0; k < M0; ++k) g0o[k] = p.gd0[2 * (k < H0 ? H0 - k : k - H0)];
#pragma unroll
        for (int k = 0; k < M1; ++k) g1o[k] = p.gd1[2 * (k < H1 ? H1 - k : k - H1)];
        f4 r1p[2][6];
        auto request = [&](int n) {         // the record rows of group n - HL2; against zero bytes where level 1 does not run
            bool sw;
            const bool gok = n - HL2 >= j0 && n - HL2 <= j1;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int rr = rec_row(2 * (n - HL2) + e, sw);
                const DtBuf br = dt_buf_n(Y0b + (int64_t)rr * C * 6, gok ? r1bytes : 0u);
#pragma unroll
                for (int m = 0; m < 6; ++m) r1p[e][m] = dt2d::dt_buf_ld4(br, 16u * (unsigned)lane + 1024u * m, 0u);
            }
        };
        request(nfirst);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int m = 0; m < 6; ++m) asm volatile("" : "+v"(r1p[e][m].x), "+v"(r1p[e][m].y), "+v"(r1p[e][m].z), "+v"(r1p[e][m].w) : : "memory");
        float PX[NPX][4];
#pragma unroll
        for (int i = 0; i < NPX; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) PX[i][c] = 0.f;
        const unsigned xv = 16u * (unsigned)(lane - HL);
        const pk2 *gd0 = reinterpret_cast<const pk2 *>(p.gd0), *gd1 = reinterpret_cast<const pk2 *>(p.gd1);

        for (int ms = 0; ms < nms; ++ms) {
            const int n = nfirst + ms, j = n - HL2;
            bool sw1[2];
            (void)rec_row(2 * j, sw1[0]); (void)rec_row(2 * j + 1, sw1[1]);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int m = 0; m < 6; ++m) slab[e][6 * lmin + lane + 64 * m] = r1p[e][m];
            request(n + 1);
            DTM_PAIR_BARRIER();                      // group j is in xb[ms & 1] (and this wavefront's slab writes have landed)
            const bool grp = j >= j0 && j <= j1;           // uniform
            f4 zg[4];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) zg[r4] = xb[ms & 1][r4][lane];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rho = 4 * j + 2 * h;
                if (grp) {
                    float q05[2][4], q23[2][4], q14[2][4];
                    {
                        const f4 *sp = slab[h] + 6 * sl;
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
                        if (sw1[h]) {               // a reflected record row (image top / bottom): its quads upside down
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                float t_;
                                t_ = q05[0][c]; q05[0][c] = q05[1][c]; q05[1][c] = t_;
                                t_ = q23[0][c]; q23[0][c] = q23[1][c]; q23[1][c] = t_;
                                t_ = q14[0][c]; q14[0][c] = q14[1][c]; q14[1][c] = t_;
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            if (edge_strip) {       // mirrored lanes: the mirror lane's four columns in reverse
                                float t_;
#define DTM_REV4(x_) t_ = x_[e][0]; x_[e][0] = mir ? x_[e][3] : t_; x_[e][3] = mir ? t_ : x_[e][3]; \
                     t_ = x_[e][1]; x_[e][1] = mir ? x_[e][2] : t_; x_[e][2] = mir ? t_ : x_[e][2];
                                DTM_REV4(q05) DTM_REV4(q23) DTM_REV4(q14)
#undef DTM_REV4
                            }
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        // (Z1, q05) through g0o and (q23, q14) through g1o, the filters symmetric: V[c] = (v0[c], v1[c])
                        const f4 &zr = zg[2 * h + e];
                        const float zv[4] = {zr.x, zr.y, zr.z, zr.w};
                        pk2 Wa[4 + 2 * H0], Wb[4 + 2 * H1], V[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            Wa[H0 + c] = pk2{zv[c], q05[e][c]};
                            Wb[H1 + c] = pk2{q23[e][c], q14[e][c]};
                        }
#pragma unroll
                        for (int i = 0; i < H0; ++i) {
                            Wa[i] = pk2{dpp_from_left(Wa[4 + i].x), dpp_from_left(Wa[4 + i].y)};
                            Wa[H0 + 4 + i] = pk2{dpp_from_right(Wa[H0 + i].x), dpp_from_right(Wa[H0 + i].y)};
                        }
#pragma unroll
                        for (int i = 0; i < H1; ++i) {
                            Wb[i] = pk2{dpp_from_left(Wb[4 + i].x), dpp_from_left(Wb[4 + i].y)};
                            Wb[H1 + 4 + i] = pk2{dpp_from_right(Wb[H1 + i].x), dpp_from_right(Wb[H1 + i].y)};
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            pk2 a_ = gd0[0] * Wa[H0 + c];
#pragma unroll
                            for (int d = 1; d <= H0; ++d) a_ += gd0[d] * (Wa[H0 + c - d] + Wa[H0 + c + d]);
                            a_ += gd1[0] * Wb[H1 + c];
#pragma unroll
                            for (int d = 1; d <= H1; ++d) a_ += gd1[d] * (Wb[H1 + c - d] + Wb[H1 + c + d]);
                            V[c] = a_;
                        }
                        const float v0[4] = {V[0].x, V[1].x, V[2].x, V[3].x}, v1[4] = {V[0].y, V[1].y, V[2].y, V[3].y};
                        // columns, transposed: PX[i] is row rho - H0 + i; the first row of a half-step also moves the pending rows up by two
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (e == 0) {
#pragma unroll
                                for (int i = 0; i < NPX; ++i) {
                                    float acc;
                                    if (i + 2 < NPX && i < M0) {
                                        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(acc) : "s"(g0o[i]), "v"(v0[c]), "v"(PX[i + 2][c]));
                                    } else {
                                        acc = i + 2 < NPX ? PX[i + 2][c] : 0.f;
                                        if (i < M0) acc += g0o[i] * v0[c];
                                    }
                                    if (i >= H0 - H1 && i - (H0 - H1) < M1) acc += g1o[i - (H0 - H1)] * v1[c];
                                    PX[i][c] = acc;
                                }
                            } else {
#pragma unroll
                                for (int k = 0; k < M0; ++k) PX[e + k][c] += g0o[k] * v0[c];
#pragma unroll
                                for (int k = 0; k < M1; ++k) PX[e + (H0 - H1) + k][c] += g1o[k] * v1[c];
                            }
                        }
                    }
                }
                // rows rho - H0, rho - H0 + 1 are complete
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int x = rho - H0 + e;
                    const bool ok = x >= rb && x < rb + nrow;
                    const int xo = ok ? x : 0;
                    const DtBuf bo = dt_buf_n(Xb + (int64_t)xo * C, ok ? 16u * nv : 0u);
                    dt2d::dt_buf_st4<true>(bo, xv, 0u, f4{PX[e][0], PX[e][1], PX[e][2], PX[e][3]});
                }
            }
            DT_WAVE_LDS_SYNC();
        }
    }
#endif
}

// ======================================================================================================================
// Level 2 of the inverse ALONE as a march (transform2d.py:242-273 at the second level): the level-2 wavefront of k_inv21p with its
// completed groups of four Z1 rows stored instead of handed over -- for the sets whose level 1 no pair takes (near_sym_b's 19 / 13-tap
// synthesis filters run as k_inv1m, march2d_l1.hpp).  Z2 [B][R/2][C/2] + Yh[1] -> Z1 = p.X [B][R][C]; no level-1 halo lane: HL = HL2.
// Per pixel of Z1: 1 B of Z2 + 3 B of records read, 4 B written -- the bytes of the tile program k_inv2, in one pass without barriers.
// ======================================================================================================================
template <int M>
struct Inv2m {
    static constexpr int HL2 = (M - 2) / 4, HL = HL2, VL = 64 - 2 * HL;
    static_assert((M - 2) % 4 == 0 && M >= 10 && M <= MAXT2, "q-shift lengths the level-2 march is built for");
};

template <int M, int WPS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, WPS))) k_inv2m(const Inv21mParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = Inv2m<M>;
    constexpr int HL = G::HL, HL2 = G::HL2, VL = G::VL;
    const int lane = threadIdx.x;
    int strip, band, b;
    if (!dtm_job(p.jb, blockIdx.x, strip, band, b)) return;
    const int R = p.R, C = p.C;
    const int nv = (C - strip * (4 * VL)) / 4 < VL ? (C - strip * (4 * VL)) / 4 : VL;
    const int rb = band * p.jb.band_rows;
    const int nrow = R - rb < p.jb.band_rows ? R - rb : p.jb.band_rows;
    const int g0 = rb / 4, g1 = (rb + nrow) / 4 - 1;     // the band's groups of four Z1 rows
    float *const Xb = p.X + (int64_t)b * R * C + strip * (4 * VL);
    const unsigned xv = 16u * (unsigned)(lane - HL);
    inv2_wave<M, HL>(p, lane, strip, b, g0 - HL2, g1 - g0 + 2 * HL2 + 1, [&](int, int n, const pk2 (&E)[4], const pk2 (&O)[4]) {
        // plain stores: the level-1 launch reads these rows next
        const int j = n - HL2;
        const bool ok = j >= g0 && j <= g1;            // uniform; otherwise the stores are dropped
        const int jo = ok ? j : g0;
        const f4 rows4[4] = {f4{E[0].x, E[1].x, E[2].x, E[3].x}, f4{O[0].x, O[1].x, O[2].x, O[3].x},
                             f4{E[0].y, E[1].y, E[2].y, E[3].y}, f4{O[0].y, O[1].y, O[2].y, O[3].y}};
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const DtBuf bo = dt_buf_n(Xb + (int64_t)(4 * jo + r4) * C, ok ? 16u * nv : 0u);
            dt2d::dt_buf_st4<false>(bo, xv, 0u, rows4[r4]);
        }
    });
#endif
}

}  // namespace dtm
