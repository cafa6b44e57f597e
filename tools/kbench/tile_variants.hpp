// Earlier generations and experimental variants of the 2-D tile programs.  They are NOT part
// of libdtcwt_hip.so: tools/kbench/kbench.hip times them against the product kernels
// (profiles/r01/kbench*.txt record what each of them measured and why it was dropped).
#pragma once
#include "fused2d_tiles.hpp"
#include "fused2d_tiles_v2.hpp"

namespace dt2d {

// ======================================================================================
// first generation: window staged in LDS before the column pass
// ======================================================================================
template <int TR_, int TC_, int M0_, int M1_>
struct Fwd1Cfg {
    static constexpr int TR = TR_, TC = TC_, M0 = M0_, M1 = M1_;
    static constexpr int H0 = M0 / 2, H1 = M1 / 2, HH = cmax(H0, H1);
    static constexpr int HC = (HH + 1) & ~1;          // even column halo: aligned float2 windows
    static constexpr int W = TC + 2 * HC;             // LDS row length (even)
    static constexpr int NR = TR + 2 * HH;            // window rows
    static constexpr int RS = 8;                      // rows per column-pass strip
    static constexpr int SX = NR * W, SL = TR * W;
    static constexpr int LDS_FLOATS = SX + 2 * SL;
    static_assert(TR % RS == 0 && TR % 2 == 0 && TC % 2 == 0, "tile shape");
    static_assert(M0 % 2 == 1 && M1 % 2 == 1, "biort filters must have odd length");
};

template <class C>
DT_HD void fwd1_load(const Fwd1Params &p, float *sx, int tid, int b, int r0, int c0) {
    const float *Xb = p.X + (int64_t)b * p.inR * p.inC;
    const int ro = r0 - C::HH, co = c0 - C::HC;
    const bool interior = ro >= 0 && ro + C::NR <= p.inR && co >= 0 && co + C::W <= p.inC;
    for (int e = tid; e < C::SX; e += DT_NT) {
        int rr = e / C::W, cc = e - rr * C::W;
        int gr = ro + rr, gc = co + cc;
        if (!interior) {
            gr = reflect_i(gr, p.LR); if (gr > p.inR - 1) gr = p.inR - 1;
            gc = reflect_i(gc, p.LC); if (gc > p.inC - 1) gc = p.inC - 1;
        }
        sx[e] = Xb[(int64_t)gr * p.inC + gc];
    }
}

// Lo[r] = sum_k h0[k] X[r + H0 - k],  Hi[r] = sum_k h1[k] X[r + H1 - k]   (A.1, m odd)
template <class C>
DT_HD void fwd1_cols(const Fwd1Params &p, const float *sx, float *sLo, float *sHi, int tid) {
    constexpr int NS = C::TR / C::RS;
    for (int task = tid; task < NS * C::W; task += DT_NT) {
        int strip = task / C::W, cc = task - strip * C::W;
        float w[C::RS + 2 * C::HH];
#pragma unroll
        for (int j = 0; j < C::RS + 2 * C::HH; ++j) w[j] = sx[(strip * C::RS + j) * C::W + cc];
#pragma unroll
        for (int q = 0; q < C::RS; ++q) {
            float lo = 0.f, hi = 0.f;
#pragma unroll
            for (int k = 0; k < C::M0; ++k) lo += p.h0[k] * w[q + C::HH + C::H0 - k];
#pragma unroll
            for (int k = 0; k < C::M1; ++k) hi += p.h1[k] * w[q + C::HH + C::H1 - k];
            sLo[(strip * C::RS + q) * C::W + cc] = lo;
            sHi[(strip * C::RS + q) * C::W + cc] = hi;
        }
    }
}

template <class C>
DT_HD void fwd1_rows(const Fwd1Params &p, const float *sLo, const float *sHi, int tid, int b,
                     int r0, int c0) {
    constexpr int NV = C::TC / 2, NU = C::TR / 2;
    constexpr int WL = 2 * C::HC + 2;             // window length (even)
    const int HR = p.LR / 2, HCc = p.LC / 2;
    for (int task = tid; task < NU * NV; task += DT_NT) {
        int u = task / NV, v = task - u * NV;
        int R = r0 + 2 * u, Cc = c0 + 2 * v;
        if (R >= p.LR || Cc >= p.LC) continue;
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
#pragma unroll
            for (int ec = 0; ec < 2; ++ec) {
                float s_ll = 0.f, s_hl = 0.f, s_lh = 0.f, s_hh = 0.f;
#pragma unroll
                for (int k = 0; k < C::M0; ++k) {
                    s_ll += p.h0[k] * wl[ec + C::HC + C::H0 - k];
                    s_hl += p.h0[k] * wh[ec + C::HC + C::H0 - k];
                }
#pragma unroll
                for (int k = 0; k < C::M1; ++k) {
                    s_lh += p.h1[k] * wl[ec + C::HC + C::H1 - k];
                    s_hh += p.h1[k] * wh[ec + C::HC + C::H1 - k];
                }
                ll[er][ec] = s_ll; hl[er][ec] = s_hl; lh[er][ec] = s_lh; hh[er][ec] = s_hh;
            }
        }
        float *L = p.LoLo + ((int64_t)b * p.LR + R) * p.LC + Cc;
        *reinterpret_cast<f2 *>(L) = f2{ll[0][0], ll[0][1]};
        *reinterpret_cast<f2 *>(L + p.LC) = f2{ll[1][0], ll[1][1]};
        float *rec = p.Yh + (((int64_t)b * HR + R / 2) * HCc + Cc / 2) * 12;
        store_record(rec, hl, lh, hh);
    }
}

template <int TR_, int TC_, int M_>
struct Fwd2Cfg {
    static constexpr int TR = TR_, TC = TC_, M = M_;       // TR x TC outputs of LoLo'
    static constexpr int TI = TR / 2, TJ = TC / 2;         // (A,B) pairs per axis
    static constexpr int NRI = 2 * TR + 2 * M - 4;         // input window rows
    static constexpr int NCI = 2 * TC + 2 * M - 4;         // input window cols (% 4 == 0)
    static constexpr int SX = NRI * NCI, SL = TR * NCI;
    static constexpr int LDS_FLOATS = SX + 2 * SL;
    static_assert(M % 2 == 0 && TR % 2 == 0 && TC % 2 == 0, "even taps / tile");
};

template <class C>
DT_HD void fwd2_load(const Fwd2Params &p, float *sx, int tid, int b, int r0, int c0) {
    // output tile origin (r0, c0) in LoLo' coordinates -> pair index i0 = r0/2 -> logical
    // input rows start at 4 i0 - M + 2 = 2 r0 - M + 2.
    const float *Xb = p.X + (int64_t)b * p.inR * p.inC;
    const int ro = 2 * r0 - C::M + 2, co = 2 * c0 - C::M + 2;
    const bool interior = ro - p.padR >= 0 && ro + C::NRI - p.padR <= p.inR &&
                          co - p.padC >= 0 && co + C::NCI - p.padC <= p.inC;
    for (int e = tid; e < C::SX; e += DT_NT) {
        int rr = e / C::NCI, cc = e - rr * C::NCI;
        int gr = ro + rr, gc = co + cc;
        if (interior) {
            gr -= p.padR; gc -= p.padC;
        } else {
            gr = clamp_i(reflect_i(gr, p.LR) - p.padR, 0, p.inR - 1);
            gc = clamp_i(reflect_i(gc, p.LC) - p.padC, 0, p.inC - 1);
        }
        sx[e] = Xb[(int64_t)gr * p.inC + gc];
    }
}

template <class C>
DT_HD void fwd2_cols(const Fwd2Params &p, const float *sx, float *sLo, float *sHi, int tid) {
    for (int task = tid; task < C::TI * C::NCI; task += DT_NT) {
        int il = task / C::NCI, cc = task - il * C::NCI;
        float w[2 * C::M];
#pragma unroll
        for (int j = 0; j < 2 * C::M; ++j) w[j] = sx[(4 * il + j) * C::NCI + cc];
        float A, Bv;
        dfilt_pair<C::M>(w, p.l_a, p.l_b, A, Bv);
        sLo[(2 * il) * C::NCI + cc] = p.lo_a_first ? A : Bv;
        sLo[(2 * il + 1) * C::NCI + cc] = p.lo_a_first ? Bv : A;
        dfilt_pair<C::M>(w, p.h_a, p.h_b, A, Bv);
        sHi[(2 * il) * C::NCI + cc] = p.hi_a_first ? A : Bv;
        sHi[(2 * il + 1) * C::NCI + cc] = p.hi_a_first ? Bv : A;
    }
}

template <class C>
DT_HD void fwd2_rows(const Fwd2Params &p, const float *sLo, const float *sHi, int tid, int b,
                     int r0, int c0) {
    const int OR = p.LR / 2, OC = p.LC / 2;       // LoLo' size
    const int HR = OR / 2, HCc = OC / 2;          // Yh size
    for (int task = tid; task < C::TI * C::TJ; task += DT_NT) {
        int il = task / C::TJ, jl = task - il * C::TJ;
        int R = r0 + 2 * il, Cc = c0 + 2 * jl;
        if (R >= OR || Cc >= OC) continue;
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
            float A, Bv;
            dfilt_pair<C::M>(wl, p.l_a, p.l_b, A, Bv);      // Lo rows, lo filter -> LoLo'
            ll[er][0] = p.lo_a_first ? A : Bv; ll[er][1] = p.lo_a_first ? Bv : A;
            dfilt_pair<C::M>(wh, p.l_a, p.l_b, A, Bv);      // Hi rows, lo filter -> HL
            hl[er][0] = p.lo_a_first ? A : Bv; hl[er][1] = p.lo_a_first ? Bv : A;
            dfilt_pair<C::M>(wl, p.h_a, p.h_b, A, Bv);      // Lo rows, hi filter -> LH
            lh[er][0] = p.hi_a_first ? A : Bv; lh[er][1] = p.hi_a_first ? Bv : A;
            dfilt_pair<C::M>(wh, p.h_a, p.h_b, A, Bv);      // Hi rows, hi filter -> HH
            hh[er][0] = p.hi_a_first ? A : Bv; hh[er][1] = p.hi_a_first ? Bv : A;
        }
        float *L = p.LoLo + ((int64_t)b * OR + R) * OC + Cc;
        *reinterpret_cast<f2 *>(L) = f2{ll[0][0], ll[0][1]};
        *reinterpret_cast<f2 *>(L + OC) = f2{ll[1][0], ll[1][1]};
        float *rec = p.Yh + (((int64_t)b * HR + R / 2) * HCc + Cc / 2) * 12;
        store_record(rec, hl, lh, hh);
    }
}

// ======================================================================================
// Inverse: shared record -> quad-plane loader (c2q with gains, A.4)
// ======================================================================================
// Fills window quads of the three planes  s1 = c2q(Yh[...,[0,5]]) ("lh" in the reference:
// filtered with g1 down axis 0, g0 along axis 1), s2 = c2q(Yh[...,[2,3]]) ("hl"),
// s3 = c2q(Yh[...,[1,4]]) ("hh").  Window origin (ro, co) is even, window is NR x NC
// (both even), planes have row stride NC.  zr, zc: plane size (even).  g[6] already
// includes the sqrt(1/2) of c2q.
DT_HD void inv_load_quads(const float *Yhb, int zr, int zc, const float *g, float *s1, float *s2,
                          float *s3, int NR, int NC, int ro, int co, int tid) {
    const int QR = NR / 2, QC = NC / 2, hc = zc / 2;
    for (int q = tid; q < QR * QC; q += DT_NT) {
        int uw = q / QC, vw = q - uw * QC;
        int ra = reflect_i(ro + 2 * uw, zr), ca = reflect_i(co + 2 * vw, zc);
        int U = ra >> 1, fr = ra & 1, V = ca >> 1, fc = ca & 1;
        const f4 *rec = reinterpret_cast<const f4 *>(Yhb + ((int64_t)U * hc + V) * 12);
        f4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
        // slots: 0=(r0.x,r0.y) 1=(r0.z,r0.w) 2=(r1.x,r1.y) 3=(r1.z,r1.w) 4=(r2.x,r2.y) 5=(r2.z,r2.w)
        // c2q of one subband pair: a = Re P, b = Im P, c = Im Q, d = -Re Q with
        // P = w0 + w1, Q = w0 - w1 (gains and sqrt(1/2) folded into g).  The window may
        // be a mirrored copy of the quad: swap rows when fr, columns when fc (selects,
        // no runtime-indexed arrays: those would live in scratch).
        int base = (2 * uw) * NC + 2 * vw;
#define DT_QUAD(S, W0R, W0I, W1R, W1I, G0, G1)                                     \
        {                                                                           \
            float ar = (W0R) * (G0), ai = (W0I) * (G0), br = (W1R) * (G1), bi = (W1I) * (G1); \
            float qa = ar + br, qb = ai + bi, qc = ai - bi, qd = -(ar - br);        \
            float t0 = fr ? qc : qa, t1 = fr ? qd : qb;                             \
            float b0 = fr ? qa : qc, b1 = fr ? qb : qd;                             \
            *reinterpret_cast<f2 *>((S) + base) = fc ? f2{t1, t0} : f2{t0, t1};     \
            *reinterpret_cast<f2 *>((S) + base + NC) = fc ? f2{b1, b0} : f2{b0, b1}; \
        }
        DT_QUAD(s1, r0.x, r0.y, r2.z, r2.w, g[0], g[5])      // subbands (0, 5)
        DT_QUAD(s2, r1.x, r1.y, r1.z, r1.w, g[2], g[3])      // subbands (2, 3)
        DT_QUAD(s3, r0.z, r0.w, r2.x, r2.y, g[1], g[4])      // subbands (1, 4)
#undef DT_QUAD
    }
}

DT_HD void inv_load_low(const float *Zb, int zr, int zc, float *s0, int NR, int NC, int ro,
                        int co, int tid) {
    const bool interior = ro >= 0 && ro + NR <= zr && co >= 0 && co + NC <= zc;
    for (int e = tid; e < NR * NC; e += DT_NT) {
        int rr = e / NC, cc = e - rr * NC;
        int gr = ro + rr, gc = co + cc;
        if (!interior) { gr = reflect_i(gr, zr); gc = reflect_i(gc, zc); }
        s0[e] = Zb[(int64_t)gr * zc + gc];
    }
}

template <int TR_, int TC_, int M0_, int M1_>
struct Inv1Cfg {
    static constexpr int TR = TR_, TC = TC_, M0 = M0_, M1 = M1_;
    static constexpr int H0 = M0 / 2, H1 = M1 / 2, HH = cmax(H0, H1);
    static constexpr int HE = (HH + 1) & ~1;          // even halo: windows made of whole quads
    static constexpr int NR = TR + 2 * HE, NC = TC + 2 * HE;
    static constexpr int RS = 8;
    static constexpr int SP = NR * NC;                // one input plane
    static constexpr int SY = TR * NC;                // one column-pass plane
    static constexpr int LDS_FLOATS = 4 * SP + 2 * SY;
    static_assert(TR % RS == 0 && TR % 2 == 0 && TC % 4 == 0, "tile shape");
    static_assert(M0 % 2 == 1 && M1 % 2 == 1, "biort filters must have odd length");
};

template <class C>
DT_HD void inv1_load(const Inv1Params &p, float *s0, float *s1, float *s2, float *s3, int tid,
                     int b, int r0, int c0) {
    const int ro = r0 - C::HE, co = c0 - C::HE;
    inv_load_low(p.Z + (int64_t)b * p.R * p.C, p.R, p.C, s0, C::NR, C::NC, ro, co, tid);
    inv_load_quads(p.Yh + (int64_t)b * (p.R / 2) * (p.C / 2) * 12, p.R, p.C, p.g, s1, s2, s3,
                   C::NR, C::NC, ro, co, tid);
}

template <class C>
DT_HD void inv1_cols(const Inv1Params &p, const float *s0, const float *s1, const float *s2,
                     const float *s3, float *y1, float *y2, int tid) {
    constexpr int NS = C::TR / C::RS;
    constexpr int WN = C::RS + 2 * C::HE;
    for (int task = tid; task < NS * C::NC; task += DT_NT) {
        int strip = task / C::NC, cc = task - strip * C::NC;
        float w0[WN], w1[WN], w2[WN], w3[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            int idx = (strip * C::RS + j) * C::NC + cc;
            w0[j] = s0[idx]; w1[j] = s1[idx]; w2[j] = s2[idx]; w3[j] = s3[idx];
        }
#pragma unroll
        for (int q = 0; q < C::RS; ++q) {
            float a = 0.f, bq = 0.f;
#pragma unroll
            for (int k = 0; k < C::M0; ++k) {
                a += p.g0[k] * w0[q + C::HE + C::H0 - k];
                bq += p.g0[k] * w2[q + C::HE + C::H0 - k];
            }
#pragma unroll
            for (int k = 0; k < C::M1; ++k) {
                a += p.g1[k] * w1[q + C::HE + C::H1 - k];
                bq += p.g1[k] * w3[q + C::HE + C::H1 - k];
            }
            y1[(strip * C::RS + q) * C::NC + cc] = a;
            y2[(strip * C::RS + q) * C::NC + cc] = bq;
        }
    }
}

template <class C>
DT_HD void inv1_rows(const Inv1Params &p, const float *y1, const float *y2, int tid, int b,
                     int r0, int c0) {
    constexpr int NQ = C::TC / 4;                 // 4 outputs per task
    constexpr int WL = 4 + 2 * C::HE;             // window (multiple of 2)
    for (int task = tid; task < C::TR * NQ; task += DT_NT) {
        int r = task / NQ, q = task - r * NQ;
        int R = r0 + r, Cc = c0 + 4 * q;
        if (R >= p.R || Cc >= p.C) continue;
        float wa[WL], wb[WL];
        const f2 *pa = reinterpret_cast<const f2 *>(y1 + r * C::NC + 4 * q);
        const f2 *pb = reinterpret_cast<const f2 *>(y2 + r * C::NC + 4 * q);
#pragma unroll
        for (int j = 0; j < WL / 2; ++j) {
            f2 a = pa[j], c = pb[j];
            wa[2 * j] = a.x; wa[2 * j + 1] = a.y;
            wb[2 * j] = c.x; wb[2 * j + 1] = c.y;
        }
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < C::M0; ++k) s += p.g0[k] * wa[e + C::HE + C::H0 - k];
#pragma unroll
            for (int k = 0; k < C::M1; ++k) s += p.g1[k] * wb[e + C::HE + C::H1 - k];
            o[e] = s;
        }
        float *X = p.X + ((int64_t)b * p.R + R) * p.C + Cc;
        if (Cc + 3 < p.C && (p.C & 3) == 0) {
            *reinterpret_cast<f4 *>(X) = f4{o[0], o[1], o[2], o[3]};
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (Cc + e < p.C) X[e] = o[e];
        }
    }
}

template <int TR_, int TC_, int M_>
struct Inv2Cfg {
    static constexpr int TR = TR_, TC = TC_, M = M_;       // TR x TC INPUT samples per tile
    static constexpr int M2 = M / 2;
    static constexpr bool ODD = (M2 % 2) == 1;
    static constexpr int WN = ODD ? M : M + 2;             // per-j window length
    static constexpr int ORG = ODD ? 1 - M2 : -M2;         // window origin rel. to 2j (even)
    static constexpr int NR = TR + WN - 2, NC = TC + WN - 2;   // input window (even)
    static constexpr int SP = NR * NC;                     // one input plane
    static constexpr int SY = 2 * TR * NC;                 // one column-pass plane
    static constexpr int LDS_FLOATS = 4 * SP + 2 * SY;
    static_assert(M % 2 == 0 && TR % 2 == 0 && TC % 2 == 0, "even taps / tile");
};

template <class C>
DT_HD void inv2_load(const Inv2Params &p, float *s0, float *s1, float *s2, float *s3, int tid,
                     int b, int r0, int c0) {
    // (r0, c0): tile origin in INPUT coordinates (even)
    const int ro = r0 + C::ORG, co = c0 + C::ORG;
    inv_load_low(p.Z + (int64_t)b * p.zr * p.zc, p.zr, p.zc, s0, C::NR, C::NC, ro, co, tid);
    inv_load_quads(p.Yh + (int64_t)b * (p.zr / 2) * (p.zc / 2) * 12, p.zr, p.zc, p.g, s1, s2,
                   s3, C::NR, C::NC, ro, co, tid);
}

template <class C>
DT_HD void inv2_cols(const Inv2Params &p, const float *s0, const float *s1, const float *s2,
                     const float *s3, float *y1, float *y2, int tid) {
    constexpr int NJ = C::TR / 2;
    for (int task = tid; task < NJ * C::NC; task += DT_NT) {
        int jl = task / C::NC, cc = task - jl * C::NC;
        float w[C::WN];
        float a[4], t[4];
#pragma unroll
        for (int j = 0; j < C::WN; ++j) w[j] = s0[(2 * jl + j) * C::NC + cc];
        ifilt4<C>(w, p.l_a, p.l_b, p.lo_pos, a);
#pragma unroll
        for (int j = 0; j < C::WN; ++j) w[j] = s1[(2 * jl + j) * C::NC + cc];
        ifilt4<C>(w, p.h_a, p.h_b, p.hi_pos, t);
#pragma unroll
        for (int e = 0; e < 4; ++e) y1[(4 * jl + e) * C::NC + cc] = a[e] + t[e];
#pragma unroll
        for (int j = 0; j < C::WN; ++j) w[j] = s2[(2 * jl + j) * C::NC + cc];
        ifilt4<C>(w, p.l_a, p.l_b, p.lo_pos, a);
#pragma unroll
        for (int j = 0; j < C::WN; ++j) w[j] = s3[(2 * jl + j) * C::NC + cc];
        ifilt4<C>(w, p.h_a, p.h_b, p.hi_pos, t);
#pragma unroll
        for (int e = 0; e < 4; ++e) y2[(4 * jl + e) * C::NC + cc] = a[e] + t[e];
    }
}

// ======================================================================================
// second generation variants
// ======================================================================================
// Inverse, second generation: coalesced record loads.
// Reading a 48-byte record per lane (3 x 16 B at a 48-byte lane stride) has the same
// poor efficiency as writing it that way.  Instead every wavefront loads 21 consecutive
// records as 63 consecutive 16-byte pieces (one coalesced 1008-byte run per instruction),
// parks them in a small private LDS slab, and then lane L turns (record L/3, subband pair
// L%3) into one 2x2 quad of the corresponding plane (c2q + gain, A.4).  Two wave-iterations
// share a round so that two loads are in flight per lane.  The two halves are separate
// functions for the same reason as the staged stores above.
constexpr int REC_PER_ITER = 21;
constexpr int REC_ITERS_PER_ROUND = 2;
constexpr int REC_SLAB_FLOATS_PER_WAVE = REC_ITERS_PER_ROUND * 64 * 4;
// ---- wave-level record loader: ALL loads of a wavefront in flight at once --------------
// The round-by-round loader above pays one HBM latency per round.  Here a wavefront first
// issues every 16-byte piece it is responsible for (ITERS loads per lane, held in
// registers), then walks them through its 1 KiB slab one iteration at a time.  Written as
// a per-WAVE function: on the device each lane executes the body once; the host emulator
// executes it with an explicit loop over the 64 lanes (DT_LANE_LOOP), per-lane registers
// becoming arrays.
#if defined(__HIP_DEVICE_COMPILE__)
#define DT_NL 1
#define DT_LANE_LOOP(l) if (const int l = (int)(threadIdx.x & 63); true)
#define DT_LI(l) 0
#else
#define DT_NL 64
#define DT_LANE_LOOP(l) for (int l = 0; l < 64; ++l)
#define DT_LI(l) (l)
#endif
constexpr int rec_iters(int nrec) { return (nrec + REC_PER_ITER * 4 - 1) / (REC_PER_ITER * 4); }

// Row pass + q2c.  STAGE = false: every lane stores its own 48-byte record (3 x 16 B at a
// 48 B stride).  STAGE = true: records of a wavefront's 64 tasks are bounced through a
// private LDS slab so that each store instruction writes 1 KiB of consecutive bytes.
template <class C, bool STAGE>
DT_HD void fwd1d_rows(const Fwd1Params &p, const float *sLo, const float *sHi, float *stage, int tid,
                      int b, int r0, int c0) {
    constexpr int NV = C::TC / 2, NU = C::TR / 2;
    constexpr int WL = 2 * C::HC + 2;
    const int HR = p.LR / 2, HCc = p.LC / 2;
    for (int task = tid; task < NU * NV; task += DT_NT) {
        int u = task / NV, v = task - u * NV;
        int R = r0 + 2 * u, Cc = c0 + 2 * v;
        if (R >= p.LR || Cc >= p.LC) continue;
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
#pragma unroll
            for (int ec = 0; ec < 2; ++ec) {
                float s_ll = 0.f, s_hl = 0.f, s_lh = 0.f, s_hh = 0.f;
#pragma unroll
                for (int k = 0; k < C::M0; ++k) {
                    s_ll += p.h0[k] * wl[ec + C::HC + C::H0 - k];
                    s_hl += p.h0[k] * wh[ec + C::HC + C::H0 - k];
                }
#pragma unroll
                for (int k = 0; k < C::M1; ++k) {
                    s_lh += p.h1[k] * wl[ec + C::HC + C::H1 - k];
                    s_hh += p.h1[k] * wh[ec + C::HC + C::H1 - k];
                }
                ll[er][ec] = s_ll; hl[er][ec] = s_hl; lh[er][ec] = s_lh; hh[er][ec] = s_hh;
            }
        }
        float *L = p.LoLo + ((int64_t)b * p.LR + R) * p.LC + Cc;
        *reinterpret_cast<f2 *>(L) = f2{ll[0][0], ll[0][1]};
        *reinterpret_cast<f2 *>(L + p.LC) = f2{ll[1][0], ll[1][1]};
        float *rec = p.Yh + (((int64_t)b * HR + R / 2) * HCc + Cc / 2) * 12;
        store_record(rec, hl, lh, hh);
    }
    (void)stage;
}

constexpr int rec_rounds(int nrec) {
    return (nrec + REC_PER_ITER * 4 * REC_ITERS_PER_ROUND - 1) / (REC_PER_ITER * 4 * REC_ITERS_PER_ROUND);
}

// window: NR x NC samples (even), origin (ro, co) (even); zr x zc: plane size (even)
DT_HD void inv_rec_fetch(const float *Yhb, int zr, int zc, float *slab_all, int NR, int NC, int ro, int co,
                         int tid, int round) {
    const int lane = tid & 63, wave = tid >> 6;
    const int QC = NC / 2, nrec = (NR / 2) * QC, hc = zc / 2;
    f4 *slab = reinterpret_cast<f4 *>(slab_all + wave * REC_SLAB_FLOATS_PER_WAVE);
    if (lane >= 63) return;
#pragma unroll
    for (int s = 0; s < REC_ITERS_PER_ROUND; ++s) {
        int q = round * REC_ITERS_PER_ROUND + s;
        int ridx = (q * 4 + wave) * REC_PER_ITER + lane / 3;
        if (ridx < nrec) {
            int uw = ridx / QC, vw = ridx - uw * QC;
            int U = reflect_i(ro + 2 * uw, zr) >> 1, V = reflect_i(co + 2 * vw, zc) >> 1;
            const f4 *src = reinterpret_cast<const f4 *>(Yhb + ((int64_t)U * hc + V) * 12);
            slab[s * 64 + lane] = src[lane % 3];
        }
    }
}

DT_HD void inv_rec_expand(const float *slab_all, int zr, int zc, const float *g, float *s1, float *s2,
                          float *s3, int NR, int NC, int ro, int co, int tid, int round) {
    const int lane = tid & 63, wave = tid >> 6;
    const int QC = NC / 2, nrec = (NR / 2) * QC;
    const float *slab = slab_all + wave * REC_SLAB_FLOATS_PER_WAVE;
    if (lane >= 63) return;
    const int rl = lane / 3, plane = lane - 3 * rl;
    // float offsets of the two subbands of this plane inside a record:
    //   plane 0 "lh": slots (0, 5)   plane 1 "hl": slots (2, 3)   plane 2 "hh": slots (1, 4)
    const int o0 = plane == 0 ? 0 : (plane == 1 ? 4 : 2);
    const int o1 = plane == 0 ? 10 : (plane == 1 ? 6 : 8);
    const float g0 = plane == 0 ? g[0] : (plane == 1 ? g[2] : g[1]);
    const float g1 = plane == 0 ? g[5] : (plane == 1 ? g[3] : g[4]);
    float *dst = plane == 0 ? s1 : (plane == 1 ? s2 : s3);
#pragma unroll
    for (int s = 0; s < REC_ITERS_PER_ROUND; ++s) {
        int q = round * REC_ITERS_PER_ROUND + s;
        int ridx = (q * 4 + wave) * REC_PER_ITER + rl;
        if (ridx < nrec) {
            int uw = ridx / QC, vw = ridx - uw * QC;
            int fr = reflect_i(ro + 2 * uw, zr) & 1, fc = reflect_i(co + 2 * vw, zc) & 1;
            const float *rec = slab + (s * 64 + 3 * rl) * 4;
            f2 w0 = *reinterpret_cast<const f2 *>(rec + o0);
            f2 w1 = *reinterpret_cast<const f2 *>(rec + o1);
            float ar = w0.x * g0, ai = w0.y * g0, br = w1.x * g1, bi = w1.y * g1;
            float qa = ar + br, qb = ai + bi, qc = ai - bi, qd = -(ar - br);
            float t0 = fr ? qc : qa, t1 = fr ? qd : qb;
            float b0 = fr ? qa : qc, b1 = fr ? qb : qd;
            int base = (2 * uw) * NC + 2 * vw;
            *reinterpret_cast<f2 *>(dst + base) = fc ? f2{t1, t0} : f2{t0, t1};
            *reinterpret_cast<f2 *>(dst + base + NC) = fc ? f2{b1, b0} : f2{b0, b1};
        }
    }
}

template <int NR, int NC>
DT_HD void inv_rec_load_wave(const float *Yhb, int zr, int zc, const float *g, float *slab_all, float *s1,
                             float *s2, float *s3, int ro, int co, int wave) {
    constexpr int QC = NC / 2, NREC = (NR / 2) * QC, ITERS = rec_iters(NREC);
    const int hc = zc / 2;
    f4 *slab = reinterpret_cast<f4 *>(slab_all + wave * 256);
    const bool interior = ro >= 0 && ro + NR <= zr && co >= 0 && co + NC <= zc;
    float pcx[ITERS][DT_NL], pcy[ITERS][DT_NL], pcz[ITERS][DT_NL], pcw[ITERS][DT_NL];
    DT_LANE_LOOP(l) {
        if (l < 63) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                int ridx = (it * 4 + wave) * REC_PER_ITER + l / 3;
                if (ridx < NREC) {
                    int uw = ridx / QC, vw = ridx - uw * QC;
                    int ur = ro + 2 * uw, vc = co + 2 * vw;
                    if (!interior) { ur = reflect_i(ur, zr); vc = reflect_i(vc, zc); }
                    const f4 *src = reinterpret_cast<const f4 *>(Yhb + ((int64_t)(ur >> 1) * hc + (vc >> 1)) * 12);
                    f4 v = src[l % 3];
                    pcx[it][DT_LI(l)] = v.x; pcy[it][DT_LI(l)] = v.y; pcz[it][DT_LI(l)] = v.z; pcw[it][DT_LI(l)] = v.w;
                }
            }
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        DT_LANE_LOOP(l) {
            if (l < 63 && (it * 4 + wave) * REC_PER_ITER + l / 3 < NREC)
                slab[l] = f4{pcx[it][DT_LI(l)], pcy[it][DT_LI(l)], pcz[it][DT_LI(l)], pcw[it][DT_LI(l)]};
        }
        DT_LANE_LOOP(l) {
            const int rl = l / 3, plane = l - 3 * rl;
            const int ridx = (it * 4 + wave) * REC_PER_ITER + rl;
            if (l < 63 && ridx < NREC) {
                const int o0 = plane == 0 ? 0 : (plane == 1 ? 4 : 2);
                const int o1 = plane == 0 ? 10 : (plane == 1 ? 6 : 8);
                const float g0 = plane == 0 ? g[0] : (plane == 1 ? g[2] : g[1]);
                const float g1 = plane == 0 ? g[5] : (plane == 1 ? g[3] : g[4]);
                float *dst = plane == 0 ? s1 : (plane == 1 ? s2 : s3);
                int uw = ridx / QC, vw = ridx - uw * QC;
                const float *rec = reinterpret_cast<const float *>(slab) + 12 * rl;
                f2 w0 = *reinterpret_cast<const f2 *>(rec + o0);
                f2 w1 = *reinterpret_cast<const f2 *>(rec + o1);
                float ar = w0.x * g0, ai = w0.y * g0, br = w1.x * g1, bi = w1.y * g1;
                float qa = ar + br, qb = ai + bi, qc = ai - bi, qd = -(ar - br);
                int base = (2 * uw) * NC + 2 * vw;
                if (interior) {
                    *reinterpret_cast<f2 *>(dst + base) = f2{qa, qb};
                    *reinterpret_cast<f2 *>(dst + base + NC) = f2{qc, qd};
                } else {
                    int fr = reflect_i(ro + 2 * uw, zr) & 1, fc = reflect_i(co + 2 * vw, zc) & 1;
                    float t0 = fr ? qc : qa, t1 = fr ? qd : qb;
                    float b0 = fr ? qa : qc, b1 = fr ? qb : qd;
                    *reinterpret_cast<f2 *>(dst + base) = fc ? f2{t1, t0} : f2{t0, t1};
                    *reinterpret_cast<f2 *>(dst + base + NC) = fc ? f2{b1, b0} : f2{b0, b1};
                }
            }
        }
    }
}

// ---- level 1 inverse ------------------------------------------------------------------
template <int TR_, int TC_, int RS_, int M0_, int M1_>
struct Inv1DCfg {
    static constexpr int TR = TR_, TC = TC_, RS = RS_, M0 = M0_, M1 = M1_;
    static constexpr int H0 = M0 / 2, H1 = M1 / 2, HH = cmax(H0, H1);
    static constexpr int HE = (HH + 1) & ~1;
    static constexpr int NR = TR + 2 * HE, NC = TC + 2 * HE;
    static constexpr int NS = TR / RS;
    static constexpr int WN = RS + 2 * HE;
    static constexpr int SP = NR * NC;                // one quad plane
    static constexpr int SY = TR * NC;                // one column-pass plane
    static constexpr int SLAB = 4 * REC_SLAB_FLOATS_PER_WAVE;
    static constexpr int YF = 2 * SY > SLAB ? 2 * SY : SLAB;   // y1|y2 alias the record slab
    static constexpr int LDS_FLOATS = 3 * SP + YF;
    static constexpr int ROUNDS = rec_rounds((NR / 2) * (NC / 2));
    static_assert(TR % RS == 0 && TR % 2 == 0 && TC % 4 == 0, "tile shape");
    static_assert(M0 % 2 == 1 && M1 % 2 == 1, "biort filters must have odd length");
};

template <class C>
DT_HD void inv1d_cols(const Inv1Params &p, const float *s1, const float *s2, const float *s3, float *y1,
                      float *y2, int tid, int b, int r0, int c0) {
    const float *Zb = p.Z + (int64_t)b * p.R * p.C;
    const int ro = r0 - C::HE, co = c0 - C::HE;
    const bool interior = ro >= 0 && ro + C::NR <= p.R && co >= 0 && co + C::NC <= p.C;
    for (int task = tid; task < C::NS * C::NC; task += DT_NT) {
        int strip = task / C::NC, cc = task - strip * C::NC;
        float w0[C::WN], w1[C::WN], w2[C::WN], w3[C::WN];
        if (interior) {
            const float *src = Zb + (int64_t)(ro + strip * C::RS) * p.C + (co + cc);
#pragma unroll
            for (int j = 0; j < C::WN; ++j) w0[j] = src[(int64_t)j * p.C];
        } else {
            int gc = reflect_i(co + cc, p.C);
#pragma unroll
            for (int j = 0; j < C::WN; ++j)
                w0[j] = Zb[(int64_t)reflect_i(ro + strip * C::RS + j, p.R) * p.C + gc];
        }
#pragma unroll
        for (int j = 0; j < C::WN; ++j) {
            int idx = (strip * C::RS + j) * C::NC + cc;
            w1[j] = s1[idx]; w2[j] = s2[idx]; w3[j] = s3[idx];
        }
#pragma unroll
        for (int q = 0; q < C::RS; ++q) {
            float a = 0.f, bq = 0.f;
#pragma unroll
            for (int k = 0; k < C::M0; ++k) {
                a += p.g0[k] * w0[q + C::HE + C::H0 - k];
                bq += p.g0[k] * w2[q + C::HE + C::H0 - k];
            }
#pragma unroll
            for (int k = 0; k < C::M1; ++k) {
                a += p.g1[k] * w1[q + C::HE + C::H1 - k];
                bq += p.g1[k] * w3[q + C::HE + C::H1 - k];
            }
            y1[(strip * C::RS + q) * C::NC + cc] = a;
            y2[(strip * C::RS + q) * C::NC + cc] = bq;
        }
    }
}

// Prefetching variant: the lowpass window of the thread's (single) column-pass task is
// requested before the record phase, so that records and lowpass share ONE memory latency.
// Requires NS * NC <= DT_NT (one task per thread).
template <class C>
DT_HD void inv1p_fetch(const Inv1Params &p, float (&w0)[C::WN], int tid, int b, int r0, int c0) {
    static_assert(C::NS * C::NC <= DT_NT, "one column-pass task per thread");
    if (tid >= C::NS * C::NC) return;
    const float *Zb = p.Z + (int64_t)b * p.R * p.C;
    const int ro = r0 - C::HE, co = c0 - C::HE;
    const bool interior = ro >= 0 && ro + C::NR <= p.R && co >= 0 && co + C::NC <= p.C;
    int strip = tid / C::NC, cc = tid - strip * C::NC;
    if (interior) {
        const float *src = Zb + (int64_t)(ro + strip * C::RS) * p.C + (co + cc);
#pragma unroll
        for (int j = 0; j < C::WN; ++j) w0[j] = src[(int64_t)j * p.C];
    } else {
        int gc = reflect_i(co + cc, p.C);
#pragma unroll
        for (int j = 0; j < C::WN; ++j)
            w0[j] = Zb[(int64_t)reflect_i(ro + strip * C::RS + j, p.R) * p.C + gc];
    }
}

template <class C>
DT_HD void inv1p_cols(const Inv1Params &p, const float (&w0)[C::WN], const float *s1, const float *s2,
                      const float *s3, float *y1, float *y2, int tid) {
    if (tid >= C::NS * C::NC) return;
    int strip = tid / C::NC, cc = tid - strip * C::NC;
    float w1[C::WN], w2[C::WN], w3[C::WN];
#pragma unroll
    for (int j = 0; j < C::WN; ++j) {
        int idx = (strip * C::RS + j) * C::NC + cc;
        w1[j] = s1[idx]; w2[j] = s2[idx]; w3[j] = s3[idx];
    }
#pragma unroll
    for (int q = 0; q < C::RS; ++q) {
        float a = 0.f, bq = 0.f;
#pragma unroll
        for (int k = 0; k < C::M0; ++k) {
            a += p.g0[k] * w0[q + C::HE + C::H0 - k];
            bq += p.g0[k] * w2[q + C::HE + C::H0 - k];
        }
#pragma unroll
        for (int k = 0; k < C::M1; ++k) {
            a += p.g1[k] * w1[q + C::HE + C::H1 - k];
            bq += p.g1[k] * w3[q + C::HE + C::H1 - k];
        }
        y1[(strip * C::RS + q) * C::NC + cc] = a;
        y2[(strip * C::RS + q) * C::NC + cc] = bq;
    }
}

// ---- level >= 2 inverse ---------------------------------------------------------------
template <int TR_, int TC_, int JS_, int M_>
struct Inv2DCfg {
    static constexpr int TR = TR_, TC = TC_, JS = JS_, M = M_;     // TR x TC INPUT samples per tile
    static constexpr int M2 = M / 2;
    static constexpr bool ODD = (M2 % 2) == 1;
    static constexpr int WN = ODD ? M : M + 2;                     // window of one j
    static constexpr int ORG = ODD ? 1 - M2 : -M2;
    static constexpr int NR = TR + WN - 2, NC = TC + WN - 2;
    static constexpr int NJ = TR / 2;                              // j's per tile column
    static constexpr int NS = NJ / JS;                             // strips of JS j's
    static constexpr int WS = 2 * JS + WN - 2;                     // window of a strip
    static constexpr int SP = NR * NC;
    static constexpr int SY = 2 * TR * NC;
    static constexpr int SLAB = 4 * REC_SLAB_FLOATS_PER_WAVE;
    static constexpr int YF = 2 * SY > SLAB ? 2 * SY : SLAB;
    static constexpr int LDS_FLOATS = 3 * SP + YF;
    static constexpr int ROUNDS = rec_rounds((NR / 2) * (NC / 2));
    static_assert(M % 2 == 0 && TR % 2 == 0 && TC % 2 == 0 && NJ % JS == 0, "even taps / tile");
};

template <class C>
DT_HD void inv2d_cols(const Inv2Params &p, const float *s1, const float *s2, const float *s3, float *y1,
                      float *y2, int tid, int b, int r0, int c0) {
    const float *Zb = p.Z + (int64_t)b * p.zr * p.zc;
    const int ro = r0 + C::ORG, co = c0 + C::ORG;
    const bool interior = ro >= 0 && ro + C::NR <= p.zr && co >= 0 && co + C::NC <= p.zc;
    for (int task = tid; task < C::NS * C::NC; task += DT_NT) {
        int strip = task / C::NC, cc = task - strip * C::NC;
        const int rs = 2 * C::JS * strip;                   // first window row of the strip
        float w[C::WS];
        float a[4], t[4];
        // ---- y1 = colifilt(Z, lo pair) + colifilt(lh, hi pair) ----
        if (interior) {
            const float *src = Zb + (int64_t)(ro + rs) * p.zc + (co + cc);
#pragma unroll
            for (int j = 0; j < C::WS; ++j) w[j] = src[(int64_t)j * p.zc];
        } else {
            int gc = reflect_i(co + cc, p.zc);
#pragma unroll
            for (int j = 0; j < C::WS; ++j) w[j] = Zb[(int64_t)reflect_i(ro + rs + j, p.zr) * p.zc + gc];
        }
        float acc[C::JS][4];
#pragma unroll
        for (int q = 0; q < C::JS; ++q) {
            ifilt4<C>(w + 2 * q, p.l_a, p.l_b, p.lo_pos, a);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[q][e] = a[e];
        }
#pragma unroll
        for (int j = 0; j < C::WS; ++j) w[j] = s1[(rs + j) * C::NC + cc];
#pragma unroll
        for (int q = 0; q < C::JS; ++q) {
            ifilt4<C>(w + 2 * q, p.h_a, p.h_b, p.hi_pos, t);
#pragma unroll
            for (int e = 0; e < 4; ++e) y1[(4 * (strip * C::JS + q) + e) * C::NC + cc] = acc[q][e] + t[e];
        }
        // ---- y2 = colifilt(hl, lo pair) + colifilt(hh, hi pair) ----
#pragma unroll
        for (int j = 0; j < C::WS; ++j) w[j] = s2[(rs + j) * C::NC + cc];
#pragma unroll
        for (int q = 0; q < C::JS; ++q) {
            ifilt4<C>(w + 2 * q, p.l_a, p.l_b, p.lo_pos, a);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[q][e] = a[e];
        }
#pragma unroll
        for (int j = 0; j < C::WS; ++j) w[j] = s3[(rs + j) * C::NC + cc];
#pragma unroll
        for (int q = 0; q < C::JS; ++q) {
            ifilt4<C>(w + 2 * q, p.h_a, p.h_b, p.hi_pos, t);
#pragma unroll
            for (int e = 0; e < 4; ++e) y2[(4 * (strip * C::JS + q) + e) * C::NC + cc] = acc[q][e] + t[e];
        }
    }
}

// Register-staged variant of inv_rec_stage for software pipelining: fetch the pieces of
// the NEXT tile into registers while the current tile is being computed, write them to
// LDS once the current tile's column pass is done.  NP = pieces per thread.
template <int QR, int QC>
struct RecRegs {
    static constexpr int NPIECE = 3 * QR * QC;
    static constexpr int NP = (NPIECE + DT_NT - 1) / DT_NT;
    float x[NP], y[NP], z[NP], w[NP];
};

template <int QR, int QC>
DT_HD void inv_rec_fetch_regs(const float *Yhb, int zr, int zc, RecRegs<QR, QC> &rg, int ro, int co, int tid) {
    const int hc = zc / 2;
    const bool interior = ro >= 0 && ro + 2 * QR <= zr && co >= 0 && co + 2 * QC <= zc;
#pragma unroll
    for (int k = 0; k < RecRegs<QR, QC>::NP; ++k) {
        int piece = tid + k * DT_NT;
        if (piece < RecRegs<QR, QC>::NPIECE) {
            int rec = piece / 3, part = piece - 3 * rec;
            int uw = rec / QC, vw = rec - uw * QC;
            int ur = ro + 2 * uw, vc = co + 2 * vw;
            if (!interior) { ur = reflect_i(ur, zr); vc = reflect_i(vc, zc); }
            f4 v = reinterpret_cast<const f4 *>(Yhb + ((int64_t)(ur >> 1) * hc + (vc >> 1)) * 12)[part];
            rg.x[k] = v.x; rg.y[k] = v.y; rg.z[k] = v.z; rg.w[k] = v.w;
        }
    }
}

template <int QR, int QC>
DT_HD void inv_rec_store_regs(float *srec, const RecRegs<QR, QC> &rg, int tid) {
    f4 *dst = reinterpret_cast<f4 *>(srec);
#pragma unroll
    for (int k = 0; k < RecRegs<QR, QC>::NP; ++k) {
        int piece = tid + k * DT_NT;
        if (piece < RecRegs<QR, QC>::NPIECE) dst[piece] = f4{rg.x[k], rg.y[k], rg.z[k], rg.w[k]};
    }
}

template <class C, int E>
DT_HD void inv1r_cols_e(const Inv1Params &p, const float (&w0)[C::WN], const float *srec, float *y1,
                        float *y2, int tid, int r0, int c0) {
    const ColTask t = inv_col_task<C>(tid);
    if (!t.valid) return;
    const int ro = r0 - C::HE, co = c0 - C::HE;
    const bool interior = ro >= 0 && ro + C::NR <= p.R && co >= 0 && co + C::NC <= p.C;
    const int cc = 2 * t.i + t.e;
    float w1[C::WN], w2[C::WN], w3[C::WN];
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
#pragma unroll
    for (int q = 0; q < C::RS; ++q) {
        float a = 0.f, bq = 0.f;
#pragma unroll
        for (int k = 0; k < C::M0; ++k) {
            a += p.g0[k] * w0[q + C::HE + C::H0 - k];
            bq += p.g0[k] * w2[q + C::HE + C::H0 - k];
        }
#pragma unroll
        for (int k = 0; k < C::M1; ++k) {
            a += p.g1[k] * w1[q + C::HE + C::H1 - k];
            bq += p.g1[k] * w3[q + C::HE + C::H1 - k];
        }
        y1[(t.strip * C::RS + q) * C::NC + cc] = a;
        y2[(t.strip * C::RS + q) * C::NC + cc] = bq;
    }
}

template <class C>
DT_HD void inv1r_cols(const Inv1Params &p, const float (&w0)[C::WN], const float *srec, float *y1,
                      float *y2, int tid, int r0, int c0) {
    if (DT_WAVE_UNIFORM((tid >> 6) & 1)) inv1r_cols_e<C, 1>(p, w0, srec, y1, y2, tid, r0, c0);
    else inv1r_cols_e<C, 0>(p, w0, srec, y1, y2, tid, r0, c0);
}

// lowpass window from an LDS plane s0[NR][NC] (staged by inv_load_low) instead of global:
// direct global reads re-fetch every lowpass row (RS + 2*HE)/RS times through L1, which is
// the scarcer resource (measured ~40 B/clk/CU) -- see DESIGN.md section 3.
template <class C>
DT_HD void inv1r_fetch_lds(const float *s0, float (&w0)[C::WN], int tid) {
    const ColTask t = inv_col_task<C>(tid);
    if (!t.valid) return;
    const int cc = 2 * t.i + t.e;
#pragma unroll
    for (int j = 0; j < C::WN; ++j) w0[j] = s0[(t.strip * C::RS + j) * C::NC + cc];
}

// prefetching variant of the level >= 2 inverse column pass (see inv1p_fetch)
template <class C>
DT_HD void inv2p_fetch(const Inv2Params &p, float (&w0)[C::WS], int tid, int b, int r0, int c0) {
    static_assert(C::NS * C::NC <= DT_NT, "one column-pass task per thread");
    if (tid >= C::NS * C::NC) return;
    const float *Zb = p.Z + (int64_t)b * p.zr * p.zc;
    const int ro = r0 + C::ORG, co = c0 + C::ORG;
    const bool interior = ro >= 0 && ro + C::NR <= p.zr && co >= 0 && co + C::NC <= p.zc;
    int strip = tid / C::NC, cc = tid - strip * C::NC;
    const int rs = 2 * C::JS * strip;
    if (interior) {
        const float *src = Zb + (int64_t)(ro + rs) * p.zc + (co + cc);
#pragma unroll
        for (int j = 0; j < C::WS; ++j) w0[j] = src[(int64_t)j * p.zc];
    } else {
        int gc = reflect_i(co + cc, p.zc);
#pragma unroll
        for (int j = 0; j < C::WS; ++j) w0[j] = Zb[(int64_t)reflect_i(ro + rs + j, p.zr) * p.zc + gc];
    }
}

template <class C>
DT_HD void inv2p_cols(const Inv2Params &p, const float (&w0)[C::WS], const float *s1, const float *s2,
                      const float *s3, float *y1, float *y2, int tid) {
    if (tid >= C::NS * C::NC) return;
    int strip = tid / C::NC, cc = tid - strip * C::NC;
    const int rs = 2 * C::JS * strip;
    float w[C::WS];
    float a[4], t[4];
    float acc[C::JS][4];
#pragma unroll
    for (int q = 0; q < C::JS; ++q) {
        ifilt4<C>(w0 + 2 * q, p.l_a, p.l_b, p.lo_pos, a);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[q][e] = a[e];
    }
#pragma unroll
    for (int j = 0; j < C::WS; ++j) w[j] = s1[(rs + j) * C::NC + cc];
#pragma unroll
    for (int q = 0; q < C::JS; ++q) {
        ifilt4<C>(w + 2 * q, p.h_a, p.h_b, p.hi_pos, t);
#pragma unroll
        for (int e = 0; e < 4; ++e) y1[(4 * (strip * C::JS + q) + e) * C::NC + cc] = acc[q][e] + t[e];
    }
#pragma unroll
    for (int j = 0; j < C::WS; ++j) w[j] = s2[(rs + j) * C::NC + cc];
#pragma unroll
    for (int q = 0; q < C::JS; ++q) {
        ifilt4<C>(w + 2 * q, p.l_a, p.l_b, p.lo_pos, a);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[q][e] = a[e];
    }
#pragma unroll
    for (int j = 0; j < C::WS; ++j) w[j] = s3[(rs + j) * C::NC + cc];
#pragma unroll
    for (int q = 0; q < C::JS; ++q) {
        ifilt4<C>(w + 2 * q, p.h_a, p.h_b, p.hi_pos, t);
#pragma unroll
        for (int e = 0; e < 4; ++e) y2[(4 * (strip * C::JS + q) + e) * C::NC + cc] = acc[q][e] + t[e];
    }
}

template <class C, int E>
DT_HD void inv2r_cols_e(const Inv2Params &p, const float (&w0)[C::WS], const float *srec, float *y1,
                        float *y2, int tid, int r0, int c0) {
    const ColTask t = inv_col_task<C>(tid);
    if (!t.valid) return;
    const int ro = r0 + C::ORG, co = c0 + C::ORG;
    const bool interior = ro >= 0 && ro + C::NR <= p.zr && co >= 0 && co + C::NC <= p.zc;
    const int cc = 2 * t.i + t.e, rs = C::RS * t.strip;
    float w1[C::WS], w2[C::WS], w3[C::WS];
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
    float a[4], tt[4];
#pragma unroll
    for (int q = 0; q < C::JS; ++q) {
        ifilt4<C>(w0 + 2 * q, p.l_a, p.l_b, p.lo_pos, a);
        ifilt4<C>(w1 + 2 * q, p.h_a, p.h_b, p.hi_pos, tt);
#pragma unroll
        for (int e = 0; e < 4; ++e) y1[(4 * (t.strip * C::JS + q) + e) * C::NC + cc] = a[e] + tt[e];
        ifilt4<C>(w2 + 2 * q, p.l_a, p.l_b, p.lo_pos, a);
        ifilt4<C>(w3 + 2 * q, p.h_a, p.h_b, p.hi_pos, tt);
#pragma unroll
        for (int e = 0; e < 4; ++e) y2[(4 * (t.strip * C::JS + q) + e) * C::NC + cc] = a[e] + tt[e];
    }
}

template <class C>
DT_HD void inv2r_cols(const Inv2Params &p, const float (&w0)[C::WS], const float *srec, float *y1,
                      float *y2, int tid, int r0, int c0) {
    if (DT_WAVE_UNIFORM((tid >> 6) & 1)) inv2r_cols_e<C, 1>(p, w0, srec, y1, y2, tid, r0, c0);
    else inv2r_cols_e<C, 0>(p, w0, srec, y1, y2, tid, r0, c0);
}

}  // namespace dt2d
