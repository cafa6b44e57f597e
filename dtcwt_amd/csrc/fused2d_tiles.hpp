// Fused per-level tile programs of the float32 2-D DT-CWT (forward and inverse).
//
// One workgroup (256 threads, 4 wavefronts) owns one output tile of one image and runs
// three phases separated by workgroup barriers, everything in between living in LDS:
//
//   forward:  load input window (+halo, symmetric reflection / edge replication done by
//             index math)  ->  column pass (filter down axis 0: Lo, Hi)  ->  row pass
//             (filter along axis 1) fused with q2c, writing LoLo and whole 48-byte
//             6-subband records of Yh.
//   inverse:  load lowpass window and Yh records (c2q + gains applied on the fly into
//             three quad planes)  ->  column pass (y1, y2)  ->  row pass writing Z.
//
// Each phase is a __host__ __device__ function of (params, LDS pointers, thread id, tile
// origin) so that exactly the same index algebra can be stepped through on the host by
// the test-only emulator (tests/emu/), which is how it was debugged without a GPU.  The
// product only ever calls them from the __global__ wrappers in fused2d.hip.
//
// Index algebra: SURVEY.md Appendix A.1-A.4, A.6; reference: dtcwt/numpy/lowlevel.py
// (colfilter :47-80, coldfilt :82-154, colifilt :156-260) and
// dtcwt/numpy/transform2d.py (level loops :112-160, :242-293; q2c :301-322; c2q :324-350).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#define DT_HD __host__ __device__ __forceinline__
#define DT_NT 256                       // threads per workgroup
#define DT_MAXT 40                      // == DTCWT_HIP_MAX_TAPS

namespace dt2d {

struct alignas(16) f4 { float x, y, z, w; };
struct alignas(8) f2 { float x, y; };

// Half-sample symmetric reflection, ONE bounce: valid for -n <= u < 2n.  Branch-free on
// purpose: a data-dependent branch per index splits the load sequences of the tile
// programs into basic blocks and serialises their memory latencies (profiles/ round 1).
// The plan only uses the fused kernels when every level is at least DT_MIN_FUSED_DIM
// samples wide (halos are < 40), so one bounce always suffices; smaller images take the
// generic kernels (filters.hip), whose reflection is the multi-bounce modulo form.
#define DT_MIN_FUSED_DIM 40
DT_HD int reflect_i(int u, int n) {
    u = u < 0 ? -1 - u : u;
    u = u >= n ? 2 * n - 1 - u : u;
    // rows/columns of partial edge tiles beyond one bounce are computed but never stored:
    // keep their addresses inside the array
    return u < 0 ? 0 : (u > n - 1 ? n - 1 : u);
}
DT_HD int clamp_i(int u, int lo, int hi) { return u < lo ? lo : (u > hi ? hi : u); }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// ======================================================================================
// Level 1 forward: odd-length biort filters h0 (M0 taps), h1 (M1 taps), no decimation.
// ======================================================================================
struct Fwd1Params {
    const float *X;       // [B][inR][inC]
    float *LoLo;          // [B][LR][LC]
    float *Yh;            // [B][LR/2][LC/2][12 floats]
    int B, inR, inC;      // real input size (may be odd)
    int LR, LC;           // even-extended logical size (transform2d.py:86-94)
    int tilesR, tilesC;
    int xcd_order;        // 1: XCD-contiguous tile runs, 0: linear order
    float h0[DT_MAXT], h1[DT_MAXT];
};

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

// q2c of quad (a b / c d): z0 = s((a-d) + j(b+c)), z1 = s((a+d) + j(b-c))   (A.4)
DT_HD void q2c_pair(float a, float b, float c, float d, float &z0r, float &z0i, float &z1r,
                    float &z1i) {
    const float s = 0.70710678118654752440f;
    z0r = s * (a - d); z0i = s * (b + c);
    z1r = s * (a + d); z1i = s * (b - c);
}

// Store one Yh record: subband slots 0..5 = HLz0, HHz0, LHz0, LHz1, HHz1, HLz1 where
// HL = hi on axis 0 / lo on axis 1 (transform2d.py:122-127: slots [0,5], [2,3], [1,4]).
DT_HD void store_record(float *rec, const float (&hl)[2][2], const float (&lh)[2][2],
                        const float (&hh)[2][2]) {
    float a0r, a0i, a1r, a1i, b0r, b0i, b1r, b1i, c0r, c0i, c1r, c1i;
    q2c_pair(hl[0][0], hl[0][1], hl[1][0], hl[1][1], a0r, a0i, a1r, a1i);
    q2c_pair(hh[0][0], hh[0][1], hh[1][0], hh[1][1], b0r, b0i, b1r, b1i);
    q2c_pair(lh[0][0], lh[0][1], lh[1][0], lh[1][1], c0r, c0i, c1r, c1i);
    f4 *o = reinterpret_cast<f4 *>(rec);
    o[0] = f4{a0r, a0i, b0r, b0i};
    o[1] = f4{c0r, c0i, c1r, c1i};
    o[2] = f4{b1r, b1i, a1r, a1i};
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

// ======================================================================================
// Level >= 2 forward: even-length q-shift pairs, decimation by 2 per axis (coldfilt).
// ======================================================================================
struct Fwd2Params {
    const float *X;       // [B][inR][inC]   LoLo of the previous level
    float *LoLo;          // [B][LR/2][LC/2]
    float *Yh;            // [B][LR/4][LC/4][12]
    int B, inR, inC;
    int padR, padC;       // 0/1: one replicated row/col each side (transform2d.py:134-140)
    int LR, LC;           // inR + 2 padR, multiples of 4
    int tilesR, tilesC;
    int xcd_order;        // 1: XCD-contiguous tile runs, 0: linear order
    int lo_a_first, hi_a_first;   // sign of sum(ha*hb) of the lo / hi pair (lowlevel.py:143)
    // coldfilt(X, ha, hb) is called as (h0b, h0a) and (h1b, h1a): "a" arrays hold the
    // first argument, "b" the second.
    float l_a[DT_MAXT], l_b[DT_MAXT], h_a[DT_MAXT], h_b[DT_MAXT];
};

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

// One (A, B) pair from a 2M window w[0..2M) whose element j is logical sample
// 4i - M + 2 + j (A.2):  A = sum_k ha[2k] w[2M-2-4k] + ha[2k+1] w[2M-4-4k]
//                        B = sum_k hb[2k] w[2M-1-4k] + hb[2k+1] w[2M-3-4k]
template <int M>
DT_HD void dfilt_pair(const float *w, const float *ha, const float *hb, float &A, float &B) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < M / 2; ++k) {
        a += ha[2 * k] * w[2 * M - 2 - 4 * k];
        a += ha[2 * k + 1] * w[2 * M - 4 - 4 * k];
        b += hb[2 * k] * w[2 * M - 1 - 4 * k];
        b += hb[2 * k + 1] * w[2 * M - 3 - 4 * k];
    }
    A = a; B = b;
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

// ======================================================================================
// Level 1 inverse: odd-length g0 (M0 taps), g1 (M1 taps), no interpolation.
//   y1 = colfilter(Z, g0) + colfilter(lh, g1);  y2 = colfilter(hl, g0) + colfilter(hh, g1)
//   X  = rowfilter(y1, g0) + rowfilter(y2, g1)            (transform2d.py:275-293)
// ======================================================================================
struct Inv1Params {
    const float *Z;       // [B][R][C]
    const float *Yh;      // [B][R/2][C/2][12]
    float *X;             // [B][R][C]
    int B, R, C;
    int tilesR, tilesC;
    int xcd_order;        // 1: XCD-contiguous tile runs, 0: linear order
    float g[6];           // gain_mask column * sqrt(1/2)
    float g0[DT_MAXT], g1[DT_MAXT];
};

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

// ======================================================================================
// Level >= 2 inverse: even-length q-shift pairs, interpolation by 2 per axis (colifilt).
//   y1 = colifilt(Z, g0b, g0a) + colifilt(lh, g1b, g1a)
//   y2 = colifilt(hl, g0b, g0a) + colifilt(hh, g1b, g1a)
//   Z' = rowifilt(y1, g0b, g0a) + rowifilt(y2, g1b, g1a), cropped by 1 where the forward
//   level padded                                               (transform2d.py:242-273)
// ======================================================================================
struct Inv2Params {
    const float *Z;       // [B][zr][zc]
    const float *Yh;      // [B][zr/2][zc/2][12]
    float *Out;           // [B][2zr - 2cropR][2zc - 2cropC]
    int B, zr, zc;
    int cropR, cropC;     // 0/1
    int tilesR, tilesC;
    int xcd_order;        // 1: XCD-contiguous tile runs, 0: linear order
    int lo_pos, hi_pos;   // sum(ha*hb) > 0 of the g0 / g1 pair
    float g[6];
    // colifilt(X, ha, hb) is called as (g0b, g0a) and (g1b, g1a)
    float l_a[DT_MAXT], l_b[DT_MAXT], h_a[DT_MAXT], h_b[DT_MAXT];
};

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

// Four output phases 4j..4j+3 from window w (element j' = sample 2j + ORG + j')   (A.3)
template <class C>
DT_HD void ifilt4(const float *w, const float *ha, const float *hb, int pos, float (&y)[4]) {
    float y0 = 0.f, y1 = 0.f, y2 = 0.f, y3 = 0.f;
    if (C::ODD) {
        // offsets from 2j: ta = m2 - 2k (pos) ; tb = m2 - 2k - 1  -> w index = off - ORG
#pragma unroll
        for (int k = 0; k < C::M2; ++k) {
            float hi = w[C::M - 1 - 2 * k], lo = w[C::M - 2 - 2 * k];
            float xa = pos ? hi : lo, xb = pos ? lo : hi;
            y0 += ha[2 * k] * xb;          // hao
            y1 += hb[2 * k] * xa;          // hbo
            y2 += ha[2 * k + 1] * xb;      // hae
            y3 += hb[2 * k + 1] * xa;      // hbe
        }
    } else {
        // offsets from 2j: t = m2 + 1 - 2k ; w index = off + m2
#pragma unroll
        for (int k = 0; k < C::M2; ++k) {
            float t0 = w[C::M + 1 - 2 * k];     // t
            float t1 = w[C::M - 2 * k];         // t - 1
            float t2 = w[C::M - 1 - 2 * k];     // t - 2
            float t3 = w[C::M - 2 - 2 * k];     // t - 3
            float xa = pos ? t0 : t1, xb = pos ? t1 : t0;        // ta, tb
            float xa2 = pos ? t2 : t3, xb2 = pos ? t3 : t2;      // ta-2, tb-2
            y0 += ha[2 * k + 1] * xb2;     // hae X[tb-2]
            y1 += hb[2 * k + 1] * xa2;     // hbe X[ta-2]
            y2 += ha[2 * k] * xb;          // hao X[tb]
            y3 += hb[2 * k] * xa;          // hbo X[ta]
        }
    }
    y[0] = y0; y[1] = y1; y[2] = y2; y[3] = y3;
}

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

template <class C>
DT_HD void inv2_rows(const Inv2Params &p, const float *y1, const float *y2, int tid, int b,
                     int r0, int c0) {
    constexpr int NJ = C::TC / 2;
    const int OR = 2 * p.zr - 2 * p.cropR, OC = 2 * p.zc - 2 * p.cropC;
    for (int task = tid; task < 2 * C::TR * NJ; task += DT_NT) {
        int r = task / NJ, jl = task - r * NJ;
        int Rl = 2 * r0 + r;                 // logical output row
        int Cl = 2 * c0 + 4 * jl;            // logical output col of phase 0
        if (Rl >= 2 * p.zr || Cl >= 2 * p.zc) continue;
        int Rw = Rl - p.cropR;
        if (Rw < 0 || Rw >= OR) continue;
        float w[C::WN];
        float a[4], t[4];
        const f2 *pa = reinterpret_cast<const f2 *>(y1 + r * C::NC + 2 * jl);
        const f2 *pb = reinterpret_cast<const f2 *>(y2 + r * C::NC + 2 * jl);
#pragma unroll
        for (int j = 0; j < C::WN / 2; ++j) { f2 v = pa[j]; w[2 * j] = v.x; w[2 * j + 1] = v.y; }
        ifilt4<C>(w, p.l_a, p.l_b, p.lo_pos, a);
#pragma unroll
        for (int j = 0; j < C::WN / 2; ++j) { f2 v = pb[j]; w[2 * j] = v.x; w[2 * j + 1] = v.y; }
        ifilt4<C>(w, p.h_a, p.h_b, p.hi_pos, t);
        float *O = p.Out + ((int64_t)b * OR + Rw) * OC;
        if (p.cropC == 0 && (OC & 3) == 0) {
            *reinterpret_cast<f4 *>(O + Cl) = f4{a[0] + t[0], a[1] + t[1], a[2] + t[2], a[3] + t[3]};
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int cw = Cl + e - p.cropC;
                if (cw >= 0 && cw < OC) O[cw] = a[e] + t[e];
            }
        }
    }
}

// XCD-aware tile order: consecutive workgroup ids round-robin over the 8 XCDs
// (MI355X_MICROARCH.md: block b -> XCD b % 8), so give each XCD a contiguous run of
// tiles; neighbouring tiles then share halos through the same 4 MiB L2.
DT_HD int xcd_tile(int bid, int ntiles) {
    int per = (ntiles + 7) / 8;
    int t = (bid % 8) * per + bid / 8;
    return t;       // may be >= ntiles: caller must skip
}
// Measured on MI355X (tools/kbench): for these write-heavy kernels the plain linear order
// is FASTER than the XCD-contiguous one (82.6 vs 93.1 us for level-1 forward at 4096^2) --
// neighbouring tiles in flight together keep the HBM write streams dense -- so linear is
// the default and the remap stays selectable (DTCWT_HIP_XCD_ORDER=1) for experiments.
DT_HD int tile_of(int bid, int ntiles, int xcd_order) {
    return xcd_order ? xcd_tile(bid, ntiles) : bid;
}

}  // namespace dt2d
