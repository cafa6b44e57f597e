// Second-generation level-1 tile programs (same arithmetic as fused2d_tiles.hpp, different
// data movement): the column pass reads its sliding window straight from global memory
// (coalesced rows, many independent loads in flight per lane) instead of staging the input
// window in LDS first.  That removes one LDS plane and one workgroup barrier per tile, so
// more workgroups fit per CU and more bytes are in flight per CU -- the level-1 kernels
// are latency-, not bandwidth-limited in the first-generation layout (profiles/ round 1).
#pragma once
#include "fused2d_tiles.hpp"

namespace dt2d {

// ======================================================================================
// Level 1 forward, direct column pass.
// ======================================================================================
template <int TR_, int TC_, int RS_, int M0_, int M1_>
struct Fwd1DCfg {
    static constexpr int TR = TR_, TC = TC_, RS = RS_, M0 = M0_, M1 = M1_;
    static constexpr int H0 = M0 / 2, H1 = M1 / 2, HH = cmax(H0, H1);
    static constexpr int HC = (HH + 1) & ~1;
    static constexpr int W = TC + 2 * HC;
    static constexpr int NS = TR / RS;
    static constexpr int WN = RS + 2 * HH;            // register window per task
    static constexpr int SL = TR * W;
    static constexpr int LDS_FLOATS = 2 * SL;
    static_assert(TR % RS == 0 && TR % 2 == 0 && TC % 2 == 0, "tile shape");
    static_assert(M0 % 2 == 1 && M1 % 2 == 1, "biort filters must have odd length");
};

template <class C>
DT_HD void fwd1d_cols(const Fwd1Params &p, float *sLo, float *sHi, int tid, int b, int r0, int c0) {
    const float *Xb = p.X + (int64_t)b * p.inR * p.inC;
    const int ro = r0 - C::HH, co = c0 - C::HC;
    const bool interior = ro >= 0 && ro + C::TR + 2 * C::HH <= p.inR && co >= 0 && co + C::W <= p.inC;
    for (int task = tid; task < C::NS * C::W; task += DT_NT) {
        int strip = task / C::W, cc = task - strip * C::W;
        float w[C::WN];
        if (interior) {
            const float *src = Xb + (int64_t)(ro + strip * C::RS) * p.inC + (co + cc);
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

}  // namespace dt2d
