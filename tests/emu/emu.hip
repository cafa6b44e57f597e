// TEST-ONLY host emulator of the fused 2-D tile programs.
//
// Steps through the very same __host__ __device__ phase functions the gfx950 kernels call
// (dtcwt_amd/csrc/fused2d_tiles.hpp, same tile configurations from fused2d_table.hpp), one
// workgroup at a time, thread by thread, phase by phase, on the CPU, with plain arrays in
// place of LDS.  It exists so the index algebra of the kernels can be checked against the
// oracle in the CPU test-suite (no GPU in the build container).  It is not linked into
// libdtcwt_hip.so, not declared in include/dtcwt_hip.h and never loaded by dtcwt_amd.
#include <cstring>
#include <vector>

#include "fused2d_tiles.hpp"
#include "fused2d_tiles_v2.hpp"
#include "march2d.hpp"
#include "fused2d_table.hpp"
#include "fused3d_tiles.hpp"
#include "fused3d_inv_tiles.hpp"

using namespace dt2d;

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static void put_taps(float *dst, const double *src, int m) {
    for (int k = 0; k < DT_MAXT; ++k) dst[k] = k < m ? (float)src[k] : 0.f;
}
static double dotd(const double *a, const double *b, int m) {
    double s = 0;
    for (int k = 0; k < m; ++k) s += a[k] * b[k];
    return s;
}

template <class C>
static int run_fwd1(Fwd1Params p) {
    p.tilesR = cdiv(p.LR, C::TR); p.tilesC = cdiv(p.LC, C::TC);
    dt_pack_c01<C::M0, C::M1>(p);
    std::vector<float> smem(C::LDS_FLOATS + 4 * STAGE_FLOATS_PER_WAVE + 4);
    float *base = smem.data();
    while (((uintptr_t)base) & 15) ++base;
    float *sLo = base, *sHi = sLo + C::SL, *sBa = sHi + C::SL, *stage = base + C::LDS_FLOATS;
    constexpr int NQ = (C::TR / 2) * (C::TC / 2);
    for (int b = 0; b < p.B; ++b)
        for (int tr = 0; tr < p.tilesR; ++tr)
            for (int tc = 0; tc < p.tilesC; ++tc) {
                int r0 = tr * C::TR, c0 = tc * C::TC;
                for (int t = 0; t < DT_NT; ++t) fwd1d_cols<C>(p, sLo, sHi, t, b, r0, c0, sBa);
                for (int q = 0; q < NQ; q += DT_NT) {      // a wave's two halves run as two passes
                    for (int t = 0; t < DT_NT; ++t) fwd1s_rows_compute<C>(p, sLo, sHi, stage, t, q, b, r0, c0, sBa);
                    for (int t = 0; t < DT_NT; ++t) fwd1s_rows_flush<C>(p, stage, t, q, b, r0, c0);
                }
            }
    return 0;
}

template <class C>
static int run_fwd2(Fwd2Params p) {
    p.tilesR = cdiv(p.LR / 2, C::TR); p.tilesC = cdiv(p.LC / 2, C::TC);
    dt_pack_lh(p);
    std::vector<float> smem(C::LDS_FLOATS + 4 * STAGE_FLOATS_PER_WAVE + 4);
    float *base = smem.data();
    while (((uintptr_t)base) & 15) ++base;
    float *sLo = base, *sHi = sLo + C::SL, *sBa = sHi + C::SL, *stage = base + C::LDS_FLOATS;
    for (int b = 0; b < p.B; ++b)
        for (int tr = 0; tr < p.tilesR; ++tr)
            for (int tc = 0; tc < p.tilesC; ++tc) {
                int r0 = tr * C::TR, c0 = tc * C::TC;
                for (int t = 0; t < DT_NT; ++t) fwd2d_cols<C>(p, sLo, sHi, t, b, r0, c0, sBa);
                for (int q = 0; q < C::TI * C::TJ; q += DT_NT) {
                    for (int t = 0; t < DT_NT; ++t) fwd2s_rows_compute<C>(p, sLo, sHi, stage, t, q, b, r0, c0, sBa);
                    for (int t = 0; t < DT_NT; ++t) fwd2s_rows_flush<C>(p, stage, t, q, b, r0, c0);
                }
            }
    return 0;
}

template <class C>
static int run_inv1(Inv1Params p) {
    p.tilesR = cdiv(p.R, C::TR); p.tilesC = cdiv(p.C, C::TC);
    dt_pack_g01<C::M0, C::M1>(p);
    std::vector<float> smem(C::LDS_ALIASED + 4);
    float *base = smem.data();
    while (((uintptr_t)base) & 15) ++base;
    float *srec = base, *y1 = base, *y2 = y1 + C::SY, *y3 = y2 + C::SY;    // y planes alias the records
    static float wz[DT_NT][C::WN], w1[DT_NT][C::WN], w2[DT_NT][C::WN], w3[DT_NT][C::WN];
    for (int b = 0; b < p.B; ++b)
        for (int tr = 0; tr < p.tilesR; ++tr)
            for (int tc = 0; tc < p.tilesC; ++tc) {
                int r0 = tr * C::TR, c0 = tc * C::TC;
                const float *Yhb = p.Yh + (int64_t)b * (p.R / 2) * (p.C / 2) * 12;
                for (int t = 0; t < DT_NT; ++t) inv1r_fetch<C>(p, wz[t], t, b, r0, c0);
                for (int t = 0; t < DT_NT; ++t)
                    inv_rec_stage<C::QR, C::QC>(Yhb, p.R, p.C, srec, r0 - C::HE, c0 - C::HE, t);
                for (int t = 0; t < DT_NT; ++t) inv1r_gather<C>(p, srec, w1[t], w2[t], w3[t], t, r0, c0);
                for (int t = 0; t < DT_NT; ++t) inv1r_fir<C>(p, wz[t], w1[t], w2[t], w3[t], y1, y2, t, y3);
                for (int t = 0; t < DT_NT; ++t) inv1d_rows<C>(p, y1, y2, t, b, r0, c0, y3);
            }
    return 0;
}

template <class C>
static int run_inv2(Inv2Params p) {
    p.tilesR = cdiv(p.zr, C::TR); p.tilesC = cdiv(p.zc, C::TC);
    std::vector<float> smem(C::LDS_ALIASED + 4);
    float *base = smem.data();
    while (((uintptr_t)base) & 15) ++base;
    float *srec = base, *y1 = base, *y2 = y1 + C::SY, *y3 = y2 + C::SY;
    static float wz[DT_NT][C::WS], w1[DT_NT][C::WS], w2[DT_NT][C::WS], w3[DT_NT][C::WS];
    for (int b = 0; b < p.B; ++b)
        for (int tr = 0; tr < p.tilesR; ++tr)
            for (int tc = 0; tc < p.tilesC; ++tc) {
                int r0 = tr * C::TR, c0 = tc * C::TC;
                const float *Yhb = p.Yh + (int64_t)b * (p.zr / 2) * (p.zc / 2) * 12;
                for (int t = 0; t < DT_NT; ++t) inv2r_fetch<C>(p, wz[t], t, b, r0, c0);
                for (int t = 0; t < DT_NT; ++t)
                    inv_rec_stage<C::QR, C::QC>(Yhb, p.zr, p.zc, srec, r0 + C::ORG, c0 + C::ORG, t);
                const bool std_set = p.lo_pos && !p.hi_pos && (!C::BP || !p.bp_pos);     // as launch_inv2 decides
                if (!C::BP) {
                    // k_inv2s: the column phase in two halves, y1 in a plane of its own, y2 over the records
                    static float y1s[C::SY];
                    for (int t = 0; t < DT_NT; ++t) inv2r_gather_half<C, 0>(p, srec, w1[t], w3[t], t, r0, c0);
                    for (int t = 0; t < DT_NT; ++t) {
                        if (std_set) inv2r_fir_plane<C, true>(p, wz[t], w1[t], y1s, t);
                        else inv2r_fir_plane<C, false>(p, wz[t], w1[t], y1s, t);
                    }
                    for (int t = 0; t < DT_NT; ++t) inv2r_gather_half<C, 1>(p, srec, w2[t], w2[t], t, r0, c0);
                    for (int t = 0; t < DT_NT; ++t) {
                        if (std_set) inv2r_fir_plane<C, true>(p, w2[t], w3[t], y1, t);       // y2 = the record buffer
                        else inv2r_fir_plane<C, false>(p, w2[t], w3[t], y1, t);
                    }
                    for (int t = 0; t < DT_NT; ++t) {
                        if (std_set) inv2_rows<C, true>(p, y1s, y1, t, b, r0, c0, nullptr);
                        else inv2_rows<C, false>(p, y1s, y1, t, b, r0, c0, nullptr);
                    }
                    continue;
                }
                for (int t = 0; t < DT_NT; ++t) inv2r_gather<C>(p, srec, w1[t], w2[t], w3[t], t, r0, c0);
                for (int t = 0; t < DT_NT; ++t) {
                    if (std_set) inv2r_fir<C, false, true>(p, wz[t], w1[t], w2[t], w3[t], y1, y2, t, y3);
                    else inv2r_fir<C, false, false>(p, wz[t], w1[t], w2[t], w3[t], y1, y2, t, y3);
                }
                for (int t = 0; t < DT_NT; ++t) {
                    if (std_set) inv2_rows<C, true>(p, y1, y2, t, b, r0, c0, y3);
                    else inv2_rows<C, false>(p, y1, y2, t, b, r0, c0, y3);
                }
            }
    return 0;
}

template <class C>
static int run_fwd3_l1(dt3d::Fwd3L1Params p, int chunk) {
    using namespace dt3d;
    p.tilesJ = cdiv(p.n1, C::TJ); p.tilesK = cdiv(p.n2, C::TK);
    p.chunk = chunk; p.chunks = cdiv(p.n0, chunk);
    dt3d::f3l1_pack_taps<C>(p);
    std::vector<float> smem(C::LDS_FLOATS + 4);
    float *base = smem.data();
    while (((uintptr_t)base) & 15) ++base;
    float *S0 = base, *S1 = base + C::S0F, *stage = S1 + C::S1F, *XR = base + C::XR0;
    static Fwd3L1State<C> st[C::NT];
    for (int ch = 0; ch < p.chunks; ++ch)
        for (int tj = 0; tj < p.tilesJ; ++tj)
            for (int tk = 0; tk < p.tilesK; ++tk) {
                int j0 = tj * C::TJ, k0 = tk * C::TK, i0 = ch * p.chunk;
                int iend = i0 + p.chunk < p.n0 ? i0 + p.chunk : p.n0;
                const bool full = j0 + C::TJ <= p.n1 && k0 + C::TK <= p.n2;
                for (int t = 0; t < C::NT; ++t) {
                    f3l1_init<C>(p, st[t], t, j0, k0); f3l1_prologue<C>(p, st[t], XR, t, i0); f3l1_prefetch<C>(p, st[t], t, i0);
                }
                static float od[C::NT][8][4];
                int rot = 0;
                for (int i = i0; i < iend; i += 2) {
                    for (int t = 0; t < C::NT; ++t) f3l1_axis0<C, 0>(p, st[t], S0, XR, t, rot);
                    for (int t = 0; t < C::NT; ++t) f3l1_axis2<C>(p, S0, S1, t);
                    for (int t = 0; t < C::NT; ++t) {
                        if (full) f3l1_axis1<C, true>(p, st[t].ev, S1, t, i, j0, k0);
                        else f3l1_axis1<C, false>(p, st[t].ev, S1, t, i, j0, k0);
                    }
                    for (int t = 0; t < C::NT; ++t) f3l1_axis0<C, 1>(p, st[t], S0, XR, t, rot);
                    for (int t = 0; t < C::NT; ++t) f3l1_axis2<C>(p, S0, S1, t);
                    for (int t = 0; t < C::NT; ++t) {
                        f3l1_rotate2<C>(st[t], XR, t, rot);
                        f3l1_prefetch<C>(p, st[t], t, i + 2);
                        if (full) f3l1_axis1<C, true>(p, od[t], S1, t, i + 1, j0, k0);
                        else f3l1_axis1<C, false>(p, od[t], S1, t, i + 1, j0, k0);
                    }
                    rot = rot + 2 >= C::MR ? rot + 2 - C::MR : rot + 2;
                    for (int pass = 0; pass < C::SP; ++pass) {
                        for (int t = 0; t < C::NT; ++t) f3l1_pack_stage<C>(st[t].ev, od[t], stage, t, pass);
                        for (int t = 0; t < C::NT; ++t) {
                            if (full) f3l1_pack_flush<C, true>(p, stage, t, pass, i + 1, j0, k0);
                            else f3l1_pack_flush<C, false>(p, stage, t, pass, i + 1, j0, k0);
                        }
                    }
                }
            }
    return 0;
}

template <class C>
static int run_fwd3_l2(Fwd2Params a, dt3d::Fwd3L2Params b, float *planes) {
    a.tilesR = cdiv(a.LR / 2, C::TR); a.tilesC = cdiv(a.LC / 2, C::TC);
    dt_pack_lh(a); dt_pack_lh(b);
    std::vector<float> smem(C::LDS_FLOATS + 4);
    float *base = smem.data();
    while (((uintptr_t)base) & 15) ++base;
    float *sLo = base, *sHi = sLo + C::SL;
    for (int s = 0; s < a.B; ++s)
        for (int tr = 0; tr < a.tilesR; ++tr)
            for (int tc = 0; tc < a.tilesC; ++tc) {
                int r0 = tr * C::TR, c0 = tc * C::TC;
                for (int t = 0; t < DT_NT; ++t) fwd2d_cols<C>(a, sLo, sHi, t, s, r0, c0);
                for (int q = 0; q < C::TI * C::TJ; q += DT_NT)
                    for (int t = 0; t < DT_NT; ++t) dt3d::fwd2p_rows<C>(a, sLo, sHi, planes, b.pstride, t, q, s, r0, c0);
            }
    int cells = (b.O0 / 2) * (b.O1 / 2) * (b.O2 / 2);
    std::vector<float> slab(64 * dt3d::REC_LDS + 4);
    float *ws = slab.data();
    while (((uintptr_t)ws) & 15) ++ws;
    for (int first = 0; first < cells; first += 32) {      // two lanes per cell, as the kernel runs coarse levels
        for (int l = 0; l < 64; ++l)
            dt3d::f3l2_axis0_stage<C::M, 2>(b, first + (l & 31), ws + (l & 31) * dt3d::REC_LDS, 2 * (l >> 5));
        for (int l = 0; l < 64; ++l) dt3d::f3l2_axis0_flush<32>(b, first, l, ws);
    }
    return 0;
}

// ---- 3-D inverse: pass A (march) and pass B (2-D tile passes over the four planes)
template <class F>
static void run_inv3_axis0(dt3d::Inv3AParams p, int chunk) {
    using namespace dt3d;
    p.tilesJ = cdiv(p.n1 / 2, I3_CJ); p.tilesK = cdiv(p.n2 / 2, I3_CK);
    p.chunk = chunk; p.chunks = cdiv(p.n0 / 2, chunk);
    std::vector<float> mem(2 * I3_SLAB + 4);
    float *base = mem.data();
    while (((uintptr_t)base) & 15) ++base;
    float *slab[2] = {base, base + I3_SLAB};
    static Inv3AState<F> st[DT_NT];
    static float out[DT_NT][F::NOUT][4];
    for (int ch = 0; ch < p.chunks; ++ch)
        for (int tj = 0; tj < p.tilesJ; ++tj)
            for (int tk = 0; tk < p.tilesK; ++tk) {
                const int cj0 = tj * I3_CJ, ck0 = tk * I3_CK, c0 = ch * p.chunk;
                const int c1 = c0 + p.chunk < p.n0 / 2 ? c0 + p.chunk : p.n0 / 2;
                const int cs = c0 - (2 * F::HP + 1);
                for (int t = 0; t < DT_NT; ++t) {
                    memset(&st[t], 0, sizeof(st[t]));
                    i3a_issue_rec<F>(p, st[t], t, cj0, ck0, cs + F::HP + 1);
                    i3a_issue_low<F>(p, st[t], t, cj0, ck0, cs + F::HP + 1);
                    i3a_slab_write<F>(st[t], slab[0], t);
                    i3a_issue_rec<F>(p, st[t], t, cj0, ck0, cs + F::HP + 2);
                }
                for (int c = cs; c < c1; ++c) {
                    const int buf = (c - cs) & 1;
                    for (int t = 0; t < DT_NT; ++t) {
                        if (c >= c0) F::compute(p, st[t].ra, st[t].rb, out[t]);
                        i3a_push<F>(p, st[t], slab[buf], t, c + F::HP + 1);
                        i3a_slab_write<F>(st[t], slab[buf ^ 1], t);
                        if (c + 1 < c1) {
                            i3a_issue_rec<F>(p, st[t], t, cj0, ck0, c + F::HP + 3);
                            i3a_issue_low<F>(p, st[t], t, cj0, ck0, c + F::HP + 2);
                        }
                        if (c >= c0) i3a_store<F>(p, out[t], t, cj0, ck0, c);
                    }
                }
            }
}

// level 1: the march with the axis-2 merge (k_inv3_l1_axis02), then the axis-1 merge (k_inv3_l1_axis1)
template <class F, class G>
static void run_inv3_l1_axis02(dt3d::Inv3AParams p, int chunk) {
    using namespace dt3d;
    const int e2 = p.n2 / 2;
    p.hal = e2 > G::CK ? 2 : 0;
    p.tilesJ = cdiv(p.n1 / 2, G::CJ); p.tilesK = p.hal ? cdiv(e2, G::CK - 4) : 1;
    p.chunk = chunk; p.chunks = cdiv(p.n0 / 2, chunk);
    std::vector<float> mem(2 * G::SLAB + I3Ex<G>::FLOATS + 4);
    float *base = mem.data();
    while (((uintptr_t)base) & 15) ++base;
    float *slab[2] = {base, base + G::SLAB}, *E = base + 2 * G::SLAB;
    static Inv3TState<F, G> st[G::NT];
    static float out[G::NT][2][4];
    for (int ch = 0; ch < p.chunks; ++ch)
        for (int tj = 0; tj < p.tilesJ; ++tj)
            for (int tk = 0; tk < p.tilesK; ++tk) {
                const int cj0 = tj * G::CJ, ck0 = tk * (G::CK - 2 * p.hal) - p.hal, c0 = ch * p.chunk;
                const int c1 = c0 + p.chunk < p.n0 / 2 ? c0 + p.chunk : p.n0 / 2;
                const int q0 = c0 - F::HP, q1 = c1 - 1 + F::HP;
                for (int t = 0; t < G::NT; ++t) {
                    memset(&st[t], 0, sizeof(st[t]));
                    i3a_issue_rec<F, G>(p, st[t], t, cj0, ck0, q0);
                    i3a_issue_low<F, G>(p, st[t], t, cj0, ck0, q0);
                    i3a_slab_write<F, G>(st[t], slab[0], t);
                    i3a_issue_rec<F, G>(p, st[t], t, cj0, ck0, q0 + 1);
                }
                for (int q = q0; q <= q1; ++q) {
                    const int buf = (q - q0) & 1, c = q - F::HP;
                    for (int i = 0; i < I3Ex<G>::FLOATS; ++i) E[i] = NAN;       // nothing stale may be read
                    for (int t = 0; t < G::NT; ++t) {
                        i3a_accumulate<F, G>(p, st[t], slab[buf], t, q, out[t]);
                        i3a_slab_write<F, G>(st[t], slab[buf ^ 1], t);
                        if (q < q1) {
                            i3a_issue_rec<F, G>(p, st[t], t, cj0, ck0, q + 2);
                            i3a_issue_low<F, G>(p, st[t], t, cj0, ck0, q + 1);
                        }
                        if (c >= c0) i3a_exchange<F, G>(p, out[t], E, t, ck0);
                    }
                    if (c >= c0) for (int t = 0; t < G::NT; ++t) i3a_merge_k<F, G>(p, E, t, cj0, ck0, c);
                }
            }
}

template <class F>
static void run_inv3_l1(dt3d::Inv3AParams a, dt3d::Inv3BParams b, int chunk) {
    using namespace dt3d;
    const int e2 = a.n2 / 2;
    if (e2 <= 32) run_inv3_l1_axis02<F, I3Geo<512, 32>>(a, chunk);
    else if (e2 <= 64) run_inv3_l1_axis02<F, I3Geo<512, 64>>(a, chunk);
    else run_inv3_l1_axis02<F, I3Geo<512, 128>>(a, chunk);
    const int vec = (b.n2 & 3) == 0 ? 4 : 2;
    b.kvecs = b.n2 / vec; b.strips = cdiv(b.n1, 8);
    const int64_t tasks = (int64_t)b.kvecs * b.strips * b.S;
    for (int64_t t = 0; t < tasks; ++t) {
        if (vec == 4) i3b_axis1<F, 4, 8>(b, t);
        else i3b_axis1<F, 2, 8>(b, t);
    }
}

template <class C>
static void run_inv3_l2_planes(Inv2Params p, const float *planes, int64_t ps) {
    p.tilesR = cdiv(p.zr, C::TR); p.tilesC = cdiv(p.zc, C::TC);
    std::vector<float> smem(2 * C::SY + 4);
    float *base = smem.data();
    while (((uintptr_t)base) & 15) ++base;
    float *y1 = base, *y2 = y1 + C::SY;
    static float wz[DT_NT][C::WS], w1[DT_NT][C::WS], w2[DT_NT][C::WS], w3[DT_NT][C::WS];
    for (int b = 0; b < p.B; ++b)
        for (int tr = 0; tr < p.tilesR; ++tr)
            for (int tc = 0; tc < p.tilesC; ++tc) {
                int r0 = tr * C::TR, c0 = tc * C::TC;
                for (int t = 0; t < DT_NT; ++t) {
                    inv2r_fetch_from<C, true>(p, planes, wz[t], t, b, r0, c0);
                    inv2r_fetch_from<C, true>(p, planes + 2 * ps, w1[t], t, b, r0, c0);
                    inv2r_fetch_from<C, true>(p, planes + ps, w2[t], t, b, r0, c0);
                    inv2r_fetch_from<C, true>(p, planes + 3 * ps, w3[t], t, b, r0, c0);
                    inv2r_fir<C, true>(p, wz[t], w1[t], w2[t], w3[t], y1, y2, t);
                }
                for (int t = 0; t < DT_NT; ++t) inv2_rows<C>(p, y1, y2, t, b, r0, c0);
            }
}

#define EMU_FWD1(TR, TC, RS, A, B_) if (m0 == A && m1 == B_) return run_fwd1<Fwd1DCfg<TR, TC, RS, A, B_>>(p);
#define EMU_INV1(TR, TC, RS, A, B_) if (m0 == A && m1 == B_) return run_inv1<Inv1RCfg<TR, TC, RS, A, B_>>(p);
#define EMU_FWD2(TR, TC, PS, M) if (m == M) return run_fwd2<Fwd2DCfg<TR, TC, PS, M>>(p);
#define EMU_INV2(TR, TC, JS, M) if (m == M) return run_inv2<Inv2RCfg<TR, TC, JS, M>>(p);

extern "C" {

// the level-2 taps of the marching forward kernel as dtm_pack_qshift() lays them out: ta / tb by window offset
// (A = sum_t ta[t] w[2t], B = sum_t tb[t] w[2t + 1] over the 2M-sample window starting at sample 4i - M + 2)
int emu_march_pack_qshift(int M, const float *l_a, const float *l_b, const float *h_a, const float *h_b, float *out) {
    dtm::Fwd12mParams p{};
    dtm::dtm_pack_qshift(p, M, l_a, l_b, h_a, h_b);
    for (int t = 0; t < dtm::MAXT2; ++t) {
        out[t] = p.ta_lo[t]; out[dtm::MAXT2 + t] = p.tb_lo[t]; out[2 * dtm::MAXT2 + t] = p.ta_hi[t]; out[3 * dtm::MAXT2 + t] = p.tb_hi[t];
        if (p.ta2[2 * t] != p.ta_lo[t] || p.ta2[2 * t + 1] != p.ta_hi[t] || p.tb2[2 * t] != p.tb_lo[t] || p.tb2[2 * t + 1] != p.tb_hi[t]) return -1;
    }
    return dtm::MAXT2;
}

// the job order of the marching launches (march2d.hpp): for workgroup w of the grid dtm_set_jobs() asks for, which
// (strip, band, image) it marches, or -1 x 3 when it leaves at once; returns the grid size
int emu_march_jobs(int B, int R, int nstrip, int band_rows, int *out, int cap) {
    dtm::MarchJobs j{};
    const int grid = (int)dtm::dtm_set_jobs(j, B, R, nstrip, band_rows);
    for (int w = 0; w < grid && w < cap; ++w) {
        int s, bd, b;
        if (dtm::dtm_job(j, w, s, bd, b)) { out[3 * w] = s; out[3 * w + 1] = bd; out[3 * w + 2] = b; }
        else out[3 * w] = out[3 * w + 1] = out[3 * w + 2] = -1;
    }
    return grid;
}

// all pointers are HOST pointers
int emu_fwd1(int m0, int m1, const float *X, float *LoLo, float *Yh, int B, int inR, int inC,
             const double *h0, const double *h1) {
    Fwd1Params p{};
    p.X = X; p.LoLo = LoLo; p.Yh = Yh; p.B = B; p.inR = inR; p.inC = inC;
    p.LR = inR + (inR & 1); p.LC = inC + (inC & 1);
    put_taps(p.h0, h0, m0); put_taps(p.h1, h1, m1);
    DT_FWD1_TABLE(EMU_FWD1)
    return -3;
}

int emu_fwd2(int m, const float *X, float *LoLo, float *Yh, int B, int inR, int inC,
             const double *la, const double *lb, const double *ha, const double *hb) {
    Fwd2Params p{};
    p.X = X; p.LoLo = LoLo; p.Yh = Yh; p.B = B; p.inR = inR; p.inC = inC;
    p.padR = (inR % 4) ? 1 : 0; p.padC = (inC % 4) ? 1 : 0;
    p.LR = inR + 2 * p.padR; p.LC = inC + 2 * p.padC;
    put_taps(p.l_a, la, m); put_taps(p.l_b, lb, m); put_taps(p.h_a, ha, m); put_taps(p.h_b, hb, m);
    p.lo_a_first = dotd(la, lb, m) > 0; p.hi_a_first = dotd(ha, hb, m) > 0;
    DT_FWD2_TABLE(EMU_FWD2)
    return -3;
}

int emu_inv1(int m0, int m1, const float *Z, const float *Yh, float *X, int B, int R, int C,
             const double *gain6, const double *g0, const double *g1) {
    Inv1Params p{};
    p.Z = Z; p.Yh = Yh; p.X = X; p.B = B; p.R = R; p.C = C;
    for (int d = 0; d < 6; ++d) p.g[d] = (float)(0.70710678118654752440 * gain6[d]);
    put_taps(p.g0, g0, m0); put_taps(p.g1, g1, m1);
    DT_INV1_TABLE(EMU_INV1)
    return -3;
}

int emu_inv2(int m, const float *Z, const float *Yh, float *Out, int B, int zr, int zc, int cropR,
             int cropC, const double *gain6, const double *la, const double *lb, const double *ha,
             const double *hb) {
    Inv2Params p{};
    p.Z = Z; p.Yh = Yh; p.Out = Out; p.B = B; p.zr = zr; p.zc = zc; p.cropR = cropR; p.cropC = cropC;
    for (int d = 0; d < 6; ++d) p.g[d] = (float)(0.70710678118654752440 * gain6[d]);
    put_taps(p.l_a, la, m); put_taps(p.l_b, lb, m); put_taps(p.h_a, ha, m); put_taps(p.h_b, hb, m);
    p.lo_pos = dotd(la, lb, m) > 0; p.hi_pos = dotd(ha, hb, m) > 0;
    // the library's choice: the small tiles where the level has fewer than two 16 x 56 tiles per CU of an MI355X
    if ((int64_t)cdiv(zr, 16) * cdiv(zc, 56) * B < DT_INV2_SMALL_BELOW) { DT_INV2_SMALL_TABLE(EMU_INV2) }
    DT_INV2_TABLE(EMU_INV2)
    return -3;
}

// the same level with the large tiles whatever the size
int emu_inv2_large(int m, const float *Z, const float *Yh, float *Out, int B, int zr, int zc, int cropR,
                   int cropC, const double *gain6, const double *la, const double *lb, const double *ha,
                   const double *hb) {
    Inv2Params p{};
    p.Z = Z; p.Yh = Yh; p.Out = Out; p.B = B; p.zr = zr; p.zc = zc; p.cropR = cropR; p.cropC = cropC;
    for (int d = 0; d < 6; ++d) p.g[d] = (float)(0.70710678118654752440 * gain6[d]);
    put_taps(p.l_a, la, m); put_taps(p.l_b, lb, m); put_taps(p.h_a, ha, m); put_taps(p.h_b, hb, m);
    p.lo_pos = dotd(la, lb, m) > 0; p.hi_pos = dotd(ha, hb, m) > 0;
    DT_INV2_TABLE(EMU_INV2)
    return -3;
}

// band-pass variants: h2 / g2 (level 1) and the (b, a) band-pass q-shift pairs (level >= 2)
int emu_fwd1_bp(int m0, int m1, int m2, const float *X, float *LoLo, float *Yh, int B, int inR, int inC,
                const double *h0, const double *h1, const double *h2) {
    Fwd1Params p{};
    p.X = X; p.LoLo = LoLo; p.Yh = Yh; p.B = B; p.inR = inR; p.inC = inC;
    p.LR = inR + (inR & 1); p.LC = inC + (inC & 1);
    put_taps(p.h0, h0, m0); put_taps(p.h1, h1, m1); put_taps(p.h2, h2, m2);
#define EMU_FWD1_BP(TR, TC, RS, A, B_, C2) if (m0 == A && m1 == B_ && m2 == C2) return run_fwd1<Fwd1DCfg<TR, TC, RS, A, B_, C2>>(p);
    DT_FWD1_BP_TABLE(EMU_FWD1_BP)
    return -3;
}

int emu_fwd2_bp(int m, const float *X, float *LoLo, float *Yh, int B, int inR, int inC, const double *la,
                const double *lb, const double *ha, const double *hb, const double *ba, const double *bb) {
    Fwd2Params p{};
    p.X = X; p.LoLo = LoLo; p.Yh = Yh; p.B = B; p.inR = inR; p.inC = inC;
    p.padR = (inR % 4) ? 1 : 0; p.padC = (inC % 4) ? 1 : 0;
    p.LR = inR + 2 * p.padR; p.LC = inC + 2 * p.padC;
    put_taps(p.l_a, la, m); put_taps(p.l_b, lb, m); put_taps(p.h_a, ha, m); put_taps(p.h_b, hb, m);
    put_taps(p.b_a, ba, m); put_taps(p.b_b, bb, m);
    p.lo_a_first = dotd(la, lb, m) > 0; p.hi_a_first = dotd(ha, hb, m) > 0; p.bp_a_first = dotd(ba, bb, m) > 0;
#define EMU_FWD2_BP(TR, TC, PS, M) if (m == M) return run_fwd2<Fwd2DCfg<TR, TC, PS, M, true>>(p);
    DT_FWD2_BP_TABLE(EMU_FWD2_BP)
    return -3;
}

int emu_inv1_bp(int m0, int m1, int m2, const float *Z, const float *Yh, float *X, int B, int R, int C,
                const double *gain6, const double *g0, const double *g1, const double *g2) {
    Inv1Params p{};
    p.Z = Z; p.Yh = Yh; p.X = X; p.B = B; p.R = R; p.C = C;
    for (int d = 0; d < 6; ++d) p.g[d] = (float)(0.70710678118654752440 * gain6[d]);
    put_taps(p.g0, g0, m0); put_taps(p.g1, g1, m1); put_taps(p.g2, g2, m2);
#define EMU_INV1_BP(TR, TC, RS, A, B_, C2) if (m0 == A && m1 == B_ && m2 == C2) return run_inv1<Inv1RCfg<TR, TC, RS, A, B_, C2>>(p);
    DT_INV1_BP_TABLE(EMU_INV1_BP)
    return -3;
}

int emu_inv2_bp(int m, const float *Z, const float *Yh, float *Out, int B, int zr, int zc, int cropR, int cropC,
                const double *gain6, const double *la, const double *lb, const double *ha, const double *hb,
                const double *ba, const double *bb) {
    Inv2Params p{};
    p.Z = Z; p.Yh = Yh; p.Out = Out; p.B = B; p.zr = zr; p.zc = zc; p.cropR = cropR; p.cropC = cropC;
    for (int d = 0; d < 6; ++d) p.g[d] = (float)(0.70710678118654752440 * gain6[d]);
    put_taps(p.l_a, la, m); put_taps(p.l_b, lb, m); put_taps(p.h_a, ha, m); put_taps(p.h_b, hb, m);
    put_taps(p.b_a, ba, m); put_taps(p.b_b, bb, m);
    p.lo_pos = dotd(la, lb, m) > 0; p.hi_pos = dotd(ha, hb, m) > 0; p.bp_pos = dotd(ba, bb, m) > 0;
#define EMU_INV2_BP(TR, TC, JS, M) if (m == M) return run_inv2<Inv2RCfg<TR, TC, JS, M, true>>(p);
    DT_INV2_BP_TABLE(EMU_INV2_BP)
    return -3;
}

int emu_fwd3_l1(int m0, int m1, const float *X, float *LLL, float *Yh, int n0, int n1, int n2, int chunk,
                const double *h0, const double *h1) {
    dt3d::Fwd3L1Params p{};
    p.X = X; p.LLL = LLL; p.Yh = Yh; p.n0 = n0; p.n1 = n1; p.n2 = n2;
    put_taps(p.h0, h0, m0); put_taps(p.h1, h1, m1);
    if (m0 == 5 && m1 == 7) return run_fwd3_l1<dt3d::Fwd3L1Cfg<5, 7>>(p, chunk);
    if (m0 == 9 && m1 == 7) return run_fwd3_l1<dt3d::Fwd3L1Cfg<9, 7>>(p, chunk);
    if (m0 == 5 && m1 == 3) {       // as dtcwt_hip_fwd3_level1: centred zero-padded 7 taps
        for (int k = 0; k < DT_MAXT; ++k) p.h1[k] = 0.f;
        for (int k = 0; k < 3; ++k) p.h1[k + 2] = (float)h1[k];
        return run_fwd3_l1<dt3d::Fwd3L1Cfg<5, 7>>(p, chunk);
    }
    return -3;
}

int emu_fwd3_l2(int m, const float *X, float *planes, float *LLL, float *Yh, int n0, int n1, int n2, int pad0,
                int pad1, int pad2, const double *h0b, const double *h0a, const double *h1b, const double *h1a) {
    Fwd2Params a{};
    a.X = X; a.B = n0; a.inR = n1; a.inC = n2; a.padR = pad1; a.padC = pad2;
    a.LR = n1 + 2 * pad1; a.LC = n2 + 2 * pad2;
    put_taps(a.l_a, h0b, m); put_taps(a.l_b, h0a, m); put_taps(a.h_a, h1b, m); put_taps(a.h_b, h1a, m);
    a.lo_a_first = dotd(h0b, h0a, m) > 0; a.hi_a_first = dotd(h1b, h1a, m) > 0;
    dt3d::Fwd3L2Params b{};
    b.P = planes; b.LLL = LLL; b.Yh = Yh; b.n0 = n0; b.pad0 = pad0; b.L0 = n0 + 2 * pad0;
    b.O0 = b.L0 / 2; b.O1 = a.LR / 2; b.O2 = a.LC / 2;
    b.pstride = (int64_t)n0 * b.O1 * b.O2;
    b.lo_a_first = a.lo_a_first; b.hi_a_first = a.hi_a_first;
    put_taps(b.l_a, h0b, m); put_taps(b.l_b, h0a, m); put_taps(b.h_a, h1b, m); put_taps(b.h_b, h1a, m);
#define EMU_L2(TR, TC, PS, M) if (m == M) return run_fwd3_l2<Fwd2DCfg<TR, TC, PS, M>>(a, b, planes);
    DT_FWD2_TABLE(EMU_L2)
    return -3;
}

int emu_inv3_l1(int m0, int m1, const float *LLL, const float *Yh, float *planes, float *Z, int n0, int n1,
                int n2, int chunk, const double *g0, const double *g1) {
    dt3d::Inv3AParams a{};
    a.LLL = LLL; a.Yh = Yh; a.P = planes; a.n0 = n0; a.n1 = n1; a.n2 = n2; a.S = n0; a.crop0 = 0;
    a.pstride = (int64_t)n0 * n1 * n2;
    put_taps(a.l_a, g0, m0); put_taps(a.h_a, g1, m1);
    dt3d::Inv3BParams b{};
    b.Q = planes; b.pstride = a.pstride; b.Z = Z; b.S = n0; b.n1 = n1; b.n2 = n2;
    put_taps(b.g0, g0, m0); put_taps(b.g1, g1, m1);
    if (m0 == 7 && m1 == 5) { run_inv3_l1<dt3d::Inv3L1<7, 5>>(a, b, chunk); return 0; }
    if (m0 == 7 && m1 == 9) { run_inv3_l1<dt3d::Inv3L1<7, 9>>(a, b, chunk); return 0; }
    if (m0 == 3 && m1 == 5) { run_inv3_l1<dt3d::Inv3L1<3, 5>>(a, b, chunk); return 0; }
    return -3;
}

int emu_inv3_l2(int m, const float *LLL, const float *Yh, float *planes, float *Z, int n0, int n1, int n2,
                int crop0, int crop1, int crop2, int chunk, const double *g0b, const double *g0a,
                const double *g1b, const double *g1a) {
    dt3d::Inv3AParams a{};
    a.LLL = LLL; a.Yh = Yh; a.P = planes; a.n0 = n0; a.n1 = n1; a.n2 = n2; a.S = 2 * n0 - 2 * crop0; a.crop0 = crop0;
    a.pstride = (int64_t)a.S * n1 * n2;
    a.lo_pos = dotd(g0b, g0a, m) > 0; a.hi_pos = dotd(g1b, g1a, m) > 0;
    put_taps(a.l_a, g0b, m); put_taps(a.l_b, g0a, m); put_taps(a.h_a, g1b, m); put_taps(a.h_b, g1a, m);
    Inv2Params b{};
    b.Out = Z; b.B = a.S; b.zr = n1; b.zc = n2; b.cropR = crop1; b.cropC = crop2;
    b.lo_pos = a.lo_pos; b.hi_pos = a.hi_pos;
    put_taps(b.l_a, g0b, m); put_taps(b.l_b, g0a, m); put_taps(b.h_a, g1b, m); put_taps(b.h_b, g1a, m);
    if (m == 10) {
        run_inv3_axis0<dt3d::Inv3L2<10>>(a, chunk);
        // planes whose width is a multiple of 64 go through the 8 x 64 tiles the library uses at coarse levels
        if (n2 % 64 == 0) run_inv3_l2_planes<Inv2RCfg<12, 64, 2, 10>>(b, planes, a.pstride);
        else run_inv3_l2_planes<Inv2RCfg<16, 56, 2, 10>>(b, planes, a.pstride);
        return 0;
    }
    if (m == 14) { run_inv3_axis0<dt3d::Inv3L2<14>>(a, chunk); run_inv3_l2_planes<Inv2RCfg<16, 52, 2, 14>>(b, planes, a.pstride); return 0; }
    return -3;
}

}  // extern "C"
