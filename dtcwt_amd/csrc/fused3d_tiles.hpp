// Fused per-level tile programs of the float32 3-D DT-CWT forward transform.
//
// Level 1 (dtcwt/numpy/transform3d.py:208-289, odd-length biort filters, no decimation)
// ------------------------------------------------------------------------------------
// One workgroup (256 threads) owns a TJ x TK (axis 1 x axis 2) column of the volume and
// MARCHES along axis 0 over a chunk of slices.  Per thread a register ring holds the last
// 2H+1 raw input slices of a handful of (j, k) positions of the haloed tile; each step
//
//   axis 0   lo/hi of the ring (registers)                      -> S0[2][PJ][PK]   (LDS)
//   axis 2   lo/hi along k of both, 4 outputs per task          -> S1[4][PJ][TK]   (LDS)
//   axis 1   lo/hi along j of all four for the thread's 2x2 (j,k) cell  -> 8 octants
//
// and every second slice the 2x2x2 cells of the seven highpass octants are packed by
// cube2c (:532-579) into whole 224-byte records of Yh[n0/2][n1/2][n2/2][28] complex64,
// while the LLL octant goes to the level-1 lowpass volume.  The input is read once (plus
// halo, served by L2), every output is written once: 36 B/voxel against the ~130 B/voxel
// of three separate axis passes plus seven cube2c passes.
//
// The filters are separable and linear, so applying axis 0 first (the reference filters
// axis 2, 1, 0) changes float32 rounding only; symmetric extension commutes with the
// other axes' filters, so halo positions are simply reflected loads of X.
//
// Same conventions as fused2d_tiles.hpp: every phase is a __host__ __device__ function so
// the test-only emulator (tests/emu/) steps the identical index algebra on the CPU.
#pragma once
#include "fused2d_tiles.hpp"
#include "fused2d_tiles_v2.hpp"

// Keeps the consumption of a prefetched value (and so the s_waitcnt on it) at this program
// point: left alone, the compiler sinks it below the record stores of the step, where the
// wait also drains those stores.
#if defined(__HIP_DEVICE_COMPILE__)
#define DT_PIN_HERE(x) asm volatile("" : "+v"(x) : : "memory")
#else
#define DT_PIN_HERE(x) (void)(x)
#endif


namespace dt3d {

using dt2d::cmax;
using dt2d::f2;
using dt2d::f4;
using dt2d::reflect_i;

// A 224-byte record is 56 floats; in the LDS slabs consecutive records are 60 floats apart.
// 56 floats = 14 sixteen-byte slots makes lanes 4 apart (writes, 32 banks) or every second
// lane (16-byte reads, 64 banks) share banks; with 15 slots the lanes of a bank group all
// land on different banks (MI355X_MICROARCH.md, LDS lane groups).
constexpr int REC_LDS = 60;
// 16-byte piece `piece` (0 .. 14 n) of n consecutive records -> its float4 index in a slab
DT_HD int slab_f4(int piece) { int r = piece / 14; return r * (REC_LDS / 4) + (piece - 14 * r); }

struct Fwd3L1Params {
    const float *X;      // [n0][n1][n2]
    float *LLL;          // [n0][n1][n2]
    float *Yh;           // [n0/2][n1/2][n2/2][56 floats]
    int n0, n1, n2;      // all even
    int chunk;           // slices marched by one workgroup (even)
    int tilesJ, tilesK, chunks;
    float h0[DT_MAXT], h1[DT_MAXT];
    // cube2c's factor 1/2 rides on the filter taps (exact: a power of two), set up by f3l1_pack_taps(): the a0 = 1
    // half-volume takes it at axis 0 (h1s = h1 / 2 there), the a1 = 1 octants of the a0 = 0 half at axis 1 (h1s
    // again); only octant (0, 0, 1), which shares its axis-1 chain with the unscaled lowpass volume, is halved in
    // cube2c itself.  One scaled table instead of three: the kernel is short of scalar registers.
    float h1s[DT_MAXT];
    // (h0, h1) by window offset d = 0 .. 2H (the shorter filter zero-padded): the pairs of the axis-2 chains, and
    // the scalars of the axis-0 / axis-1 chains, from one table
    float c01[2 * DT_MAXT] __attribute__((aligned(8)));
};
template <class C>
inline void f3l1_pack_taps(Fwd3L1Params &p) {
    for (int k = 0; k < DT_MAXT; ++k) {
        p.h1s[k] = 0.5f * p.h1[k];
        const int k0 = C::H + C::H0 - k, k1 = C::H + C::H1 - k;       // k as window offset d
        p.c01[2 * k] = (k0 >= 0 && k0 < C::M0) ? p.h0[k0] : 0.f;
        p.c01[2 * k + 1] = (k1 >= 0 && k1 < C::M1) ? p.h1[k1] : 0.f;
    }
}

typedef float f3_v2f __attribute__((ext_vector_type(2)));

template <int M0_, int M1_>
struct Fwd3L1Cfg {
    static constexpr int M0 = M0_, M1 = M1_;
    static constexpr int H0 = M0 / 2, H1 = M1 / 2, H = cmax(H0, H1);
    static constexpr int MR = 2 * H + 1;                  // ring depth (slices i-H .. i+H)
    static constexpr int TJ = 16, TK = 64;                // outputs per slice
    static constexpr int PJ = TJ + 2 * H, PK = TK + 2 * H;
    static constexpr int S0S = TK + 8;                    // S0 row stride: 16-byte aligned windows
    static constexpr int NT = 256;                        // threads per workgroup (4 wavefronts)
    static constexpr int NPOS = PJ * PK;
    // The ring lives in registers as PAIRS of k-adjacent positions (one v_pk_fma_f32 filters both); every thread
    // holds NPP pairs, and the NLEFT pairs that do not divide evenly (2 for H = 3, 96 for H = 4) belong to the
    // first NLEFT threads, which keep their raw slices in a small LDS ring (XR) instead of a fourth register slot.
    static constexpr int NPAIR = NPOS / 2;
    static constexpr int NPP = NPAIR / NT;
    static constexpr int NLEFT = NPAIR - NPP * NT;
    static constexpr int XRF = 2 * NLEFT * MR;            // floats of the leftover ring
    static constexpr int S0F = 2 * PJ * S0S, S1F = 4 * PJ * TK;
#ifndef DT_F3L1_STAGE_PASSES
#define DT_F3L1_STAGE_PASSES 2
#endif
    // the records of a wavefront's 64 cells leave through its slab in SP passes of 64 / SP records: 2 passes need
    // 30 KB of slabs per workgroup (66 KB of LDS in all: two workgroups per CU), 4 passes 15 KB (51 KB: three) --
    // but the kernel needs ~250 VGPRs (register rings of 7 slices x 6 positions + 8 x 4 octant cells), i.e. two
    // waves per SIMD whatever the LDS says: at the 168-register cap of three it spills 84-194 VGPRs.  So 2.
    static constexpr int SP = DT_F3L1_STAGE_PASSES, SREC = 64 / SP;
    static constexpr int STAGE_W = SREC * REC_LDS;        // floats: SREC records per wavefront
    static constexpr int XR0 = S0F + S1F + (NT / 64) * STAGE_W;
    static constexpr int LDS_FLOATS = XR0 + ((XRF + 3) & ~3);
    static constexpr int NT2 = 2 * PJ * (TK / 4);         // axis-2 tasks (4 outputs each)
    static constexpr int WK = 4 + 2 * H;                  // axis-2 window (<= 12)
    static_assert(M0 % 2 == 1 && M1 % 2 == 1, "biort filters must have odd length");
    static_assert(H <= 4, "axis-2 window must fit three float4");
    static_assert(PK % 2 == 0 && NLEFT <= NT, "position pairs must not straddle rows");
    static constexpr int NCELL = (TJ / 2) * (TK / 2);
    static_assert(NCELL == NT && TK / 2 == 32, "one 2x2 cell per thread, one cell row per half wavefront");
};

// The lowpass and the highpass filter of axis 2 and axis 1 run as ONE chain of packed FMAs: window element d of
// 2H+1 meets the pair (h0, h1) of taps that multiply it, the shorter filter padded with zeros -- 7 v_pk_fma_f32
// per output pair instead of 5 + 7 scalar FMAs for near_sym_a.  (The kernel is bound by instruction issue and
// LDS latency at two waves per SIMD, not by HBM: 590 VALU instructions per thread and slice before this.)
template <class C>
DT_HD f3_v2f f3l1_tap_pair(const Fwd3L1Params &p, int d) {
    return reinterpret_cast<const f3_v2f *>(p.c01)[d];
}

// per-thread registers carried across the steps of the march
template <class C>
struct Fwd3L1State {
    f3_v2f ring[C::NPP][C::MR];
    f3_v2f nxa[C::NPP], nxb[C::NPP];      // the next two slices, loaded one slice pair ahead
    int goff[C::NPP][2];                  // j*n2 + k of the two positions (reflected), constant over slices
    int soff[C::NPP];                     // pj*S0S + pk of the pair
    f3_v2f exa, exb;                      // leftover pair (threads < NLEFT)
    int egoff[2], esoff;
    float ev[8][4];           // [a0*4 + a1*2 + a2][dj*2 + dk] of the even slice of the pair
};

template <class C>
DT_HD void f3l1_pair_offsets(const Fwd3L1Params &p, int qp, int j0, int k0, int (&goff)[2], int &soff) {
    const int q = 2 * qp;
    const int pj = q / C::PK, pk = q - pj * C::PK;
    const int j = reflect_i(j0 - C::H + pj, p.n1);
    goff[0] = j * p.n2 + reflect_i(k0 - C::H + pk, p.n2);
    goff[1] = j * p.n2 + reflect_i(k0 - C::H + pk + 1, p.n2);
    soff = pj * C::S0S + pk;
}

template <class C>
DT_HD void f3l1_init(const Fwd3L1Params &p, Fwd3L1State<C> &st, int tid, int j0, int k0) {
#pragma unroll
    for (int s = 0; s < C::NPP; ++s) f3l1_pair_offsets<C>(p, tid + C::NT * s, j0, k0, st.goff[s], st.soff[s]);
    if (tid < C::NLEFT) f3l1_pair_offsets<C>(p, tid + C::NT * C::NPP, j0, k0, st.egoff, st.esoff);
}

// slices i+H+1 and i+H+2 -> nxa / nxb.  Called BEFORE the stores of a slice pair are issued: s_waitcnt vmcnt
// counts loads and stores in order, so a load issued after the record stores cannot be waited for without
// draining them, while one issued before them only has to let `the stores issued since` stay in flight.
template <class C>
DT_HD void f3l1_prefetch(const Fwd3L1Params &p, Fwd3L1State<C> &st, int tid, int i) {
    const int64_t ss = (int64_t)p.n1 * p.n2;
    const float *sa = p.X + ss * reflect_i(i + C::H + 1, p.n0), *sb = p.X + ss * reflect_i(i + C::H + 2, p.n0);
#pragma unroll
    for (int s = 0; s < C::NPP; ++s) {
        st.nxa[s] = f3_v2f{sa[st.goff[s][0]], sa[st.goff[s][1]]};
        st.nxb[s] = f3_v2f{sb[st.goff[s][0]], sb[st.goff[s][1]]};
    }
    if (tid < C::NLEFT) {
        st.exa = f3_v2f{sa[st.egoff[0]], sa[st.egoff[1]]};
        st.exb = f3_v2f{sb[st.egoff[0]], sb[st.egoff[1]]};
    }
}

// ring <- slices i0-H .. i0+H (all loads in one batch), leftover pairs into XR[slice][thread]
template <class C>
DT_HD void f3l1_prologue(const Fwd3L1Params &p, Fwd3L1State<C> &st, float *XR, int tid, int i0) {
    const int64_t ss = (int64_t)p.n1 * p.n2;
#pragma unroll
    for (int t = 0; t < C::MR; ++t) {
        const float *sl = p.X + ss * reflect_i(i0 - C::H + t, p.n0);
#pragma unroll
        for (int s = 0; s < C::NPP; ++s) st.ring[s][t] = f3_v2f{sl[st.goff[s][0]], sl[st.goff[s][1]]};
    }
    if (tid < C::NLEFT) {
        f3_v2f e[C::MR];                                 // all loads first, then the LDS writes (one wait)
#pragma unroll
        for (int t = 0; t < C::MR; ++t) {
            const float *sl = p.X + ss * reflect_i(i0 - C::H + t, p.n0);
            e[t] = f3_v2f{sl[st.egoff[0]], sl[st.egoff[1]]};
        }
#pragma unroll
        for (int t = 0; t < C::MR; ++t) *reinterpret_cast<f3_v2f *>(XR + 2 * (t * C::NLEFT + tid)) = e[t];
    }
}

// both filters of axis 0 over a k-adjacent position pair: M0 + M1 packed FMAs, lo / hi pairs to S0
template <class C>
DT_HD void f3l1_axis0_pair(const Fwd3L1Params &p, const f3_v2f (&r)[C::MR], float *S0, int soff) {
    // the two chains side by side: a packed FMA that feeds the next instruction costs a wait state (s_nop)
    f3_v2f lo = {0.f, 0.f}, hi = {0.f, 0.f};
#pragma unroll
    for (int d = 0; d < C::MR; ++d) {
        if (d >= C::H - C::H0 && d <= C::H + C::H0) lo += p.c01[2 * d] * r[d];
        if (d >= C::H - C::H1 && d <= C::H + C::H1) hi += p.h1s[C::H + C::H1 - d] * r[d];    // the a0 = 1 half carries cube2c's 1/2 from here
    }
    *reinterpret_cast<f3_v2f *>(S0 + soff) = lo;
    *reinterpret_cast<f3_v2f *>(S0 + C::PJ * C::S0S + soff) = hi;
}

// filter the ring along axis 0 into S0.  ODD = 0: slices i-H .. i+H = the ring; ODD = 1 (the second slice of a pair):
// slices i-H+1 .. i+H+1 = ring[1 ..] and the first prefetched slice, so that the ring moves once per slice PAIR (by two
// slots: 42 register moves per thread and pair instead of 84).  rot: the slot of the leftover ring (XR) that holds
// the oldest slice.
template <class C, int ODD>
DT_HD void f3l1_axis0(const Fwd3L1Params &p, Fwd3L1State<C> &st, float *S0, const float *XR, int tid, int rot) {
#pragma unroll
    for (int s = 0; s < C::NPP; ++s) {
        if (ODD) {
            DT_PIN_HERE(st.nxa[s]);                       // the wait for the prefetched slice belongs here
            f3_v2f r[C::MR];
#pragma unroll
            for (int t = 0; t < C::MR - 1; ++t) r[t] = st.ring[s][t + 1];
            r[C::MR - 1] = st.nxa[s];
            f3l1_axis0_pair<C>(p, r, S0, st.soff[s]);
        } else {
            f3l1_axis0_pair<C>(p, st.ring[s], S0, st.soff[s]);
        }
    }
    if (tid < C::NLEFT) {
        f3_v2f r[C::MR];
#pragma unroll
        for (int t = 0; t < C::MR - ODD; ++t) {
            int slot = rot + t + ODD;
            if (slot >= C::MR) slot -= C::MR;
            r[t] = *reinterpret_cast<const f3_v2f *>(XR + 2 * (slot * C::NLEFT + tid));
        }
        if (ODD) r[C::MR - 1] = st.exa;
        f3l1_axis0_pair<C>(p, r, S0, st.esoff);
    }
}

// ring <- two slices further: the prefetched pair behind what is left
template <class C>
DT_HD void f3l1_rotate2(Fwd3L1State<C> &st, float *XR, int tid, int rot) {
#pragma unroll
    for (int s = 0; s < C::NPP; ++s) {
#pragma unroll
        for (int t = 0; t < C::MR - 2; ++t) st.ring[s][t] = st.ring[s][t + 2];
        st.ring[s][C::MR - 2] = st.nxa[s];
        st.ring[s][C::MR - 1] = st.nxb[s];
        DT_PIN_HERE(st.ring[s][C::MR - 1]);
    }
    if (tid < C::NLEFT) {
        const int r1 = rot + 1 >= C::MR ? rot + 1 - C::MR : rot + 1;
        *reinterpret_cast<f3_v2f *>(XR + 2 * (rot * C::NLEFT + tid)) = st.exa;
        *reinterpret_cast<f3_v2f *>(XR + 2 * (r1 * C::NLEFT + tid)) = st.exb;
    }
}

// S1[2*a0 + a2][pj][k] = (axis-2 filter a2) of S0[a0][pj][.]
template <class C>
DT_HD void f3l1_axis2(const Fwd3L1Params &p, const float *S0, float *S1, int tid) {
    constexpr int G = C::TK / 4;
#pragma unroll
    for (int r = 0; r < (C::NT2 + C::NT - 1) / C::NT; ++r) {
        int it = tid + C::NT * r;
        if (it >= C::NT2) break;
        int vol = it / (C::PJ * G), rem = it - vol * (C::PJ * G);
        int pj = rem / G, c = rem - pj * G;
        const f4 *src = reinterpret_cast<const f4 *>(S0 + (vol * C::PJ + pj) * C::S0S + 4 * c);
        float w[12];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            f4 v = src[q];
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
        // (a2 = 0, a2 = 1) of outputs k = 4c .. 4c + 3, the even ones (0, 2) and the odd ones (1, 3) each in ONE
        // four-register value: they leave as one 16-byte LDS write each, and register pairs that were accumulated
        // apart had to be moved together first (4 v_mov per write).  The four chains side by side: a packed FMA
        // that feeds its neighbour costs a wait state.
        dt_v4f ev = {0.f, 0.f, 0.f, 0.f}, od = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < C::MR; ++d) {
            const f3_v2f tp = f3l1_tap_pair<C>(p, d);
            ev.lo += tp * f3_v2f{w[d], w[d]};
            od.lo += tp * f3_v2f{w[1 + d], w[1 + d]};
            ev.hi += tp * f3_v2f{w[2 + d], w[2 + d]};
            od.hi += tp * f3_v2f{w[3 + d], w[3 + d]};
        }
        // S1[a0][pj][k parity][k / 2] pairs: the even and the odd outputs of the task are 16 bytes each, at a 16-byte
        // lane stride (conflict-free), and stay packed for the axis-1 chains
        float *d = S1 + (((vol * C::PJ + pj) * 2) * (C::TK / 2) + 2 * c) * 2;
        *reinterpret_cast<dt_v4f *>(d) = ev;
        *reinterpret_cast<dt_v4f *>(d + C::TK) = od;
    }
}

// cube2c of one octant (transform3d.py:532-579): cube corners A..H = (di, dj, dk) in
// {000, 010, 100, 110, 001, 011, 101, 111}; ev/od hold [dj*2+dk] of slice 2u / 2u+1
DT_HD void cube2c_record(float *rec, const float (&ev)[4], const float (&od)[4]) {
    const float A = ev[0], B = ev[2], Cc = od[0], D = od[2], E = ev[1], F = ev[3], G = od[1], Hh = od[3];
    const float h = 0.5f;
    f4 *o = reinterpret_cast<f4 *>(rec);
    o[0] = f4{(A - G - D - F) * h, (B - Hh + Cc + E) * h, (A - G + D + F) * h, (-B + Hh + Cc + E) * h};
    o[1] = f4{(A + G + D - F) * h, (B + Hh - Cc + E) * h, (A + G - D + F) * h, (-B - Hh - Cc + E) * h};
}

// the same for octants that already carry the factor 1/2 (f3l1_axis1)
DT_HD void cube2c_record_scaled(float *rec, const float (&ev)[4], const float (&od)[4]) {
    const float A = ev[0], B = ev[2], Cc = od[0], D = od[2], E = ev[1], F = ev[3], G = od[1], Hh = od[3];
    f4 *o = reinterpret_cast<f4 *>(rec);
    o[0] = f4{A - G - D - F, B - Hh + Cc + E, A - G + D + F, -B + Hh + Cc + E};
    o[1] = f4{A + G + D - F, B + Hh - Cc + E, A + G - D + F, -B - Hh - Cc + E};
}

// record slot of octant idx = a0*4 + a1*2 + a2 (reference order 010 100 110 001 011 101 111)
DT_HD int octant_slot(int idx) { return (idx & 1) ? 3 + (idx >> 1) : (idx >> 1) - 1; }

// axis-1 filters for the thread's 2x2 (j, k) cell -> out[a0*4 + a1*2 + a2][dj*2 + dk], LLL store.
// FULL: the tile lies inside the volume (stores without conditions: the compiler can then count them when it
// waits for the prefetched slices, see f3l1_prefetch)
template <class C, bool FULL>
DT_HD void f3l1_axis1(const Fwd3L1Params &p, float (&out)[8][4], const float *S1, int tid, int i, int j0,
                      int k0) {
    const int cj = tid / (C::TK / 2), ck = tid - cj * (C::TK / 2);
    const int j = j0 + 2 * cj, k = k0 + 2 * ck;
    // per a0: the rows of the cell's two columns as (a2 = 0, a2 = 1) pairs; the lowpass chain of a position gives
    // octants (a1 = 0; a2 = 0, 1), the highpass chain (a1 = 1; a2 = 0, 1): M0 + M1 packed FMAs for four values.
    // cube2c's 1/2: see Fwd3L1Params::h1s.
#pragma unroll
    for (int a0 = 0; a0 < 2; ++a0) {
        f3_v2f u[2 * C::H + 2][2];
#pragma unroll
        for (int r = 0; r < 2 * C::H + 2; ++r) {
            const float *q = S1 + ((((a0 * C::PJ + 2 * cj + r) * 2) * (C::TK / 2)) + ck) * 2;
            u[r][0] = *reinterpret_cast<const f3_v2f *>(q);
            u[r][1] = *reinterpret_cast<const f3_v2f *>(q + C::TK);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            // the four chains of a row (lo / hi x two columns) side by side: no packed FMA feeds its neighbour
            f3_v2f lo[2] = {{0.f, 0.f}, {0.f, 0.f}}, hi[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int d = 0; d < 2 * C::H + 1; ++d)          // window offset: row e + d
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (d >= C::H - C::H0 && d <= C::H + C::H0) lo[c] += p.c01[2 * d] * u[e + d][c];
                    if (d >= C::H - C::H1 && d <= C::H + C::H1) {
                        if (a0 == 0) hi[c] += p.h1s[C::H + C::H1 - d] * u[e + d][c];
                        else hi[c] += p.c01[2 * d + 1] * u[e + d][c];
                    }
                }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                out[a0 * 4 + 0][e * 2 + c] = lo[c].x; out[a0 * 4 + 1][e * 2 + c] = lo[c].y;
                out[a0 * 4 + 2][e * 2 + c] = hi[c].x; out[a0 * 4 + 3][e * 2 + c] = hi[c].y;
            }
        }
    }
    if (FULL || (j < p.n1 && k < p.n2)) {
        float *L = p.LLL + ((int64_t)i * p.n1 + j) * p.n2 + k;
        *reinterpret_cast<f2 *>(L) = f2{out[0][0], out[0][1]};
        *reinterpret_cast<f2 *>(L + p.n2) = f2{out[0][2], out[0][3]};
    }
}

// ---- staged record stores (odd slices) -------------------------------------------------
// A lane-owned 224-byte record written as fourteen 16-byte stores at a 224-byte lane
// stride turns every store instruction into 64 separate 16-byte write requests, and the
// L2 request rate -- not bytes -- then bounds the kernel.  Each half wavefront owns one
// row of 32 cells = 7168 contiguous bytes of Yh, so the halves take turns bouncing their
// records through a wave-private LDS slab and the whole wavefront writes the row out as
// seven consecutive 1 KiB runs.  Two functions because the host emulator runs them as two
// passes; on the device they are called back to back (LDS operations of one wavefront
// execute in order, no barrier needed).
template <class C>
DT_HD void f3l1_pack_stage(const float (&ev)[8][4], const float (&od)[8][4], float *stage, int tid, int pass) {
    const int lane = tid & 63, wave = tid >> 6;
    DT_WAVE_LDS_SYNC();                                  // the previous pass has read its records out of the slab
    if (lane / C::SREC != pass) return;
    float *rec = stage + wave * C::STAGE_W + (lane % C::SREC) * REC_LDS;
    // record slots in the reference's concatenation order (transform3d.py:278-289):
    // (a0,a1,a2) = 010, 100, 110, 001, 011, 101, 111
    cube2c_record_scaled(rec + 0, ev[2], od[2]);
    cube2c_record_scaled(rec + 8, ev[4], od[4]);
    cube2c_record_scaled(rec + 16, ev[6], od[6]);
    cube2c_record(rec + 24, ev[1], od[1]);               // (0, 0, 1): not scaled yet
    cube2c_record_scaled(rec + 32, ev[3], od[3]);
    cube2c_record_scaled(rec + 40, ev[5], od[5]);
    cube2c_record_scaled(rec + 48, ev[7], od[7]);
}

// pass: which SREC consecutive cells of the wavefront's 64 (two rows of 32) are in the slab
template <class C, bool FULL>
DT_HD void f3l1_pack_flush(const Fwd3L1Params &p, const float *stage, int tid, int pass, int i, int j0,
                           int k0) {
    const int lane = tid & 63, wave = tid >> 6;
    const int cell0 = pass * C::SREC;                    // first cell of the pass within the wavefront
    const int half = cell0 >> 5, c0 = cell0 & 31;        // its cell row (half wavefront) and first cell in the row
    const int j = j0 + 2 * (2 * wave + half);
    DT_WAVE_LDS_SYNC();                                  // the records other lanes staged
    if (!FULL && j >= p.n1) return;
    const f4 *slab = reinterpret_cast<const f4 *>(stage + wave * C::STAGE_W);
    f4 *row = reinterpret_cast<f4 *>(p.Yh + (((int64_t)(i >> 1) * (p.n1 / 2) + (j >> 1)) * (p.n2 / 2) + (k0 >> 1) + c0) * 56);
    int ncell = C::SREC;
    if (!FULL) {
        ncell = (p.n2 - k0) / 2 - c0;
        if (ncell > C::SREC) ncell = C::SREC;
    }
#pragma unroll
    for (int it = 0; it < (C::SREC * 14 + 63) / 64; ++it) {
        int piece = it * 64 + lane;                      // 16-byte piece of the pass's records
        if (FULL ? (C::SREC * 14 % 64 == 0 || piece < C::SREC * 14) : piece < ncell * 14)
            DT_STREAM_STORE_F4(row + piece, slab[slab_f4(piece)]);
    }
}

// ======================================================================================
// Level >= 2 (transform3d.py:317-383, even-length q-shift filters, decimation by 2).
// ======================================================================================
// A 2M-sample window per axis makes a single 3-D tile far larger than LDS, so a level is
// two launches:
//   pass A  every slice of the volume goes through the 2-D level-2 tile program
//           (fwd2d_cols of fused2d_tiles_v2.hpp: axis 1 down the rows, then axis 2 along
//           the columns), but instead of q2c records it stores the four (a1, a2) planes:
//           P[2*a1 + a2][n0][O1][O2];
//   pass B  one thread per 2x2x2 output cell filters the four plane-volumes down axis 0
//           (window of 2M slices, lanes along axis 2) and packs all eight octants: LLL
//           and the whole 224-byte highpass record.
// Edge padding (ext_mode 4: one replicated plane per side, 8: two; :322-335) is index
// arithmetic in both passes.

template <class C>
DT_HD void fwd2p_rows(const dt2d::Fwd2Params &p, const float *sLo, const float *sHi, float *planes,
                      int64_t pstride, int tid, int base, int b, int r0, int c0) {
    using dt2d::dfilt_pair;
    const int OR = p.LR / 2, OC = p.LC / 2;
    const int task = base + tid;
    const int il = task / C::TJ, jl = task - il * C::TJ;
    const int R = r0 + 2 * il, Cc = c0 + 2 * jl;
    if (task >= C::TI * C::TJ || R >= OR || Cc >= OC) return;
    float *o = planes + ((int64_t)b * OR + R) * OC + Cc;
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
        float *q = o + (int64_t)er * OC;
        dt2d::dfilt_pair2<C::M>(wl, p.lh_a, p.lh_b, A, Bv, Ah, Bh);      // a1 = 0; a2 = 0, 1
        *reinterpret_cast<f2 *>(q) = p.lo_a_first ? f2{A, Bv} : f2{Bv, A};
        *reinterpret_cast<f2 *>(q + pstride) = p.hi_a_first ? f2{Ah, Bh} : f2{Bh, Ah};
        dt2d::dfilt_pair2<C::M>(wh, p.lh_a, p.lh_b, A, Bv, Ah, Bh);      // a1 = 1; a2 = 0, 1
        *reinterpret_cast<f2 *>(q + 2 * pstride) = p.lo_a_first ? f2{A, Bv} : f2{Bv, A};
        *reinterpret_cast<f2 *>(q + 3 * pstride) = p.hi_a_first ? f2{Ah, Bh} : f2{Bh, Ah};
    }
}

struct Fwd3L2Params {
    const float *P;       // [4][n0][O1][O2]
    int64_t pstride;      // n0*O1*O2
    float *LLL;           // [O0][O1][O2]
    float *Yh;            // [O0/2][O1/2][O2/2][56 floats]
    int n0, pad0, L0;     // real slices, replicated planes per side, L0 = n0 + 2 pad0 (% 4 == 0)
    int O0, O1, O2;       // octant extents (L/2, all even)
    int lo_a_first, hi_a_first;
    int id0, id1;         // pass B: the cells [id0, id1) of this launch (0, 0: all of them)
    float l_a[DT_MAXT], l_b[DT_MAXT], h_a[DT_MAXT], h_b[DT_MAXT];
    float lh_a[2 * DT_MAXT] __attribute__((aligned(8))), lh_b[2 * DT_MAXT] __attribute__((aligned(8)));   // dt_pack_lh()
};

// Planes v0 .. v0 + nv - 1 (of the four (a1, a2) planes) of output cell `id`: their octants go to the cell's
// 56-float record in a wave-private LDS slab, and f3l2_axis0_flush writes the wavefront's consecutive records
// (cell ids are record indices) as 1 KiB runs.  Two lanes share a cell (two planes each: 32 cells and a 7.7 KB
// slab per wavefront, five workgroups per CU, 80 dependent loads per thread) or one lane does all four (64
// cells, 15 KB: two workgroups per CU, 160 loads per thread).
// INNER: the 2M slices of the window are slices r0 .. r0 + 2M - 1 of the planes as they are (no reflection, no
// replicated planes): one multiply-add per address instead of a table of 2M reflected offsets
template <int M, bool INNER>
DT_HD void f3l2_axis0_planes(const Fwd3L2Params &p, int c0, int base, int r0, float *rec, int v0, int nv) {
    const int ss = p.O1 * p.O2;
    int soff[INNER ? 1 : 2 * M];
    if (!INNER) {
#pragma unroll
        for (int j = 0; j < 2 * M; ++j) {
            int r = r0 + p.pad0 + j;                         // logical slice (A.2)
            while ((unsigned)r >= (unsigned)p.L0) r = r < 0 ? -1 - r : 2 * p.L0 - 1 - r;
            r -= p.pad0;
            r = r < 0 ? 0 : (r > p.n0 - 1 ? p.n0 - 1 : r);
            soff[j] = r * ss;
        }
    }
#pragma unroll 1
    for (int v = v0; v < v0 + nv; ++v) {
        const float *Pv = p.P + v * p.pstride + base;
        const float *P0 = Pv + (int64_t)r0 * ss;
        float w[4][2 * M];                               // [dj*2 + dk][slice]
#pragma unroll
        for (int j = 0; j < 2 * M; ++j) {
            const float *src = INNER ? P0 + j * ss : Pv + soff[j];
            f2 a = *reinterpret_cast<const f2 *>(src);
            f2 b = *reinterpret_cast<const f2 *>(src + p.O2);
            w[0][j] = a.x; w[1][j] = a.y; w[2][j] = b.x; w[3][j] = b.y;
        }
        float lev[4], lod[4], hev[4], hod[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float A, Bv, Ah, Bh;
            dt2d::dfilt_pair2<M>(w[q], p.lh_a, p.lh_b, A, Bv, Ah, Bh);
            lev[q] = p.lo_a_first ? A : Bv; lod[q] = p.lo_a_first ? Bv : A;
            hev[q] = p.hi_a_first ? Ah : Bh; hod[q] = p.hi_a_first ? Bh : Ah;
        }
        if (v == 0) {
            float *L = p.LLL + (int64_t)(2 * c0) * ss + base;
            *reinterpret_cast<f2 *>(L) = f2{lev[0], lev[1]};
            *reinterpret_cast<f2 *>(L + p.O2) = f2{lev[2], lev[3]};
            *reinterpret_cast<f2 *>(L + ss) = f2{lod[0], lod[1]};
            *reinterpret_cast<f2 *>(L + ss + p.O2) = f2{lod[2], lod[3]};
        } else {
            cube2c_record(rec + 8 * octant_slot(v), lev, lod);
        }
        cube2c_record(rec + 8 * octant_slot(4 + v), hev, hod);
    }
}

// The INNER case with buffer addressing and the planes software-pipelined: the four plane volumes' base sits in a
// scalar descriptor, a lane holds ONE byte offset per row of its cell, the slice offsets j * ss are scalar offsets
// (no vector instruction per address: they were two thirds of this kernel's 1576 vector instructions per
// wavefront), and the window of plane v + 1 is requested before plane v is filtered -- with two wavefronts per
// SIMD (the record slabs fill the LDS) nothing else hides the latency of 40 dependent loads per plane, four times
// per cell.  Needs the four plane volumes inside one 4 GiB descriptor.
template <int M, int NV>
DT_HD void f3l2_axis0_planes_buf(const Fwd3L2Params &p, int c0, int base, int r0, float *rec, int v0) {
    const int ss = p.O1 * p.O2;
    const dt2d::DtBuf pb = dt2d::dt_buf(p.P);
    const unsigned ps4 = 4u * (unsigned)p.pstride, row = 4u * (unsigned)p.O2;
    const unsigned vo = 4u * (unsigned)(base + r0 * ss) + (unsigned)v0 * ps4;
    float w[2][4][2 * M];                                // two windows: [dj*2 + dk][slice]
#pragma unroll
    for (int j = 0; j < 2 * M; ++j) {
        const f2 a = dt2d::dt_buf_ld2(pb, vo, 4u * (unsigned)(j * ss)), b = dt2d::dt_buf_ld2(pb, vo + row, 4u * (unsigned)(j * ss));
        w[0][0][j] = a.x; w[0][1][j] = a.y; w[0][2][j] = b.x; w[0][3][j] = b.y;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (i + 1 < NV) {
            const unsigned vn = vo + (unsigned)(i + 1) * ps4;
#pragma unroll
            for (int j = 0; j < 2 * M; ++j) {
                const f2 a = dt2d::dt_buf_ld2(pb, vn, 4u * (unsigned)(j * ss)), b = dt2d::dt_buf_ld2(pb, vn + row, 4u * (unsigned)(j * ss));
                w[(i + 1) & 1][0][j] = a.x; w[(i + 1) & 1][1][j] = a.y; w[(i + 1) & 1][2][j] = b.x; w[(i + 1) & 1][3][j] = b.y;
            }
        }
        const int v = v0 + i;
        float lev[4], lod[4], hev[4], hod[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float A, Bv, Ah, Bh;
            dt2d::dfilt_pair2<M>(w[i & 1][q], p.lh_a, p.lh_b, A, Bv, Ah, Bh);
            lev[q] = p.lo_a_first ? A : Bv; lod[q] = p.lo_a_first ? Bv : A;
            hev[q] = p.hi_a_first ? Ah : Bh; hod[q] = p.hi_a_first ? Bh : Ah;
        }
        if (v == 0) {
            float *L = p.LLL + (int64_t)(2 * c0) * ss + base;
            *reinterpret_cast<f2 *>(L) = f2{lev[0], lev[1]};
            *reinterpret_cast<f2 *>(L + p.O2) = f2{lev[2], lev[3]};
            *reinterpret_cast<f2 *>(L + ss) = f2{lod[0], lod[1]};
            *reinterpret_cast<f2 *>(L + ss + p.O2) = f2{lod[2], lod[3]};
        } else {
            cube2c_record(rec + 8 * octant_slot(v), lev, lod);
        }
        cube2c_record(rec + 8 * octant_slot(4 + v), hev, hod);
    }
}

template <int M, int NV = 4>
DT_HD void f3l2_axis0_stage(const Fwd3L2Params &p, int id, float *rec, int v0 = 0, int nv = NV) {
    const int e2 = p.O2 / 2, e1 = p.O1 / 2, e0 = p.O0 / 2;
    if (id >= (p.id1 ? p.id1 : e0 * e1 * e2)) return;
    const int c2 = id % e2, t = id / e2, c1 = t % e1, c0 = t / e1;
    const int base = (2 * c1) * p.O2 + 2 * c2;
    const int r0 = 4 * c0 - M + 2 - p.pad0;              // first slice of the window in the unpadded planes
    const bool inner = r0 >= 0 && r0 + 2 * M <= p.n0;
    if (inner && nv == NV && 16 * p.pstride < ((int64_t)1 << 32)) f3l2_axis0_planes_buf<M, NV>(p, c0, base, r0, rec, v0);
    else if (inner) f3l2_axis0_planes<M, true>(p, c0, base, r0, rec, v0, nv);
    else f3l2_axis0_planes<M, false>(p, c0, base, r0, rec, v0, nv);
}

// ---- pass B, two cells along axis 0 per lane pair ---------------------------------------------------------------
// What pass B waits for is the CU's vector-memory path: every plane slice is wanted by five cells along axis 0, so
// 335 MB go through L1 for 67 MB of planes (profiles/r03/c4_passB.txt).  Cells (2u, 2u + 1) share 2M - 4 of their 2M
// slices: a lane that filters BOTH from one window of 2M + 4 slices loads 24 instead of 40 slices per pair (M = 10).
// A wavefront = 32 cell columns x 2 cells; lane l and lane l + 32 share a column and take planes (0, 1) / (2, 3).  Only
// cells whose windows lie inside the volume (no reflection, no pad planes); the boundary layers go to the one-cell
// kernel.  Buffer addressing as in f3l2_axis0_planes_buf.
template <int M>
DT_HD void f3l2_axis0_pair_stage(const Fwd3L2Params &p, int u, int col, int half, float *slab, int cl) {
    constexpr int WP = 2 * M + 4;
    const int ss = p.O1 * p.O2, e2 = p.O2 / 2;
    const int c1 = col / e2, c2 = col - c1 * e2;
    const int base = (2 * c1) * p.O2 + 2 * c2;
    const int c0 = 2 * u, r0 = 4 * c0 - M + 2 - p.pad0;
    const dt2d::DtBuf pb = dt2d::dt_buf(p.P);
    const unsigned ps4 = 4u * (unsigned)p.pstride, row = 4u * (unsigned)p.O2;
    const unsigned vo = 4u * (unsigned)(base + r0 * ss) + (unsigned)(2 * half) * ps4;
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {
        const int v = 2 * half + i;
        const unsigned vv = vo + (unsigned)i * ps4;
        float w[4][WP];
#pragma unroll
        for (int j = 0; j < WP; ++j) {
            const f2 a = dt2d::dt_buf_ld2(pb, vv, 4u * (unsigned)(j * ss)), b = dt2d::dt_buf_ld2(pb, vv + row, 4u * (unsigned)(j * ss));
            w[0][j] = a.x; w[1][j] = a.y; w[2][j] = b.x; w[3][j] = b.y;
        }
#pragma unroll
        for (int cell = 0; cell < 2; ++cell) {
            float lev[4], lod[4], hev[4], hod[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float A, Bv, Ah, Bh;
                dt2d::dfilt_pair2<M>(w[q] + 4 * cell, p.lh_a, p.lh_b, A, Bv, Ah, Bh);
                lev[q] = p.lo_a_first ? A : Bv; lod[q] = p.lo_a_first ? Bv : A;
                hev[q] = p.hi_a_first ? Ah : Bh; hod[q] = p.hi_a_first ? Bh : Ah;
            }
            float *rec = slab + (cell * 32 + cl) * REC_LDS;
            if (v == 0) {
                float *L = p.LLL + (int64_t)(2 * (c0 + cell)) * ss + base;
                *reinterpret_cast<f2 *>(L) = f2{lev[0], lev[1]};
                *reinterpret_cast<f2 *>(L + p.O2) = f2{lev[2], lev[3]};
                *reinterpret_cast<f2 *>(L + ss) = f2{lod[0], lod[1]};
                *reinterpret_cast<f2 *>(L + ss + p.O2) = f2{lod[2], lod[3]};
            } else {
                cube2c_record(rec + 8 * octant_slot(v), lev, lod);
            }
            cube2c_record(rec + 8 * octant_slot(4 + v), hev, hod);
        }
    }
}

// the wavefront's 2 x 32 records: two runs of 32 consecutive records (cells col0 .. col0 + 31 of layers 2u and 2u + 1)
DT_HD void f3l2_axis0_pair_flush(const Fwd3L2Params &p, int u, int col0, int lane, const float *slab) {
    const int E = (p.O1 / 2) * (p.O2 / 2);
    const f4 *src = reinterpret_cast<const f4 *>(slab);
#pragma unroll
    for (int cell = 0; cell < 2; ++cell) {
        f4 *dst = reinterpret_cast<f4 *>(p.Yh + ((int64_t)(2 * u + cell) * E + col0) * 56);
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int piece = it * 64 + lane;                   // 32 records = 448 pieces
            DT_STREAM_STORE_F4(dst + piece, src[slab_f4(cell * 32 * 14 + piece)]);
        }
    }
}

// first: id of the wavefront's first cell, CPW: cells per wavefront (64 or 32)
template <int CPW = 64>
DT_HD void f3l2_axis0_flush(const Fwd3L2Params &p, int first, int lane, const float *slab) {
    const int ncell = p.id1 ? p.id1 : (p.O0 / 2) * (p.O1 / 2) * (p.O2 / 2);
    const int n = ncell - first < CPW ? ncell - first : CPW;
    f4 *dst = reinterpret_cast<f4 *>(p.Yh + (int64_t)first * 56);
    const f4 *src = reinterpret_cast<const f4 *>(slab);
#pragma unroll
    for (int it = 0; it < 14 * CPW / 64; ++it) {
        int piece = it * 64 + lane;
        if (piece < n * 14) DT_STREAM_STORE_F4(dst + piece, src[slab_f4(piece)]);
    }
}

}  // namespace dt3d
