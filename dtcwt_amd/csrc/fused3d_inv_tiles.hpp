// Fused per-level tile programs of the float32 3-D DT-CWT inverse transform.
//
// One level of dtcwt/numpy/transform3d.py:385-440 (`_level1_ifm`) / :460-526
// (`_level2_ifm`) -- c2cube of the seven highpass octants (:581-619) and the three
// axis-wise merges filter(lo-branch) + filter(hi-branch) -- is two launches:
//
//   pass A  unpack + axis-0 merge.  A workgroup owns 2 x 32 cells of the (axis 1, axis 2)
//           plane and MARCHES along axis 0, one highpass record (= two slices) per step.
//           Its four wavefronts are the four (a1, a2) combinations; each thread keeps a
//           register ring of the last few slices of its two octants (a0 = 0 / 1) at the
//           2x2 positions of its cell, filters them down axis 0 and writes the plane-volume
//           P[2*a1 + a2].  Records travel global -> registers -> LDS slab (coalesced
//           16-byte pieces, double buffered, prefetched two steps ahead) -> octant pieces;
//           the (0, 0, 0) octant is read straight from the lowpass volume.
//   pass B  every output slice goes through the column / row passes of the 2-D inverse tile
//           programs (fused2d_tiles_v2.hpp) with the four planes in place of the lowpass
//           and the three c2q quad planes.
//
// The merges are linear and separable, so doing axis 0 first (the reference merges axes
// 1, 0, 2) changes float32 rounding only.  Output cropping (ext_mode 4 / 8, :505-524) is
// index arithmetic: axis 0 in pass A, axes 1 and 2 in pass B.
#pragma once
#include "fused3d_tiles.hpp"

namespace dt3d {

struct Inv3AParams {
    const float *LLL;     // [n0][n1][n2]
    const float *Yh;      // [n0/2][n1/2][n2/2][56 floats]
    float *P;             // [4][S][n1][n2]
    int64_t pstride;      // S*n1*n2
    int n0, n1, n2;       // all even
    int S;                // slices written: n0 (level 1) or 2 n0 - 2 crop0 (level >= 2)
    int crop0;
    int chunk;            // records (slice pairs) marched by one workgroup
    int tilesJ, tilesK, chunks;
    // slab launches (level 1 of large volumes: the planes of one slab stay in the Infinity Cache for pass B and the
    // buffer is reused by the next slab): this launch marches chunks ch0 .. ch0 + gridDim / tiles - 1 and P holds
    // output slices so0 .. so0 + slabS - 1 only.  slabS == 0: one launch, P holds all S slices.
    int ch0, so0, slabS;
    int lo_pos, hi_pos;   // level >= 2: sum(ha*hb) > 0 of the g0 / g1 pair (lowlevel.py:205,232)
    // level 1: l_a = g0o, h_a = g1o.  level >= 2: colifilt(., g0b, g0a) + colifilt(., g1b, g1a):
    // l_a = g0b, l_b = g0a, h_a = g1b, h_b = g1a
    float l_a[DT_MAXT], l_b[DT_MAXT], h_a[DT_MAXT], h_b[DT_MAXT];
};

constexpr int I3_CJ = 2, I3_CK = 32;                 // cells per workgroup tile
constexpr int I3_SLAB = I3_CJ * I3_CK * REC_LDS;     // floats of one staged record tile
constexpr int I3_NPIECE = I3_CJ * I3_CK * 14;        // 16-byte pieces (896)
constexpr int I3_RP = (I3_NPIECE + DT_NT - 1) / DT_NT;

// level 1: odd-length biort filters g0o (M0 taps) on the a0 = 0 branch, g1o (M1) on a0 = 1
template <int M0_, int M1_>
struct Inv3L1 {
    static constexpr int M0 = M0_, M1 = M1_;
    static constexpr int H0 = M0 / 2, H1 = M1 / 2, H = cmax(H0, H1);
    static constexpr int HP = (H + 1) / 2;           // pairs before / after the current one
    static constexpr int NP = 2 * HP + 1, NS = 2 * NP;   // ring: pairs, slices
    static constexpr int NOUT = 2;                   // output slices per step
    static_assert(M0 % 2 == 1 && M1 % 2 == 1, "biort filters must have odd length");
    // ring slot of slice s at step c is s - 2(c - HP)
    static DT_HD void compute(const Inv3AParams &p, const float (&ra)[4][NS], const float (&rb)[4][NS],
                              float (&out)[NOUT][4]) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < M0; ++t) s += p.l_a[t] * ra[q][e + H0 - t + 2 * HP];
#pragma unroll
                for (int t = 0; t < M1; ++t) s += p.h_a[t] * rb[q][e + H1 - t + 2 * HP];
                out[e][q] = s;
            }
    }
};

// level >= 2: even-length q-shift filters, interpolation by 2 (A.3): the ring IS the window
// of ifilt4 (sample 2c + ORG + j' sits in slot j')
template <int M_>
struct Inv3L2 {
    static constexpr int M = M_, M2 = M / 2;
    static constexpr bool ODD = (M2 % 2) == 1;
    static constexpr int WN = ODD ? M : M + 2;
    static constexpr int HP = ODD ? (M2 - 1) / 2 : M2 / 2;
    static constexpr int NP = 2 * HP + 1, NS = 2 * NP;
    static constexpr int NOUT = 4;
    static_assert(M % 2 == 0 && NS == WN, "ring = colifilt window");
    static DT_HD void compute(const Inv3AParams &p, const float (&ra)[4][NS], const float (&rb)[4][NS],
                              float (&out)[NOUT][4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float a[4], t[4];
            dt2d::ifilt4<Inv3L2>(ra[q], p.l_a, p.l_b, p.lo_pos, a);
            dt2d::ifilt4<Inv3L2>(rb[q], p.h_a, p.h_b, p.hi_pos, t);
#pragma unroll
            for (int e = 0; e < 4; ++e) out[e][q] = a[e] + t[e];
        }
    }
};

template <class F>
struct Inv3AState {
    float ra[4][F::NS], rb[4][F::NS];   // [dj*2 + dk][ring slot] of the a0 = 0 / 1 octant
    float R[I3_RP][4];                  // record pieces in flight
    float L[2][4];                      // lowpass pair in flight (wavefront 0 only)
};

// virtual record q (slices 2q, 2q+1 with symmetric extension): real record, and whether the
// two slices arrive swapped
DT_HD void i3_vrec(int q, int n0, int &rq, int &swap) {
    int r = reflect_i(2 * q, n0);
    rq = r >> 1; swap = r & 1;
}

// issue the loads of virtual record q (all threads)
template <class F>
DT_HD void i3a_issue_rec(const Inv3AParams &p, Inv3AState<F> &st, int tid, int cj0, int ck0, int q) {
    const int e1 = p.n1 / 2, e2 = p.n2 / 2;
    int rq, sw;
    i3_vrec(q, p.n0, rq, sw);
    const int ncell = e2 - ck0 < I3_CK ? e2 - ck0 : I3_CK;
#pragma unroll
    for (int s = 0; s < I3_RP; ++s) {
        int piece = tid + DT_NT * s;
        int row = piece / (I3_CK * 14), within = piece - row * (I3_CK * 14);
        bool ok = piece < I3_NPIECE && cj0 + row < e1 && within < ncell * 14;
        if (!ok) { row = 0; within = 0; }           // harmless in-range address, value unused
        const f4 *src = reinterpret_cast<const f4 *>(p.Yh + (((int64_t)rq * e1 + (cj0 + row)) * e2 + ck0) * 56);
        const dt_v4f v = DT_STREAM_LOAD_F4(src + within);
        st.R[s][0] = v.x; st.R[s][1] = v.y; st.R[s][2] = v.z; st.R[s][3] = v.w;
    }
}

// issue the loads of lowpass slices 2q, 2q+1 (wavefront 0: the (0,0,0) octant)
template <class F>
DT_HD void i3a_issue_low(const Inv3AParams &p, Inv3AState<F> &st, int tid, int cj0, int ck0, int q) {
    if ((tid >> 6) != 0) return;
    const int lane = tid & 63;
    int j = 2 * (cj0 + (lane >> 5)), k = 2 * (ck0 + (lane & 31));
    if (j >= p.n1) j = 0;
    if (k >= p.n2) k = 0;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const float *sl = p.LLL + ((int64_t)reflect_i(2 * q + d, p.n0) * p.n1 + j) * p.n2 + k;
        f2 a = *reinterpret_cast<const f2 *>(sl), b = *reinterpret_cast<const f2 *>(sl + p.n2);
        st.L[d][0] = a.x; st.L[d][1] = a.y; st.L[d][2] = b.x; st.L[d][3] = b.y;
    }
}

template <class F>
DT_HD void i3a_slab_write(Inv3AState<F> &st, float *slab, int tid) {
    f4 *dst = reinterpret_cast<f4 *>(slab);
#pragma unroll
    for (int s = 0; s < I3_RP; ++s) {
        int piece = tid + DT_NT * s;
        DT_PIN_HERE(st.R[s][0]);
        if (piece < I3_NPIECE) dst[slab_f4(piece)] = f4{st.R[s][0], st.R[s][1], st.R[s][2], st.R[s][3]};
    }
}

// c2cube of one octant piece (transform3d.py:581-619): ev/od = [dj*2 + dk] of slice 2u / 2u+1
DT_HD void c2cube_piece(const float *piece, float (&ev)[4], float (&od)[4]) {
    const f4 a = reinterpret_cast<const f4 *>(piece)[0], b = reinterpret_cast<const f4 *>(piece)[1];
    const float pr = a.x, pi = a.y, qr = a.z, qi = a.w, rr = b.x, ri = b.y, sr = b.z, si = b.w;
    const float h = 0.5f;
    ev[0] = (pr + qr + rr + sr) * h;      // A (0,0,0)
    ev[1] = (pi + qi + ri + si) * h;      // E (0,0,1)
    ev[2] = (pi - qi + ri - si) * h;      // B (0,1,0)
    ev[3] = (-pr + qr - rr + sr) * h;     // F (0,1,1)
    od[0] = (pi + qi - ri - si) * h;      // C (1,0,0)
    od[1] = (-pr - qr + rr + sr) * h;     // G (1,0,1)
    od[2] = (-pr + qr + rr - sr) * h;     // D (1,1,0)
    od[3] = (-pi + qi + ri - si) * h;     // H (1,1,1)
}

// rotate the rings by one pair and push virtual record qv (from the slab; octant (0,0,0)
// from the lowpass pair in flight)
template <class F>
DT_HD void i3a_push(const Inv3AParams &p, Inv3AState<F> &st, const float *slab, int tid, int qv) {
    const int v = tid >> 6, lane = tid & 63;
    int rq, sw;
    i3_vrec(qv, p.n0, rq, sw);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < F::NS - 2; ++t) { st.ra[q][t] = st.ra[q][t + 2]; st.rb[q][t] = st.rb[q][t + 2]; }
    const float *rec = slab + lane * REC_LDS;
    float ev[4], od[4];
    c2cube_piece(rec + 8 * octant_slot(4 + v), ev, od);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        st.rb[q][F::NS - 2] = sw ? od[q] : ev[q];
        st.rb[q][F::NS - 1] = sw ? ev[q] : od[q];
    }
    if (v == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            DT_PIN_HERE(st.L[0][q]);
            DT_PIN_HERE(st.L[1][q]);
            st.ra[q][F::NS - 2] = st.L[0][q];       // the loads already followed the reflection
            st.ra[q][F::NS - 1] = st.L[1][q];
        }
    } else {
        c2cube_piece(rec + 8 * octant_slot(v), ev, od);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            st.ra[q][F::NS - 2] = sw ? od[q] : ev[q];
            st.ra[q][F::NS - 1] = sw ? ev[q] : od[q];
        }
    }
}

template <class F>
DT_HD void i3a_store(const Inv3AParams &p, const float (&out)[F::NOUT][4], int tid, int cj0, int ck0, int c) {
    const int v = tid >> 6, lane = tid & 63;
    const int j = 2 * (cj0 + (lane >> 5)), k = 2 * (ck0 + (lane & 31));
    if (j >= p.n1 || k >= p.n2) return;
#pragma unroll
    for (int e = 0; e < F::NOUT; ++e) {
        int so = F::NOUT * c + e - p.crop0;
        if (so < 0 || so >= p.S) continue;
        if (p.slabS) {
            so -= p.so0;
            if (so < 0 || so >= p.slabS) continue;
        }
        float *o = p.P + v * p.pstride + ((int64_t)so * p.n1 + j) * p.n2 + k;
        *reinterpret_cast<f2 *>(o) = f2{out[e][0], out[e][1]};
        *reinterpret_cast<f2 *>(o + p.n2) = f2{out[e][2], out[e][3]};
    }
}

}  // namespace dt3d
