// Fused per-level tile programs of the float32 3-D DT-CWT inverse transform.
//
// One level of dtcwt/numpy/transform3d.py:385-440 (`_level1_ifm`) / :460-526
// (`_level2_ifm`) -- c2cube of the seven highpass octants (:581-619) and the three
// axis-wise merges filter(lo-branch) + filter(hi-branch) -- is two launches.
//
// Levels >= 2:
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
// Level 1 (no decimation, so four plane-volumes would be 32 of 68 B/voxel): the march merges
// axis 2 as well and hands two volumes to an axis-1 column filter -- see the second half of
// this file.
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
    int hal;              // level 1 (k_inv3_l1_axis02): halo cells on either side of a k tile (0: the tile is the row)
    int lo_pos, hi_pos;   // level >= 2: sum(ha*hb) > 0 of the g0 / g1 pair (lowlevel.py:205,232)
    // level 1: l_a = g0o, h_a = g1o.  level >= 2: colifilt(., g0b, g0a) + colifilt(., g1b, g1a):
    // l_a = g0b, l_b = g0a, h_a = g1b, h_b = g1a
    float l_a[DT_MAXT], l_b[DT_MAXT], h_a[DT_MAXT], h_b[DT_MAXT];
};

// Tile geometry of the march: NT threads = the four (a1, a2) combinations (wavefront index mod 4) x NT / 4 cells,
// CK cells per tile row (a power of two).  ck0, the first cell column of a tile, may be negative (halo cells of the
// level-1 kernel below): cells outside the volume are not loaded.
template <int NT_, int CK_>
struct I3Geo {
    static constexpr int NT = NT_, NCELL = NT / 4, CK = CK_, CJ = NCELL / CK;
    static constexpr int SLAB = NCELL * REC_LDS;         // floats of one staged record tile
    static constexpr int NPIECE = NCELL * 14;            // 16-byte pieces
    static constexpr int RP = (NPIECE + NT - 1) / NT;
    static_assert(NT % 256 == 0 && CJ * CK == NCELL && (CK & (CK - 1)) == 0, "tile geometry");
    static DT_HD int combo(int tid) { return (tid >> 6) & 3; }                 // 2 a1 + a2
    static DT_HD int cell(int tid) { return (tid >> 8) * 64 + (tid & 63); }    // row * CK + column
};
typedef I3Geo<256, 32> I3GeoA;                       // levels >= 2 (k_inv3_axis0): 2 x 32 cells
constexpr int I3_CJ = I3GeoA::CJ, I3_CK = I3GeoA::CK, I3_SLAB = I3GeoA::SLAB;

// level 1: odd-length biort filters g0o (M0 taps) on the a0 = 0 branch, g1o (M1) on a0 = 1
template <int M0_, int M1_>
struct Inv3L1 {
    static constexpr int M0 = M0_, M1 = M1_;
    static constexpr int H0 = M0 / 2, H1 = M1 / 2, H = cmax(H0, H1);
    static constexpr int HP = (H + 1) / 2;           // record pairs before / after the current one that reach it
    static_assert(M0 % 2 == 1 && M1 % 2 == 1, "biort filters must have odd length");
    // Transposed form: instead of a ring of the last 4 HP + 2 input slices of both octants (what Inv3L2 keeps), the
    // partial sums of the output slices they reach -- slice s adds tap t to output s - H + t -- which is one set
    // of NA registers per position whatever the number of input octants: 36 instead of 68 live registers for
    // near_sym_a.  acc[i] belongs to output slice 2 q - HE + i while pair q is being added; after it, outputs
    // 2 q - HE and 2 q - HE + 1 are complete.
    static constexpr int HE = 2 * HP, NA = HE + H + 2;
    static DT_HD void accumulate(const Inv3AParams &p, float (&acc)[4][NA], const float (&xa)[2][4],
                                 const float (&xb)[2][4]) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int t = 0; t < M0; ++t) acc[q][HE + e - H0 + t] += p.l_a[t] * xa[e][q];
#pragma unroll
                for (int t = 0; t < M1; ++t) acc[q][HE + e - H1 + t] += p.h_a[t] * xb[e][q];
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

template <class F, class G = I3GeoA>
struct Inv3AState {
    float ra[4][F::NS], rb[4][F::NS];   // [dj*2 + dk][ring slot] of the a0 = 0 / 1 octant
    float R[G::RP][4];                  // record pieces in flight
    float L[2][4];                      // lowpass pair in flight (wavefront 0 only)
};

template <class F, class G>
struct Inv3TState {
    float acc[4][F::NA];                // [dj*2 + dk][pending output slice]: Inv3L1::accumulate
    float R[G::RP][4];
    float L[2][4];
};

// virtual record q (slices 2q, 2q+1 with symmetric extension): real record, and whether the
// two slices arrive swapped
DT_HD void i3_vrec(int q, int n0, int &rq, int &swap) {
    int r = reflect_i(2 * q, n0);
    rq = r >> 1; swap = r & 1;
}

// issue the loads of virtual record q (all threads)
template <class F, class G = I3GeoA, class ST>
DT_HD void i3a_issue_rec(const Inv3AParams &p, ST &st, int tid, int cj0, int ck0, int q) {
    const int e1 = p.n1 / 2, e2 = p.n2 / 2;
    int rq, sw;
    i3_vrec(q, p.n0, rq, sw);
    // pieces [lo14, hi14) of a tile row belong to cells inside the volume
    const int lo14 = ck0 < 0 ? -14 * ck0 : 0;
    const int hi14 = 14 * (e2 - ck0 < G::CK ? e2 - ck0 : G::CK);
#pragma unroll
    for (int s = 0; s < G::RP; ++s) {
        int piece = tid + G::NT * s;
        int row = piece / (G::CK * 14), within = piece - row * (G::CK * 14);
        bool ok = piece < G::NPIECE && cj0 + row < e1 && within >= lo14 && within < hi14;
        if (!ok) { row = 0; within = lo14; }        // harmless in-range address, value unused
        const f4 *src = reinterpret_cast<const f4 *>(p.Yh + (((int64_t)rq * e1 + (cj0 + row)) * e2 + ck0) * 56);
        const dt_v4f v = DT_STREAM_LOAD_F4(src + within);
        st.R[s][0] = v.x; st.R[s][1] = v.y; st.R[s][2] = v.z; st.R[s][3] = v.w;
    }
}

// issue the loads of lowpass slices 2q, 2q+1 (the wavefronts of combination 0: the (0,0,0) octant)
template <class F, class G = I3GeoA, class ST>
DT_HD void i3a_issue_low(const Inv3AParams &p, ST &st, int tid, int cj0, int ck0, int q) {
    if (G::combo(tid) != 0) return;
    const int idx = G::cell(tid);
    int j = 2 * (cj0 + idx / G::CK), k = 2 * (ck0 + idx % G::CK);
    if (j >= p.n1) j = 0;
    if (k >= p.n2 || k < 0) k = 0;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const float *sl = p.LLL + ((int64_t)reflect_i(2 * q + d, p.n0) * p.n1 + j) * p.n2 + k;
        f2 a = *reinterpret_cast<const f2 *>(sl), b = *reinterpret_cast<const f2 *>(sl + p.n2);
        st.L[d][0] = a.x; st.L[d][1] = a.y; st.L[d][2] = b.x; st.L[d][3] = b.y;
    }
}

template <class F, class G = I3GeoA, class ST>
DT_HD void i3a_slab_write(ST &st, float *slab, int tid) {
    f4 *dst = reinterpret_cast<f4 *>(slab);
#pragma unroll
    for (int s = 0; s < G::RP; ++s) {
        int piece = tid + G::NT * s;
        DT_PIN_HERE(st.R[s][0]);
        if (piece < G::NPIECE) dst[slab_f4(piece)] = f4{st.R[s][0], st.R[s][1], st.R[s][2], st.R[s][3]};
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
template <class F, class G = I3GeoA>
DT_HD void i3a_push(const Inv3AParams &p, Inv3AState<F, G> &st, const float *slab, int tid, int qv) {
    const int v = G::combo(tid), lane = G::cell(tid);
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

// transposed form: add virtual record qv (from the slab; octant (0,0,0) from the lowpass pair in flight) to the
// pending outputs, hand out the finished pair (slices 2 (qv - HP), + 1) and move on
template <class F, class G>
DT_HD void i3a_accumulate(const Inv3AParams &p, Inv3TState<F, G> &st, const float *slab, int tid, int qv,
                          float (&out)[2][4]) {
    const int v = G::combo(tid);
    int rq, sw;
    i3_vrec(qv, p.n0, rq, sw);
    const float *rec = slab + G::cell(tid) * REC_LDS;
    float xa[2][4], xb[2][4], ev[4], od[4];
    c2cube_piece(rec + 8 * octant_slot(4 + v), ev, od);
#pragma unroll
    for (int q = 0; q < 4; ++q) { xb[0][q] = sw ? od[q] : ev[q]; xb[1][q] = sw ? ev[q] : od[q]; }
    if (v == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            DT_PIN_HERE(st.L[0][q]);
            DT_PIN_HERE(st.L[1][q]);
            xa[0][q] = st.L[0][q]; xa[1][q] = st.L[1][q];       // the loads already followed the reflection
        }
    } else {
        c2cube_piece(rec + 8 * octant_slot(v), ev, od);
#pragma unroll
        for (int q = 0; q < 4; ++q) { xa[0][q] = sw ? od[q] : ev[q]; xa[1][q] = sw ? ev[q] : od[q]; }
    }
    F::accumulate(p, st.acc, xa, xb);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        out[0][q] = st.acc[q][0]; out[1][q] = st.acc[q][1];
#pragma unroll
        for (int i = 0; i < F::NA - 2; ++i) st.acc[q][i] = st.acc[q][i + 2];
        st.acc[q][F::NA - 2] = 0.f; st.acc[q][F::NA - 1] = 0.f;
    }
}

template <class F, class G = I3GeoA>
DT_HD void i3a_store(const Inv3AParams &p, const float (&out)[F::NOUT][4], int tid, int cj0, int ck0, int c) {
    const int v = G::combo(tid), idx = G::cell(tid);
    const int j = 2 * (cj0 + idx / G::CK), k = 2 * (ck0 + idx % G::CK);
    if (j >= p.n1 || k >= p.n2) return;
#pragma unroll
    for (int e = 0; e < F::NOUT; ++e) {
        int so = F::NOUT * c + e - p.crop0;
        if (so < 0 || so >= p.S) continue;
        float *o = p.P + v * p.pstride + ((int64_t)so * p.n1 + j) * p.n2 + k;
        *reinterpret_cast<f2 *>(o) = f2{out[e][0], out[e][1]};
        *reinterpret_cast<f2 *>(o + p.n2) = f2{out[e][2], out[e][3]};
    }
}


// ======================================================================================
// Level 1 with the axis-2 merge inside the march (k_inv3_l1_axis02 + k_inv3_l1_axis1).
// ======================================================================================
// Level 1 does not decimate, so the four (a1, a2) plane-volumes of the two-launch scheme above are 16 B/voxel
// written and 16 read back -- 32 of the level's 68 B/voxel.  Here the march also merges axis 2: the wavefronts
// leave their slices of the four planes in an LDS exchange buffer (E), and after a barrier every thread filters
// four consecutive k of one row of Q[a1] = colfilter(P[a1, 0], g0o) + colfilter(P[a1, 1], g1o) along axis 2
// and stores them: 8 B/voxel of intermediates each way, and what is left for the second launch is the axis-1
// merge alone, a column filter over coalesced rows (52 B/voxel in all).  A tile is NT / 4 = 128 cells of CJ cell
// rows (one row of 128 cells, 2 x 64, 4 x 32: the narrowest that holds a row); rows longer than 128 cells are cut
// into tiles with two halo cells on either side (3 % more record reads).  Symmetric extension along axis 2 is
// written into E by the threads of the two cells at either end of a row.
template <class G>
struct I3Ex {
    static constexpr int RL = 2 * G::CK + 8;             // floats per row: 4 halo + 2 CK + 4 halo
    static constexpr int ROWS = 2 * 4 * 2 * G::CJ;       // [slice parity e][a1, a2][2 r + dj]
    static constexpr int FLOATS = ROWS * RL;
};

// out[e][dj*2 + dk] of plane (a1, a2) = combo(tid) -> E, with the mirror images of the row ends
template <class F, class G>
DT_HD void i3a_exchange(const Inv3AParams &p, const float (&out)[2][4], float *E, int tid, int ck0) {
    typedef I3Ex<G> X;
    const int v = G::combo(tid), idx = G::cell(tid), row = idx / G::CK, col = idx % G::CK;
    const int gc = ck0 + col, e2 = p.n2 / 2;
    if (gc < 0 || gc >= e2) return;
    // mirror of voxel k: -1 - k at the left end, 2 n2 - 1 - k at the right one; E position of voxel k: 4 + k - 2 ck0
    int mpos = -1;
    if (gc < 2) mpos = 4 + (-1 - 2 * gc) - 2 * ck0;
    else if (gc >= e2 - 2) mpos = 4 + (2 * p.n2 - 1 - 2 * gc) - 2 * ck0;
    if (mpos < 1 || mpos >= X::RL) mpos = -1;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
            float *r = E + ((e * 4 + v) * 2 * G::CJ + 2 * row + dj) * X::RL;
            *reinterpret_cast<f2 *>(r + 4 + 2 * col) = f2{out[e][2 * dj], out[e][2 * dj + 1]};
            if (mpos >= 0) { r[mpos] = out[e][2 * dj]; r[mpos - 1] = out[e][2 * dj + 1]; }
        }
}

// E -> four consecutive k of one row of Q[a1] (slices 2c, 2c + 1), stored
template <class F, class G>
DT_HD void i3a_merge_k(const Inv3AParams &p, const float *E, int tid, int cj0, int ck0, int c) {
    typedef I3Ex<G> X;
    constexpr int QPR = G::CK / 2;                       // quads per row
    const int quad = tid % QPR, rowid = tid / QPR;       // rowid < 8 CJ
    const int a1 = rowid / (4 * G::CJ), e = (rowid / (2 * G::CJ)) & 1, jr = rowid % (2 * G::CJ);
    const int j = 2 * cj0 + jr, k = 2 * ck0 + 4 * quad;
    const int hq = p.hal / 2;
    if (j >= p.n1 || k >= p.n2 || quad < hq || quad >= QPR - hq) return;
    int so = 2 * c + e;
    if (so >= p.S) return;
    const float *ea = E + ((e * 4 + 2 * a1) * 2 * G::CJ + jr) * X::RL + 4 * quad;     // window: voxels k - 4 .. k + 7
    const float *eb = ea + 2 * G::CJ * X::RL;
    float wa[12], wb[12];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const f4 a = reinterpret_cast<const f4 *>(ea)[q], b = reinterpret_cast<const f4 *>(eb)[q];
        wa[4 * q] = a.x; wa[4 * q + 1] = a.y; wa[4 * q + 2] = a.z; wa[4 * q + 3] = a.w;
        wb[4 * q] = b.x; wb[4 * q + 1] = b.y; wb[4 * q + 2] = b.z; wb[4 * q + 3] = b.w;
    }
    float o[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < F::M0; ++t) s += p.l_a[t] * wa[4 + m + F::H0 - t];
#pragma unroll
        for (int t = 0; t < F::M1; ++t) s += p.h_a[t] * wb[4 + m + F::H1 - t];
        o[m] = s;
    }
    float *dst = p.P + a1 * p.pstride + ((int64_t)so * p.n1 + j) * p.n2 + k;
    if ((p.n2 & 3) == 0) *reinterpret_cast<f4 *>(dst) = f4{o[0], o[1], o[2], o[3]};
    else {
        *reinterpret_cast<f2 *>(dst) = f2{o[0], o[1]};
        if (k + 2 < p.n2) *reinterpret_cast<f2 *>(dst + 2) = f2{o[2], o[3]};
    }
}

// ---- the axis-1 merge: Z = colfilter(Q[0], g0o) + colfilter(Q[1], g1o) down the rows of every slice ----------
struct Inv3BParams {
    const float *Q;       // [2][S][n1][n2]
    int64_t pstride;
    float *Z;             // [S][n1][n2]
    int S, n1, n2;
    int kvecs, strips;    // tasks per row / per column
    float g0[DT_MAXT], g1[DT_MAXT];
};

// one task: VEC consecutive k of RS consecutive rows
template <class F, int VEC, int RS>
DT_HD void i3b_axis1(const Inv3BParams &p, int64_t task) {
    const int kv = (int)(task % p.kvecs);
    const int64_t t2 = task / p.kvecs;
    const int strip = (int)(t2 % p.strips), s = (int)(t2 / p.strips);
    if (s >= p.S) return;
    const int k = kv * VEC, j0 = strip * RS;
    constexpr int WN = RS + 2 * F::H;
    const bool interior = j0 - F::H >= 0 && j0 + RS + F::H <= p.n1;
    float acc[RS][VEC];
#pragma unroll
    for (int q = 0; q < RS; ++q)
#pragma unroll
        for (int x = 0; x < VEC; ++x) acc[q][x] = 0.f;
#pragma unroll
    for (int a1 = 0; a1 < 2; ++a1) {
        const float *src = p.Q + a1 * p.pstride + (int64_t)s * p.n1 * p.n2 + k;
        float w[WN][VEC];
#pragma unroll
        for (int r = 0; r < WN; ++r) {
            const int jr = interior ? j0 - F::H + r : reflect_i(j0 - F::H + r, p.n1);
            const float *row = src + (int64_t)jr * p.n2;
            if (VEC == 4) {
                const f4 t = *reinterpret_cast<const f4 *>(row);
                w[r][0] = t.x; w[r][1] = t.y; w[r][VEC > 2 ? 2 : 0] = t.z; w[r][VEC > 2 ? 3 : 0] = t.w;
            } else {
                const f2 t = *reinterpret_cast<const f2 *>(row);
                w[r][0] = t.x; w[r][1] = t.y;
            }
        }
#pragma unroll
        for (int q = 0; q < RS; ++q)
#pragma unroll
            for (int x = 0; x < VEC; ++x) {
                if (a1 == 0) {
#pragma unroll
                    for (int t = 0; t < F::M0; ++t) acc[q][x] += p.g0[t] * w[q + F::H + F::H0 - t][x];
                } else {
#pragma unroll
                    for (int t = 0; t < F::M1; ++t) acc[q][x] += p.g1[t] * w[q + F::H + F::H1 - t][x];
                }
            }
    }
    float *dst = p.Z + ((int64_t)s * p.n1 + j0) * p.n2 + k;
#pragma unroll
    for (int q = 0; q < RS; ++q) {
        if (j0 + q >= p.n1) break;
        if (VEC == 4) *reinterpret_cast<f4 *>(dst + (int64_t)q * p.n2) = f4{acc[q][0], acc[q][1], acc[q][VEC > 2 ? 2 : 0], acc[q][VEC > 2 ? 3 : 0]};
        else *reinterpret_cast<f2 *>(dst + (int64_t)q * p.n2) = f2{acc[q][0], acc[q][1]};
    }
}

}  // namespace dt3d
