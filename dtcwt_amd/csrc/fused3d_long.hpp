// Level 1 of the float32 3-D DT-CWT for LONG level-1 filters (near_sym_b: 13 / 19 taps) as two launches per direction (gfx950).
//
// The one-launch level 1 of fused3d_march.hpp / fused3d_tiles.hpp keeps a window of 2 H + 2 slices of a tile in registers
// and filters the other two axes inside the same workgroup: at H = 9 that is a ring of 20 slices and a halo of 9 on every
// side of a tile, which neither the registers nor the LDS hold.  Until round 5 these filters ran axis by axis on the
// generic kernels (84 B/voxel in seven launches, the cube2c ones writing 32-byte pieces of the 224-byte records: 755 us
// forward / 965 us inverse at 256^3 -- profiles/r05/c4_qbgn.txt).  Here a level is TWO launches around four plane volumes
// P[2 a1 + a2] (the in-slice subbands: a1 = lo | hi along axis 1, a2 = lo | hi along axis 2; 16 B/voxel):
//
//   forward   k_fwd1m<.., PLANES>   (march2d_l1.hpp)   X -> P         every slice is an image of the 2-D level-1 march
//             k_fwd3l_axis0         (this file)        P -> LLL, Yh   axis 0 + cube2c
//   inverse   k_inv3l_axis0         (fused3d.hip, the march of fused3d_inv_tiles.hpp)   LLL, Yh -> P   c2cube + axis 0
//             k_inv1m<.., PLANES>   (march2d_l1.hpp)   P -> X
//
// 4 + 16 | 16 + 32 = 68 B/voxel each way against the 36 of a one-launch level.
//
// k_fwd3l_axis0: a workgroup = four wavefronts = the four planes of one job (a row of cells along axis 1, a strip of 64
// cells along axis 2, a chunk of slices); a lane owns ONE 2 x 2 cell column.  Axis 2 is the lane axis and needs no
// neighbours any more, so there are no halo lanes and no DPP: a wavefront marches down axis 0 over a register ring of the
// last 2 H + 2 slices of its plane (2 rows x 2 columns: 80 registers at H = 9), the loop unrolled over the ring's period,
// filters (lo0, hi0) with the column filter of the 2-D march (col_lohi2), runs cube2c on its two octants (a0 = 0 / 1) and
// puts them into the record row the four wavefronts assemble in a shared LDS slab; the row leaves as one contiguous run of
// 16-byte pieces (two LDS-only barriers per slice pair).  Wavefront 0 stores the lowpass volume.  cube2c's 1/2 rides on
// the taps (exact).
//
// Reference: dtcwt/numpy/transform3d.py:208-289 (level 1: colfilter along axes 2, 1, 0 + cube2c :532-579).
#pragma once
#include "fused3d_march.hpp"
#include "march2d_l1.hpp"

namespace dt3l {

using dt2d::DtBuf;
using dt2d::f2;
using dt2d::f4;

struct Fwd3lParams {
    const float *P;       // [4][n0][n1][n2]
    int64_t pstride;      // n0 * n1 * n2
    float *LLL;           // [n0][n1][n2]
    float *Yh;            // [n0/2][n1/2][n2/2][56 floats]
    int n0, n1, n2;       // all even
    int nstrip, ncr, nchunk, chunk;       // strips of 64 cells along axis 2, rows of cells (n1 / 2), chunks of `chunk` slices (even)
    // axis 0: (h0, h1) pairs by distance from the centre with cube2c's 1/2: [0] plane 0 -- octant (0, 0, 0) is the lowpass
    // volume and stays unscaled: (h0, h1 / 2) --, [1] planes 1 .. 3: (h0, h1) / 2
    float hp[2][2 * (dtm::MAXH1 + 1)] __attribute__((aligned(8)));
};
inline void pack_fwd3l(Fwd3lParams &p, const double *h0, int m0, const double *h1, int m1) {
    for (int d = 0; d <= dtm::MAXH1; ++d) {
        const double a = d <= m0 / 2 ? h0[m0 / 2 - d] : 0.0, b = d <= m1 / 2 ? h1[m1 / 2 - d] : 0.0;
        p.hp[0][2 * d] = (float)a; p.hp[0][2 * d + 1] = (float)(0.5 * b);
        p.hp[1][2 * d] = (float)(0.5 * a); p.hp[1][2 * d + 1] = (float)(0.5 * b);
    }
}

template <int M0, int M1>
__global__ void __launch_bounds__(256) k_fwd3l_axis0(const Fwd3lParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = dtm::Fwd1m<M0, M1>;
    using dtm::pk2;
    constexpr int HH = G::HH, WR = G::WR, PER = G::PER;
    __shared__ __attribute__((aligned(16))) f4 slab[64 * 15 + 8];           // a record = 14 pieces, 15 apart (bank spread)
    const int tid = threadIdx.x, lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);                 // plane 2 a1 + a2
    const int w = blockIdx.x;
    const int cr = w % p.ncr, rest = w / p.ncr;
    const int strip = rest % p.nstrip, chunk = rest / p.nstrip;
    if (chunk >= p.nchunk) return;
    const int n0 = p.n0, n1 = p.n1, n2 = p.n2;
    const int ncell = n2 / 2 - strip * 64 < 64 ? n2 / 2 - strip * 64 : 64;         // cells of this strip
    const bool owns = lane < ncell;
    const int k0 = 2 * (strip * 64 + (owns ? lane : 0));
    const int j0 = 2 * cr;
    const DtBuf bx = dtm::dt_buf2g(p.P + v * p.pstride);
    const unsigned voff[2] = {((unsigned)j0 * (unsigned)n2 + (unsigned)k0) * 4u, ((unsigned)(j0 + 1) * (unsigned)n2 + (unsigned)k0) * 4u};
    const int64_t spitch = (int64_t)n1 * n2;
    const int s0 = chunk * p.chunk;
    const int ns = n0 - s0 < p.chunk ? n0 - s0 : p.chunk;           // slices of this job (even)
    const int nst = ns / 2;                                         // steps; the loop leaves inside a period of the ring (uniform)
    const int last_slice = s0 + ns - 1 + HH;
    auto soff = [&](int s) -> unsigned {
        s = s > last_slice ? last_slice : s;
        s = s < 0 ? -1 - s : s; s = s >= n0 ? 2 * n0 - 1 - s : s;
        return (unsigned)s * (unsigned)spitch * 4u;                 // bytes (a volume is < 2 GiB: launcher)
    };
    auto load_slice = [&](int s, pk2 (&o)[2]) {
        const unsigned so = soff(s);
#pragma unroll
        for (int h = 0; h < 2; ++h) { const f2 t = dt2d::dt_buf_ld2(bx, voff[h], so); o[h] = pk2{t.x, t.y}; }
    };

    pk2 ring[WR][2], pre[2][2];
#pragma unroll
    for (int i = 0; i < WR; ++i) load_slice(s0 - HH + i, ring[i]);
    load_slice(s0 - HH + WR, pre[0]);
    load_slice(s0 - HH + WR + 1, pre[1]);
#pragma unroll
    for (int i = 0; i < WR; ++i) asm volatile("" : "+v"(ring[i][0].x), "+v"(ring[i][0].y), "+v"(ring[i][1].x), "+v"(ring[i][1].y) : : "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(pre[i][0].x), "+v"(pre[i][0].y), "+v"(pre[i][1].x), "+v"(pre[i][1].y) : : "memory");

    const pk2 *hpp = reinterpret_cast<const pk2 *>(p.hp[v == 0 ? 0 : 1]);
    // record slots of this plane's two octants (oracle _OCTANTS order: (a0, a1, a2) = (0,1,0) (1,0,0) (1,1,0) (0,0,1) (0,1,1)
    // (1,0,1) (1,1,1)): plane 0 -> (lowpass volume, 1), plane 1 = (a1, a2) = (0, 1) -> (3, 5), plane 2 -> (0, 2), plane 3 -> (4, 6)
    const int slot_lo = v == 1 ? 3 : (v == 2 ? 0 : 4), slot_hi = v == 0 ? 1 : (v == 1 ? 5 : (v == 2 ? 2 : 6));
    const int64_t rec_row = (int64_t)(n2 / 2) * 56;                 // floats per row of cells
    float *const Lb = p.LLL + (int64_t)j0 * n2 + 2 * strip * 64;
    float *const Yb = p.Yh + (int64_t)cr * rec_row + (int64_t)strip * 64 * 56;
    const int npiece = ncell * 14;
    int pslab[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int piece = tid + 256 * i; pslab[i] = (piece / 14) * 15 + piece % 14; }

    for (int t0 = 0; t0 < nst; t0 += PER) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int m = t0 + k, s = s0 + 2 * m;                  // output slices s, s + 1
            if (m >= nst) return;                                  // the whole workgroup at once
            pk2 in0[2] = {pre[0][0], pre[0][1]}, in1[2] = {pre[1][0], pre[1][1]};
            load_slice(s - HH + WR + 2, pre[0]);
            load_slice(s - HH + WR + 3, pre[1]);
            constexpr bool ok = true;
            // ---- axis 0: O[q][e][c] = (lo0, hi0) at slice s + q, row j0 + e, column k0 + c
            pk2 O[2][2][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                pk2 win[WR];
#pragma unroll
                for (int j = 0; j < WR; ++j) win[j] = ring[(2 * k + j) % WR][e];
#pragma unroll
                for (int q = 0; q < 2; ++q) dtm::col_lohi2<HH>(&win[q + HH], hpp, O[q][e][0], O[q][e][1]);
            }
            // ---- the slab must be free: the flush of the previous step has read it
            DT3M_LDS_BARRIER();
            {
                f4 o0, o1;
                dt3m::cube2c_cell(O[0][0][0].y, O[0][1][0].y, O[1][0][0].y, O[1][1][0].y, O[0][0][1].y, O[0][1][1].y, O[1][0][1].y, O[1][1][1].y, o0, o1);
                if (owns) { slab[lane * 15 + 2 * slot_hi] = o0; slab[lane * 15 + 2 * slot_hi + 1] = o1; }
                dt3m::cube2c_cell(O[0][0][0].x, O[0][1][0].x, O[1][0][0].x, O[1][1][0].x, O[0][0][1].x, O[0][1][1].x, O[1][0][1].x, O[1][1][1].x, o0, o1);
                if (owns && v != 0) { slab[lane * 15 + 2 * slot_lo] = o0; slab[lane * 15 + 2 * slot_lo + 1] = o1; }
            }
            // the lowpass volume (every wavefront issues the same four stores: against zero bytes unless it is wavefront 0)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int so = ok ? s + q : s0;
                    const DtBuf bl = dtm::dt_buf_n(Lb + (int64_t)so * spitch + (int64_t)e * n2, (ok && v == 0) ? 8u * ncell : 0u);
                    dt2d::dt_buf_st2<false>(bl, 8u * (unsigned)lane, 0u, f2{O[q][e][0].x, O[q][e][1].x});
                }
            DT3M_LDS_BARRIER();
            // ---- flush: the strip's row of cells as one run of 16-byte pieces
            {
                const int so = ok ? s / 2 : s0 / 2;
                const DtBuf by = dtm::dt_buf_n(Yb + (int64_t)so * (n1 / 2) * rec_row, ok ? 16u * npiece : 0u);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int piece = tid + 256 * i;
                    if (i < 3 || piece < 64 * 14) {
                        const f4 val = slab[pslab[i]];
                        dt2d::dt_buf_st4<true>(by, 16u * (unsigned)piece, 0u, val);
                    }
                }
            }
            ring[(2 * k) % WR][0] = in0[0]; ring[(2 * k) % WR][1] = in0[1];
            ring[(2 * k + 1) % WR][0] = in1[0]; ring[(2 * k + 1) % WR][1] = in1[1];
        }
    }
#endif
}


// ======================================================================================================================
// Round 6: the level cut the OTHER way -- axis 0 first, then both in-slice axes + cube2c in one launch -- so that only TWO
// intermediate volumes reach HBM instead of four:
//
//   forward   colfilter2 along axis 0 (generic2d.hip: the marching pair kernel)   X -> V[lo0], V[hi0]          4 -> 8 B/voxel
//             k_fwd3l_slices (below)                                              V -> LLL, Yh                 8 -> 32 B/voxel
//
// 52 B/voxel instead of 68 (the three filters commute: the same sums in another order, agreeing with the axis-2-first order
// of transform3d.py:256-273 to float32 rounding).  k_fwd3l_slices: a workgroup = FOUR wavefronts = (axis-0 band f0 = lo | hi)
// x (slice parity a0) of one slice PAIR, one strip of columns and one band of rows: each wavefront runs the 2-D level-1 march
// of march2d_l1.hpp on its own slice image (register ring of 20 rows down axis 1, DPP / in-lane mirror halo along axis 2) and
// holds, per step of two rows, its four in-slice planes (f1, f2).  cube2c needs both slices of a cell: the two wavefronts of
// a band exchange half of their planes through LDS -- a0 = 0 takes the f1 = lo octants of both slices, a0 = 1 the f1 = hi ones
// -- run cube2c on the lane's two cells and put the results into the record row that all four wavefronts assemble in one slab
// (7 octants x 8 floats per cell); the row leaves as one contiguous run of 16-byte pieces.  The f0 = lo wavefronts store their
// (lo, lo) plane as the lowpass volume.  Two LDS-only barriers per step.  cube2c's 1/2 rides on the row taps.
// Reference: dtcwt/numpy/transform3d.py:208-289, cube2c :532-579.
// ======================================================================================================================
struct Fwd3sParams {
    const float *V;       // [2][n0][n1][n2]: axis 0 lowpass / highpass of X
    int64_t vstride;      // n0 * n1 * n2
    float *LLL;           // [n0][n1][n2]
    float *Yh;            // [n0/2][n1/2][n2/2][56 floats]
    int n0, n1, n2;       // n0, n1 even, n2 % 4 == 0
    dtm::MarchJobs jb;    // strips along axis 2, bands of rows of axis 1, "images" = slice pairs (n0 / 2)
    float hp[2 * (dtm::MAXH1 + 1)] __attribute__((aligned(8)));          // axis 1: (h0, h1) by distance from the centre
    // axis 2 with cube2c's 1/2: (h0, h1) / 2 for every plane of every wavefront (one table, no indexing by f0: a run-time index
    // into the parameter block put it on the stack); the (lo, lo, lo) plane is the lowpass volume and is doubled back on its
    // way out -- exact, the halved taps are exact halves
    float hq[2 * (dtm::MAXH1 + 1)] __attribute__((aligned(8)));
};
inline void pack_fwd3s(Fwd3sParams &p, const double *h0, int m0, const double *h1, int m1) {
    for (int d = 0; d <= dtm::MAXH1; ++d) {
        const double a = d <= m0 / 2 ? h0[m0 / 2 - d] : 0.0, b = d <= m1 / 2 ? h1[m1 / 2 - d] : 0.0;
        p.hp[2 * d] = (float)a; p.hp[2 * d + 1] = (float)b;
        p.hq[2 * d] = 0.5f * (float)a; p.hq[2 * d + 1] = 0.5f * (float)b;
    }
}

#if defined(__HIP_DEVICE_COMPILE__)
// c ? a : b by components (`c ? a : b` on two f4 LVALUES is a pointer select: both operands go to the stack)
__device__ __forceinline__ f4 sel4(bool c, const f4 &a, const f4 &b) { return f4{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w}; }
#endif

template <int M0, int M1, int P, bool EDGE>
__global__ void __launch_bounds__(256, 2) k_fwd3l_slices(const Fwd3sParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = dtm::Fwd1m<M0, M1>;
    using dtm::pk2; using dtm::dt_buf2g; using dtm::dt_buf_n;
    constexpr int HH = G::HH, WR = G::WR, HL = EDGE ? 0 : G::HL, VL = EDGE ? 64 : G::VL, PER = G::PER;
    static_assert(PER % P == 0, "prefetch depth divides the ring period");
    __shared__ __attribute__((aligned(16))) f4 slab[128 * 15 + 8];           // a row of 128 cells, 14 pieces each, 15 apart
    __shared__ __attribute__((aligned(16))) f4 xch[4][4 * 64];               // what a wavefront hands its partner: [plane][row][lane]
    const int tid = threadIdx.x, lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f0 = v >> 1, a0 = v & 1;
    int strip, band, b;
    if (!dtm::dtm_job(p.jb, blockIdx.x, strip, band, b)) return;           // the whole workgroup
    const int R = p.n1, C = p.n2;

    const int c0 = strip * (4 * VL) - 4 * HL + 4 * lane;
    const bool rev = c0 < 0 || c0 >= C;
    int lc = c0 < 0 ? -c0 - 4 : (c0 >= C ? 2 * C - 4 - c0 : c0);
    lc = lc < 0 ? 0 : (lc > C - 4 ? C - 4 : lc);
    const bool edge_strip = !EDGE && (strip == 0 || (strip + 1) * (4 * VL) + 4 * HL >= C);
    const int nv = (C - strip * (4 * VL)) / 4 < VL ? (C - strip * (4 * VL)) / 4 : VL;     // owning lanes
    const bool owns = lane >= HL && lane < HL + nv;

    const int64_t slice = (int64_t)R * C;
    const int64_t img = (int64_t)(2 * b + a0) * slice;
    const DtBuf bx = dt_buf2g(p.V + f0 * p.vstride + img);
    float *const Lb = p.LLL + img + strip * (4 * VL);
    const int64_t rec_row = (int64_t)(C / 2) * 56;                  // floats per row of cells
    float *const Yb = p.Yh + (int64_t)b * (R / 2) * rec_row + (int64_t)strip * (2 * VL) * 56;
    const unsigned pitch = (unsigned)C * 4u;

    const int rb = band * p.jb.band_rows;
    const int nrow = R - rb < p.jb.band_rows ? R - rb : p.jb.band_rows;
    const int nst = (nrow / 2 + PER - 1) / PER * PER;
    const int last_row = rb + nrow - 1 + HH;

    auto ldrow = [&](int u) -> f4 {
        u = u > last_row ? last_row : u;
        u = u < 0 ? -1 - u : u;
        u = u >= R ? 2 * R - 1 - u : u;
        return dt2d::dt_buf_ld4(bx, (unsigned)lc * 4u, (unsigned)u * pitch);
    };
    auto fix = [&](f4 &x) { if (edge_strip) x = rev ? dtm::rev4(x) : x; };

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

    const unsigned lv = 16u * (unsigned)(lane - HL);        // halo lanes: out of range either side
    const pk2 *hpp = reinterpret_cast<const pk2 *>(p.hp);
    const pk2 *hq = reinterpret_cast<const pk2 *>(p.hq);
    // record slots (oracle _OCTANTS order, (f0, f1, f2)): (0,1,0) (1,0,0) (1,1,0) (0,0,1) (0,1,1) (1,0,1) (1,1,1).  This
    // wavefront runs cube2c on the planes f1 = a0: (f1, lo) -> slot_a, (f1, hi) -> slot_b; (0, 0, lo) is the lowpass volume
    const int slot_a = a0 == 0 ? 1 : (f0 ? 2 : 0), slot_b = a0 == 0 ? (f0 ? 5 : 3) : (f0 ? 6 : 4);
    const bool has_a = a0 == 1 || f0 == 1;
    const int cell0 = 2 * (lane - HL);
    const int npiece = 2 * nv * 14;
    int pslab[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) { const int piece = tid + 256 * i; pslab[i] = (piece / 14) * 15 + piece % 14; }
    f4 *const xs = xch[v], *const xr = xch[v ^ 1];

    for (int t0 = 0; t0 < nst; t0 += PER) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int r = rb + 2 * (t0 + k);
            const f4 in0 = pre[(2 * k) % (2 * P)], in1 = pre[(2 * k + 1) % (2 * P)];
            pre[(2 * k) % (2 * P)] = ldrow(r - HH + WR + 2 * P);
            pre[(2 * k + 1) % (2 * P)] = ldrow(r - HH + WR + 2 * P + 1);
            const bool in_band = r < rb + nrow;             // uniform
            pk2 wp[2][WR];
#pragma unroll
            for (int j = 0; j < WR; ++j) {
                const f4 &w = ring[(2 * k + j) % WR];
                wp[0][j] = pk2{w.x, w.y}; wp[1][j] = pk2{w.z, w.w};
            }
            // pl[f1][f2][q]: the lane's four columns of row r + q of in-slice plane (f1, f2), cube2c's 1/2 included
            f4 pl[2][2][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                pk2 W[4 + 2 * HH];
                dtm::col_lohi2<HH>(&wp[0][q + HH], hpp, W[HH], W[HH + 1]);
                dtm::col_lohi2<HH>(&wp[1][q + HH], hpp, W[HH + 2], W[HH + 3]);
                dtm::halo_pairs<HH>(W);
                if constexpr (EDGE) dtm::edge_mirror<HH>(W, lane, nv);
                pk2 ol[4], oh[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) dtm::row_lohi_s<HH>(&W[c + HH], hq, hq, ol[c], oh[c]);
                pl[0][0][q] = f4{ol[0].x, ol[1].x, ol[2].x, ol[3].x}; pl[0][1][q] = f4{ol[0].y, ol[1].y, ol[2].y, ol[3].y};
                pl[1][0][q] = f4{oh[0].x, oh[1].x, oh[2].x, oh[3].x}; pl[1][1][q] = f4{oh[0].y, oh[1].y, oh[2].y, oh[3].y};
            }
            // ---- hand the partner the planes it packs: f1 = 1 - a0 (the exchange area is free: the partner read it before the
            // second barrier of the previous step)
#pragma unroll
            for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
                for (int q = 0; q < 2; ++q) xs[(2 * f2 + q) * 64 + lane] = sel4(a0 == 0, pl[1][f2][q], pl[0][f2][q]);
            // the lowpass volume: plane (lo, lo) of the f0 = lo wavefronts, rows r, r + 1 of slice 2 b + a0 (every wavefront
            // issues the same stores, against zero bytes where they are not its business: march2d.hpp on vmcnt)
            {
                const int ro = in_band ? r : rb;
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    dt2d::dt_buf_st4<false>(dt_buf_n(Lb + (int64_t)(ro + q) * C, (in_band && f0 == 0) ? 16u * nv : 0u), lv, 0u,
                                            f4{2.f * pl[0][0][q].x, 2.f * pl[0][0][q].y, 2.f * pl[0][0][q].z, 2.f * pl[0][0][q].w});
            }
            DT3M_LDS_BARRIER();            // the exchange is written; the slab is free (the flush of the previous step has read it)
            {
                f4 mine[2][2], other[2][2];         // [f2][row] of plane f1 = a0: this slice's, the partner slice's
#pragma unroll
                for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
                    for (int q = 0; q < 2; ++q) { mine[f2][q] = sel4(a0 == 0, pl[0][f2][q], pl[1][f2][q]); other[f2][q] = xr[(2 * f2 + q) * 64 + lane]; }
#pragma unroll
                for (int f2 = 0; f2 < 2; ++f2) {
                    // S[slice parity][row parity]: the four columns
                                        const f4 s00 = sel4(a0 == 0, mine[f2][0], other[f2][0]), s01 = sel4(a0 == 0, mine[f2][1], other[f2][1]);
                    const f4 s10 = sel4(a0 == 0, other[f2][0], mine[f2][0]), s11 = sel4(a0 == 0, other[f2][1], mine[f2][1]);
                    const int slot = f2 == 0 ? slot_a : slot_b;
                    f4 o0, o1;
                    dt3m::cube2c_cell(s00.x, s01.x, s10.x, s11.x, s00.y, s01.y, s10.y, s11.y, o0, o1);
                    if (owns && (f2 == 1 || has_a)) { slab[cell0 * 15 + 2 * slot] = o0; slab[cell0 * 15 + 2 * slot + 1] = o1; }
                    dt3m::cube2c_cell(s00.z, s01.z, s10.z, s11.z, s00.w, s01.w, s10.w, s11.w, o0, o1);
                    if (owns && (f2 == 1 || has_a)) { slab[(cell0 + 1) * 15 + 2 * slot] = o0; slab[(cell0 + 1) * 15 + 2 * slot + 1] = o1; }
                }
            }
            DT3M_LDS_BARRIER();            // the record row is complete
            {
                const int ro = in_band ? r : rb;
                const DtBuf by = dt_buf_n(Yb + (int64_t)(ro >> 1) * rec_row, in_band ? 16u * npiece : 0u);
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    const f4 val = slab[pslab[i]];
                    dt2d::dt_buf_st4<true>(by, 16u * (unsigned)(tid + 256 * i), 0u, val);
                }
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
// The inverse the same way round (round 6): c2cube + both in-slice axes in one launch, then axis 0 --
//
//   inverse   k_inv3l_slices (below)                                               LLL, Yh -> V[lo0], V[hi0]     32 + 4 -> 8 B/voxel
//             colfilter_sum2 along axis 0 (generic2d.hip: the marching sum kernel)  V -> X                        8 -> 4 B/voxel
//
// 52 B/voxel instead of 68 (k_inv3l_axis0 + k_inv1m<PLANES> around four plane volumes).  k_inv3l_slices: a workgroup = FOUR
// wavefronts = (axis-0 band f0) x (slice parity a0) of one slice pair, one strip, one band of rows; per step the 256 threads
// bring ONE row of cells (one record row of the slice pair: 224 bytes per cell) into a shared slab, requested a step ahead;
// every wavefront takes from it the c2cube samples of ITS slice (a0) of ITS four octants (f0, f1, f2) -- the (lo, lo, lo)
// plane of the f0 = lo wavefronts is the lowpass volume instead -- and runs the level-1 inverse march of march2d_l1.hpp on
// them (row filters through DPP / in-lane mirrors, transposed column filters into 20 pending rows), storing its slice of
// V[f0].  Two LDS-only barriers per step, no exchange between the wavefronts.
// Reference: dtcwt/numpy/transform3d.py:385-440, c2cube :581-619.
// ======================================================================================================================
struct Inv3sParams {
    const float *LLL;     // [n0][n1][n2]
    const float *Yh;      // [n0/2][n1/2][n2/2][56 floats]
    float *V;             // [2][n0][n1][n2]: what the axis-0 synthesis filters g0o / g1o take
    int64_t vstride;
    int n0, n1, n2;
    dtm::MarchJobs jb;    // strips along axis 2, bands of rows of axis 1, "images" = slice pairs
    float gd0[2 * (dtm::MAXH1 + 1)] __attribute__((aligned(8))), gd1[2 * (dtm::MAXH1 + 1)] __attribute__((aligned(8)));
};
inline void pack_inv3s(Inv3sParams &p, int m0, int m1, const double *g0, const double *g1) {
    for (int d = 0; d <= dtm::MAXH1; ++d) {
        p.gd0[2 * d] = p.gd0[2 * d + 1] = d <= m0 / 2 ? (float)g0[m0 / 2 - d] : 0.f;
        p.gd1[2 * d] = p.gd1[2 * d + 1] = d <= m1 / 2 ? (float)g1[m1 / 2 - d] : 0.f;
    }
}

#if defined(__HIP_DEVICE_COMPILE__)
// c2cube of one octant piece (two f4 = four complex numbers of a cell, transform3d.py:581-619), the half a wavefront needs: the
// four samples v[dj][dk] of slice parity a0 (uniform)
__device__ __forceinline__ void c2cube_half(const f4 &a, const f4 &b, bool odd, float (&v)[2][2]) {
    const float pr = a.x, pi = a.y, qr = a.z, qi = a.w, rr = b.x, ri = b.y, sr = b.z, si = b.w, h = 0.5f;
    if (!odd) {
        v[0][0] = (pr + qr + rr + sr) * h; v[0][1] = (pi + qi + ri + si) * h;
        v[1][0] = (pi - qi + ri - si) * h; v[1][1] = (-pr + qr - rr + sr) * h;
    } else {
        v[0][0] = (pi + qi - ri - si) * h; v[0][1] = (-pr - qr + rr + sr) * h;
        v[1][0] = (-pr + qr + rr - sr) * h; v[1][1] = (-pi + qi + ri - si) * h;
    }
}
#endif

template <int M0, int M1, bool EDGE>
__global__ void __launch_bounds__(256, 2) k_inv3l_slices(const Inv3sParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = dtm::Inv1m<M0, M1>;
    using dtm::pk2; using dtm::dt_buf2g; using dtm::dt_buf_n;
    constexpr int H0 = G::H0, H1 = G::H1, HM = G::HM, HL = EDGE ? 0 : G::HL, VL = EDGE ? 64 : G::VL, NPX = G::NPX, WARM = G::WARM;
    __shared__ __attribute__((aligned(16))) f4 slab[128 * 15 + 8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f0 = v >> 1, a0 = v & 1;
    int strip, band, b;
    if (!dtm::dtm_job(p.jb, blockIdx.x, strip, band, b)) return;           // the whole workgroup
    const int R = p.n1, C = p.n2;
    const int cb = strip * (4 * VL) - 4 * HL;             // column of lane 0
    const int c0 = cb + 4 * lane;
    const bool mir = c0 < 0 || c0 >= C;
    const bool edge_strip = !EDGE && (cb < 0 || cb + 256 > C);      // uniform: some lane is mirrored
    const int nv = (C - strip * (4 * VL)) / 4 < VL ? (C - strip * (4 * VL)) / 4 : VL;
    int lc = c0 < 0 ? -c0 - 4 : (c0 >= C ? 2 * C - 4 - c0 : c0);
    lc = lc < 0 ? 0 : (lc > C - 4 ? C - 4 : lc);
    int sl = c0 < 0 ? (-c0 - 4 - cb) / 4 : (c0 >= C ? (2 * C - 4 - c0 - cb) / 4 : lane);      // the slab lane that holds this lane's cells
    sl = sl < 0 ? 0 : (sl > 63 ? 63 : sl);
    const int lmin = cb < 0 ? -cb / 4 : 0, lmax = (C - cb) / 4 - 1 < 63 ? (C - cb) / 4 - 1 : 63;     // lanes inside the volume

    const int64_t slice = (int64_t)R * C;
    const int64_t img = (int64_t)(2 * b + a0) * slice;
    // the lowpass volume feeds plane (lo, lo) of the f0 = lo wavefronts only: the others load against zero bytes
    const DtBuf bz = dt_buf_n(p.LLL + img, f0 == 0 ? (unsigned)slice * 4u : 0u);
    const int64_t rec_row_f = (int64_t)(C / 2) * 56;                 // floats per row of cells
    const float *const Yb = p.Yh + (int64_t)b * (R / 2) * rec_row_f + (int64_t)((cb + 4 * lmin) / 2) * 56;    // first cell inside, row 0
    float *const Xb = p.V + f0 * p.vstride + img + strip * (4 * VL);
    const unsigned pitch = (unsigned)C * 4u;
    const unsigned npiece = (unsigned)(2 * (lmax - lmin + 1)) * 14u;

    const int rb = band * p.jb.band_rows;
    const int nrow = R - rb < p.jb.band_rows ? R - rb : p.jb.band_rows;
    const int rr0 = rb / 2 - WARM, nst = nrow / 2 + 2 * WARM;      // record rows rr0 .. rr0 + nst - 1

    auto zrow = [&](int u) { u = u < 0 ? -1 - u : u; u = u >= R ? 2 * R - 1 - u : u; return u < 0 ? 0 : (u > R - 1 ? R - 1 : u); };
    auto rec_row = [&](int rr, bool &sw) { sw = rr < 0 || rr >= R / 2; rr = rr < 0 ? -1 - rr : rr; rr = rr >= R / 2 ? R - 1 - rr : rr; return rr < 0 ? 0 : (rr > R / 2 - 1 ? R / 2 - 1 : rr); };

    int pdst[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) { const int piece = tid + 256 * i; pdst[i] = (2 * lmin + piece / 14) * 15 + piece % 14; }
    f4 zp[2], rp[7];
    auto request = [&](int rr) {
        bool sw;
        zp[0] = dt2d::dt_buf_ld4(bz, (unsigned)lc * 4u + (unsigned)zrow(2 * rr) * pitch, 0u);
        zp[1] = dt2d::dt_buf_ld4(bz, (unsigned)lc * 4u + (unsigned)zrow(2 * rr + 1) * pitch, 0u);
        const DtBuf br = dt_buf_n(Yb + (int64_t)rec_row(rr, sw) * rec_row_f, 16u * npiece);
#pragma unroll
        for (int i = 0; i < 7; ++i) rp[i] = dt2d::dt_buf_ld4(br, 16u * (unsigned)(tid + 256 * i), 0u);
    };
    request(rr0);
    asm volatile("" : "+v"(zp[0].x), "+v"(zp[0].y), "+v"(zp[0].z), "+v"(zp[0].w), "+v"(zp[1].x), "+v"(zp[1].y), "+v"(zp[1].z), "+v"(zp[1].w) : : "memory");
#pragma unroll
    for (int i = 0; i < 7; ++i) asm volatile("" : "+v"(rp[i].x), "+v"(rp[i].y), "+v"(rp[i].z), "+v"(rp[i].w) : : "memory");

    pk2 PX[NPX][2];
#pragma unroll
    for (int i = 0; i < NPX; ++i) { PX[i][0] = pk2{0.f, 0.f}; PX[i][1] = pk2{0.f, 0.f}; }
    const unsigned xv = 16u * (unsigned)(lane - HL);
    const pk2 *gd0 = reinterpret_cast<const pk2 *>(p.gd0), *gd1 = reinterpret_cast<const pk2 *>(p.gd1);
    // record slots of this wavefront's octants (f0, f1, f2): idx = 4 f0 + 2 f1 + f2, slot = idx odd ? 3 + idx / 2 : idx / 2 - 1
    const int s00 = f0 ? 1 : 0 /* unused for f0 = lo */, s01 = f0 ? 5 : 3, s10 = f0 ? 2 : 0, s11 = f0 ? 6 : 4;

    for (int st = 0; st < nst; ++st) {
        const int rr = rr0 + st, rho = 2 * rr;
        bool sw;
        (void)rec_row(rr, sw);
        // ---- what was requested a step ago: the record row to the slab, the lowpass rows in place
#pragma unroll
        for (int i = 0; i < 7; ++i) if ((unsigned)(tid + 256 * i) < npiece) slab[pdst[i]] = rp[i];
        f4 zz[2] = {zp[0], zp[1]};
        if (edge_strip) { zz[0] = mir ? dtm::rev4(zz[0]) : zz[0]; zz[1] = mir ? dtm::rev4(zz[1]) : zz[1]; }
        request(rr + 1);
        DT3M_LDS_BARRIER();
        float qz[2][4], q05[2][4], q23[2][4], q14[2][4];
        {
            const f4 *ca = slab + (2 * sl) * 15, *cbp = ca + 15;        // the lane's two cells
            float t[2][2];
#define DT3L_OCT(dst_, slot_)                                                                   \
            c2cube_half(ca[2 * (slot_)], ca[2 * (slot_) + 1], a0 != 0, t);                       \
            dst_[0][0] = t[0][0]; dst_[0][1] = t[0][1]; dst_[1][0] = t[1][0]; dst_[1][1] = t[1][1]; \
            c2cube_half(cbp[2 * (slot_)], cbp[2 * (slot_) + 1], a0 != 0, t);                     \
            dst_[0][2] = t[0][0]; dst_[0][3] = t[0][1]; dst_[1][2] = t[1][0]; dst_[1][3] = t[1][1];
            DT3L_OCT(q23, s01) DT3L_OCT(q05, s10) DT3L_OCT(q14, s11)
            if (f0) { DT3L_OCT(qz, s00) }
            else {
#pragma unroll
                for (int c = 0; c < 4; ++c) { qz[0][c] = 0.f; qz[1][c] = 0.f; }
            }
#undef DT3L_OCT
            if (sw) {               // a reflected record row (top / bottom of the slices): its cells upside down
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float t_;
                    t_ = q05[0][c]; q05[0][c] = q05[1][c]; q05[1][c] = t_;
                    t_ = q23[0][c]; q23[0][c] = q23[1][c]; q23[1][c] = t_;
                    t_ = q14[0][c]; q14[0][c] = q14[1][c]; q14[1][c] = t_;
                    t_ = qz[0][c]; qz[0][c] = qz[1][c]; qz[1][c] = t_;
                }
            }
            if (edge_strip) {       // mirrored lanes: the mirror lane's four columns in reverse
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float t_;
#define DTM_REV4(x_) t_ = x_[e][0]; x_[e][0] = mir ? x_[e][3] : t_; x_[e][3] = mir ? t_ : x_[e][3]; \
                     t_ = x_[e][1]; x_[e][1] = mir ? x_[e][2] : t_; x_[e][2] = mir ? t_ : x_[e][2];
                    DTM_REV4(q05) DTM_REV4(q23) DTM_REV4(q14) DTM_REV4(qz)
#undef DTM_REV4
                }
            }
        }
        DT3M_LDS_BARRIER();            // every wavefront has read the row: the slab may take the next one
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float zl[4] = {zz[e].x, zz[e].y, zz[e].z, zz[e].w};
            pk2 Wa[4 + 2 * H0], Wb[4 + 2 * H1], Vv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                Wa[H0 + c] = pk2{f0 ? qz[e][c] : zl[c], q05[e][c]};
                Wb[H1 + c] = pk2{q23[e][c], q14[e][c]};
            }
            dtm::halo_pairs<H0>(Wa);
            dtm::halo_pairs<H1>(Wb);
            if constexpr (EDGE) { dtm::edge_mirror<H0>(Wa, lane, nv); dtm::edge_mirror<H1>(Wb, lane, nv); }
#pragma unroll
            for (int c = 0; c < 4; ++c) Vv[c] = dtm::sym_gg<H0>(&Wa[H0 + c], gd0) + dtm::sym_gg<H1>(&Wb[H1 + c], gd1);
            const pk2 v0p[2] = {pk2{Vv[0].x, Vv[1].x}, pk2{Vv[2].x, Vv[3].x}}, v1p[2] = {pk2{Vv[0].y, Vv[1].y}, pk2{Vv[2].y, Vv[3].y}};
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
        // rows rho - HM, rho - HM + 1 of this wavefront's slice of V[f0] are complete
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int x = rho - HM + e;
            const bool ok = x >= rb && x < rb + nrow;
            const int xo = ok ? x : 0;
            const DtBuf bo = dt_buf_n(Xb + (int64_t)xo * C, ok ? 16u * nv : 0u);
            dt2d::dt_buf_st4<false>(bo, xv, 0u, f4{PX[e][0].x, PX[e][0].y, PX[e][1].x, PX[e][1].y});
        }
    }
#endif
}

}  // namespace dt3l
