// Level 1 of the float32 3-D DT-CWT forward transform as a marching PAIR of wavefronts (gfx950): k_fwd3m_l1.
//
// The tile program k_fwd3_l1 (fused3d_tiles.hpp) hands two of its three axis passes through LDS planes and is bound by the
// instructions its wavefronts issue (1352 per wavefront and slice pair for 512 voxels, 256 VGPRs with spills; 128-136 us at
// 256^3 for a 604 MB stream that takes ~100).  Here the lessons of the 2-D marching launches (march2d.hpp) are applied to the
// volume:
//   * a workgroup = TWO wavefronts = one job = (a strip of 64 lanes x 4 columns of axis 2, ONE PAIR OF ROWS of axis 1, a chunk
//     of slices of axis 0); wavefront b of the pair owns the a1 = b half of the eight octants;
//   * axis 1 first, straight from the eight raw rows its two rows reach (loaded by index -- the neighbouring row pairs load
//     them as well, from the L2): each wavefront keeps only ITS half (lowpass or highpass along axis 1);
//   * axis 0 over a register ring of the last 8 filtered slices (2 rows x 4 columns: 64 registers per wavefront), the march
//     loop unrolled over the ring's period of four slice pairs;
//   * axis 2 last, the neighbouring columns from the neighbouring LANES (DPP wave shifts, one halo lane at strip
//     boundaries; at the edges of the volume the mirror columns are the lane's own, selected by a lane predicate -- a
//     256-wide row is exactly one strip with no halo lanes);
//   * cube2c is lane-local (a lane owns two 2 x 2 x 2 cells of each of its four octants); the 224-byte records of a row of
//     cells are assembled in an LDS slab the two wavefronts share and leave as one contiguous run (two LDS-only barriers
//     per slice pair); wavefront 0 stores the lowpass volume directly.
// Every factor 1/2 of cube2c rides on filter taps (exact).  Traffic: X 4 -> LLL 4 + Yh[0] 28 B/voxel; the raw rows of a
// slice are requested by four row pairs x two wavefronts, once from HBM.
//
// Reference: dtcwt/numpy/transform3d.py:208-289 (level 1: colfilter along axes 2, 1, 0 + cube2c :532-579).
#pragma once
#include "march2d.hpp"

namespace dt3m {

using dt2d::DtBuf;
using dt2d::f4;
#if defined(__HIP_DEVICE_COMPILE__)
using dtm::pk2;
#endif

struct Fwd3mParams {
    const float *X;      // [n0][n1][n2]
    float *LLL;          // [n0][n1][n2]
    float *Yh;           // [n0/2][n1/2][n2/2][56 floats]
    int n0, n1, n2;      // n0, n1 even, n2 % 4 == 0
    int nstrip, nrp, nchunk, chunk;      // strips along axis 2, row pairs (n1 / 2), chunks of `chunk` slices (% 8 == 0)
    // axis 1, per branch b: taps by distance d from the centre, each twice (t, t): b = 0 the lowpass h0, b = 1 the highpass
    // h1 / 2 (cube2c's 1/2: every octant of the a1 = 1 half is a highpass octant)
    float a1[2][8] __attribute__((aligned(8)));
    // axis 0: (h0, h1) pairs by distance
    float hp[8] __attribute__((aligned(8)));
    // axis 2, per branch: over the a0 = 0 plane (h0, h1) x (1, 1/2) for b = 0 -- octant (0, 0, 0) is the lowpass volume and
    // stays unscaled --, over the a0 = 1 plane (h0, h1) / 2; for b = 1 both unscaled (the 1/2 is in the axis-1 taps)
    float hpl[2][8] __attribute__((aligned(8))), hph[2][8] __attribute__((aligned(8)));
};
inline void pack_fwd3m(Fwd3mParams &p, const double *h0, int m0, const double *h1, int m1) {
    for (int d = 0; d < 4; ++d) {
        const double a = d <= m0 / 2 ? h0[m0 / 2 - d] : 0.0, b = d <= m1 / 2 ? h1[m1 / 2 - d] : 0.0;
        p.a1[0][2 * d] = p.a1[0][2 * d + 1] = (float)a;
        p.a1[1][2 * d] = p.a1[1][2 * d + 1] = (float)(0.5 * b);
        p.hp[2 * d] = (float)a; p.hp[2 * d + 1] = (float)b;
        p.hpl[0][2 * d] = (float)a; p.hpl[0][2 * d + 1] = (float)(0.5 * b);
        p.hph[0][2 * d] = (float)(0.5 * a); p.hph[0][2 * d + 1] = (float)(0.5 * b);
        p.hpl[1][2 * d] = p.hph[1][2 * d] = (float)a; p.hpl[1][2 * d + 1] = p.hph[1][2 * d + 1] = (float)b;
    }
}

// strips along axis 2: a halo lane at every INTERIOR strip boundary, none at the edges of the volume
inline int fwd3m_nstrip(int n2) {
    const int lanes = n2 / 4;
    if (lanes <= 64) return 1;
    return 2 + (lanes - 126 + 61) / 62;       // two edge strips of 63 owning lanes, interior strips of 62
}
DT_HD void fwd3m_strip(int strip, int nstrip, int n2, int &lane0_col, int &first_own, int &nown) {
    // owning lanes of the strips before this one
    const int before = strip == 0 ? 0 : 63 + (strip - 1) * 62;
    first_own = strip == 0 ? 0 : 1;
    lane0_col = 4 * (before - first_own);
    const int cap = nstrip == 1 ? 64 : ((strip == 0 || strip == nstrip - 1) ? 63 : 62);
    const int left = n2 / 4 - before;
    nown = left < cap ? left : cap;
}

#if defined(__HIP_DEVICE_COMPILE__)
#define DT3M_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// cube2c of one octant's 2 x 2 x 2 cell (transform3d.py:532-579; the 1/2 is already in the taps):
//   A = y[0,0,0] B = y[0,1,0] C = y[1,0,0] D = y[1,1,0] E = y[0,0,1] F = y[0,1,1] G = y[1,0,1] H = y[1,1,1]   (slice, row, column)
__device__ __forceinline__ void cube2c_cell(float A, float B, float C, float D, float E, float F, float G, float H, f4 &o0, f4 &o1) {
    const float amg = A - G, apg = A + G, dpf = D + F, dmf = D - F;
    const float bmh = B - H, bph = B + H, cpe = C + E, emc = E - C;
    o0 = f4{amg - dpf, bmh + cpe, amg + dpf, cpe - bmh};
    o1 = f4{apg + dmf, bph + emc, apg - dmf, emc - bph};
}
#endif

// OCC = wavefronts per SIMD the register allocation is made for.  The march holds the ring (64), two raw slices in flight
// (64), the (lo0, hi0) pairs of a slice pair (32), the four-octant outputs (64) and a 10-column window (20): OCC = 1 gets
// 256 + 20 registers and spills nothing; OCC = 2 (256) spills 24 of them to scratch.  Measured equal at 256^3
// (profiles/r05/ab_fwd3m.txt: 0.103-0.117 against 0.114-0.120 ms alone, 0.199 against 0.203 ms per forward in flight): the
// kernel is within 15 % of its 604 MB at the copy rate either way, and one wavefront per SIMD has its loads a whole slice
// pair ahead.  The launcher takes OCC = 1 (DTCWT_HIP_FWD3_OCC=2 for the other).
template <int M0, int M1, int OCC>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) k_fwd3m_l1(const Fwd3mParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(M0 <= 7 && M1 <= 7 && M0 % 2 == 1 && M1 % 2 == 1, "level-1 filters of at most 7 taps");
    constexpr int HH = 3, WR = 8;
    using dtm::dpp_from_left; using dtm::dpp_from_right; using dtm::dt_buf2g; using dtm::dt_buf_n;
    __shared__ __attribute__((aligned(16))) f4 slab[128 * 14 + 8];            // one row of cells: 64 lanes x 2 cells x 224 bytes
    const int lane = threadIdx.x & 63;
    const int br = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // 0: the a1 = 0 octants + LLL, 1: the a1 = 1 octants
    // job: XCD x takes a contiguous run of row pairs (neighbouring pairs share six of their eight raw rows)
    const int w = blockIdx.x, x = w & 7, i = w >> 3;
    const int per = (p.nrp + 7) / 8;
    const int rp = x * per + i % per, rest = i / per;
    const int strip = rest % p.nstrip, chunk = rest / p.nstrip;
    if (rp >= p.nrp || chunk >= p.nchunk) return;
    const int n0 = p.n0, n1 = p.n1, n2 = p.n2;
    int lane0_col, first_own, nown;
    fwd3m_strip(strip, p.nstrip, n2, lane0_col, first_own, nown);
    int c0 = lane0_col + 4 * lane;
    c0 = c0 < 0 ? 0 : (c0 > n2 - 4 ? n2 - 4 : c0);
    // lanes whose left / right neighbours lie beyond the volume take the mirror columns from themselves
    const bool isL = strip == 0 && lane == 0, isR = strip == p.nstrip - 1 && lane == first_own + nown - 1;

    const DtBuf bx = dt_buf2g(p.X);
    const unsigned rpitch = (unsigned)n2 * 4u;
    const int j0 = 2 * rp;
    unsigned roff[8];                    // byte offsets of the eight raw rows within a slice (reflected at the faces)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int u = j0 - HH + r;
        u = u < 0 ? -1 - u : u; u = u >= n1 ? 2 * n1 - 1 - u : u;
        roff[r] = (unsigned)u * rpitch;
    }
    const int s0 = chunk * p.chunk;
    const int ns = n0 - s0 < p.chunk ? n0 - s0 : p.chunk;          // slices of this job (even)
    const int nms = (ns / 2 + 3) / 4 * 4;                           // macro-steps, whole periods of the ring
    const int last_slice = s0 + ns - 1 + HH + 1;
    const int64_t spitch = (int64_t)n1 * n2;                        // elements per slice (< 2^31: checked by the launcher)
    auto soff = [&](int s) -> unsigned {
        s = s > last_slice ? last_slice : s;
        s = s < 0 ? -1 - s : s; s = s >= n0 ? 2 * n0 - 1 - s : s;
        return (unsigned)s * (unsigned)spitch * 4u;                 // bytes (a volume is < 2 GiB: launcher)
    };
    auto load_slice = [&](int s, f4 (&raw)[8]) {
        const unsigned so = soff(s);
#pragma unroll
        for (int r = 0; r < 8; ++r) raw[r] = dt2d::dt_buf_ld4(bx, (unsigned)c0 * 4u + roff[r], so);
    };
    // axis 1 on the raw rows of a slice -> this wavefront's half, rows j0, j0 + 1 as column pairs
    const pk2 *t1 = reinterpret_cast<const pk2 *>(p.a1[br]);
    auto axis1 = [&](const f4 (&raw)[8], pk2 (&out)[2][2]) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                auto col = [&](int r) { return h ? pk2{raw[r].z, raw[r].w} : pk2{raw[r].x, raw[r].y}; };
                pk2 a = t1[0] * col(e + HH);
#pragma unroll
                for (int d = 1; d <= HH; ++d) a += t1[d] * (col(e + HH - d) + col(e + HH + d));
                out[e][h] = a;
            }
    };

    // ring of the last eight filtered slices: ring[slot][row][column pair]
    pk2 ring[WR][2][2];
    f4 raw[2][8];
    // prologue: slices s0 - 3 .. s0 + 2 into slots 0 .. 5, then the two slices of the first macro-step in flight
#pragma unroll
    for (int q = 0; q < 6; q += 2) {
        load_slice(s0 - HH + q, raw[0]);
        load_slice(s0 - HH + q + 1, raw[1]);
        axis1(raw[0], ring[q]);
        axis1(raw[1], ring[q + 1]);
    }
    load_slice(s0 + 3, raw[0]);
    load_slice(s0 + 4, raw[1]);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 8; ++r) asm volatile("" : "+v"(raw[q][r].x), "+v"(raw[q][r].y), "+v"(raw[q][r].z), "+v"(raw[q][r].w) : : "memory");

    const pk2 *hp = reinterpret_cast<const pk2 *>(p.hp);
    const pk2 *hpl = reinterpret_cast<const pk2 *>(p.hpl[br]), *hph = reinterpret_cast<const pk2 *>(p.hph[br]);
    const int64_t rec_row = (int64_t)(n2 / 2) * 56;                  // floats per row of cells
    const int own0 = first_own;
    float *const Lb = p.LLL + (int64_t)j0 * n2 + (lane0_col + 4 * own0);
    const unsigned lv = 16u * (unsigned)(lane - own0);
    // the strip's part of a row of cells: first cell = (lane0_col + 4 own0) / 2, 2 nown cells of 224 bytes
    float *const Yb = p.Yh + (int64_t)rp * rec_row + (int64_t)((lane0_col + 4 * own0) / 2) * 56;
    const int npiece = nown * 28;                                   // 16-byte pieces of the strip's record row
    // where this lane's records sit in the slab: cell 2 (lane - own0) + k, octant slot o -> f4 index (cell * 14 + 2 o)
    const int cell0 = 2 * (lane - own0);
    const bool owns = lane >= own0 && lane < own0 + nown;

    for (int m0 = 0; m0 < nms; m0 += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = m0 + k, s = s0 + 2 * m;            // output slices s, s + 1
            // what was requested a macro-step ago joins the ring; the next two slices are requested
            axis1(raw[0], ring[(2 * k + 6) % WR]);
            axis1(raw[1], ring[(2 * k + 7) % WR]);
            load_slice(s + 5, raw[0]);
            load_slice(s + 6, raw[1]);
            const bool ok = 2 * m < ns;                      // uniform; surplus macro-steps store nothing
            // ---- axis 0: (lo0, hi0) of the four columns, rows e, slices q
            pk2 Wq[2][2][4];
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    pk2 win[WR];
#pragma unroll
                    for (int j = 0; j < WR; ++j) win[j] = ring[(2 * k + j) % WR][e][h];
#pragma unroll
                    for (int q = 0; q < 2; ++q) dtm::col_lohi2<HH>(&win[q + HH], hp, Wq[q][e][2 * h], Wq[q][e][2 * h + 1]);
                }
            // ---- axis 2 + cube2c, one a0 at a time: v[a2][q][e][c]
            pk2 ol[2][2][4], oh[2][2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    pk2 W[4 + 2 * HH];
#pragma unroll
                    for (int c = 0; c < 4; ++c) W[HH + c] = Wq[q][e][c];
#pragma unroll
                    for (int j = 0; j < HH; ++j) {
                        // columns -3 .. -1 from the left neighbour's columns 1 .. 3 -- or, at the face, the lane's own 2 .. 0
                        const pk2 dl = pk2{dpp_from_left(W[HH + 1 + j].x), dpp_from_left(W[HH + 1 + j].y)};
                        const pk2 ml = W[HH + 2 - j];
                        W[j] = pk2{isL ? ml.x : dl.x, isL ? ml.y : dl.y};
                        // columns 4 .. 6 from the right neighbour's 0 .. 2 -- or the lane's own 3 .. 1
                        const pk2 dr = pk2{dpp_from_right(W[HH + j].x), dpp_from_right(W[HH + j].y)};
                        const pk2 mr = W[HH + 3 - j];
                        W[HH + 4 + j] = pk2{isR ? mr.x : dr.x, isR ? mr.y : dr.y};
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) dtm::row_lohi_s<HH>(&W[c + HH], hpl, hph, ol[q][e][c], oh[q][e][c]);
                }
            // ---- the slab must be free: the flush of the previous macro-step has read it
            DT3M_LDS_BARRIER();
            // octant slots of the record (oracle _OCTANTS order): (a0, a1, a2) = (0,1,0) (1,0,0) (1,1,0) (0,0,1) (0,1,1) (1,0,1) (1,1,1)
            auto put_octant = [&](int slot, auto val) {
#pragma unroll
                for (int cell = 0; cell < 2; ++cell) {
                    f4 o0, o1;
                    cube2c_cell(val(0, 0, 2 * cell), val(0, 1, 2 * cell), val(1, 0, 2 * cell), val(1, 1, 2 * cell),
                                val(0, 0, 2 * cell + 1), val(0, 1, 2 * cell + 1), val(1, 0, 2 * cell + 1), val(1, 1, 2 * cell + 1), o0, o1);
                    if (owns) { slab[(cell0 + cell) * 14 + 2 * slot] = o0; slab[(cell0 + cell) * 14 + 2 * slot + 1] = o1; }
                }
            };
            if (br == 0) {
                put_octant(3, [&](int q, int e, int c) { return ol[q][e][c].y; });        // (0, 0, 1)
                put_octant(1, [&](int q, int e, int c) { return oh[q][e][c].x; });        // (1, 0, 0)
                put_octant(5, [&](int q, int e, int c) { return oh[q][e][c].y; });        // (1, 0, 1)
            } else {
                put_octant(0, [&](int q, int e, int c) { return ol[q][e][c].x; });        // (0, 1, 0)
                put_octant(4, [&](int q, int e, int c) { return ol[q][e][c].y; });        // (0, 1, 1)
                put_octant(2, [&](int q, int e, int c) { return oh[q][e][c].x; });        // (1, 1, 0)
                put_octant(6, [&](int q, int e, int c) { return oh[q][e][c].y; });        // (1, 1, 1)
            }
            // the lowpass volume: rows j0, j0 + 1 of slices s, s + 1 (every path issues the same four stores: against
            // zero bytes for wavefront 1 and for surplus macro-steps -- march2d.hpp on vmcnt)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int so = ok ? s + q : s0;
                    const DtBuf bl = dt_buf_n(Lb + (int64_t)so * spitch + (int64_t)e * n2, (ok && br == 0) ? 16u * nown : 0u);
                    dt2d::dt_buf_st4<false>(bl, lv, 0u, f4{ol[q][e][0].x, ol[q][e][1].x, ol[q][e][2].x, ol[q][e][3].x});
                }
            DT3M_LDS_BARRIER();
            // ---- flush: the strip's row of cells as one run of 16-byte pieces, half to each wavefront
            {
                const int so = ok ? s / 2 : s0 / 2;
                const DtBuf by = dt_buf_n(Yb + (int64_t)so * (n1 / 2) * rec_row, ok ? 16u * npiece : 0u);
#pragma unroll
                for (int t = 0; t < 14; ++t) {
                    const int piece = lane + 64 * (2 * t + br);
                    const f4 v = slab[piece];
                    dt2d::dt_buf_st4<true>(by, 16u * (unsigned)piece, 0u, v);
                }
            }
        }
    }
#endif
}

}  // namespace dt3m
