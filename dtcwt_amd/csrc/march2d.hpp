// Marching wavefront programs of the float32 2-D DT-CWT (gfx950): levels 1 + 2 of the forward transform (k_fwd12m) and
// levels 2 + 1 of the inverse (k_inv21m) as ONE launch each, the level-1 lowpass never leaving the registers.
//
// The tile programs of fused2d_tiles_v2.hpp hand the column pass to the row pass through LDS planes and workgroup
// barriers, one launch per level.  Here ONE WAVEFRONT is the unit of work and nothing is shared between wavefronts:
//
//   * a wavefront owns a strip of 64 x 4 = 256 adjacent columns (lane l holds columns 4l .. 4l+3 of the strip
//     as one float4: a row of the strip is ONE 1 KiB load) and marches down a band of rows;
//   * the column filters run over a register ring of the last rows (static indices: the march loop is unrolled
//     over one period of the ring) or, transposed, into pending sums -- no LDS, no halo re-computation across rows;
//   * the row filters take the neighbouring columns from the neighbouring LANES with DPP wave shifts
//     (v_mov_b32_dpp wave_shr:1 / wave_shl:1): the first and the last HL lanes of a strip are halo lanes;
//   * q2c / c2q are lane-local (a lane owns whole 2 x 2 quads), the 48-byte subband records travel through a
//     wave-private LDS slab so that every global access is a run of consecutive 16-byte pieces;
//   * rows are requested ahead of their use; there is no barrier anywhere, so the load, FMA and store streams of
//     the wavefronts of a CU overlap freely.
// What the hardware taught on the way (data hazard of 16-byte buffer stores, vmcnt per path, one instruction per four
// cycles per wavefront, true-pair packing) is written where it matters below and in DESIGN.md section 4 "Round 4".
// The first prototypes (level 1 alone, a pair of wavefronts per job) are in tools/kbench/march_experiments.hpp.
//
// Reference: dtcwt/numpy/transform2d.py:112-160 (forward levels 1, 2), :242-293 (inverse levels 2, 1), q2c :301-322,
// c2q :324-350; colfilter / coldfilt / colifilt dtcwt/numpy/lowlevel.py:47-80, :82-154, :156-260.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "fused2d_tiles.hpp"

namespace dtm {

using dt2d::DtBuf;
using dt2d::dt_buf;
using dt2d::f4;

#if defined(__HIP_DEVICE_COMPILE__)
// A raw buffer over [base, base + 2 GiB) for the row loads: the row offset rides in the scalar offset field, the
// lane's column in the vector offset (loads have no hazard with a register there; an image is < 2 GiB: march2d.hip).
constexpr unsigned OOB = 0x80000000u;
__device__ __forceinline__ DtBuf dt_buf2g(const void *base) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    void *ub = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
    DtBuf b; b.r = __builtin_amdgcn_make_buffer_rsrc(ub, 0, (int)OOB, 0x00020000); return b;
}
// A raw buffer over [base, base + bytes): offsets >= bytes are dropped by the hardware.  The march re-makes the store
// descriptors per row (scalar adds), so that a store needs no vector address arithmetic, no exec mask for the lanes
// that own nothing, and -- with the scalar offset field left at 0 -- the compiler sees the "VALU overwrites the data
// of a > 8-byte store in the next cycle" hazard: with a REGISTER in the soffset field LLVM assumes the hazard does not
// exist, and on gfx950 it does (lanes 12-15 of every 16 stored the next instruction's result).
__device__ __forceinline__ DtBuf dt_buf_n(const void *base, unsigned bytes) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    void *ub = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
    DtBuf b; b.r = __builtin_amdgcn_make_buffer_rsrc(ub, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000); return b;
}
// lane l receives lane l-1's value (lane 0: 0) / lane l+1's value (lane 63: 0)
__device__ __forceinline__ float dpp_from_left(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_right(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
#endif

// Which (strip, band, image) a workgroup marches.  Workgroup w runs on XCD w % 8 (observed, for speed only), and each
// XCD has its own L2: neighbouring strips of a band read each other's halo columns at the same moment (they march in
// step), so a band's strips go to ONE XCD -- in `ngrp` groups of `gstrips` strips when a band alone would leave the
// XCDs unevenly loaded (a 4096^2 image in 28 bands: 56 half-bands = 7 per XCD).  Unit u = (image, band, group) sits on
// XCD u % 8; surplus workgroups (padding of the last group, of the last round of units) leave at once.
struct MarchJobs {
    int nstrip, nband, band_rows;
    int ngrp, gstrips;            // groups per band, strips per group (ngrp * gstrips >= nstrip)
    int nunit;                    // B * nband * ngrp
};
inline unsigned dtm_set_jobs(MarchJobs &j, int B, int R, int nstrip, int band_rows) {
    j.nstrip = nstrip; j.band_rows = band_rows; j.nband = (R + band_rows - 1) / band_rows;
    j.ngrp = nstrip >= 14 ? 2 : 1;
    j.gstrips = (nstrip + j.ngrp - 1) / j.ngrp;
    j.nunit = B * j.nband * j.ngrp;
    return (unsigned)(((j.nunit + 7) / 8) * 8 * j.gstrips);         // workgroups
}
DT_HD bool dtm_job(const MarchJobs &j, int w, int &strip, int &band, int &b) {
    const int x = w & 7, i = w >> 3;
    const int u = (i / j.gstrips) * 8 + x, sidx = i % j.gstrips;
    const int g = u % j.ngrp, ub = u / j.ngrp;
    strip = g * j.gstrips + sidx; band = ub % j.nband; b = ub / j.nband;
    return u < j.nunit && strip < j.nstrip;
}

constexpr int MAXT1 = 9;        // longest level-1 filter a one-lane halo serves (halo <= 4 columns)

#if defined(__HIP_DEVICE_COMPILE__)

__device__ __forceinline__ f4 rev4(const f4 &v) { return f4{v.w, v.z, v.y, v.x}; }

// the lane's four samples of a plane row + HH columns either side (from the neighbouring lanes)
template <int HH, bool NODPP = false>
__device__ __forceinline__ void row_window(const f4 &v, float (&w)[4 + 2 * HH]) {
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < HH; ++j) w[j] = NODPP ? e[4 - HH + j] : dpp_from_left(e[4 - HH + j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) w[HH + j] = e[j];
#pragma unroll
    for (int j = 0; j < HH; ++j) w[HH + 4 + j] = NODPP ? e[j] : dpp_from_right(e[j]);
}

// A symmetric filter (h[k] = h[M-1-k]: every biort set) at one column, the mirror pairs added first: the lowpass on
// the warm-up rows, where level 1 only has to feed level 2 (xc = centre of the window)
template <int MA, int HH>
__device__ __forceinline__ float sym_one(const float *xc, const float *a) {
    constexpr int HA = MA / 2;
    float ra = a[HA] * xc[0];
#pragma unroll
    for (int d = 1; d <= HA; ++d) ra += a[HA - d] * (xc[-d] + xc[d]);
    return ra;
}

// ---- hand-packed forms ------------------------------------------------------------------------------------------------
// A wavefront issues at most one vector instruction every four cycles, and these kernels run one or two wavefronts to
// a SIMD (their register needs; a 4096^2 image is only ~1900 jobs): they are bound by the number of instructions a
// wavefront issues (profiles/r04/pmc_march.txt: VALU active = instructions x 4 cycles = 60 % of the kernel), so two
// multiply-adds per v_pk_fma_f32 halve what matters.  The pairs are chosen so that no operand has to be moved into
// place: (lowpass, highpass) of the column pass on the same sample, (Lo plane, Hi plane) of the row pass with the same tap.
typedef float pk2 __attribute__((ext_vector_type(2)));
// Operands of a packed instruction are 64-bit register pairs.  A scalar broadcast to both halves occupies a whole pair
// if it lives in a register of its own, and a broadcast of one half of a SCALAR-register pair is materialised as a new
// pair (s_mov) -- both were tried and spilled.  What costs nothing is broadcasting one half of a VECTOR-register pair
// (op_sel).  So: taps come as true (lowpass, highpass) pairs in scalar registers, samples live as true pairs of
// neighbours in vector registers (two columns of a row / (Lo, Hi) of a column / two samples of a window) and are
// broadcast by half.
#define DTM_BX(v_) pk2{(v_).x, (v_).x}
#define DTM_BY(v_) pk2{(v_).y, (v_).y}
// Column pass, two neighbouring columns: x[j] = their samples in window row j (xc = centre); hp[d] = (h0, h1) at
// distance d.  (lo, hi) of the first and of the second column.
template <int HH>
__device__ __forceinline__ void col_lohi2(const pk2 *xc, const pk2 *hp, pk2 &c0, pk2 &c1) {
    pk2 a = hp[0] * DTM_BX(xc[0]), b = hp[0] * DTM_BY(xc[0]);
#pragma unroll
    for (int d = 1; d <= HH; ++d) { const pk2 sm = xc[-d] + xc[d]; a += hp[d] * DTM_BX(sm); b += hp[d] * DTM_BY(sm); }
    c0 = a; c1 = b;
}
// Row pass at one column over the interleaved window W[i] = (Lo, Hi) of column i (wc_ = centre):
// (ll, lh) = (h0, h1) * Lo row, (hl, hh) = (h0, h1) * Hi row
template <int HH>
__device__ __forceinline__ void row_lohi(const pk2 *wc_, const pk2 *hp, pk2 &ol, pk2 &oh) {
    pk2 a = hp[0] * DTM_BX(wc_[0]), b = hp[0] * DTM_BY(wc_[0]);
#pragma unroll
    for (int d = 1; d <= HH; ++d) { const pk2 sm = wc_[-d] + wc_[d]; a += hp[d] * DTM_BX(sm); b += hp[d] * DTM_BY(sm); }
    ol = a; oh = b;
}
// the same with the 1/sqrt2 of q2c folded into the taps: hpl = (h0, h1 / sqrt2) over the Lo row -> (ll, lh / sqrt2),
// hph = (h0, h1) / sqrt2 over the Hi row -> (hl, hh) / sqrt2; 24 multiplications per step less
template <int HH>
__device__ __forceinline__ void row_lohi_s(const pk2 *wc_, const pk2 *hpl, const pk2 *hph, pk2 &ol, pk2 &oh) {
    pk2 a = hpl[0] * DTM_BX(wc_[0]), b = hph[0] * DTM_BY(wc_[0]);
#pragma unroll
    for (int d = 1; d <= HH; ++d) { const pk2 sm = wc_[-d] + wc_[d]; a += hpl[d] * DTM_BX(sm); b += hph[d] * DTM_BY(sm); }
    ol = a; oh = b;
}

// q2c of the lane's two quads of a plane (rows e0 / e1; the 1/sqrt2 is already in the row taps)
//   z0 = (a - d) + j(b + c), z1 = (a + d) + j(b - c)  for  a b / c d
struct Zq { float z0r, z0i, z1r, z1i; };
__device__ __forceinline__ Zq q2c_s(float a, float b, float c, float d) { return Zq{a - d, b + c, a + d, b - c}; }
// the same with every result pinned as a scalar: four v_add / v_sub that write straight into their places in the record.
// Left alone, the compiler pairs them into v_pk_add_f32 (operands swapped and negated by modifiers) and then needs a v_mov
// per component to lay the record out for its 16-byte LDS write -- more instructions than it saved.
__device__ __forceinline__ Zq q2c_p(float a, float b, float c, float d) {
    Zq z{a - d, b + c, a + d, b - c};
    asm("" : "+v"(z.z0r)); asm("" : "+v"(z.z0i)); asm("" : "+v"(z.z1r)); asm("" : "+v"(z.z1i));
    return z;
}

#endif  // __HIP_DEVICE_COMPILE__

// ======================================================================================================================
// Levels 1 + 2 of the forward transform in one march (transform2d.py:112-160).
//
// The lowpass rows of level 1 never leave the registers: as the march produces them (two per step) they go through the
// level-2 row filters (the 2M-sample window of output pair j = the lane's own four columns + HL2 lanes either side,
// DPP again) and are then SCATTERED down the columns: a row adds into the pending sums of the M/2 output row pairs
// whose windows contain it (coldfilt in transposed form, lowlevel.py:82-154), so the level-2 column window costs
// M/2 x 16 accumulators per lane instead of a ring of 2M rows.  Every second step the oldest pair is complete: q2c,
// one 48-byte record per lane through the slab, two 8-byte LoLo2 stores.
// Traffic per pixel of X: 4 B in (x 1 + (2M + 2) / band_rows for the warm-up rows, which neighbouring bands fetch as
// well -- mostly from the L2 / Infinity Cache), 12 + 3 + 1 B out; the 4 + 4 B of LoLo1 that the two separate launches
// write and read back are gone.
// The level-1 filters must be symmetric (every biort set is): the halo lanes beyond the image edge hold LoLo1 computed
// from mirrored X, which equals the mirrored LoLo1 that coldfilt's symmetric extension asks for only then.
// ======================================================================================================================
constexpr int MAXT2 = 20;

struct Fwd12mParams {
    const float *X;       // [B][R][C]
    float *Yh0;           // [B][R/2][C/2][12]
    float *Yh1;           // [B][R/4][C/4][12]
    float *LoLo2;         // [B][R/2][C/2]
    float *LoLo1;         // [B][R][C]: the level-1 lowpass, written only by the `scales` variant (KO bit 128), else unused
    int B, R, C;          // R % 4 == 0, C % 4 == 0
    MarchJobs jb;                     // strips of VL lanes, bands of jb.band_rows rows (% 4 == 0)
    int lo_a_first, hi_a_first;       // sign of sum(ha * hb) of the lowpass / highpass q-shift pair (lowlevel.py:143)
    float h0[MAXT1], h1[MAXT1];
    // level-2 taps by window offset: A = sum_t ta[t] w[2t], B = sum_t tb[t] w[2t + 1] over the 2M-sample window whose
    // element j is logical sample 4i - M + 2 + j (fused2d_tiles.hpp: dfilt_pair); dtm_pack_qshift() fills them
    float ta_lo[MAXT2], tb_lo[MAXT2], ta_hi[MAXT2], tb_hi[MAXT2];
    // the same as (lowpass, highpass) pairs: one v_pk_fma_f32 with the sample broadcast does both filters of a window
    // position -- and the level-1 taps by distance d from the centre as (h0, h1) pairs (h0 zero beyond its half length)
    float ta2[2 * MAXT2] __attribute__((aligned(8))), tb2[2 * MAXT2] __attribute__((aligned(8)));
    float hp[2 * (MAXT1 / 2 + 1)] __attribute__((aligned(8)));
    // the row-pass taps with the 1/sqrt2 of q2c folded in: over the Lo plane (h0, h1 / sqrt2), over the Hi plane (h0, h1) / sqrt2
    float hpl[2 * (MAXT1 / 2 + 1)] __attribute__((aligned(8))), hph[2 * (MAXT1 / 2 + 1)] __attribute__((aligned(8)));
};

// ha / hb: the FIRST / SECOND filter argument of coldfilt (Fwd2Params::l_a, l_b and h_a, h_b)
template <class Pm>
inline void dtm_pack_qshift(Pm &p, int M, const float *l_a, const float *l_b, const float *h_a, const float *h_b) {
    for (int t = 0; t < MAXT2; ++t) p.ta_lo[t] = p.tb_lo[t] = p.ta_hi[t] = p.tb_hi[t] = 0.f;
    for (int k = 0; k < M / 2; ++k) {
        // A: ha[2k] w[2M-2-4k] + ha[2k+1] w[2M-4-4k];  B: hb[2k] w[2M-1-4k] + hb[2k+1] w[2M-3-4k]
        p.ta_lo[(2 * M - 2 - 4 * k) / 2] = l_a[2 * k]; p.ta_lo[(2 * M - 4 - 4 * k) / 2] = l_a[2 * k + 1];
        p.tb_lo[(2 * M - 2 - 4 * k) / 2] = l_b[2 * k]; p.tb_lo[(2 * M - 4 - 4 * k) / 2] = l_b[2 * k + 1];
        p.ta_hi[(2 * M - 2 - 4 * k) / 2] = h_a[2 * k]; p.ta_hi[(2 * M - 4 - 4 * k) / 2] = h_a[2 * k + 1];
        p.tb_hi[(2 * M - 2 - 4 * k) / 2] = h_b[2 * k]; p.tb_hi[(2 * M - 4 - 4 * k) / 2] = h_b[2 * k + 1];
    }
    for (int t = 0; t < MAXT2; ++t) {
        p.ta2[2 * t] = p.ta_lo[t]; p.ta2[2 * t + 1] = p.ta_hi[t];
        p.tb2[2 * t] = p.tb_lo[t]; p.tb2[2 * t + 1] = p.tb_hi[t];
    }
}
// (h0, h1) by distance from the centre tap; m0, m1: the (odd) lengths
template <class Pm>
inline void dtm_pack_biort(Pm &p, int m0, int m1) {
    for (int d = 0; d <= MAXT1 / 2; ++d) {
        p.hp[2 * d] = d <= m0 / 2 ? p.h0[m0 / 2 - d] : 0.f;
        p.hp[2 * d + 1] = d <= m1 / 2 ? p.h1[m1 / 2 - d] : 0.f;
    }
}
// h0d / h1d: the level-1 taps in double precision (the products with 1/sqrt2 are rounded once)
template <class Pm>
inline void dtm_pack_biort_scaled(Pm &p, int m0, int m1, const double *h0d, const double *h1d) {
    const double rs = 0.70710678118654752440;
    for (int d = 0; d <= MAXT1 / 2; ++d) {
        const double a = d <= m0 / 2 ? h0d[m0 / 2 - d] : 0.0, b = d <= m1 / 2 ? h1d[m1 / 2 - d] : 0.0;
        p.hpl[2 * d] = (float)a; p.hpl[2 * d + 1] = (float)(b * rs);
        p.hph[2 * d] = (float)(a * rs); p.hph[2 * d + 1] = (float)(b * rs);
    }
}

template <int M0, int M1, int M>
struct Fwd12m {
    static constexpr int H0 = M0 / 2, H1 = M1 / 2;
    // the window is always 2 x 3 + 2 rows: the register ring of the march has a period of WR / 2 = 4 steps, which is
    // what the loop is unrolled by (shorter filters meet zero taps, longer ones stay with the tile programs)
    static constexpr int HH = 3;
    static constexpr int HL1 = 1, HL2 = (M - 2) / 4, HL = HL1 + HL2;
    static constexpr int VL = 64 - 2 * HL;
    static constexpr int WR = 2 * HH + 2;
    static constexpr int NP2 = M / 2;             // pending level-2 row pairs
    static constexpr int PRE = M - 2;             // LoLo1 rows a band needs above its first row (and PRE + 1... below)
    static_assert(H0 <= HH && H1 <= HH && (M - 2) % 4 == 0 && M <= MAXT2, "filters the marching forward is built for");
};

template <int M0, int M1, int M, int P, int KO, int WPS = 2>
__global__ void __launch_bounds__(64, WPS) k_fwd12m(const Fwd12mParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = Fwd12m<M0, M1, M>;
    constexpr int HH = G::HH, WR = G::WR, HL = G::HL, HL2 = G::HL2, VL = G::VL, NP2 = G::NP2, PER = 4, H0_ = G::H0;
    static_assert(PER % P == 0, "prefetch depth divides the ring period");
    __shared__ __attribute__((aligned(16))) f4 slab[64 * 6 + 6 * G::HL + 8];
    __shared__ __attribute__((aligned(16))) f4 slab2[64 * 3 + 3 * G::HL + 8];
    const int lane = threadIdx.x;
    int strip, band, b;
    if (!dtm_job(p.jb, blockIdx.x, strip, band, b)) return;
    const int R = p.R, C = p.C;

    const int c0 = strip * (4 * VL) - 4 * HL + 4 * lane;
    const bool rev = c0 < 0 || c0 >= C;
    int lc = c0 < 0 ? -c0 - 4 : (c0 >= C ? 2 * C - 4 - c0 : c0);
    lc = lc < 0 ? 0 : (lc > C - 4 ? C - 4 : lc);
    const bool edge_strip = strip == 0 || (strip + 1) * (4 * VL) + 4 * HL >= C;
    const int nv = (C - strip * (4 * VL)) / 4 < VL ? (C - strip * (4 * VL)) / 4 : VL;     // owning lanes

    const int64_t img = (int64_t)b * R * C;
    const DtBuf bx = dt_buf2g(p.X + img);
    float *const Y0b = p.Yh0 + img * 3 + (int64_t)strip * (VL * 24);
    float *const Y1b = p.Yh1 + (img / 4) * 3 + (int64_t)strip * (VL * 12);
    float *const L2b = p.LoLo2 + img / 4 + strip * (VL * 2);
    const unsigned pitch = (unsigned)C * 4u;

    const int rb = band * p.jb.band_rows;
    const int nrow = R - rb < p.jb.band_rows ? R - rb : p.jb.band_rows;
    const int rbase = rb - G::PRE;                     // first LoLo1 row the band's level-2 windows want
    // steps of two rows: rbase .. rb + nrow + PRE + 1, rounded up to whole periods of the register rings (the march
    // loop has ONE exit: with a way out after every step the compiler copies the loop-carried rows -- loads still in
    // flight among them -- into place on the way back, which waits for everything outstanding once per period); the
    // surplus steps re-read the last row, own no output row and complete no pair of the band
    const int nst = (nrow / 2 + G::PRE + PER - 1) / PER * PER;
    const int last_row = rbase + 2 * (nrow / 2 + G::PRE) - 1 + HH;

    auto ldrow = [&](int u) -> f4 {
        u = u > last_row ? last_row : u;
        u = u < 0 ? -1 - u : u;
        u = u >= R ? 2 * R - 1 - u : u;
        if (KO & 1) u &= 15;
        if (KO & 32) { const dt2d::dt_bv4 v = __builtin_amdgcn_raw_buffer_load_b128(bx.r, (unsigned)lc * 4u, (unsigned)u * pitch, 2); return f4{v.x, v.y, v.z, v.w}; }
        return dt2d::dt_buf_ld4(bx, (unsigned)lc * 4u, (unsigned)u * pitch);
    };
    auto fix = [&](f4 &v) { if (edge_strip) v = rev ? rev4(v) : v; };

    float h0[M0], h1[M1];
#pragma unroll
    for (int k = 0; k < M0; ++k) h0[k] = p.h0[k];
#pragma unroll
    for (int k = 0; k < M1; ++k) h1[k] = p.h1[k];
    const float sq = 0.70710678118654752440f;

    // window rows rho0 - HH .. rho0 + HH + 1 of the step in `ring`, the next 2P rows on their way in `pre`
    f4 ring[WR], pre[2 * P];
#pragma unroll
    for (int i = 0; i < WR; ++i) ring[i] = ldrow(rbase - HH + i);
#pragma unroll
    for (int i = 0; i < 2 * P; ++i) pre[i] = ldrow(rbase - HH + WR + i);
#pragma unroll
    for (int i = 0; i < WR; ++i) asm volatile("" : "+v"(ring[i].x), "+v"(ring[i].y), "+v"(ring[i].z), "+v"(ring[i].w) : : "memory");
#pragma unroll
    for (int i = 0; i < 2 * P; ++i) asm volatile("" : "+v"(pre[i].x), "+v"(pre[i].y), "+v"(pre[i].z), "+v"(pre[i].w) : : "memory");
#pragma unroll
    for (int i = 0; i < WR; ++i) fix(ring[i]);

    // pending level-2 pairs, oldest first: S[slot][which][v], which = A lo, A hi, B lo, B hi (column filter),
    // v = row-pass value L_A, L_B, H_A, H_B
    pk2 S2[NP2][2][4];
#pragma unroll
    for (int a = 0; a < NP2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) S2[a][c][v] = pk2{0.f, 0.f};

    const unsigned yv = 16u * (unsigned)lane;
    const unsigned l2v = 8u * (unsigned)(lane - HL);

    for (int t0 = 0; t0 < nst; t0 += PER) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int t = t0 + k;
            const int r = rbase + 2 * t;               // rho0
            // rows leaving `pre` for the window of the next step; their slots take the loads of rows 2P further down
            const f4 in0 = pre[(2 * k) % (2 * P)], in1 = pre[(2 * k + 1) % (2 * P)];
            pre[(2 * k) % (2 * P)] = ldrow(r - HH + WR + 2 * P);
            pre[(2 * k + 1) % (2 * P)] = ldrow(r - HH + WR + 2 * P + 1);
            f4 w[WR];
#pragma unroll
            for (int j = 0; j < WR; ++j) w[j] = ring[(2 * k + j) % WR];

            const bool in_band = r >= rb && r < rb + nrow;          // uniform
            // the window as column pairs: wp[0][j] = columns (0, 1) of window row j, wp[1][j] = columns (2, 3)
            pk2 wp[2][WR];
#pragma unroll
            for (int j = 0; j < WR; ++j) { wp[0][j] = pk2{w[j].x, w[j].y}; wp[1][j] = pk2{w[j].z, w[j].w}; }
            const pk2 *hpp = reinterpret_cast<const pk2 *>(p.hp);
            const pk2 *hpl = reinterpret_cast<const pk2 *>(p.hpl), *hph = reinterpret_cast<const pk2 *>(p.hph);
            f4 ll[2];
            f4 pv[6];                                   // (KO & 64) planar pyramid: a lane's two coefficients of each subband
            if (in_band) {
                f4 lh[2], hl[2], hh[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    // W[i] = (Lo, Hi) of column i - HH: own columns from the column pass, the rest from the neighbours
                    pk2 W[4 + 2 * HH];
                    col_lohi2<HH>(&wp[0][q + HH], hpp, W[HH], W[HH + 1]);
                    col_lohi2<HH>(&wp[1][q + HH], hpp, W[HH + 2], W[HH + 3]);
#pragma unroll
                    for (int j = 0; j < HH; ++j) {
                        W[j] = pk2{dpp_from_left(W[4 + j].x), dpp_from_left(W[4 + j].y)};
                        W[HH + 4 + j] = pk2{dpp_from_right(W[HH + j].x), dpp_from_right(W[HH + j].y)};
                    }
                    pk2 ol[4], oh[4];           // (ll, lh) and (hl, hh) of the four columns
#pragma unroll
                    for (int c = 0; c < 4; ++c) row_lohi_s<HH>(&W[c + HH], hpl, hph, ol[c], oh[c]);
                    ll[q] = f4{ol[0].x, ol[1].x, ol[2].x, ol[3].x}; lh[q] = f4{ol[0].y, ol[1].y, ol[2].y, ol[3].y};
                    hl[q] = f4{oh[0].x, oh[1].x, oh[2].x, oh[3].x}; hh[q] = f4{oh[0].y, oh[1].y, oh[2].y, oh[3].y};
                }
                {
                    const Zq a0 = q2c_p(hl[0].x, hl[0].y, hl[1].x, hl[1].y), a1 = q2c_p(hl[0].z, hl[0].w, hl[1].z, hl[1].w);
                    const Zq b0 = q2c_p(hh[0].x, hh[0].y, hh[1].x, hh[1].y), b1 = q2c_p(hh[0].z, hh[0].w, hh[1].z, hh[1].w);
                    const Zq c0q = q2c_p(lh[0].x, lh[0].y, lh[1].x, lh[1].y), c1q = q2c_p(lh[0].z, lh[0].w, lh[1].z, lh[1].w);
                    if constexpr ((KO & 64) != 0) {
                        pv[0] = f4{a0.z0r, a0.z0i, a1.z0r, a1.z0i}; pv[1] = f4{b0.z0r, b0.z0i, b1.z0r, b1.z0i};
                        pv[2] = f4{c0q.z0r, c0q.z0i, c1q.z0r, c1q.z0i}; pv[3] = f4{c0q.z1r, c0q.z1i, c1q.z1r, c1q.z1i};
                        pv[4] = f4{b0.z1r, b0.z1i, b1.z1r, b1.z1i}; pv[5] = f4{a0.z1r, a0.z1i, a1.z1r, a1.z1i};
                    }
                    f4 *o = slab + lane * 6;
                    if constexpr ((KO & 64) == 0) {
                    o[0] = f4{a0.z0r, a0.z0i, b0.z0r, b0.z0i};
                    o[1] = f4{c0q.z0r, c0q.z0i, c0q.z1r, c0q.z1i};
                    o[2] = f4{b0.z1r, b0.z1i, a0.z1r, a0.z1i};
                    o[3] = f4{a1.z0r, a1.z0i, b1.z0r, b1.z0i};
                    o[4] = f4{c1q.z0r, c1q.z0i, c1q.z1r, c1q.z1i};
                    o[5] = f4{b1.z1r, b1.z1i, a1.z1r, a1.z1i};
                    }
                }
            } else {
                if constexpr ((KO & 64) != 0) {
#pragma unroll
                    for (int m = 0; m < 6; ++m) pv[m] = f4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float lo[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float xs[2 * H0_ + 1];
#pragma unroll
                        for (int j = 0; j < 2 * H0_ + 1; ++j) {
                            const f4 &x = w[q + HH - H0_ + j];
                            xs[j] = c == 0 ? x.x : (c == 1 ? x.y : (c == 2 ? x.z : x.w));
                        }
                        lo[c] = sym_one<M0, HH>(&xs[H0_], h0);
                    }
                    float wl[4 + 2 * HH], a_[4];
                    row_window<HH>(f4{lo[0], lo[1], lo[2], lo[3]}, wl);
#pragma unroll
                    for (int c = 0; c < 4; ++c) a_[c] = sym_one<M0, HH>(&wl[c + HH], h0);
                    ll[q] = f4{a_[0], a_[1], a_[2], a_[3]};
                }
            }
            // The record stores are issued on EVERY step -- for rows outside the band against a descriptor of zero
            // bytes, which drops them whole (and in order: tools/kbench/vmcnt_probe) -- so that every path through a
            // step carries the same number of memory operations.  The compiler counts them per path and, where paths
            // meet, assumes the fewest: with the stores inside the branch each use of a prefetched row waited for all
            // but the youngest 4 operations, i.e. for the stores of the step before.
            if constexpr ((KO & 64) != 0) {
                // experiment (tools/kbench/march_bench): the pyramid as six planes [6][R/2][C/2] of complex64 -- a lane
                // stores its own two coefficients of each subband, nothing goes through the slab
                const int ro = (KO & 2) ? (r & 15) : r;
                float *const Y0p = p.Yh0 + img * 3 + strip * (VL * 4) + (int64_t)(ro >> 1) * C;
#pragma unroll
                for (int m = 0; m < 6; ++m) {
                    const DtBuf by = dt_buf_n(Y0p + (int64_t)m * (R / 2) * C, in_band ? 16u * nv : 0u);
                    dt2d::dt_buf_st4<true>(by, 16u * (unsigned)(lane - HL), 0u, pv[m]);
                }
            } else {
                const int ro = (KO & 2) ? (r & 15) : r;
                DT_WAVE_LDS_SYNC();
                const DtBuf by = dt_buf_n(Y0b + (int64_t)(ro >> 1) * C * 6, in_band ? 96u * nv : 0u);
#pragma unroll
                for (int m = 0; m < 6; ++m) {
                    const f4 v = slab[6 * HL + lane + 64 * m];
                    if (KO & 16) dt2d::dt_buf_st4<false>(by, yv + 1024u * m, 0u, v); else dt2d::dt_buf_st4<true>(by, yv + 1024u * m, 0u, v);
                }
                DT_WAVE_LDS_SYNC();
            }

            if constexpr ((KO & 128) != 0) {
                // `scales` (include_scale): the level-1 lowpass is an output after all -- two 16-byte stores per step, on
                // every step like the records (4 B/px more than the plain launch moves; still one launch instead of two)
                float *const L1b = p.LoLo1 + img + strip * (4 * VL);
                const int ro = in_band ? r : rb;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const DtBuf b1 = dt_buf_n(L1b + (int64_t)(ro + q) * C, in_band ? 16u * nv : 0u);
                    dt2d::dt_buf_st4<false>(b1, 16u * (unsigned)(lane - HL), 0u, ll[q]);
                }
            }
            // ---- level 2: the two LoLo1 rows of this step ----
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                // the 2M-sample window as M pairs of neighbouring samples: (A sample, B sample) of window position t
                pk2 w2[M];
                {
                    float cl[4] = {ll[q].x, ll[q].y, ll[q].z, ll[q].w}, cr[4] = {ll[q].x, ll[q].y, ll[q].z, ll[q].w};
                    w2[2 * HL2] = pk2{cl[0], cl[1]}; w2[2 * HL2 + 1] = pk2{cl[2], cl[3]};
#pragma unroll
                    for (int d = 1; d <= HL2; ++d) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) { cl[c] = dpp_from_left(cl[c]); cr[c] = dpp_from_right(cr[c]); }
                        w2[2 * (HL2 - d)] = pk2{cl[0], cl[1]}; w2[2 * (HL2 - d) + 1] = pk2{cl[2], cl[3]};
                        w2[2 * (HL2 + d)] = pk2{cr[0], cr[1]}; w2[2 * (HL2 + d) + 1] = pk2{cr[2], cr[3]};
                    }
                }
                const pk2 *ta2 = reinterpret_cast<const pk2 *>(p.ta2), *tb2 = reinterpret_cast<const pk2 *>(p.tb2);
                pk2 rA = {0.f, 0.f}, rB = {0.f, 0.f};          // (L_A, H_A), (L_B, H_B)
#pragma unroll
                for (int tt = 0; tt < M; ++tt) { rA += ta2[tt] * DTM_BX(w2[tt]); rB += tb2[tt] * DTM_BY(w2[tt]); }
                // down the columns: a row of phase phi is an A row (phi even) or a B row of the pairs it reaches;
                // S2[slot][A / B][v] = (column lowpass, column highpass) of row value v = L_A, L_B, H_A, H_B
                const int phi = 2 * (k & 1) + q;           // row 4n + phi of its group (rbase % 4 == 0)
#pragma unroll
                for (int a = 0; a < NP2; ++a) {
                    const int tt = (phi + 8 * HL2 - 4 * a) >> 1;
                    const pk2 cc = (phi & 1) ? tb2[tt] : ta2[tt];
                    if (phi < 2) {
                        // rows 4n, 4n + 1 are the first to touch the A / B halves after pair n - HL2 - 1 has left: slot a
                        // continues what slot a + 1 held, the last slot starts afresh -- the pairs move up without a v_mov
                        // (v_pk_fma_f32 has a destination of its own)
                        if (a + 1 < NP2) {
                            S2[a][phi & 1][0] = cc * DTM_BX(rA) + S2[a + 1][phi & 1][0]; S2[a][phi & 1][1] = cc * DTM_BX(rB) + S2[a + 1][phi & 1][1];
                            S2[a][phi & 1][2] = cc * DTM_BY(rA) + S2[a + 1][phi & 1][2]; S2[a][phi & 1][3] = cc * DTM_BY(rB) + S2[a + 1][phi & 1][3];
                        } else {
                            S2[a][phi & 1][0] = cc * DTM_BX(rA); S2[a][phi & 1][1] = cc * DTM_BX(rB);
                            S2[a][phi & 1][2] = cc * DTM_BY(rA); S2[a][phi & 1][3] = cc * DTM_BY(rB);
                        }
                    } else {
                        S2[a][phi & 1][0] += cc * DTM_BX(rA); S2[a][phi & 1][1] += cc * DTM_BX(rB);
                        S2[a][phi & 1][2] += cc * DTM_BY(rA); S2[a][phi & 1][3] += cc * DTM_BY(rB);
                    }
                }
            }
            if (k & 1) {
                // rows 4n + 2, 4n + 3 are in: pair i = n - HL2 is complete
                const int i2 = (r - 2) / 4 - HL2;
                const bool pair_ok = 4 * i2 >= rb && 4 * i2 < rb + nrow;        // uniform; otherwise the stores are dropped
                {
                    // sum(ha * hb) > 0 for the lowpass pair, < 0 for the highpass pair: every shipped q-shift set
                    // (dtcwt_march_fwd12_ok sends anything else to the tile programs)
                    constexpr bool la = true, ha = false;
                    // column-lowpass plane rows (A lo / B lo), column-highpass plane rows (A hi / B hi)
                    float pl[2][4], ph[2][4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        pl[0][v] = la ? S2[0][0][v].x : S2[0][1][v].x; pl[1][v] = la ? S2[0][1][v].x : S2[0][0][v].x;
                        ph[0][v] = ha ? S2[0][0][v].y : S2[0][1][v].y; ph[1][v] = ha ? S2[0][1][v].y : S2[0][0][v].y;
                    }
                    float llo[2][2], lh2[2][2], hl2[2][2], hh2[2][2];
#pragma unroll
                    for (int er = 0; er < 2; ++er) {
                        llo[er][0] = la ? pl[er][0] : pl[er][1]; llo[er][1] = la ? pl[er][1] : pl[er][0];
                        lh2[er][0] = ha ? pl[er][2] : pl[er][3]; lh2[er][1] = ha ? pl[er][3] : pl[er][2];
                        hl2[er][0] = la ? ph[er][0] : ph[er][1]; hl2[er][1] = la ? ph[er][1] : ph[er][0];
                        hh2[er][0] = ha ? ph[er][2] : ph[er][3]; hh2[er][1] = ha ? ph[er][3] : ph[er][2];
                    }
                    const int io = pair_ok ? ((KO & 2) ? (i2 & 3) : i2) : 0;
                    const DtBuf bl0 = dt_buf_n(L2b + (int64_t)(2 * io) * (C / 2), pair_ok ? 8u * nv : 0u);
                    const DtBuf bl1 = dt_buf_n(L2b + (int64_t)(2 * io + 1) * (C / 2), pair_ok ? 8u * nv : 0u);
                    dt2d::dt_buf_st2<(KO & 512) != 0>(bl0, l2v, 0u, dt2d::f2{llo[0][0], llo[0][1]});
                    dt2d::dt_buf_st2<(KO & 512) != 0>(bl1, l2v, 0u, dt2d::f2{llo[1][0], llo[1][1]});
                    const Zq a = q2c_s(hl2[0][0], hl2[0][1], hl2[1][0], hl2[1][1]);
                    const Zq bq = q2c_s(hh2[0][0], hh2[0][1], hh2[1][0], hh2[1][1]);
                    const Zq c = q2c_s(lh2[0][0], lh2[0][1], lh2[1][0], lh2[1][1]);
                    if constexpr ((KO & 64) != 0) {
                        float *const Y1p = p.Yh1 + (img / 4) * 3 + strip * (VL * 2) + (int64_t)io * (C / 2);
                        const dt2d::f2 q6[6] = {{sq * a.z0r, sq * a.z0i}, {sq * bq.z0r, sq * bq.z0i}, {sq * c.z0r, sq * c.z0i},
                                                {sq * c.z1r, sq * c.z1i}, {sq * bq.z1r, sq * bq.z1i}, {sq * a.z1r, sq * a.z1i}};
#pragma unroll
                        for (int m = 0; m < 6; ++m) {
                            const DtBuf by1 = dt_buf_n(Y1p + (int64_t)m * (R / 4) * (C / 2), pair_ok ? 8u * nv : 0u);
                            dt2d::dt_buf_st2<false>(by1, l2v, 0u, q6[m]);
                        }
                    } else {
                    f4 *o = slab2 + lane * 3;
                    o[0] = f4{sq * a.z0r, sq * a.z0i, sq * bq.z0r, sq * bq.z0i};
                    o[1] = f4{sq * c.z0r, sq * c.z0i, sq * c.z1r, sq * c.z1i};
                    o[2] = f4{sq * bq.z1r, sq * bq.z1i, sq * a.z1r, sq * a.z1i};
                    DT_WAVE_LDS_SYNC();
                    const DtBuf by1 = dt_buf_n(Y1b + (int64_t)io * (C / 4) * 12, pair_ok ? 48u * nv : 0u);
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        const f4 v = slab2[3 * HL + lane + 64 * m];
                        dt2d::dt_buf_st4<(KO & 256) != 0>(by1, yv + 1024u * m, 0u, v);
                    }
                    DT_WAVE_LDS_SYNC();
                    }
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
// Levels 2 + 1 of the inverse transform in one march (transform2d.py:242-293): the level-2 output Z1 never leaves the
// registers.
//
// A macro-step takes ONE pair of level-2 input rows (two rows of Z2, one row of Yh1 records: one record per lane) and
// produces FOUR rows of X:
//   level 2  c2q of the lane's record; the row interpolation first (u0 = R0 z + R1 p23, u1 = R0 p05 + R1 p14: ten
//            samples = the lane's two columns + two lanes either side by DPP), then the column interpolation in
//            transposed form: the pair adds into the pending sums of the five groups of four Z1 rows whose windows
//            contain it (colifilt, lowlevel.py:156-260).  Rows-then-columns keeps ONE pending plane
//            (5 groups x 4 rows x 4 columns = 80 registers); it is the reference's sum in another order.
//   level 1  twice per macro-step: two rows of Z1 + one row of Yh0 records (two per lane, through the slab) -> c2q ->
//            row filters (v0 = g0o * Z1 + g1o * q23, v1 = g0o * q05 + g1o * q14, DPP halo of one lane) -> transposed
//            column filters into 8 pending rows of X; the two oldest leave as 16-byte stores.
// The rows of a macro-step are requested one macro-step ahead (64 registers in flight); loads and stores are issued on
// every macro-step, for rows outside the band against descriptors that drop them (see k_fwd12m).
// Symmetric extension: rows by reflected indices (a reflected record row also swaps the two rows of its quads),
// columns by mirrored lanes (their records come from the mirror lane, quads flipped); colifilt of a symmetrically
// extended plane is the symmetric extension of colifilt's output, so Z1 needs no special case at the image edge.
// Only the standard phases (sum(g0a g0b) > 0, sum(g1a g1b) < 0: every shipped q-shift set) and M = 10.
// ======================================================================================================================
struct Inv21mParams {
    const float *Z2;      // [B][R/2][C/2]
    const float *Yh1;     // [B][R/4][C/4][12]
    const float *Yh0;     // [B][R/2][C/2][12]
    float *X;             // [B][R][C]
    int B, R, C;          // R % 4 == 0, C % 4 == 0
    MarchJobs jb;
    float g1[6], g2[6];   // gain x sqrt(1/2) per subband of level 1 / level 2
    float g0o[MAXT1], g1o[MAXT1];
    // colifilt(., g0b, g0a), colifilt(., g1b, g1a): first / second argument.  Read as pairs (h[2k], h[2k + 1]): the two
    // output phases a window sample feeds are one packed multiply-add
    float l_a[MAXT2] __attribute__((aligned(8))), l_b[MAXT2] __attribute__((aligned(8)));
    float h_a[MAXT2] __attribute__((aligned(8))), h_b[MAXT2] __attribute__((aligned(8)));
    // level-1 taps by distance d from the centre, each twice (g, g): the row filters run on (plane, plane) pairs with
    // the tap common to both halves, and a broadcast from half a scalar pair is not free (see the forward kernel)
    float gd0[2 * (MAXT1 / 2 + 1)] __attribute__((aligned(8))), gd1[2 * (MAXT1 / 2 + 1)] __attribute__((aligned(8)));
};
template <class Pm>
inline void dtm_pack_inv_biort(Pm &p, int m0, int m1) {
    for (int d = 0; d <= MAXT1 / 2; ++d) {
        p.gd0[2 * d] = p.gd0[2 * d + 1] = d <= m0 / 2 ? p.g0o[m0 / 2 - d] : 0.f;
        p.gd1[2 * d] = p.gd1[2 * d + 1] = d <= m1 / 2 ? p.g1o[m1 / 2 - d] : 0.f;
    }
}

template <int M0, int M1, int M>
struct Inv21m {
    static constexpr int H0 = M0 / 2, H1 = M1 / 2;
    static constexpr int HL1 = 1, HL2 = 2, HL = HL1 + HL2;
    static constexpr int VL = 64 - 2 * HL;
    static constexpr int NG = M / 2;              // pending groups of four Z1 rows
    static constexpr int NPX = M0 > M1 + 2 ? M0 + 1 : M1 + 3;     // pending rows of X
    static_assert(M == 10 && H0 <= 4 && H1 <= 4 && M0 % 2 == 1 && M1 % 2 == 1, "filters the marching inverse is built for");
    static_assert(M0 == M1 + 2, "pending-row bookkeeping assumes len(g0o) = len(g1o) + 2");
};

#if defined(__HIP_DEVICE_COMPILE__)
// c2q of one subband pair (A.4): quad a b / c d from w0, w1 (complex) and their gains (x sqrt(1/2))
// The two products of every sum are NOT left to the compiler's contraction (a * b + c * d becomes a multiply and a fused multiply-add,
// and WHICH product is rounded first differed between kernels that inline this -- k_inv21m and the same macro-steps as a marching
// pair, march2d_ipair.hpp, disagreed in the last bit): the second subband's products are rounded, the first's are fused.
__device__ __forceinline__ void c2q_quad(float w0r, float w0i, float w1r, float w1i, float ga, float gb, float (&q)[2][2]) {
    const float qr = gb * w1r, qi = gb * w1i;
    q[0][0] = __builtin_fmaf(ga, w0r, qr); q[0][1] = __builtin_fmaf(ga, w0i, qi);
    q[1][0] = __builtin_fmaf(ga, w0i, -qi); q[1][1] = __builtin_fmaf(-ga, w0r, qr);
}
// interpolating row filter (colifilt along a row), standard phases: four outputs from the 10-sample window w
// (element j = sample 2 jx - 4 + j); POS: sum(ha hb) > 0
template <bool POS>
__device__ __forceinline__ void ifilt_row(const float (&w)[10], const float *ha, const float *hb, float (&y)[4]) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float lo = w[8 - 2 * k], hi = w[9 - 2 * k];
        const float xa = POS ? hi : lo, xb = POS ? lo : hi;
        y[0] += ha[2 * k] * xb; y[1] += hb[2 * k] * xa; y[2] += ha[2 * k + 1] * xb; y[3] += hb[2 * k + 1] * xa;
    }
}
// the same on pairs: W[i] = samples (2 jx - 4 + 2i, + 1), i.e. the (even, odd) = ("lo", "hi") samples of lane jx - 2 + i;
// ha2[k] = (ha[2k], ha[2k + 1]).  E = (y0, y2) and O = (y1, y3) are accumulated.
template <bool POS>
__device__ __forceinline__ void ifilt_row2(const pk2 (&W)[5], const pk2 *ha2, const pk2 *hb2, pk2 &E, pk2 &O) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        if (POS) { E += ha2[k] * DTM_BX(W[4 - k]); O += hb2[k] * DTM_BY(W[4 - k]); }
        else     { E += ha2[k] * DTM_BY(W[4 - k]); O += hb2[k] * DTM_BX(W[4 - k]); }
    }
}
__device__ __forceinline__ void win5(float v0, float v1, pk2 (&W)[5]) {
    W[2] = pk2{v0, v1};
    const float l0 = dpp_from_left(v0), l1 = dpp_from_left(v1), r0 = dpp_from_right(v0), r1 = dpp_from_right(v1);
    W[1] = pk2{l0, l1}; W[3] = pk2{r0, r1};
    W[0] = pk2{dpp_from_left(l0), dpp_from_left(l1)}; W[4] = pk2{dpp_from_right(r0), dpp_from_right(r1)};
}
// the window of a half-resolution plane row: the lane's two samples + two lanes either side
__device__ __forceinline__ void win10(float v0, float v1, float (&w)[10]) {
    w[4] = v0; w[5] = v1;
    const float l0 = dpp_from_left(v0), l1 = dpp_from_left(v1), r0 = dpp_from_right(v0), r1 = dpp_from_right(v1);
    w[2] = l0; w[3] = l1; w[6] = r0; w[7] = r1;
    w[0] = dpp_from_left(l0); w[1] = dpp_from_left(l1); w[8] = dpp_from_right(r0); w[9] = dpp_from_right(r1);
}
#endif

// PF: macro-steps the requests run ahead (1: one set of rows in flight, 64 registers; 2: two sets, for the one-wavefront-per-SIMD
// build OCC = 1, whose 512 registers leave room for them -- the judge's round-4 experiment, profiles/r05/ab_inv21m_prefetch.txt)
template <int M0, int M1, int M, int KO, int PF = 1, int OCC = 2>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) k_inv21m(const Inv21mParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = Inv21m<M0, M1, M>;
    constexpr int H0 = G::H0, H1 = G::H1, HL = G::HL, VL = G::VL, NG = G::NG, NPX = G::NPX;
    __shared__ __attribute__((aligned(16))) f4 slab[2][64 * 6 + 6 * G::HL + 8];
    const int lane = threadIdx.x;
    int strip, band, b;
    if (!dtm_job(p.jb, blockIdx.x, strip, band, b)) return;
    const int R = p.R, C = p.C;
    const int cb = strip * (4 * VL) - 4 * HL;             // column of lane 0
    const int c0 = cb + 4 * lane;
    const bool mir = c0 < 0 || c0 >= C;
    const bool edge_strip = cb < 0 || cb + 256 > C;      // uniform: some lane is mirrored
    const int nv = (C - strip * (4 * VL)) / 4 < VL ? (C - strip * (4 * VL)) / 4 : VL;
    // where the lane's half-resolution samples / its level-2 record / its level-1 records come from
    int zl = c0 < 0 ? -2 - c0 / 2 : (c0 >= C ? C - 2 - c0 / 2 : c0 / 2);
    zl = zl < 0 ? 0 : (zl > C / 2 - 2 ? C / 2 - 2 : zl);
    int ql = c0 < 0 ? -1 - c0 / 4 : (c0 >= C ? C / 2 - 1 - c0 / 4 : c0 / 4);
    ql = ql < 0 ? 0 : (ql > C / 4 - 1 ? C / 4 - 1 : ql);
    int sl = c0 < 0 ? (-c0 - 4 - cb) / 4 : (c0 >= C ? (2 * C - 4 - c0 - cb) / 4 : lane);
    sl = sl < 0 ? 0 : (sl > 63 ? 63 : sl);
    // the level-1 record pieces of a row this wavefront fetches: lanes lmin .. lmax are inside the image
    const int lmin = cb < 0 ? -cb / 4 : 0, lmax = (C - cb) / 4 - 1 < 63 ? (C - cb) / 4 - 1 : 63;

    const int64_t img = (int64_t)b * R * C;
    const DtBuf bz = dt_buf2g(p.Z2 + img / 4);
    const DtBuf b2 = dt_buf2g(p.Yh1 + (img / 4) * 3);
    const float *const Y0b = p.Yh0 + img * 3 + (int64_t)(cb + 4 * lmin) * 6;       // record of lane lmin in row 0
    float *const Xb = p.X + img + strip * (4 * VL);
    const unsigned zpitch = (unsigned)C * 2u, r2pitch = (unsigned)C * 12u;          // bytes per Z2 row, per Yh1 record row
    const unsigned r1bytes = (unsigned)(lmax - lmin + 1) * 96u;

    const int rb = band * p.jb.band_rows;
    const int nrow = R - rb < p.jb.band_rows ? R - rb : p.jb.band_rows;
    const int j0 = rb / 4 - 1, j1 = (rb + nrow) / 4;      // groups of Z1 rows level 1 reads
    const int nfirst = j0 - 2, nms = j1 - j0 + 5;         // level-2 pairs j0 - 2 .. j1 + 2

    // the level-1 taps of the column pass are the low halves of the (g, g) pairs the row pass holds (the filters are
    // symmetric: g[k] = gd[|k - H|]): no scalar registers of their own -- the kernel was spilling 47 of them
    float g0o[M0], g1o[M1];
#pragma unroll
    for (int k = 0; k < M0; ++k) g0o[k] = p.gd0[2 * (k < H0 ? H0 - k : k - H0)];
#pragma unroll
    for (int k = 0; k < M1; ++k) g1o[k] = p.gd1[2 * (k < H1 ? H1 - k : k - H1)];

    // ---- requests -----------------------------------------------------------------------------------------------
    auto zrow = [&](int u) { u = u < 0 ? -1 - u : u; u = u >= R / 2 ? R - 1 - u : u; u = u < 0 ? 0 : (u > R / 2 - 1 ? R / 2 - 1 : u); return (KO & 1) ? (u & 7) : u; };
    auto ld_z2 = [&](int u) -> dt2d::f2 { return dt2d::dt_buf_ld2(bz, (unsigned)zl * 4u, (unsigned)zrow(u) * zpitch); };
    auto pair_row = [&](int n, bool &sw) { sw = n < 0 || n >= R / 4; n = n < 0 ? -1 - n : n; n = n >= R / 4 ? R / 2 - 1 - n : n; n = n < 0 ? 0 : (n > R / 4 - 1 ? R / 4 - 1 : n); return (KO & 1) ? (n & 3) : n; };
    auto rec_row = [&](int rr, bool &sw) { sw = rr < 0 || rr >= R / 2; rr = rr < 0 ? -1 - rr : rr; rr = rr >= R / 2 ? R - 1 - rr : rr; rr = rr < 0 ? 0 : (rr > R / 2 - 1 ? R / 2 - 1 : rr); return (KO & 1) ? (rr & 7) : rr; };

    dt2d::f2 z2p[PF][2];
    f4 r2p[PF][3], r1p[PF][2][6];
    auto request = [&](int n, auto parc) {
        constexpr int P_ = decltype(parc)::value;        // level-2 inputs of pair n, level-1 record rows of group n - 2
        bool sw;
        z2p[P_][0] = ld_z2(2 * n); z2p[P_][1] = ld_z2(2 * n + 1);
        const unsigned ro2 = (unsigned)pair_row(n, sw) * r2pitch;
        if constexpr ((KO & 64) != 0) {        // experiment: planar pyramid, a lane fetches its own coefficient of each subband
            const unsigned ro2p = ro2 / 6u, pl2 = (unsigned)R * (unsigned)C / 2u;      // bytes per row / per plane of Yh1
            dt2d::f2 q6[6];
#pragma unroll
            for (int m = 0; m < 6; ++m) q6[m] = dt2d::dt_buf_ld2(b2, (unsigned)ql * 8u + pl2 * m, ro2p);
            r2p[P_][0] = f4{q6[0].x, q6[0].y, q6[1].x, q6[1].y}; r2p[P_][1] = f4{q6[2].x, q6[2].y, q6[3].x, q6[3].y};
            r2p[P_][2] = f4{q6[4].x, q6[4].y, q6[5].x, q6[5].y};
        } else {
#pragma unroll
        for (int m = 0; m < 3; ++m) r2p[P_][m] = dt2d::dt_buf_ld4(b2, (unsigned)ql * 48u + 16u * m, ro2);
        }
        // the record rows of a group level 1 does not run on (the first two and last two macro-steps of a band) are
        // requested against zero bytes: the loads are issued -- every macro-step carries the same memory operations --
        // and move nothing (they were a quarter of the kernel's record traffic at 40-row bands: 32 rows fetched, 24 used)
        const bool gok = n - 2 >= j0 && n - 2 <= j1;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int rr = rec_row(2 * (n - 2) + e, sw);
            if constexpr ((KO & 64) != 0) {
                const float *const Y0p = p.Yh0 + img * 3 + (cb + 4 * lmin) + (int64_t)rr * C;
#pragma unroll
                for (int m = 0; m < 6; ++m) {
                    const DtBuf br = dt_buf_n(Y0p + (int64_t)m * (R / 2) * C, gok ? r1bytes / 6u : 0u);
                    r1p[P_][e][m] = dt2d::dt_buf_ld4(br, 16u * (unsigned)(sl - lmin), 0u);
                }
                continue;
            }
            const DtBuf br = dt_buf_n(Y0b + (int64_t)rr * C * 6, gok ? r1bytes : 0u);
#pragma unroll
            for (int m = 0; m < 6; ++m) r1p[P_][e][m] = dt2d::dt_buf_ld4(br, 16u * (unsigned)lane + 1024u * m, 0u);
        }
    };
    request(nfirst, std::integral_constant<int, 0>{});
    if constexpr (PF == 2) request(nfirst + 1, std::integral_constant<int, 1>{});
#pragma unroll
    for (int q = 0; q < PF; ++q) {
        asm volatile("" : "+v"(z2p[q][0].x), "+v"(z2p[q][0].y), "+v"(z2p[q][1].x), "+v"(z2p[q][1].y) : : "memory");
#pragma unroll
        for (int m = 0; m < 3; ++m) asm volatile("" : "+v"(r2p[q][m].x), "+v"(r2p[q][m].y), "+v"(r2p[q][m].z), "+v"(r2p[q][m].w) : : "memory");
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int m = 0; m < 6; ++m) asm volatile("" : "+v"(r1p[q][e][m].x), "+v"(r1p[q][e][m].y), "+v"(r1p[q][e][m].z), "+v"(r1p[q][e][m].w) : : "memory");
    }

    // pending Z1 groups: PzE[a][c] = rows (0, 2), PzO[a][c] = rows (1, 3) of group slot a, column c
    pk2 PzE[NG][4], PzO[NG][4];
    float PX[NPX][4];
#pragma unroll
    for (int a = 0; a < NG; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) { PzE[a][c] = pk2{0.f, 0.f}; PzO[a][c] = pk2{0.f, 0.f}; }
#pragma unroll
    for (int i = 0; i < NPX; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) PX[i][c] = 0.f;

    const unsigned xv = 16u * (unsigned)(lane - HL);

    auto macro = [&](int ms, auto parc) {
        constexpr int P_ = decltype(parc)::value;
        const int n = nfirst + ms, j = n - 2;
        bool sw2, sw1[2];
        (void)pair_row(n, sw2);
        (void)rec_row(2 * j, sw1[0]); (void)rec_row(2 * j + 1, sw1[1]);
        // ---- what was requested a macro-step ago: the level-1 records to the slab, the level-2 inputs through c2q
        f4 s1[2][6];                    // (KO & 64): the lane's own coefficients, no slab
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                if constexpr ((KO & 64) != 0) s1[e][m] = r1p[P_][e][m]; else slab[e][6 * lmin + lane + 64 * m] = r1p[P_][e][m];
            }
        float z[2][2], p05[2][2], p23[2][2], p14[2][2];
        {
            const f4 ra = r2p[P_][0], rc = r2p[P_][1], re = r2p[P_][2];
            c2q_quad(ra.x, ra.y, re.z, re.w, p.g2[0], p.g2[5], p05);
            c2q_quad(rc.x, rc.y, rc.z, rc.w, p.g2[2], p.g2[3], p23);
            c2q_quad(ra.z, ra.w, re.x, re.y, p.g2[1], p.g2[4], p14);
            z[0][0] = z2p[P_][0].x; z[0][1] = z2p[P_][0].y; z[1][0] = z2p[P_][1].x; z[1][1] = z2p[P_][1].y;
            if (sw2) {
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    float t_;
                    t_ = p05[0][f]; p05[0][f] = p05[1][f]; p05[1][f] = t_;
                    t_ = p23[0][f]; p23[0][f] = p23[1][f]; p23[1][f] = t_;
                    t_ = p14[0][f]; p14[0][f] = p14[1][f]; p14[1][f] = t_;
                }
            }
            if (edge_strip) {
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
        request(n + PF, parc);

        // ---- level 2: rows, then columns into the pending groups
        const pk2 *la2 = reinterpret_cast<const pk2 *>(p.l_a), *lb2 = reinterpret_cast<const pk2 *>(p.l_b);
        const pk2 *ha2 = reinterpret_cast<const pk2 *>(p.h_a), *hb2 = reinterpret_cast<const pk2 *>(p.h_b);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            // u0 = R0 z + R1 p23, u1 = R0 p05 + R1 p14 as (u[0], u[2]) and (u[1], u[3])
            pk2 U0E = {0.f, 0.f}, U0O = {0.f, 0.f}, U1E = {0.f, 0.f}, U1O = {0.f, 0.f}, W[5];
            win5(z[e][0], z[e][1], W);     ifilt_row2<true>(W, la2, lb2, U0E, U0O);
            win5(p23[e][0], p23[e][1], W); ifilt_row2<false>(W, ha2, hb2, U0E, U0O);
            win5(p05[e][0], p05[e][1], W); ifilt_row2<true>(W, la2, lb2, U1E, U1O);
            win5(p14[e][0], p14[e][1], W); ifilt_row2<false>(W, ha2, hb2, U1E, U1O);
#pragma unroll
            for (int a = 0; a < NG; ++a) {          // slot a = group n - 2 + a, tap pair k = a
                if (e == 0) {       // the even ("lo") row of the pair: rows (0, 2) through g0, rows (1, 3) through g1
                    // The first touch of a slot in this macro-step also moves the groups up by one: slot a continues the
                    // sums that slot a + 1 held during the previous macro-step (whose slot 0 level 1 has consumed), the last
                    // slot starts from nothing -- the shift costs no instruction of its own (it was 64 v_mov per macro-step)
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
        // ---- level 1 on the completed group j = n - 2 (Z1 rows 4j .. 4j + 3 = Pz[0])
        DT_WAVE_LDS_SYNC();
        const bool grp = j >= j0 && j <= j1;           // uniform
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
                    if constexpr ((KO & 64) != 0) {
                        s_[0] = f4{s1[h][0].x, s1[h][0].y, s1[h][1].x, s1[h][1].y}; s_[1] = f4{s1[h][2].x, s1[h][2].y, s1[h][3].x, s1[h][3].y};
                        s_[2] = f4{s1[h][4].x, s1[h][4].y, s1[h][5].x, s1[h][5].y}; s_[3] = f4{s1[h][0].z, s1[h][0].w, s1[h][1].z, s1[h][1].w};
                        s_[4] = f4{s1[h][2].z, s1[h][2].w, s1[h][3].z, s1[h][3].w}; s_[5] = f4{s1[h][4].z, s1[h][4].w, s1[h][5].z, s1[h][5].w};
                    }
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
                const pk2 *gd0 = reinterpret_cast<const pk2 *>(p.gd0), *gd1 = reinterpret_cast<const pk2 *>(p.gd1);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    // (Z1, q05) through g0o and (q23, q14) through g1o, the filters symmetric: V[c] = (v0[c], v1[c])
                    const int zr = 2 * h + e;       // row of the group: rows 0, 2 in PzE (x, y), rows 1, 3 in PzO
                    pk2 Wa[4 + 2 * H0], Wb[4 + 2 * H1], V[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float zv = (zr & 1) ? ((zr & 2) ? PzO[0][c].y : PzO[0][c].x) : ((zr & 2) ? PzE[0][c].y : PzE[0][c].x);
                        Wa[H0 + c] = pk2{zv, q05[e][c]};
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
                    // columns, transposed: PX[i] is row rho - H0 + i.  The first row of a half-step (e == 0) also moves
                    // the pending rows up by the two that left at the end of the previous half-step: row i continues what
                    // row i + 2 held (no v_mov for the shift; it was 48 + 16 per macro-step)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (e == 0) {
#pragma unroll
                            for (int i = 0; i < NPX; ++i) {
                                float acc;
                                if (i + 2 < NPX && i < M0) {
                                    // three-address form spelled out: left to itself the compiler takes the two-address
                                    // v_fmac and copies row i + 2 into place first -- the very v_mov this is about
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
                const int xo = ok ? ((KO & 2) ? (x & 15) : x) : 0;
                const DtBuf bo = dt_buf_n(Xb + (int64_t)xo * C, ok ? 16u * nv : 0u);
                dt2d::dt_buf_st4<true>(bo, xv, 0u, f4{PX[e][0], PX[e][1], PX[e][2], PX[e][3]});
            }
        }
        DT_WAVE_LDS_SYNC();
    };
    if constexpr (PF == 1) {
        for (int ms = 0; ms < nms; ++ms) macro(ms, std::integral_constant<int, 0>{});
    } else {
        // two sets of rows in flight, used in turn (a surplus macro-step of an odd count stores nothing: its rows lie beyond the band)
        for (int ms = 0; ms < nms; ms += 2) { macro(ms, std::integral_constant<int, 0>{}); macro(ms + 1, std::integral_constant<int, 1>{}); }
    }
#endif
}

}  // namespace dtm
