// Marching wavefront programs of the float32 2-D DT-CWT (gfx950).
//
// The tile programs of fused2d_tiles_v2.hpp hand the column pass to the row pass through LDS planes and
// workgroup barriers; profiles/r03/ko_bench.txt shows them spending 160 us of CU time per 4096^2 step on 24 us
// of FMAs.  Here ONE WAVEFRONT is the unit of work and nothing is shared between wavefronts:
//
//   * a wavefront owns a strip of 64 x 4 = 256 adjacent columns (lane l holds columns 4l .. 4l+3 of the strip
//     as one float4: a row of the strip is ONE 1 KiB global_load_dwordx4) and marches down a segment of rows;
//   * the column filters run over a register ring of the last rows (static indices: the march loop is unrolled
//     over one period of the ring), so their window costs no LDS and no halo re-computation;
//   * the row filters take the neighbouring columns from the neighbouring LANES with DPP wave shifts
//     (v_mov_b32_dpp wave_shr:1 / wave_shl:1): the first and the last HL lanes of a strip are halo lanes;
//   * q2c / c2q are lane-local (a lane owns whole 2 x 2 quads), the 48-byte subband records travel through a
//     wave-private LDS slab so that every global access is a run of consecutive 16-byte pieces;
//   * rows are requested P steps ahead of their use; there is no barrier anywhere, so the load, FMA and store
//     streams of the wavefronts of a CU overlap freely.
//
// Reference: dtcwt/numpy/transform2d.py:112-130 (level 1 forward), :275-293 (level 1 inverse), q2c :301-322,
// c2q :324-350; colfilter dtcwt/numpy/lowlevel.py:47-80.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "fused2d_tiles.hpp"

namespace dtm {

using dt2d::DtBuf;
using dt2d::dt_buf;
using dt2d::f4;

#if defined(__HIP_DEVICE_COMPILE__)
// A raw buffer over [base, base + 2 GiB): a lane offset with bit 31 set is out of range, i.e. the hardware drops the
// store (and returns 0 for a load).  Lanes that own nothing keep such an offset, so no store of the march loop sits in
// a divergent branch -- with exec-masked store blocks the compiler cannot count the outstanding memory operations of a
// path and waits for (nearly) all of them at every use of a prefetched row.
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

constexpr int MAXT1 = 9;        // longest level-1 filter a one-lane halo serves (halo <= 4 columns)

struct Fwd1mParams {
    const float *X;       // [B][R][C]
    float *LoLo;          // [B][R][C]
    float *Yh;            // [B][R/2][C/2][12 floats]
    int B, R, C;          // R even, C % 4 == 0
    int nstrip, nseg, seg_rows;   // strips of VL*4 columns, segments of seg_rows (even) rows
    float h0[MAXT1], h1[MAXT1];
};

// One step of the level-1 forward march: output rows r, r+1 from window rows w[0 .. 2 HH + 1] (row r - HH first).
template <int M0, int M1>
struct Fwd1m {
    static constexpr int H0 = M0 / 2, H1 = M1 / 2, HH = H0 > H1 ? H0 : H1;
    static constexpr int HL = 1;                  // halo lanes either side
    static constexpr int VL = 64 - 2 * HL;        // lanes that own output columns
    static constexpr int WR = 2 * HH + 2;         // window rows of a step
    static_assert(HH <= 4, "one halo lane");
    static_assert(M0 % 2 == 1 && M1 % 2 == 1, "odd-length biort filters");
};

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

// out[c] = sum_k h[k] w[c + HH + H - k]   (convolution, lowlevel.py:26-44), c = 0..3
template <int M, int HH>
__device__ __forceinline__ f4 row_fir(const float (&w)[4 + 2 * HH], const float *h) {
    constexpr int H = M / 2;
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < M; ++k) a += h[k] * w[c + HH + H - k];
        o[c] = a;
    }
    return f4{o[0], o[1], o[2], o[3]};
}

template <int M, int HH, int WR>
__device__ __forceinline__ f4 col_fir(const f4 (&w)[WR], int q, const float *h) {
    constexpr int H = M / 2;
    f4 a{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < M; ++k) {
        const f4 &x = w[q + HH + H - k];
        a.x += h[k] * x.x; a.y += h[k] * x.y; a.z += h[k] * x.z; a.w += h[k] * x.w;
    }
    return a;
}

// Both filters of a level-1 pass over the SAME samples, the filters symmetric (h[k] = h[M-1-k]: every biort set): the
// mirror pairs are added once and shared, 2 + 3 adds + 3 + 4 multiply-adds for a 5- and a 7-tap filter instead of 12
// (what separates the result from the tap-by-tap sum is the rounding of those adds: ~1 ulp either way).
//   oa = sum_k a[k] x[c + HH + HA - k],  ob = sum_k b[k] x[c + HH + HB - k]
template <int MA, int MB, int HH>
__device__ __forceinline__ void sym_pair(const float *xc, const float *a, const float *b, float &oa, float &ob) {
    constexpr int HA = MA / 2, HB = MB / 2, HM = HA > HB ? HA : HB;
    float sm[HM + 1];
    sm[0] = xc[0];
#pragma unroll
    for (int d = 1; d <= HM; ++d) sm[d] = xc[-d] + xc[d];
    float ra = a[HA] * sm[0], rb = b[HB] * sm[0];
#pragma unroll
    for (int d = 1; d <= HA; ++d) ra += a[HA - d] * sm[d];
#pragma unroll
    for (int d = 1; d <= HB; ++d) rb += b[HB - d] * sm[d];
    oa = ra; ob = rb;
}
// the lowpass one alone (warm-up rows)
template <int MA, int HH>
__device__ __forceinline__ float sym_one(const float *xc, const float *a) {
    constexpr int HA = MA / 2;
    float ra = a[HA] * xc[0];
#pragma unroll
    for (int d = 1; d <= HA; ++d) ra += a[HA - d] * (xc[-d] + xc[d]);
    return ra;
}

// q2c of the lane's two quads of a plane (rows e0 / e1; the 1/sqrt2 is already in the row taps)
//   z0 = (a - d) + j(b + c), z1 = (a + d) + j(b - c)  for  a b / c d
struct Zq { float z0r, z0i, z1r, z1i; };
__device__ __forceinline__ Zq q2c_s(float a, float b, float c, float d) { return Zq{a - d, b + c, a + d, b - c}; }

#endif  // __HIP_DEVICE_COMPILE__

// KO bit 0: every wavefront reads rows 0..15 (loads served by the caches), bit 1: stores go to rows 0..15
template <int M0, int M1, int P, int WPB, int KO>
__global__ void __launch_bounds__(64 * WPB) k_fwd1m(const Fwd1mParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = Fwd1m<M0, M1>;
    constexpr int HH = G::HH, WR = G::WR, NR = WR + 2 * P, PER = NR / 2;
    __shared__ __attribute__((aligned(16))) f4 slab_all[WPB][64 * 6 + 8];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int job = blockIdx.x * WPB + wv;
    const int njob = p.nstrip * p.nseg * p.B;
    if (job >= njob) return;
    const int strip = job % p.nstrip, sb = job / p.nstrip, seg = sb % p.nseg, b = sb / p.nseg;
    f4 *slab = slab_all[wv];

    const int R = p.R, C = p.C;                         // uniform: keep them out of the divergent code below
    // the lane's columns; mirrored blocks beyond the left / right edge are loaded reversed
    const int c0 = strip * (4 * G::VL) - 4 * G::HL + 4 * lane;
    const bool rev = c0 < 0 || c0 >= C;
    int lc = c0 < 0 ? -c0 - 4 : (c0 >= C ? 2 * C - 4 - c0 : c0);
    lc = lc < 0 ? 0 : (lc > C - 4 ? C - 4 : lc);
    const bool edge_strip = strip == 0 || (strip + 1) * (4 * G::VL) + 4 >= C;     // uniform

    const int64_t img = (int64_t)b * R * C;
    const DtBuf bx = dt_buf2g(p.X + img);
    // output rows of the strip: LoLo from its first owned column, records from its first owned quad column
    float *const Lb = p.LoLo + img + strip * (4 * G::VL);
    float *const Yb = p.Yh + img * 3 + (int64_t)strip * (G::VL * 24);
    const unsigned pitch = (unsigned)C * 4u;         // bytes per image row

    const int rb = seg * p.seg_rows;
    const int nrow = (R - rb < p.seg_rows ? R - rb : p.seg_rows);
    const int nst = nrow / 2;
    const int last_row = rb + nrow + HH - 1;            // last row any step of this segment wants

    auto ldrow = [&](int u) -> f4 {
        u = u > last_row ? last_row : u;
        u = u < 0 ? -1 - u : u;
        u = u >= R ? 2 * R - 1 - u : u;
        if (KO & 1) u &= 15;
        return dt2d::dt_buf_ld4(bx, (unsigned)lc * 4u, (unsigned)u * pitch);
    };
    // mirrored blocks of the edge strips are turned round when a row ENTERS a window (P steps after its load was
    // issued), not where it is loaded: a select on a load's result is a wait for that load
    auto fix = [&](f4 &v) { if (edge_strip) v = rev ? rev4(v) : v; };

    float h0[M0], h1[M1], h0s[M0], h1s[M1];
    const float s = 0.70710678118654752440f;
#pragma unroll
    for (int k = 0; k < M0; ++k) { h0[k] = p.h0[k]; h0s[k] = s * p.h0[k]; }
#pragma unroll
    for (int k = 0; k < M1; ++k) { h1[k] = p.h1[k]; h1s[k] = s * p.h1[k]; }

    // The whole ring is loaded and WAITED FOR before the march starts.  The compiler counts the outstanding memory
    // operations of every path and takes the minimum where paths meet: a loop entered with the ring's loads still in
    // flight would wait, at the top of every period, for all but the youngest 8 operations -- i.e. for the stores of
    // the previous step -- instead of the 30 the steady state allows.  Entered with nothing in flight, the counts of
    // the back edge stand.  (Dropped out-of-range stores as padding do NOT work: they retire at once, out of order,
    // and the counter then lets real loads through unfinished.)
    f4 ring[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) ring[i] = ldrow(rb - HH + i);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        asm volatile("" : "+v"(ring[i].x), "+v"(ring[i].y), "+v"(ring[i].z), "+v"(ring[i].w) : : "memory");
    }
#pragma unroll
    for (int i = 0; i < WR - 2; ++i) fix(ring[i]);

    // record pieces of the strip's row: piece j (16 bytes) of the wave's slab <-> byte 16 (j - 6 HL) of the strip's
    // part of the record row
    const int nv = (C - strip * (4 * G::VL)) / 4 < G::VL ? (C - strip * (4 * G::VL)) / 4 : G::VL;   // owning lanes of this strip
    const unsigned lv = 16u * (unsigned)(lane - G::HL);     // halo lanes: out of range either side
    const unsigned yv = 16u * (unsigned)lane;

    for (int t0 = 0; t0 < nst; t0 += PER) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int t = t0 + k;
            if (t >= nst) break;
            const int r = rb + 2 * t;
            const f4 n0 = ldrow(r - HH + NR), n1 = ldrow(r - HH + NR + 1);
            fix(ring[(2 * k + WR - 2) % NR]);
            fix(ring[(2 * k + WR - 1) % NR]);
            f4 w[WR];
#pragma unroll
            for (int j = 0; j < WR; ++j) w[j] = ring[(2 * k + j) % NR];

            f4 ll[2], lh[2], hl[2], hh[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f4 lo = col_fir<M0, HH, WR>(w, q, h0), hi = col_fir<M1, HH, WR>(w, q, h1);
                float wl[4 + 2 * HH], wh[4 + 2 * HH];
                row_window<HH, (KO & 8) != 0>(lo, wl);
                row_window<HH, (KO & 8) != 0>(hi, wh);
                ll[q] = row_fir<M0, HH>(wl, h0);
                lh[q] = row_fir<M1, HH>(wl, h1s);
                hl[q] = row_fir<M0, HH>(wh, h0s);
                hh[q] = row_fir<M1, HH>(wh, h1s);
            }
            const int ro = (KO & 2) ? (r & 15) : r;
            // KO bit 2: no store instruction is ever executed (the test keeps the results alive)
            const bool st_ok = !(KO & 4) || (ll[0].x == 123456.789f && lh[1].y == hh[0].z * 3.f + hl[1].w);
            if (st_ok) {
            dt2d::dt_buf_st4<false>(dt_buf_n(Lb + (int64_t)ro * C, 16u * nv), lv, 0u, ll[0]);
            dt2d::dt_buf_st4<false>(dt_buf_n(Lb + (int64_t)(ro + 1) * C, 16u * nv), lv, 0u, ll[1]);
            // records of quad columns 2 lane', 2 lane' + 1: slots HLz0 HHz0 LHz0 LHz1 HHz1 HLz1
            {
                const Zq a0 = q2c_s(hl[0].x, hl[0].y, hl[1].x, hl[1].y), a1 = q2c_s(hl[0].z, hl[0].w, hl[1].z, hl[1].w);
                const Zq b0 = q2c_s(hh[0].x, hh[0].y, hh[1].x, hh[1].y), b1 = q2c_s(hh[0].z, hh[0].w, hh[1].z, hh[1].w);
                const Zq c0q = q2c_s(lh[0].x, lh[0].y, lh[1].x, lh[1].y), c1q = q2c_s(lh[0].z, lh[0].w, lh[1].z, lh[1].w);
                f4 *o = slab + lane * 6;
                o[0] = f4{a0.z0r, a0.z0i, b0.z0r, b0.z0i};
                o[1] = f4{c0q.z0r, c0q.z0i, c0q.z1r, c0q.z1i};
                o[2] = f4{b0.z1r, b0.z1i, a0.z1r, a0.z1i};
                o[3] = f4{a1.z0r, a1.z0i, b1.z0r, b1.z0i};
                o[4] = f4{c1q.z0r, c1q.z0i, c1q.z1r, c1q.z1i};
                o[5] = f4{b1.z1r, b1.z1i, a1.z1r, a1.z1i};
            }
            DT_WAVE_LDS_SYNC();
            const DtBuf by = dt_buf_n(Yb + (int64_t)(ro >> 1) * C * 6, 96u * nv);
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                const f4 v = slab[6 * G::HL + lane + 64 * m];      // piece lane + 64 m of the owning lanes' records
                dt2d::dt_buf_st4<true>(by, yv + 1024u * m, 0u, v);
            }
            DT_WAVE_LDS_SYNC();
            }
            ring[(2 * k) % NR] = n0;
            ring[(2 * k + 1) % NR] = n1;
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}


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
    int B, R, C;          // R % 4 == 0, C % 4 == 0
    int nstrip, nband, band_rows;     // band_rows % 4 == 0
    int lo_a_first, hi_a_first;       // sign of sum(ha * hb) of the lowpass / highpass q-shift pair (lowlevel.py:143)
    float h0[MAXT1], h1[MAXT1];
    // level-2 taps by window offset: A = sum_t ta[t] w[2t], B = sum_t tb[t] w[2t + 1] over the 2M-sample window whose
    // element j is logical sample 4i - M + 2 + j (fused2d_tiles.hpp: dfilt_pair); dtm_pack_qshift() fills them
    float ta_lo[MAXT2], tb_lo[MAXT2], ta_hi[MAXT2], tb_hi[MAXT2];
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
}

template <int M0, int M1, int M>
struct Fwd12m {
    static constexpr int H0 = M0 / 2, H1 = M1 / 2, HH = H0 > H1 ? H0 : H1;
    static constexpr int HL1 = 1, HL2 = (M - 2) / 4, HL = HL1 + HL2;
    static constexpr int VL = 64 - 2 * HL;
    static constexpr int WR = 2 * HH + 2;
    static constexpr int NP2 = M / 2;             // pending level-2 row pairs
    static constexpr int PRE = M - 2;             // LoLo1 rows a band needs above its first row (and PRE + 1... below)
    static_assert(HH <= 4 && (M - 2) % 4 == 0 && M <= MAXT2, "halo lanes");
};

template <int M0, int M1, int M, int P, int KO, int WPS = 2>
__global__ void __launch_bounds__(64, WPS) k_fwd12m(const Fwd12mParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = Fwd12m<M0, M1, M>;
    constexpr int HH = G::HH, WR = G::WR, HL = G::HL, HL2 = G::HL2, VL = G::VL, NP2 = G::NP2, PER = 4;
    static_assert(PER % P == 0, "prefetch depth divides the ring period");
    __shared__ __attribute__((aligned(16))) f4 slab[64 * 6 + 6 * G::HL + 8];
    __shared__ __attribute__((aligned(16))) f4 slab2[64 * 3 + 3 * G::HL + 8];
    const int lane = threadIdx.x;
    const int job = blockIdx.x;
    const int strip = job % p.nstrip, sb = job / p.nstrip, band = sb % p.nband, b = sb / p.nband;
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

    const int rb = band * p.band_rows;
    const int nrow = R - rb < p.band_rows ? R - rb : p.band_rows;
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
    float S[NP2][4][4];
#pragma unroll
    for (int a = 0; a < NP2; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) S[a][c][v] = 0.f;

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
            // the window by component: column filters run down wc[c][.]
            float wc[4][WR];
#pragma unroll
            for (int j = 0; j < WR; ++j) { wc[0][j] = w[j].x; wc[1][j] = w[j].y; wc[2][j] = w[j].z; wc[3][j] = w[j].w; }
            f4 ll[2];
            if (in_band) {
                f4 lh[2], hl[2], hh[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float lo[4], hi[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) sym_pair<M0, M1, HH>(&wc[c][q + HH], h0, h1, lo[c], hi[c]);
                    float wl[4 + 2 * HH], wh[4 + 2 * HH];
                    row_window<HH>(f4{lo[0], lo[1], lo[2], lo[3]}, wl);
                    row_window<HH>(f4{hi[0], hi[1], hi[2], hi[3]}, wh);
                    float a_[4], b_[4], c_[4], d_[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        sym_pair<M0, M1, HH>(&wl[c + HH], h0, h1, a_[c], b_[c]);
                        sym_pair<M0, M1, HH>(&wh[c + HH], h0, h1, c_[c], d_[c]);
                    }
                    ll[q] = f4{a_[0], a_[1], a_[2], a_[3]}; lh[q] = f4{b_[0], b_[1], b_[2], b_[3]};
                    hl[q] = f4{c_[0], c_[1], c_[2], c_[3]}; hh[q] = f4{d_[0], d_[1], d_[2], d_[3]};
                }
                {
                    const Zq a0 = q2c_s(hl[0].x, hl[0].y, hl[1].x, hl[1].y), a1 = q2c_s(hl[0].z, hl[0].w, hl[1].z, hl[1].w);
                    const Zq b0 = q2c_s(hh[0].x, hh[0].y, hh[1].x, hh[1].y), b1 = q2c_s(hh[0].z, hh[0].w, hh[1].z, hh[1].w);
                    const Zq c0q = q2c_s(lh[0].x, lh[0].y, lh[1].x, lh[1].y), c1q = q2c_s(lh[0].z, lh[0].w, lh[1].z, lh[1].w);
                    f4 *o = slab + lane * 6;
                    o[0] = f4{sq * a0.z0r, sq * a0.z0i, sq * b0.z0r, sq * b0.z0i};
                    o[1] = f4{sq * c0q.z0r, sq * c0q.z0i, sq * c0q.z1r, sq * c0q.z1i};
                    o[2] = f4{sq * b0.z1r, sq * b0.z1i, sq * a0.z1r, sq * a0.z1i};
                    o[3] = f4{sq * a1.z0r, sq * a1.z0i, sq * b1.z0r, sq * b1.z0i};
                    o[4] = f4{sq * c1q.z0r, sq * c1q.z0i, sq * c1q.z1r, sq * c1q.z1i};
                    o[5] = f4{sq * b1.z1r, sq * b1.z1i, sq * a1.z1r, sq * a1.z1i};
                }
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float lo[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) lo[c] = sym_one<M0, HH>(&wc[c][q + HH], h0);
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
            {
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

            // ---- level 2: the two LoLo1 rows of this step ----
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                constexpr int NW = 2 * M;
                float w2[NW];
                {
                    float cl[4] = {ll[q].x, ll[q].y, ll[q].z, ll[q].w}, cr[4] = {ll[q].x, ll[q].y, ll[q].z, ll[q].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) w2[4 * HL2 + c] = cl[c];
#pragma unroll
                    for (int d = 1; d <= HL2; ++d) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            cl[c] = dpp_from_left(cl[c]);   w2[4 * (HL2 - d) + c] = cl[c];
                            cr[c] = dpp_from_right(cr[c]);  w2[4 * (HL2 + d) + c] = cr[c];
                        }
                    }
                }
                float rv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tt = 0; tt < M; ++tt) {
                    rv[0] += p.ta_lo[tt] * w2[2 * tt];
                    rv[1] += p.tb_lo[tt] * w2[2 * tt + 1];
                    rv[2] += p.ta_hi[tt] * w2[2 * tt];
                    rv[3] += p.tb_hi[tt] * w2[2 * tt + 1];
                }
                const int phi = 2 * (k & 1) + q;           // row 4n + phi of its group (rbase % 4 == 0)
#pragma unroll
                for (int a = 0; a < NP2; ++a) {
                    const int tt = (phi + 8 * HL2 - 4 * a) >> 1;
                    const float cl_ = (phi & 1) ? p.tb_lo[tt] : p.ta_lo[tt], ch_ = (phi & 1) ? p.tb_hi[tt] : p.ta_hi[tt];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        S[a][(phi & 1) ? 2 : 0][v] += cl_ * rv[v];
                        S[a][(phi & 1) ? 3 : 1][v] += ch_ * rv[v];
                    }
                }
            }
            if (k & 1) {
                // rows 4n + 2, 4n + 3 are in: pair i = n - HL2 is complete
                const int i2 = (r - 2) / 4 - HL2;
                const bool pair_ok = 4 * i2 >= rb && 4 * i2 < rb + nrow;        // uniform; otherwise the stores are dropped
                {
                    const bool la = p.lo_a_first != 0, ha = p.hi_a_first != 0;
                    // column-lowpass plane rows (A lo / B lo), column-highpass plane rows (A hi / B hi)
                    float pl[2][4], ph[2][4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        pl[0][v] = la ? S[0][0][v] : S[0][2][v]; pl[1][v] = la ? S[0][2][v] : S[0][0][v];
                        ph[0][v] = ha ? S[0][1][v] : S[0][3][v]; ph[1][v] = ha ? S[0][3][v] : S[0][1][v];
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
                    dt2d::dt_buf_st2<false>(bl0, l2v, 0u, dt2d::f2{llo[0][0], llo[0][1]});
                    dt2d::dt_buf_st2<false>(bl1, l2v, 0u, dt2d::f2{llo[1][0], llo[1][1]});
                    const Zq a = q2c_s(hl2[0][0], hl2[0][1], hl2[1][0], hl2[1][1]);
                    const Zq bq = q2c_s(hh2[0][0], hh2[0][1], hh2[1][0], hh2[1][1]);
                    const Zq c = q2c_s(lh2[0][0], lh2[0][1], lh2[1][0], lh2[1][1]);
                    f4 *o = slab2 + lane * 3;
                    o[0] = f4{sq * a.z0r, sq * a.z0i, sq * bq.z0r, sq * bq.z0i};
                    o[1] = f4{sq * c.z0r, sq * c.z0i, sq * c.z1r, sq * c.z1i};
                    o[2] = f4{sq * bq.z1r, sq * bq.z1i, sq * a.z1r, sq * a.z1i};
                    DT_WAVE_LDS_SYNC();
                    const DtBuf by1 = dt_buf_n(Y1b + (int64_t)io * (C / 4) * 12, pair_ok ? 48u * nv : 0u);
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        const f4 v = slab2[3 * HL + lane + 64 * m];
                        dt2d::dt_buf_st4<false>(by1, yv + 1024u * m, 0u, v);
                    }
                    DT_WAVE_LDS_SYNC();
                }
#pragma unroll
                for (int a = 0; a + 1 < NP2; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int v = 0; v < 4; ++v) S[a][c][v] = S[a + 1][c][v];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int v = 0; v < 4; ++v) S[NP2 - 1][c][v] = 0.f;
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
// The same one-launch levels 1 + 2 as a PAIR of wavefronts per (strip, band): k_fwd12m needs ~190 registers -- two
// wavefronts per SIMD -- and a 4096^2 image is only ~1900 of its jobs, so a SIMD holds one or two wavefronts that
// cannot cover each other's memory and LDS waits (tools/kbench/march_bench: 53 us with every load and store knocked
// out, 83 us with them, where its traffic needs ~65).  Here wavefront 0 of a workgroup runs level 1 and hands the two
// LoLo1 rows of a step to wavefront 1 through a double-buffered 4 KiB LDS exchange (one s_barrier per step; the rows
// are written BEFORE the level-1 highpass work of the step, which then overlaps wavefront 1's level-2 work);
// wavefront 1 reads its 2M-sample row windows straight from the exchange (no DPP chain), scatters them into the
// pending pairs and writes the level-2 outputs.  Each role fits 128 registers: four wavefronts per SIMD, every job of
// a 4096^2 image resident at once, twice the wavefronts to hide latency with.
// ======================================================================================================================
#if defined(__HIP_DEVICE_COMPILE__)
// LDS-only workgroup barrier: what precedes it are LDS writes of this wavefront (lgkmcnt), nothing in flight in the
// vector-memory queues needs to land first -- __syncthreads() would wait for the prefetched rows and the stores too
#define DTM_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

template <int M0, int M1, int M, int P, int KO>
__global__ void __launch_bounds__(128, 4) k_fwd12w(const Fwd12mParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = Fwd12m<M0, M1, M>;
    constexpr int HH = G::HH, WR = G::WR, HL = G::HL, HL2 = G::HL2, VL = G::VL, NP2 = G::NP2, PER = 4;
    static_assert(PER % P == 0, "prefetch depth divides the ring period");
    __shared__ __attribute__((aligned(16))) f4 slab[64 * 6 + 6 * G::HL + 8];
    __shared__ __attribute__((aligned(16))) f4 slab2[64 * 3 + 3 * G::HL + 8];
    __shared__ __attribute__((aligned(16))) f4 xbuf[2][2][64 + 2 * G::HL2];      // [step parity][row][HL2 + lane]
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int job = blockIdx.x;
    const int strip = job % p.nstrip, sb = job / p.nstrip, band = sb % p.nband, b = sb / p.nband;
    const int R = p.R, C = p.C;
    const int nv = (C - strip * (4 * VL)) / 4 < VL ? (C - strip * (4 * VL)) / 4 : VL;     // owning lanes
    const int64_t img = (int64_t)b * R * C;
    const int rb = band * p.band_rows;
    const int nrow = R - rb < p.band_rows ? R - rb : p.band_rows;
    const int rbase = rb - G::PRE;
    const int nst = (nrow / 2 + G::PRE + PER - 1) / PER * PER;
    const float sq = 0.70710678118654752440f;
    const unsigned yv = 16u * (unsigned)lane;

    if (role == 0) {
        // ------------------------------------------------------------------ level 1
        const int c0 = strip * (4 * VL) - 4 * HL + 4 * lane;
        const bool rev = c0 < 0 || c0 >= C;
        int lc = c0 < 0 ? -c0 - 4 : (c0 >= C ? 2 * C - 4 - c0 : c0);
        lc = lc < 0 ? 0 : (lc > C - 4 ? C - 4 : lc);
        const bool edge_strip = strip == 0 || (strip + 1) * (4 * VL) + 4 * HL >= C;
        const DtBuf bx = dt_buf2g(p.X + img);
        float *const Y0b = p.Yh0 + img * 3 + (int64_t)strip * (VL * 24);
        const unsigned pitch = (unsigned)C * 4u;
        const int last_row = rbase + 2 * (nrow / 2 + G::PRE) - 1 + HH;
        auto ldrow = [&](int u) -> f4 {
            u = u > last_row ? last_row : u;
            u = u < 0 ? -1 - u : u;
            u = u >= R ? 2 * R - 1 - u : u;
            if (KO & 1) u &= 15;
            return dt2d::dt_buf_ld4(bx, (unsigned)lc * 4u, (unsigned)u * pitch);
        };
        auto fix = [&](f4 &v) { if (edge_strip) v = rev ? rev4(v) : v; };
        float h0[M0], h1[M1];
#pragma unroll
        for (int k = 0; k < M0; ++k) h0[k] = p.h0[k];
#pragma unroll
        for (int k = 0; k < M1; ++k) h1[k] = p.h1[k];

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

        for (int t0 = 0; t0 < nst; t0 += PER) {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int t = t0 + k;
                const int r = rbase + 2 * t;
                const f4 in0 = pre[(2 * k) % (2 * P)], in1 = pre[(2 * k + 1) % (2 * P)];
                pre[(2 * k) % (2 * P)] = ldrow(r - HH + WR + 2 * P);
                pre[(2 * k + 1) % (2 * P)] = ldrow(r - HH + WR + 2 * P + 1);
                float wc[4][WR];
#pragma unroll
                for (int j = 0; j < WR; ++j) {
                    const f4 &x = ring[(2 * k + j) % WR];
                    wc[0][j] = x.x; wc[1][j] = x.y; wc[2][j] = x.z; wc[3][j] = x.w;
                }
                const bool in_band = r >= rb && r < rb + nrow;          // uniform
                float hi[2][4];
                f4 lh[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float lo[4], wl[4 + 2 * HH], a_[4], b_[4];
                    if (in_band) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) sym_pair<M0, M1, HH>(&wc[c][q + HH], h0, h1, lo[c], hi[q][c]);
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) lo[c] = sym_one<M0, HH>(&wc[c][q + HH], h0);
                    }
                    row_window<HH>(f4{lo[0], lo[1], lo[2], lo[3]}, wl);
                    if (in_band) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) sym_pair<M0, M1, HH>(&wl[c + HH], h0, h1, a_[c], b_[c]);
                        lh[q] = f4{b_[0], b_[1], b_[2], b_[3]};
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) a_[c] = sym_one<M0, HH>(&wl[c + HH], h0);
                    }
                    xbuf[t & 1][q][HL2 + lane] = f4{a_[0], a_[1], a_[2], a_[3]};
                }
                DTM_LDS_BARRIER();                      // the step's LoLo1 rows are wavefront 1's now
                if (in_band) {
                    f4 hl[2], hh[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float wh[4 + 2 * HH], c_[4], d_[4];
                        row_window<HH>(f4{hi[q][0], hi[q][1], hi[q][2], hi[q][3]}, wh);
#pragma unroll
                        for (int c = 0; c < 4; ++c) sym_pair<M0, M1, HH>(&wh[c + HH], h0, h1, c_[c], d_[c]);
                        hl[q] = f4{c_[0], c_[1], c_[2], c_[3]}; hh[q] = f4{d_[0], d_[1], d_[2], d_[3]};
                    }
                    const Zq a0 = q2c_s(hl[0].x, hl[0].y, hl[1].x, hl[1].y), a1 = q2c_s(hl[0].z, hl[0].w, hl[1].z, hl[1].w);
                    const Zq b0 = q2c_s(hh[0].x, hh[0].y, hh[1].x, hh[1].y), b1 = q2c_s(hh[0].z, hh[0].w, hh[1].z, hh[1].w);
                    const Zq c0q = q2c_s(lh[0].x, lh[0].y, lh[1].x, lh[1].y), c1q = q2c_s(lh[0].z, lh[0].w, lh[1].z, lh[1].w);
                    f4 *o = slab + lane * 6;
                    o[0] = f4{sq * a0.z0r, sq * a0.z0i, sq * b0.z0r, sq * b0.z0i};
                    o[1] = f4{sq * c0q.z0r, sq * c0q.z0i, sq * c0q.z1r, sq * c0q.z1i};
                    o[2] = f4{sq * b0.z1r, sq * b0.z1i, sq * a0.z1r, sq * a0.z1i};
                    o[3] = f4{sq * a1.z0r, sq * a1.z0i, sq * b1.z0r, sq * b1.z0i};
                    o[4] = f4{sq * c1q.z0r, sq * c1q.z0i, sq * c1q.z1r, sq * c1q.z1i};
                    o[5] = f4{sq * b1.z1r, sq * b1.z1i, sq * a1.z1r, sq * a1.z1i};
                }
                {   // stores on every step, dropped whole outside the band: see k_fwd12m
                    const int ro = (KO & 2) ? (r & 15) : r;
                    DT_WAVE_LDS_SYNC();
                    const DtBuf by = dt_buf_n(Y0b + (int64_t)(ro >> 1) * C * 6, in_band ? 96u * nv : 0u);
#pragma unroll
                    for (int m = 0; m < 6; ++m) {
                        const f4 v = slab[6 * HL + lane + 64 * m];
                        dt2d::dt_buf_st4<true>(by, yv + 1024u * m, 0u, v);
                    }
                    DT_WAVE_LDS_SYNC();
                }
                f4 e0 = in0, e1 = in1;
                fix(e0); fix(e1);
                ring[(2 * k) % WR] = e0;
                ring[(2 * k + 1) % WR] = e1;
            }
        }
    } else {
        // ------------------------------------------------------------------ level 2
        float *const Y1b = p.Yh1 + (img / 4) * 3 + (int64_t)strip * (VL * 12);
        float *const L2b = p.LoLo2 + img / 4 + strip * (VL * 2);
        const unsigned l2v = 8u * (unsigned)(lane - HL);
        float S[NP2][4][4];
#pragma unroll
        for (int a = 0; a < NP2; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int v = 0; v < 4; ++v) S[a][c][v] = 0.f;
        for (int t0 = 0; t0 < nst; t0 += 2) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int t = t0 + k;
                const int r = rbase + 2 * t;
                DTM_LDS_BARRIER();                      // the rows of step t are in xbuf[t & 1]
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float rv[4] = {0.f, 0.f, 0.f, 0.f};
                    const f4 *src = &xbuf[t & 1][q][lane];          // lanes l - HL2 .. l + HL2 (the pad either end is never used by an owning lane)
#pragma unroll
                    for (int d = 0; d < 2 * HL2 + 1; ++d) {
                        const f4 x = src[d];
                        const float e[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int j = 4 * d + c, tt = j >> 1;
                            if (j & 1) { rv[1] += p.tb_lo[tt] * e[c]; rv[3] += p.tb_hi[tt] * e[c]; }
                            else       { rv[0] += p.ta_lo[tt] * e[c]; rv[2] += p.ta_hi[tt] * e[c]; }
                        }
                    }
                    const int phi = 2 * k + q;
#pragma unroll
                    for (int a = 0; a < NP2; ++a) {
                        const int tt = (phi + 8 * HL2 - 4 * a) >> 1;
                        const float cl_ = (phi & 1) ? p.tb_lo[tt] : p.ta_lo[tt], ch_ = (phi & 1) ? p.tb_hi[tt] : p.ta_hi[tt];
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            S[a][(phi & 1) ? 2 : 0][v] += cl_ * rv[v];
                            S[a][(phi & 1) ? 3 : 1][v] += ch_ * rv[v];
                        }
                    }
                }
                if (k & 1) {
                    const int i2 = (r - 2) / 4 - HL2;
                    const bool pair_ok = 4 * i2 >= rb && 4 * i2 < rb + nrow;
                    const bool la = p.lo_a_first != 0, ha = p.hi_a_first != 0;
                    float pl[2][4], ph[2][4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        pl[0][v] = la ? S[0][0][v] : S[0][2][v]; pl[1][v] = la ? S[0][2][v] : S[0][0][v];
                        ph[0][v] = ha ? S[0][1][v] : S[0][3][v]; ph[1][v] = ha ? S[0][3][v] : S[0][1][v];
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
                    dt2d::dt_buf_st2<false>(bl0, l2v, 0u, dt2d::f2{llo[0][0], llo[0][1]});
                    dt2d::dt_buf_st2<false>(bl1, l2v, 0u, dt2d::f2{llo[1][0], llo[1][1]});
                    const Zq a = q2c_s(hl2[0][0], hl2[0][1], hl2[1][0], hl2[1][1]);
                    const Zq bq = q2c_s(hh2[0][0], hh2[0][1], hh2[1][0], hh2[1][1]);
                    const Zq c = q2c_s(lh2[0][0], lh2[0][1], lh2[1][0], lh2[1][1]);
                    f4 *o = slab2 + lane * 3;
                    o[0] = f4{sq * a.z0r, sq * a.z0i, sq * bq.z0r, sq * bq.z0i};
                    o[1] = f4{sq * c.z0r, sq * c.z0i, sq * c.z1r, sq * c.z1i};
                    o[2] = f4{sq * bq.z1r, sq * bq.z1i, sq * a.z1r, sq * a.z1i};
                    DT_WAVE_LDS_SYNC();
                    const DtBuf by1 = dt_buf_n(Y1b + (int64_t)io * (C / 4) * 12, pair_ok ? 48u * nv : 0u);
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        const f4 v = slab2[3 * HL + lane + 64 * m];
                        dt2d::dt_buf_st4<false>(by1, yv + 1024u * m, 0u, v);
                    }
                    DT_WAVE_LDS_SYNC();
#pragma unroll
                    for (int a2 = 0; a2 + 1 < NP2; ++a2)
#pragma unroll
                        for (int c2 = 0; c2 < 4; ++c2)
#pragma unroll
                            for (int v = 0; v < 4; ++v) S[a2][c2][v] = S[a2 + 1][c2][v];
#pragma unroll
                    for (int c2 = 0; c2 < 4; ++c2)
#pragma unroll
                        for (int v = 0; v < 4; ++v) S[NP2 - 1][c2][v] = 0.f;
                }
            }
        }
    }
#endif
}

}  // namespace dtm
