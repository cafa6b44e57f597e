// Marching-kernel experiments that are NOT part of libdtcwt_hip.so (measurement only, tools/kbench/march_bench.hip):
//   k_fwd1m   level 1 of the forward alone as a march -- the first prototype: 13.5-16 us with every store instruction
//             removed and the loads served by the caches (i.e. its arithmetic), 68-75 us whole against the tile
//             program's 60: a level-1 kernel is bound by its 343 MB either way, the march only pays once it fuses levels
//   k_fwd12w  levels 1 + 2 as a PAIR of wavefronts per job (level-1 wavefront -> LDS exchange -> level-2 wavefront,
//             128 registers each, four wavefronts per SIMD): 84 us against 85 for the one-wavefront form -- the jobs
//             are bound by the instructions ONE wavefront issues, not by latency hiding
#pragma once
#include "march2d.hpp"

namespace dtm {

#if defined(__HIP_DEVICE_COMPILE__)
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
#endif

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
    int strip, band, b;
    if (!dtm_job(p.jb, blockIdx.x, strip, band, b)) return;
    const int R = p.R, C = p.C;
    const int nv = (C - strip * (4 * VL)) / 4 < VL ? (C - strip * (4 * VL)) / 4 : VL;     // owning lanes
    const int64_t img = (int64_t)b * R * C;
    const int rb = band * p.jb.band_rows;
    const int nrow = R - rb < p.jb.band_rows ? R - rb : p.jb.band_rows;
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
