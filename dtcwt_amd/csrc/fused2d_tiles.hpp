// Parameter blocks and shared arithmetic of the fused per-level tile programs of the
// float32 2-D DT-CWT (forward and inverse); the tile programs themselves are in
// fused2d_tiles_v2.hpp.
//
// One workgroup (256 threads, 4 wavefronts) owns one output tile of one image and runs
// phases separated by workgroup barriers, everything in between living in LDS:
//
//   forward:  column pass straight from global memory (+halo, symmetric reflection / edge
//             replication done by index math: Lo, Hi planes)  ->  row pass fused with q2c,
//             writing LoLo and whole 48-byte 6-subband records of Yh.
//   inverse:  lowpass window into registers, Yh records verbatim into LDS  ->  gather of
//             the three quad planes (c2q + gains on the fly)  ->  column pass (y1, y2)
//             ->  row pass writing Z.
//
// Each phase is a __host__ __device__ function of (params, LDS pointers, thread id, tile
// origin) so that exactly the same index algebra can be stepped through on the host by
// the test-only emulator (tests/emu/), which is how it was debugged without a GPU.  The
// product only ever calls them from the __global__ wrappers in fused2d.hip / fused3d.hip.
//
// Index algebra: SURVEY.md Appendix A.1-A.4, A.6; reference: dtcwt/numpy/lowlevel.py
// (colfilter :47-80, coldfilt :82-154, colifilt :156-260) and
// dtcwt/numpy/transform2d.py (level loops :112-160, :242-293; q2c :301-322; c2q :324-350).
#pragma once

// Lanes of ONE wavefront hand data to each other through a wave-private LDS slab without a workgroup barrier
// (LDS operations of a wavefront execute in order).  The compiler still has to be told: a wavefront-scope
// release / acquire pair around a convergent no-op, which costs no instruction but stops it from moving the
// reads of all lanes into the divergent block in which some lanes wrote (it did, once the stores that follow
// became unconditional: the lanes outside the block then stored stale registers).
#if defined(__HIP_DEVICE_COMPILE__)
#define DT_WAVE_LDS_SYNC()                                  \
    do {                                                    \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                    \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
#else
#define DT_WAVE_LDS_SYNC() (void)0
#endif

#include <hip/hip_runtime.h>

#include <cstdint>

#define DT_HD __host__ __device__ __forceinline__
#define DT_NT 256                       // threads per workgroup
#define DT_MAXT 40                      // == DTCWT_HIP_MAX_TAPS
typedef float dt_pk2 __attribute__((ext_vector_type(2)));   // one v_pk_fma_f32 operand

namespace dt2d {

struct alignas(16) f4 { float x, y, z, w; };
struct alignas(8) f2 { float x, y; };

// ---- buffer addressing ------------------------------------------------------------------
// A window row is (row base, uniform per wavefront) + (lane offset): with flat `global_load` every address is a
// 64-bit VGPR pair, i.e. one v_lshl_add_u64 (or more) per load -- a third of all vector instructions of the level
// kernels are such integer address arithmetic (profiles/r03/valu_mix.txt).  The buffer instructions of CDNA take the
// base from a 128-bit resource descriptor in scalar registers, ONE 32-bit lane offset (VGPR, bytes) and a scalar
// offset (SGPR, bytes) -- `buffer_load_dword v, v_off, s[rsrc], s_off offen` -- so stepping down the rows of a
// window costs scalar multiplies only.  Offsets are unsigned 32-bit: a descriptor covers 4 GiB from its base, so the
// base is the TILE's (or the image's) origin, re-made per workgroup with scalar arithmetic.  Out-of-range offsets
// read 0 / drop the store, which nothing here relies on.  On the host (emulator) the same calls are pointer
// arithmetic.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float dt_bv2 __attribute__((ext_vector_type(2)));
typedef float dt_bv4 __attribute__((ext_vector_type(4)));
struct DtBuf { __amdgpu_buffer_rsrc_t r; };
// gfx9 family (gfx950 included) raw buffer: DATA_FORMAT = 32 in word 3, stride 0, num_records in bytes
// The base must be the same in every lane; it is pinned into scalar registers here (readfirstlane), otherwise a
// descriptor made inside divergent control flow lands in VGPRs and every load gets a waterfall loop around it.
DT_HD DtBuf dt_buf(const void *base) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    void *ub = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
    DtBuf b; b.r = __builtin_amdgcn_make_buffer_rsrc(ub, 0, (int)0xffffffffu, 0x00020000); return b;
}
DT_HD float dt_buf_ld(const DtBuf &b, unsigned voff, unsigned soff) { return __builtin_amdgcn_raw_buffer_load_b32(b.r, voff, soff, 0); }
DT_HD f2 dt_buf_ld2(const DtBuf &b, unsigned voff, unsigned soff) {
    const dt_bv2 v = __builtin_amdgcn_raw_buffer_load_b64(b.r, voff, soff, 0); return f2{v.x, v.y};
}
DT_HD f4 dt_buf_ld4(const DtBuf &b, unsigned voff, unsigned soff) {
    const dt_bv4 v = __builtin_amdgcn_raw_buffer_load_b128(b.r, voff, soff, 0); return f4{v.x, v.y, v.z, v.w};
}
// NT: the non-temporal hint (aux bit 1 on gfx94x / gfx950), as __builtin_nontemporal_store gives flat stores
template <bool NT = false>
DT_HD void dt_buf_st2(const DtBuf &b, unsigned voff, unsigned soff, const f2 &v) {
    __builtin_amdgcn_raw_buffer_store_b64(dt_bv2{v.x, v.y}, b.r, voff, soff, NT ? 2 : 0);
}
template <bool NT = false>
DT_HD void dt_buf_st4(const DtBuf &b, unsigned voff, unsigned soff, const f4 &v) {
    __builtin_amdgcn_raw_buffer_store_b128(dt_bv4{v.x, v.y, v.z, v.w}, b.r, voff, soff, NT ? 2 : 0);
}
#else
struct DtBuf { const char *p; };
DT_HD DtBuf dt_buf(const void *base) { DtBuf b; b.p = static_cast<const char *>(base); return b; }
DT_HD float dt_buf_ld(const DtBuf &b, unsigned voff, unsigned soff) { return *reinterpret_cast<const float *>(b.p + voff + (size_t)soff); }
DT_HD f2 dt_buf_ld2(const DtBuf &b, unsigned voff, unsigned soff) { return *reinterpret_cast<const f2 *>(b.p + voff + (size_t)soff); }
DT_HD f4 dt_buf_ld4(const DtBuf &b, unsigned voff, unsigned soff) { return *reinterpret_cast<const f4 *>(b.p + voff + (size_t)soff); }
template <bool NT = false>
DT_HD void dt_buf_st2(const DtBuf &b, unsigned voff, unsigned soff, const f2 &v) { *reinterpret_cast<f2 *>(const_cast<char *>(b.p) + voff + (size_t)soff) = v; }
template <bool NT = false>
DT_HD void dt_buf_st4(const DtBuf &b, unsigned voff, unsigned soff, const f4 &v) { *reinterpret_cast<f4 *>(const_cast<char *>(b.p) + voff + (size_t)soff) = v; }
#endif

// Half-sample symmetric reflection, ONE bounce: valid for -n <= u < 2n.  Branch-free on
// purpose: a data-dependent branch per index splits the load sequences of the tile
// programs into basic blocks and serialises their memory latencies (profiles/ round 1).
// The plan only uses the fused kernels when every level is at least DT_MIN_FUSED_DIM
// samples wide (halos are < 40), so one bounce always suffices; smaller images take the
// generic kernels (filters.hip), whose reflection is the multi-bounce modulo form.
#define DT_MIN_FUSED_DIM 40
DT_HD int reflect_i(int u, int n) {
    u = u < 0 ? -1 - u : u;
    u = u >= n ? 2 * n - 1 - u : u;
    // rows/columns of partial edge tiles beyond one bounce are computed but never stored:
    // keep their addresses inside the array
    return u < 0 ? 0 : (u > n - 1 ? n - 1 : u);
}
DT_HD int clamp_i(int u, int lo, int hi) { return u < lo ? lo : (u > hi ? hi : u); }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// ======================================================================================
// Level 1 forward: odd-length biort filters h0 (M0 taps), h1 (M1 taps), no decimation.
// ======================================================================================
struct Fwd1Params {
    const float *X;       // [B][inR][inC]
    float *LoLo;          // [B][LR][LC]
    float *Yh;            // [B][LR/2][LC/2][12 floats]
    int B, inR, inC;      // real input size (may be odd)
    int LR, LC;           // even-extended logical size (transform2d.py:86-94)
    int tilesR, tilesC;
    unsigned mgC, mgRC;   // dt_tile_magic(): multipliers that replace the divisions by tilesC and tilesC * tilesR in the tile decode (0: divide)
    int xcd_order;        // 1: XCD-contiguous tile runs, 0: linear order
    float h0[DT_MAXT], h1[DT_MAXT];
    float h2[DT_MAXT];    // band-pass biort (6-vector sets): the diagonal subbands (transform2d.py:116-129)
    // (h0, h1) pairs by window offset, the shorter filter zero-padded: element d of a window of 2 HH + 1 samples
    // meets c01[2d] (lowpass) and c01[2d + 1] (highpass) in one packed FMA; filled by dt_pack_c01 in the launch
    float c01[2 * DT_MAXT] __attribute__((aligned(8)));
};

template <int M0, int M1>
inline void dt_pack_c01(Fwd1Params &p) {
    constexpr int H0 = M0 / 2, H1 = M1 / 2, HH = H0 > H1 ? H0 : H1;
    for (int d = 0; d < DT_MAXT; ++d) {
        const int k0 = HH + H0 - d, k1 = HH + H1 - d;
        p.c01[2 * d] = (k0 >= 0 && k0 < M0) ? p.h0[k0] : 0.f;
        p.c01[2 * d + 1] = (k1 >= 0 && k1 < M1) ? p.h1[k1] : 0.f;
    }
}

// q2c of quad (a b / c d): z0 = s((a-d) + j(b+c)), z1 = s((a+d) + j(b-c))   (A.4)
DT_HD void q2c_pair(float a, float b, float c, float d, float &z0r, float &z0i, float &z1r,
                    float &z1i) {
    const float s = 0.70710678118654752440f;
    z0r = s * (a - d); z0i = s * (b + c);
    z1r = s * (a + d); z1i = s * (b - c);
}

// Store one Yh record: subband slots 0..5 = HLz0, HHz0, LHz0, LHz1, HHz1, HLz1 where
// HL = hi on axis 0 / lo on axis 1 (transform2d.py:122-127: slots [0,5], [2,3], [1,4]).
DT_HD void store_record(float *rec, const float (&hl)[2][2], const float (&lh)[2][2],
                        const float (&hh)[2][2]) {
    float a0r, a0i, a1r, a1i, b0r, b0i, b1r, b1i, c0r, c0i, c1r, c1i;
    q2c_pair(hl[0][0], hl[0][1], hl[1][0], hl[1][1], a0r, a0i, a1r, a1i);
    q2c_pair(hh[0][0], hh[0][1], hh[1][0], hh[1][1], b0r, b0i, b1r, b1i);
    q2c_pair(lh[0][0], lh[0][1], lh[1][0], lh[1][1], c0r, c0i, c1r, c1i);
    f4 *o = reinterpret_cast<f4 *>(rec);
    o[0] = f4{a0r, a0i, b0r, b0i};
    o[1] = f4{c0r, c0i, c1r, c1i};
    o[2] = f4{b1r, b1i, a1r, a1i};
}

// ======================================================================================
// Level >= 2 forward: even-length q-shift pairs, decimation by 2 per axis (coldfilt).
// ======================================================================================
struct Fwd2Params {
    const float *X;       // [B][inR][inC]   LoLo of the previous level
    float *LoLo;          // [B][LR/2][LC/2]
    float *Yh;            // [B][LR/4][LC/4][12]
    int B, inR, inC;
    int padR, padC;       // 0/1: one replicated row/col each side (transform2d.py:134-140)
    int LR, LC;           // inR + 2 padR, multiples of 4
    int tilesR, tilesC;
    unsigned mgC, mgRC;   // dt_tile_magic(): multipliers that replace the divisions by tilesC and tilesC * tilesR in the tile decode (0: divide)
    int xcd_order;        // 1: XCD-contiguous tile runs, 0: linear order
    int stream_records;   // 1: records written with the non-temporal hint (arrays too big to stay cached)
    int lo_a_first, hi_a_first;   // sign of sum(ha*hb) of the lo / hi pair (lowlevel.py:143)
    // coldfilt(X, ha, hb) is called as (h0b, h0a) and (h1b, h1a): "a" arrays hold the
    // first argument, "b" the second.
    float l_a[DT_MAXT], l_b[DT_MAXT], h_a[DT_MAXT], h_b[DT_MAXT];
    // (l_a[i], h_a[i]) and (l_b[i], h_b[i]) interleaved, 8-byte aligned: the tap pairs of dfilt_pair2, filled by
    // dt_pack_lh() in the launch functions
    float lh_a[2 * DT_MAXT] __attribute__((aligned(8))), lh_b[2 * DT_MAXT] __attribute__((aligned(8)));
    // band-pass q-shift (12-vector sets): coldfilt(X, h2b, h2a) for the diagonal subbands (:145-155)
    int bp_a_first;
    float b_a[DT_MAXT], b_b[DT_MAXT];
};

// One (A, B) pair from a 2M window w[0..2M) whose element j is logical sample
// 4i - M + 2 + j (A.2):  A = sum_k ha[2k] w[2M-2-4k] + ha[2k+1] w[2M-4-4k]
//                        B = sum_k hb[2k] w[2M-1-4k] + hb[2k+1] w[2M-3-4k]
// The lowpass and the highpass pair of a level >= 2 forward filter over the SAME window as packed FMAs: lane 0
// of a v_pk_fma_f32 carries the l_a / l_b chain, lane 1 the h_a / h_b chain, the window sample is broadcast.
// Half the FMA instructions of two dfilt_pair calls (the forward q-shift kernels spend 75-85 % of their time
// with the VALU busy: profiles/r02/pmc_valu_2d.txt).
// The (lowpass, highpass) tap pairs come INTERLEAVED from the parameter block (lh_a / lh_b: one aligned 8-byte
// scalar load per pair).  Forming the pairs in the kernel from the separate l_* / h_* kernel-argument arrays made
// the compiler copy those arrays to scratch and fetch the pairs from there (k_fwd2: 21.6 -> 66 us).
template <class P>
inline void dt_pack_lh(P &p) {
    for (int i = 0; i < DT_MAXT; ++i) {
        p.lh_a[2 * i] = p.l_a[i]; p.lh_a[2 * i + 1] = p.h_a[i];
        p.lh_b[2 * i] = p.l_b[i]; p.lh_b[2 * i + 1] = p.h_b[i];
    }
}

template <int M>
DT_HD void dfilt_pair2(const float *w, const float *lha, const float *lhb, float &Al, float &Bl, float &Ah, float &Bh) {
    const dt_pk2 *ta = reinterpret_cast<const dt_pk2 *>(lha), *tb = reinterpret_cast<const dt_pk2 *>(lhb);
    dt_pk2 a = {0.f, 0.f}, b = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < M / 2; ++k) {
        a += ta[2 * k] * w[2 * M - 2 - 4 * k];
        a += ta[2 * k + 1] * w[2 * M - 4 - 4 * k];
        b += tb[2 * k] * w[2 * M - 1 - 4 * k];
        b += tb[2 * k + 1] * w[2 * M - 3 - 4 * k];
    }
    Al = a.x; Ah = a.y; Bl = b.x; Bh = b.y;
}

template <int M>
DT_HD void dfilt_pair(const float *w, const float *ha, const float *hb, float &A, float &B) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < M / 2; ++k) {
        a += ha[2 * k] * w[2 * M - 2 - 4 * k];
        a += ha[2 * k + 1] * w[2 * M - 4 - 4 * k];
        b += hb[2 * k] * w[2 * M - 1 - 4 * k];
        b += hb[2 * k + 1] * w[2 * M - 3 - 4 * k];
    }
    A = a; B = b;
}

// ======================================================================================
// Level 1 inverse: odd-length g0 (M0 taps), g1 (M1 taps), no interpolation.
//   y1 = colfilter(Z, g0) + colfilter(lh, g1);  y2 = colfilter(hl, g0) + colfilter(hh, g1)
//   X  = rowfilter(y1, g0) + rowfilter(y2, g1)            (transform2d.py:275-293)
// ======================================================================================
struct Inv1Params {
    const float *Z;       // [B][R][C]
    const float *Yh;      // [B][R/2][C/2][12]
    float *X;             // [B][R][C]
    int B, R, C;
    int tilesR, tilesC;
    unsigned mgC, mgRC;   // dt_tile_magic(): multipliers that replace the divisions by tilesC and tilesC * tilesR in the tile decode (0: divide)
    int xcd_order;        // 1: XCD-contiguous tile runs, 0: linear order
    float g[6];           // gain_mask column * sqrt(1/2)
    float g0[DT_MAXT], g1[DT_MAXT];
    float g2[DT_MAXT];    // band-pass biort: y2bp = colfilter(hh, g2), third row filter (:283-291)
    // (g0, g1) pairs by window offset for the row pass (dt_pack_g01 in the launch functions): sample d of a window
    // of 2 HH + 1 interleaved (y1, y2) pairs meets (g01[2d], g01[2d + 1]) in one packed FMA
    float g01[2 * DT_MAXT] __attribute__((aligned(8)));
};

template <int M0, int M1>
inline void dt_pack_g01(Inv1Params &p) {
    constexpr int H0 = M0 / 2, H1 = M1 / 2, HH = H0 > H1 ? H0 : H1;
    for (int d = 0; d < DT_MAXT; ++d) {
        const int k0 = HH + H0 - d, k1 = HH + H1 - d;
        p.g01[2 * d] = (k0 >= 0 && k0 < M0) ? p.g0[k0] : 0.f;
        p.g01[2 * d + 1] = (k1 >= 0 && k1 < M1) ? p.g1[k1] : 0.f;
    }
}

// ======================================================================================
// Level >= 2 inverse: even-length q-shift pairs, interpolation by 2 per axis (colifilt).
//   y1 = colifilt(Z, g0b, g0a) + colifilt(lh, g1b, g1a)
//   y2 = colifilt(hl, g0b, g0a) + colifilt(hh, g1b, g1a)
//   Z' = rowifilt(y1, g0b, g0a) + rowifilt(y2, g1b, g1a), cropped by 1 where the forward
//   level padded                                               (transform2d.py:242-273)
// ======================================================================================
struct Inv2Params {
    const float *Z;       // [B][zr][zc]
    const float *Yh;      // [B][zr/2][zc/2][12]
    float *Out;           // [B][2zr - 2cropR][2zc - 2cropC]
    int B, zr, zc;
    int cropR, cropC;     // 0/1
    int tilesR, tilesC;
    unsigned mgC, mgRC;   // dt_tile_magic(): multipliers that replace the divisions by tilesC and tilesC * tilesR in the tile decode (0: divide)
    int xcd_order;        // 1: XCD-contiguous tile runs, 0: linear order
    int lo_pos, hi_pos;   // sum(ha*hb) > 0 of the g0 / g1 pair
    float g[6];
    // colifilt(X, ha, hb) is called as (g0b, g0a) and (g1b, g1a)
    float l_a[DT_MAXT], l_b[DT_MAXT], h_a[DT_MAXT], h_b[DT_MAXT];
    // band-pass q-shift: colifilt(., g2b, g2a) on the diagonal plane and as third row filter (:250-271)
    int bp_pos;
    float b_a[DT_MAXT], b_b[DT_MAXT];
};

// Four output phases 4j..4j+3 from window w (element j' = sample 2j + ORG + j')   (A.3)
template <class C>
DT_HD void ifilt4(const float *w, const float *ha, const float *hb, int pos, float (&y)[4]) {
    float y0 = 0.f, y1 = 0.f, y2 = 0.f, y3 = 0.f;
    if (C::ODD) {
        // offsets from 2j: ta = m2 - 2k (pos) ; tb = m2 - 2k - 1  -> w index = off - ORG
#pragma unroll
        for (int k = 0; k < C::M2; ++k) {
            float hi = w[C::M - 1 - 2 * k], lo = w[C::M - 2 - 2 * k];
            float xa = pos ? hi : lo, xb = pos ? lo : hi;
            y0 += ha[2 * k] * xb;          // hao
            y1 += hb[2 * k] * xa;          // hbo
            y2 += ha[2 * k + 1] * xb;      // hae
            y3 += hb[2 * k + 1] * xa;      // hbe
        }
    } else {
        // offsets from 2j: t = m2 + 1 - 2k ; w index = off + m2
#pragma unroll
        for (int k = 0; k < C::M2; ++k) {
            float t0 = w[C::M + 1 - 2 * k];     // t
            float t1 = w[C::M - 2 * k];         // t - 1
            float t2 = w[C::M - 1 - 2 * k];     // t - 2
            float t3 = w[C::M - 2 - 2 * k];     // t - 3
            float xa = pos ? t0 : t1, xb = pos ? t1 : t0;        // ta, tb
            float xa2 = pos ? t2 : t3, xb2 = pos ? t3 : t2;      // ta-2, tb-2
            y0 += ha[2 * k + 1] * xb2;     // hae X[tb-2]
            y1 += hb[2 * k + 1] * xa2;     // hbe X[ta-2]
            y2 += ha[2 * k] * xb;          // hao X[tb]
            y3 += hb[2 * k] * xa;          // hbo X[ta]
        }
    }
    y[0] = y0; y[1] = y1; y[2] = y2; y[3] = y3;
}

// The same four phases ACCUMULATED into two packed pairs, P = (y0, y2) and Q = (y1, y3): with an odd half
// length the two outputs of a pair multiply the SAME window sample by two neighbouring taps, which is one
// v_pk_fma_f32 (taps as a scalar register pair, the sample broadcast) instead of two v_fma_f32 -- the level >= 2
// inverse kernels are bound by VALU issue (761 VALU instructions per wavefront, a third of them FMAs, = 27 us of
// the 32 us of k_inv2 at 4096^2).  POS >= 0: sum(ha * hb) > 0 known at compile time (every shipped q-shift set
// has it 1 for the g0 pair and 0 for the g1 pair), which also removes the per-sample selects; POS < 0: `pos`.
template <class C, int POS>
DT_HD void ifilt4_acc(const float *w, const float *ha, const float *hb, int pos, dt_pk2 &P, dt_pk2 &Q) {
    const bool ps = POS < 0 ? pos != 0 : POS != 0;
    if (C::ODD) {
#pragma unroll
        for (int k = 0; k < C::M2; ++k) {
            const float hi = w[C::M - 1 - 2 * k], lo = w[C::M - 2 - 2 * k];
            const float xa = ps ? hi : lo, xb = ps ? lo : hi;
            P += dt_pk2{ha[2 * k], ha[2 * k + 1]} * dt_pk2{xb, xb};
            Q += dt_pk2{hb[2 * k], hb[2 * k + 1]} * dt_pk2{xa, xa};
        }
    } else {
        float y[4];
        ifilt4<C>(w, ha, hb, ps, y);
        P += dt_pk2{y[0], y[2]};
        Q += dt_pk2{y[1], y[3]};
    }
}

// STD: the g0 pair has sum(ha*hb) > 0 and the g1 (and band-pass) pair < 0 -- true of every shipped q-shift set --
// known at compile time (see ifilt4_acc); otherwise the flags of the parameter block decide at run time
template <class C, bool STD = false>
DT_HD void inv2_rows(const Inv2Params &p, const float *y1, const float *y2, int tid, int b,
                     int r0, int c0, const float *y3 = nullptr) {
    constexpr int NJ = C::TC / 2;
    constexpr int LP = STD ? 1 : -1, HP = STD ? 0 : -1;
    const int OR = 2 * p.zr - 2 * p.cropR, OC = 2 * p.zc - 2 * p.cropC;
    for (int task = tid; task < 2 * C::TR * NJ; task += DT_NT) {
        int r = task / NJ, jl = task - r * NJ;
        int Rl = 2 * r0 + r;                 // logical output row
        int Cl = 2 * c0 + 4 * jl;            // logical output col of phase 0
        if (Rl >= 2 * p.zr || Cl >= 2 * p.zc) continue;
        int Rw = Rl - p.cropR;
        if (Rw < 0 || Rw >= OR) continue;
        float w[C::WN];
        dt_pk2 P = {0.f, 0.f}, Q = {0.f, 0.f};
        const f2 *pa = reinterpret_cast<const f2 *>(y1 + r * C::NC + 2 * jl);
        const f2 *pb = reinterpret_cast<const f2 *>(y2 + r * C::NC + 2 * jl);
#pragma unroll
        for (int j = 0; j < C::WN / 2; ++j) { f2 v = pa[j]; w[2 * j] = v.x; w[2 * j + 1] = v.y; }
        ifilt4_acc<C, LP>(w, p.l_a, p.l_b, p.lo_pos, P, Q);
#pragma unroll
        for (int j = 0; j < C::WN / 2; ++j) { f2 v = pb[j]; w[2 * j] = v.x; w[2 * j + 1] = v.y; }
        ifilt4_acc<C, HP>(w, p.h_a, p.h_b, p.hi_pos, P, Q);
        if (C::BP) {            // third row filter on the band-pass plane
            const f2 *pc = reinterpret_cast<const f2 *>(y3 + r * C::NC + 2 * jl);
#pragma unroll
            for (int j = 0; j < C::WN / 2; ++j) { f2 v = pc[j]; w[2 * j] = v.x; w[2 * j + 1] = v.y; }
            ifilt4_acc<C, HP>(w, p.b_a, p.b_b, p.bp_pos, P, Q);
        }
        const float a[4] = {P.x, Q.x, P.y, Q.y};
        float *O = p.Out + ((int64_t)b * OR + Rw) * OC;
        if (p.cropC == 0 && (OC & 3) == 0) {
            *reinterpret_cast<f4 *>(O + Cl) = f4{a[0], a[1], a[2], a[3]};
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int cw = Cl + e - p.cropC;
                if (cw >= 0 && cw < OC) O[cw] = a[e];
            }
        }
    }
}

// XCD-aware tile order: consecutive workgroup ids round-robin over the 8 XCDs
// (MI355X_MICROARCH.md: block b -> XCD b % 8), so give each XCD a contiguous run of
// tiles; neighbouring tiles then share halos through the same 4 MiB L2.
DT_HD int xcd_tile(int bid, int ntiles) {
    int per = (ntiles + 7) / 8;
    int t = (bid % 8) * per + bid / 8;
    return t;       // may be >= ntiles: caller must skip
}
// Measured on MI355X (tools/kbench): for these write-heavy kernels the plain linear order
// is FASTER than the XCD-contiguous one (82.6 vs 93.1 us for level-1 forward at 4096^2) --
// neighbouring tiles in flight together keep the HBM write streams dense -- so linear is
// the default and the remap stays selectable (DTCWT_HIP_XCD_ORDER=1) for experiments.
// xcd_order > 1: groups of that many consecutive tiles share an XCD (and its L2) while the groups themselves
// stay in linear order -- the halo columns two horizontally adjacent tiles both read are fetched once, and
// the write streams stay as dense as in linear order.
DT_HD int tile_of(int bid, int ntiles, int xcd_order) {
    if (xcd_order <= 0) return bid;
    if (xcd_order == 1) return xcd_tile(bid, ntiles);
    const int g = xcd_order, x = bid % 8, i = bid / 8;
    if (g == 8) return (((i >> 3) * 8 + x) << 3) + (i & 7);      // the default group size: shifts, no division
    return ((i / g) * 8 + x) * g + i % g;       // may be >= ntiles: caller must skip
}

// Tile index -> (column tile, row tile, image).  Every wavefront of every workgroup used to open with two integer
// divisions by run-time values (~150 scalar instructions and three v_rcp round trips before the first load is
// issued: ~0.3 us of every workgroup's life, and most of what a coarse-level launch of one workgroup per CU costs
// beyond its memory latency).  The host hands over ceil(2^32 / d) instead: q = (n * magic) >> 32 is exact for
// n * d < 2^32 (dt_tile_magic checks that; otherwise, and for d = 1, magic = 0 and the kernel divides).
inline unsigned dt_tile_magic(int d, int64_t nmax) {
    if (d <= 1 || nmax * (int64_t)d >= ((int64_t)1 << 32)) return 0u;
    return (unsigned)((((uint64_t)1 << 32) + (uint64_t)d - 1) / (uint64_t)d);
}
DT_HD int dt_fdiv(int n, int d, unsigned magic) {
    return magic ? (int)(((unsigned long long)(unsigned)n * magic) >> 32) : n / d;
}
template <class P>
inline void dt_set_tile_magic(P &p) {
    const int64_t ntile = (int64_t)p.tilesR * p.tilesC * p.B + 64;      // + the surplus blocks of the rounded-up grid
    p.mgC = dt_tile_magic(p.tilesC, ntile);
    p.mgRC = dt_tile_magic(p.tilesC * p.tilesR, ntile);
}
template <class P>
DT_HD void dt_tile_decode(const P &p, int t, int &tc, int &tr, int &b) {
    const int row = dt_fdiv(t, p.tilesC, p.mgC);          // t / tilesC
    tc = t - row * p.tilesC;
    b = dt_fdiv(t, p.tilesC * p.tilesR, p.mgRC);
    tr = row - b * p.tilesR;
}

}  // namespace dt2d
