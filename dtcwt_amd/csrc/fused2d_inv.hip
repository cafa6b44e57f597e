// The two inverse kernels of the float32 2-D plan (fused2d.hip) in a translation unit of their own: they
// are built with -fno-slp-vectorize.  The SLP vectoriser turns the record -> quad-plane arithmetic of the
// gather phase into v_pk_mul / v_pk_fma_f32 pairs, which have no throughput advantage on gfx950 and cost a
// v_mov per operand to line the pairs up; measured in one run (profiles/r02/ab_noslp.txt): k_inv1 63.1 ->
// 60.3 us, k_inv2 32.9 -> 31.9 us at 4096^2, the forward kernels unchanged or slightly slower -- they keep SLP.
#include "common.hpp"
#include "fused2d_tiles.hpp"
#include "fused2d_tiles_v2.hpp"
#include "fused2d_table.hpp"

using namespace dt2d;

namespace {

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// every tile order of tile_of() is a bijection on a grid that is a multiple of 8 x (group size)
inline unsigned grid_for(int ntile, int order = 1) {
    const int q = 8 * (order > 1 ? order : 1);
    return (unsigned)(cdiv(ntile, q) * q);
}

// Level-1 inverse: records copied verbatim into LDS (coalesced 16-byte pieces, all requested
// up front together with the lowpass window), quad-plane samples gathered from them with
// c2q folded in (column parity uniform per wavefront), barrier, column FIR writing y1/y2
// OVER the record buffer, barrier, row pass with 16-byte stores (fused2d_tiles_v2.hpp).
template <class C>
__global__ void __launch_bounds__(DT_NT) k_inv1(Inv1Params p) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_ALIASED];
    const int ntile = p.tilesR * p.tilesC * p.B;
    int t = tile_of(blockIdx.x, ntile, p.xcd_order);
    if (t >= ntile) return;
    int tc, tr, b;
    dt_tile_decode(p, t, tc, tr, b);
    float *srec = smem, *y1 = smem, *y2 = y1 + C::SY, *y3 = y2 + C::SY;     // y3: band-pass variant only
    int r0 = tr * C::TR, c0 = tc * C::TC;
    const float *Yhb = p.Yh + (int64_t)b * (p.R / 2) * (p.C / 2) * 12;
    float wz[C::WN], w1[C::WN], w2[C::WN], w3[C::WN];
    inv1r_fetch<C>(p, wz, threadIdx.x, b, r0, c0);
    inv_rec_stage<C::QR, C::QC>(Yhb, p.R, p.C, srec, r0 - C::HE, c0 - C::HE, threadIdx.x);
    __syncthreads();
    inv1r_gather<C>(p, srec, w1, w2, w3, threadIdx.x, r0, c0);
    __syncthreads();
    inv1r_fir<C>(p, wz, w1, w2, w3, y1, y2, threadIdx.x, y3);
    __syncthreads();
    inv1d_rows<C>(p, y1, y2, threadIdx.x, b, r0, c0, y3);
}

// Level >= 2 inverse: same structure with the polyphase interpolating filters.
// STD: compile-time filter phases + packed FMAs (ifilt4_acc).  112 VGPRs (the scalar form: 72); bounding the
// registers to get the occupancy back spills (5 waves per SIMD: 50 us instead of 31) and buys nothing at 4.
template <class C, bool STD>
__global__ void __launch_bounds__(DT_NT) k_inv2(Inv2Params p) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_ALIASED];
    const int ntile = p.tilesR * p.tilesC * p.B;
    int t = tile_of(blockIdx.x, ntile, p.xcd_order);
    if (t >= ntile) return;
    int tc, tr, b;
    dt_tile_decode(p, t, tc, tr, b);
    float *srec = smem, *y1 = smem, *y2 = y1 + C::SY, *y3 = y2 + C::SY;     // y3: band-pass variant only
    int r0 = tr * C::TR, c0 = tc * C::TC;
    const float *Yhb = p.Yh + (int64_t)b * (p.zr / 2) * (p.zc / 2) * 12;
    float wz[C::WS], w1[C::WS], w2[C::WS], w3[C::WS];
    inv2r_fetch<C>(p, wz, threadIdx.x, b, r0, c0);
    inv_rec_stage<C::QR, C::QC>(Yhb, p.zr, p.zc, srec, r0 + C::ORG, c0 + C::ORG, threadIdx.x);
    __syncthreads();
    inv2r_gather<C>(p, srec, w1, w2, w3, threadIdx.x, r0, c0);
    __syncthreads();
    inv2r_fir<C, false, STD>(p, wz, w1, w2, w3, y1, y2, threadIdx.x, y3);
    __syncthreads();
    inv2_rows<C, STD>(p, y1, y2, threadIdx.x, b, r0, c0, y3);
}

// Level >= 2 inverse with the column phase in two halves (inv2r_gather_half / inv2r_fir_plane): fewer live registers,
// records + one y plane in LDS -> six workgroups per CU.  Not for the band-pass sets (a third plane).
template <class C, bool STD>
__global__ void __launch_bounds__(DT_NT) k_inv2s(Inv2Params p) {
    constexpr int SRECP = (C::LDS_ALIASED + 3) & ~3;
    __shared__ __attribute__((aligned(16))) float smem[SRECP + C::SY];
    const int ntile = p.tilesR * p.tilesC * p.B;
    int t = tile_of(blockIdx.x, ntile, p.xcd_order);
    if (t >= ntile) return;
    int tc, tr, b;
    dt_tile_decode(p, t, tc, tr, b);
    float *srec = smem, *y2 = smem, *y1 = smem + SRECP;
    const int r0 = tr * C::TR, c0 = tc * C::TC;
    const float *Yhb = p.Yh + (int64_t)b * (p.zr / 2) * (p.zc / 2) * 12;
    float wz[C::WS], wa[C::WS], wb[C::WS], wc[C::WS];
    inv2r_fetch<C>(p, wz, threadIdx.x, b, r0, c0);
    inv_rec_stage<C::QR, C::QC>(Yhb, p.zr, p.zc, srec, r0 + C::ORG, c0 + C::ORG, threadIdx.x);
    __syncthreads();
    inv2r_gather_half<C, 0>(p, srec, wa, wb, threadIdx.x, r0, c0);         // lh -> wa, hh -> wb
    inv2r_fir_plane<C, STD>(p, wz, wa, y1, threadIdx.x);                    // y1 = Z (*) g0 + lh (*) g1: a plane of its own
    inv2r_gather_half<C, 1>(p, srec, wc, wc, threadIdx.x, r0, c0);         // hl -> wc
    __syncthreads();                                                        // every record is consumed: y2 goes over them
    inv2r_fir_plane<C, STD>(p, wc, wb, y2, threadIdx.x);                    // y2 = hl (*) g0 + hh (*) g1
    __syncthreads();
    inv2_rows<C, STD>(p, y1, y2, threadIdx.x, b, r0, c0, nullptr);
}

template <class C>
int launch_inv1(Inv1Params &p, hipStream_t s) {
    p.tilesR = cdiv(p.R, C::TR); p.tilesC = cdiv(p.C, C::TC);
    dt_set_tile_magic(p);
    dt_pack_g01<C::M0, C::M1>(p);
    k_inv1<C><<<grid_for(p.tilesR * p.tilesC * p.B, p.xcd_order), DT_NT, 0, s>>>(p);
    return 0;
}
template <class C>
int launch_inv2(Inv2Params &p, hipStream_t s) {
    p.tilesR = cdiv(p.zr, C::TR); p.tilesC = cdiv(p.zc, C::TC);
    dt_set_tile_magic(p);
    // every shipped q-shift set: sum(g0a g0b) > 0 > sum(g1a g1b) (and the band-pass pair like g1) -- compile-time
    // filter phases; anything else takes the run-time flags
    const bool std_set = p.lo_pos && !p.hi_pos && (!C::BP || !p.bp_pos);
    // The column phase in two halves (k_inv2s: six workgroups per CU) where there are workgroups to overlap -- 1000 tiles
    // and more: a 2048^2 lowpass 35.4 -> 33.4 us; round 6: level 3 of a 4096^2 image (a 1024^2 lowpass, 1216 tiles: 4.75 per CU,
    // i.e. TWO rounds of the one-piece kernel's four per CU) in flight on a quarter 41 -> 37.8 us, and one transform at a time
    // 0.179 -> 0.173 ms per step, the levels-2+1 launch behind it starting 5 us earlier (profiles/r06/ab_inv2s_min.txt; the
    // threshold was 2048) -- the one-piece k_inv2 for the small launches, which are latency
    // chains (a 512^2 lowpass: 8.2 us against 8.8), for the band-pass sets (a third plane) and for filters with
    // other phases than the shipped sets.  One sweep decided it (profiles/r03/inv2_phase_stamps.txt, README of
    // profiles/r04); the environment switch of round 3 is gone, and with it the k_inv2s instantiations nothing used.
    if constexpr (!C::BP && C::TR >= 16) {
        if (std_set && p.tilesR * p.tilesC * p.B >= 1000) {
            k_inv2s<C, true><<<grid_for(p.tilesR * p.tilesC * p.B, p.xcd_order), DT_NT, 0, s>>>(p);
            return 0;
        }
    }
    if (std_set) k_inv2<C, true><<<grid_for(p.tilesR * p.tilesC * p.B, p.xcd_order), DT_NT, 0, s>>>(p);
    else k_inv2<C, false><<<grid_for(p.tilesR * p.tilesC * p.B, p.xcd_order), DT_NT, 0, s>>>(p);
    return 0;
}

#define DT_CASE_INV1(TR, TC, RS, A, B) if (m0 == A && m1 == B) return launch_inv1<Inv1RCfg<TR, TC, RS, A, B>>(p, s);
#define DT_CASE_INV2(TR, TC, JS, M) if (m == M) return launch_inv2<Inv2RCfg<TR, TC, JS, M>>(p, s);
#define DT_CASE_INV1_BP(TR, TC, RS, A, B, C2) if (m0 == A && m1 == B && m2 == C2) return launch_inv1<Inv1RCfg<TR, TC, RS, A, B, C2>>(p, s);
#define DT_CASE_INV2_BP(TR, TC, JS, M) if (m == M) return launch_inv2<Inv2RCfg<TR, TC, JS, M, true>>(p, s);

}  // namespace

// m2 / bp: length of the band-pass filter of a 6-vector biort set / a 12-vector q-shift set (0 / false: none)
int dtcwt_dispatch_inv1(int m0, int m1, int m2, Inv1Params &p, hipStream_t s) {
    if (m2) { DT_INV1_BP_TABLE(DT_CASE_INV1_BP) return -3; }
    DT_INV1_TABLE(DT_CASE_INV1) return -3;
}
int dtcwt_dispatch_inv2(int m, bool bp, Inv2Params &p, hipStream_t s, bool small) {
    if (bp) { DT_INV2_BP_TABLE(DT_CASE_INV2_BP) return -3; }
    if (small) { DT_INV2_SMALL_TABLE(DT_CASE_INV2) }
    DT_INV2_TABLE(DT_CASE_INV2) return -3;
}
