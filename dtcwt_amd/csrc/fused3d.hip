// Fused float32 3-D DT-CWT level kernels: __global__ wrappers of the marching tile
// programs in fused3d_tiles.hpp behind dtcwt_hip_fwd3_level1 (include/dtcwt_hip.h).
//
// Replaces one `_level1_xfm` of dtcwt/numpy/transform3d.py:208-289 (three axis passes
// over the whole volume in Python slice loops plus seven cube2c packings) by a single
// launch that reads the volume once and writes the lowpass volume and the 28-subband
// highpass records once.
#include "common.hpp"
#include "fused2d_table.hpp"
#include "fused3d_tiles.hpp"
#include "fused3d_inv_tiles.hpp"
#include "fused3d_march.hpp"
#include "fused3d_long.hpp"

using namespace dt3d;

namespace {

// Workgroups are handed to the 8 XCDs round-robin (block b -> XCD b % 8), each with an L2 of its own.  In linear
// order the tiles that share halos (k- and j-neighbours of a slice) land on eight different L2s and every one of
// them fetches the shared lines again: k_fwd3_l1 read 3.1 x its input, pass A of level 2 2.5 x.  With `ntile_xcd`
// = the tile count, XCD x instead takes the x-th eighth of the linear order -- whole (j, k) planes of tiles.
// 0: linear order.  The grid is rounded up to a multiple of 8; surplus blocks return.
__device__ __forceinline__ int xcd_tile3(int ntile_xcd) {
    const int bid = blockIdx.x;
    if (!ntile_xcd) return bid;
    const int per = (ntile_xcd + 7) / 8;
    const int t = (bid & 7) * per + (bid >> 3);
    return ((bid >> 3) < per && t < ntile_xcd) ? t : -1;
}
// bit mask of the kernels that take the XCD order: 1 k_fwd3_l1, 2 k_fwd3_l2_planes,
// 4 k_inv3_axis0 / k_inv3_l1_axis02, 8 (was k_inv3_l1_planes: unused), 16 k_inv3_l2_planes.  Measured at 256^3 (profiles/r02/xcd3d.txt): the
// forward kernels gain (pass A of level 2: 46 -> 34 us; the transform 252 -> 240 us); so do the plane passes of
// the inverse once its level 1 runs in slabs (345 -> 322 us); the inverse march (4) does not.
enum { XCD3_FWD_L1 = 1, XCD3_FWD_PLANES = 2, XCD3_INV_AXIS0 = 4, XCD3_INV_L2_PLANES = 16 };
inline bool xcd3_enabled(int bit) {
    constexpr int mask = XCD3_FWD_L1 | XCD3_FWD_PLANES | XCD3_INV_L2_PLANES;          // (the sweep's switch, DTCWT_HIP_XCD3D, is gone)
    return (mask & bit) != 0;
}
inline unsigned xcd3_grid(int ntile, int bit) { return xcd3_enabled(bit) ? (unsigned)(8 * ((ntile + 7) / 8)) : (unsigned)ntile; }
inline int xcd3_arg(int ntile, int bit) { return xcd3_enabled(bit) ? ntile : 0; }

// the march of one workgroup; FULL: the tile lies inside the volume
template <class C, bool FULL>
__device__ __forceinline__ void fwd3_l1_march(const Fwd3L1Params &p, float *smem, int j0, int k0, int i0, int iend) {
    float *S0 = smem, *S1 = smem + C::S0F, *stage = S1 + C::S1F, *XR = smem + C::XR0;
    const int tid = threadIdx.x;
    Fwd3L1State<C> st;
    f3l1_init<C>(p, st, tid, j0, k0);
    f3l1_prologue<C>(p, st, XR, tid, i0);
    f3l1_prefetch<C>(p, st, tid, i0);
    // settle the prologue loads here so that no wait on them lands inside the march (where
    // it would also drain the stores in flight): s_waitcnt vmcnt(0)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    int rot = 0;
    for (int i = i0; i < iend; i += 2) {             // chunks start and end on even slices
        f3l1_axis0<C, 0>(p, st, S0, XR, tid, rot);
        __syncthreads();
        f3l1_axis2<C>(p, S0, S1, tid);
        __syncthreads();
        f3l1_axis1<C, FULL>(p, st.ev, S1, tid, i, j0, k0);
        f3l1_axis0<C, 1>(p, st, S0, XR, tid, rot);
        __syncthreads();
        f3l1_axis2<C>(p, S0, S1, tid);
        __syncthreads();
        f3l1_rotate2<C>(st, XR, tid, rot);
        rot = rot + 2 >= C::MR ? rot + 2 - C::MR : rot + 2;
        f3l1_prefetch<C>(p, st, tid, i + 2);         // ahead of this pair's stores (reflected past the end: harmless)
        float od[8][4];
        f3l1_axis1<C, FULL>(p, od, S1, tid, i + 1, j0, k0);
#pragma unroll
        for (int pass = 0; pass < C::SP; ++pass) {
            f3l1_pack_stage<C>(st.ev, od, stage, tid, pass);
            f3l1_pack_flush<C, FULL>(p, stage, tid, pass, i + 1, j0, k0);
        }
    }
}

template <class C>
__global__ void __launch_bounds__(C::NT, 2) k_fwd3_l1(Fwd3L1Params p, int ntile_xcd) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_FLOATS];
    const int bid = xcd_tile3(ntile_xcd);
    if (bid < 0) return;
    const int tk = bid % p.tilesK, tj = (bid / p.tilesK) % p.tilesJ, ch = bid / (p.tilesK * p.tilesJ);
    const int j0 = tj * C::TJ, k0 = tk * C::TK, i0 = ch * p.chunk;
    const int iend = min(i0 + p.chunk, p.n0);
    if (j0 + C::TJ <= p.n1 && k0 + C::TK <= p.n2)
        fwd3_l1_march<C, true>(p, smem, j0, k0, i0, iend);
    else
        fwd3_l1_march<C, false>(p, smem, j0, k0, i0, iend);
}

// Level >= 2, pass A: the 2-D level-2 tile program over every slice, four planes out.
template <class C>
__global__ void __launch_bounds__(DT_NT) k_fwd3_l2_planes(dt2d::Fwd2Params p, float *planes, int64_t pstride, int ntile_xcd) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_FLOATS];
    const int t = xcd_tile3(ntile_xcd);
    if (t < 0) return;
    int tc, tr, b;
    dt2d::dt_tile_decode(p, t, tc, tr, b);
    float *sLo = smem, *sHi = sLo + C::SL;
    const int r0 = tr * C::TR, c0 = tc * C::TC;
    dt2d::fwd2d_cols<C>(p, sLo, sHi, threadIdx.x, b, r0, c0);
    __syncthreads();
    for (int base = 0; base < C::TI * C::TJ; base += DT_NT)
        fwd2p_rows<C>(p, sLo, sHi, planes, pstride, threadIdx.x, base, b, r0, c0);
}

// Level >= 2, pass B: axis-0 decimating filters + cube2c; CPW cells per wavefront (f3l2_axis0_stage).
template <int M, int NT, int CPW>
__global__ void __launch_bounds__(NT) k_fwd3_l2_axis0(Fwd3L2Params p) {
    __shared__ __attribute__((aligned(16))) float slab[(NT / 64) * CPW * REC_LDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *ws = slab + wave * CPW * REC_LDS;
    const int cl = lane & (CPW - 1);                               // the lane's cell within the wavefront
    const int first = p.id0 + ((int)blockIdx.x * (NT / 64) + wave) * CPW;
    if (CPW == 64) f3l2_axis0_stage<M, 4>(p, first + cl, ws + cl * REC_LDS);
    else f3l2_axis0_stage<M, 2>(p, first + cl, ws + cl * REC_LDS, 2 * (lane >> 5));
    DT_WAVE_LDS_SYNC();
    f3l2_axis0_flush<CPW>(p, first, lane, ws);
}

// Level >= 2, pass B, inner layers: two cells along axis 0 per lane pair (f3l2_axis0_pair_stage); one wavefront per
// workgroup = 32 cell columns x layers (2 u, 2 u + 1).
template <int M>
__global__ void __launch_bounds__(64) k_fwd3_l2_axis0_pair(Fwd3L2Params p, int u0, int colblocks, int ninner, int lo_cells, int hi_first) {
    __shared__ __attribute__((aligned(16))) float slab[64 * REC_LDS];
    const int lane = threadIdx.x;
    const int nbound = (int)gridDim.x - ninner;
    const int w = (int)blockIdx.x - nbound;         // the (slow, reflecting) boundary blocks are dealt FIRST: they then run beside the inner ones
    if (w >= 0) {
        const int cb = w % colblocks, u = u0 + w / colblocks;
        f3l2_axis0_pair_stage<M>(p, u, cb * 32 + (lane & 31), lane >> 5, slab, lane & 31);
        DT_WAVE_LDS_SYNC();
        f3l2_axis0_pair_flush(p, u, cb * 32, lane, slab);
    } else {
        // the boundary layers ride along in the same launch, 64 cells per wavefront with the one-cell program: cells
        // [0, lo_cells) and [hi_first, end)
        int first = (int)blockIdx.x * 64;
        if (first >= lo_cells) first = hi_first + (first - lo_cells);
        f3l2_axis0_stage<M, 4>(p, first + lane, slab + lane * REC_LDS);
        DT_WAVE_LDS_SYNC();
        f3l2_axis0_flush<64>(p, first, lane, slab);
    }
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

template <class C>
int launch_l2_planes(dt2d::Fwd2Params &p, float *planes, int64_t pstride, hipStream_t s) {
    p.tilesR = cdiv(p.LR / 2, C::TR); p.tilesC = cdiv(p.LC / 2, C::TC);
    dt2d::dt_set_tile_magic(p);
    dt2d::dt_pack_lh(p);
    const int ntile = p.tilesR * p.tilesC * p.B;
    k_fwd3_l2_planes<C><<<xcd3_grid(ntile, XCD3_FWD_PLANES), DT_NT, 0, s>>>(p, planes, pstride, xcd3_arg(ntile, XCD3_FWD_PLANES));
    return 0;
}
// Volume slices are narrow (64 .. 512 columns): when the wide tile of the 2-D table would
// overhang the plane by more than a tenth, take the narrow one.
inline int64_t padded(int n, int t) { return (int64_t)cdiv(n, t) * t; }
inline bool narrow_wins(int n, int wide, int narrow) { return 10 * padded(n, narrow) < 9 * padded(n, wide); }

template <int TR, int TC, int PS, int M>
void launch_l2_planes_best(dt2d::Fwd2Params &p, float *planes, int64_t pstride, hipStream_t s) {
    if constexpr (M == 10) {
        // slices whose lowpass is a multiple of 64 columns wide: 16 x 64 tiles (a full round of row-pass tasks and
        // 256-byte plane rows per tile; 256^3 level 2: 29.4 us against 31.6 us with 16 x 32, profiles/r03/c4_l2_planes_tiles.txt)
        if ((p.LC / 2) % 64 == 0) { launch_l2_planes<dt2d::Fwd2DCfg<16, 64, 4, 10>>(p, planes, pstride, s); return; }
        if (narrow_wins(p.LC / 2, TC, 32)) {
            launch_l2_planes<dt2d::Fwd2DCfg<16, 32, 4, 10>>(p, planes, pstride, s);
            return;
        }
    }
    if constexpr (M == 14 || M == 16 || M == 18) {
        // round 6: the same for the longer q-shift sets -- the 2-D table's 52- / 50- / 48-column tiles cut a 128-column lowpass into
        // 52 + 52 + 24 (three tiles, 18 % of their lanes idle); 16 x 64 tiles are two whole ones (profiles/r06/c4_qbgn.txt)
        if ((p.LC / 2) % 64 == 0) { launch_l2_planes<dt2d::Fwd2DCfg<16, 64, 4, M>>(p, planes, pstride, s); return; }
    }
    launch_l2_planes<dt2d::Fwd2DCfg<TR, TC, PS, M>>(p, planes, pstride, s);
}

template <class C>
int launch_l2_axis0(Fwd3L2Params &p, int cus, hipStream_t s) {
    dt2d::dt_pack_lh(p);

    int cells = (p.O0 / 2) * (p.O1 / 2) * (p.O2 / 2);
    // Large levels: the inner layers of cells two at a time along axis 0 (24 instead of 40 window slices per pair
    // through L1), the boundary layers (windows that reach outside the volume) with the one-cell kernel.
    if constexpr (C::M <= 18) {
        const int e0 = p.O0 / 2, E = (p.O1 / 2) * (p.O2 / 2);
        int lo = (C::M - 2 + p.pad0 + 3) / 4;                      // first layer whose window starts inside
        int hi = (p.n0 - C::M - 2 + p.pad0) / 4;                   // last layer whose window ends inside
        if (lo & 1) ++lo;                                          // pairs (2 u, 2 u + 1)
        if (!(hi & 1)) --hi;
        if (E % 64 == 0 && hi - lo + 1 >= 8 && cdiv(cells, DT_NT) >= 4 * cus && 16 * p.pstride < ((int64_t)1 << 32)) {
            const int npair = (hi - lo + 1) / 2, colblocks = E / 32;
            const int ninner = npair * colblocks, lo_cells = lo * E, hi_first = (hi + 1) * E;     // E % 64 == 0 below
            const int nbound = (lo_cells + (e0 * E - hi_first)) / 64;
            k_fwd3_l2_axis0_pair<C::M><<<(unsigned)(ninner + nbound), 64, 0, s>>>(p, lo / 2, colblocks, ninner, lo_cells, hi_first);
            return 0;
        }
    }
    // coarse levels: single-wavefront workgroups so that every CU gets work
    // coarse levels: single-wavefront workgroups and two lanes per cell (half the dependent loads per thread: 64^3
    // cells 8.1 -> 7.4 us), so that every CU gets work; large levels are not latency-bound (no gain from the split:
    // a knock-out build runs 19 of their 31 us without any load or store) and keep one lane per cell
    if (cdiv(cells, DT_NT) < 4 * cus)
        k_fwd3_l2_axis0<C::M, 64, 32><<<(unsigned)cdiv(cells, 32), 64, 0, s>>>(p);
    else
        k_fwd3_l2_axis0<C::M, DT_NT, 64><<<(unsigned)cdiv(cells, DT_NT), DT_NT, 0, s>>>(p);
    return 0;
}

template <class C>
int launch_fwd3_l1(Fwd3L1Params &p, int cus, hipStream_t s) {
    p.tilesJ = cdiv(p.n1, C::TJ); p.tilesK = cdiv(p.n2, C::TK);
    // slices per workgroup: long marches amortise the 2H-slice warm-up, but the launch
    // still has to put several workgroups on every CU
    int chunk = 64;
    while (chunk > 8 && (int64_t)p.tilesJ * p.tilesK * cdiv(p.n0, chunk) < 4 * (int64_t)cus) chunk /= 2;
    p.chunk = chunk;
    p.chunks = cdiv(p.n0, chunk);
    f3l1_pack_taps<C>(p);
    const int ntile = p.tilesJ * p.tilesK * p.chunks;
    k_fwd3_l1<C><<<xcd3_grid(ntile, XCD3_FWD_L1), C::NT, 0, s>>>(p, xcd3_arg(ntile, XCD3_FWD_L1));
    return 0;
}


void put_taps(float *dst, const double *src, int m) {
    for (int k = 0; k < DT_MAXT; ++k) dst[k] = k < m ? (float)src[k] : 0.f;
}

void put_centred(float *dst, const double *src, int m, int to) {
    for (int k = 0; k < DT_MAXT; ++k) dst[k] = 0.f;
    for (int k = 0; k < m; ++k) dst[k + (to - m) / 2] = (float)src[k];
}

}  // namespace

#define DT_FWD3_L1_TABLE(X) X(5, 7) X(9, 7) X(7, 5) X(7, 9)

// Level 1 as a marching pair of wavefronts (fused3d_march.hpp): filters of at most 7 taps (near_sym_a, legall), rows of
// axis 2 in fours and wide enough to fill most of a strip's lanes; DTCWT_HIP_FWD3_MARCH=0 / =1 forces the tile program / the
// march wherever it applies.
static bool symmetric_taps(const double *h, int m) {
    double mx = 0;
    for (int k = 0; k < m; ++k) mx = fmax(mx, fabs(h[k]));
    for (int k = 0; k < m / 2; ++k) if (fabs(h[k] - h[m - 1 - k]) > 1e-12 * mx) return false;
    return true;
}
// The march folds the mirror pairs of a filter (t[d] * (x[H - d] + x[H + d]), pack_fwd3m keeps one half of the taps): only
// SYMMETRIC filters, and only the length pairs of the shipped sets it is tested on (near_sym_a 5 / 7, legall 5 / 3; the
// synthesis pairs 7 / 5 and 3 / 5 handed in as analysis filters) -- anything else keeps the tile program's full tap vector.
static bool fwd3m_ok(int64_t n0, int64_t n1, int64_t n2, const double *h0, int m0, const double *h1, int m1) {
    const int mode = [] { const char *e = getenv("DTCWT_HIP_FWD3_MARCH"); return e ? atoi(e) : -1; }();    // read per call: the tests switch it
    if (mode == 0) return false;
    if (!((m0 == 5 && m1 == 7) || (m0 == 7 && m1 == 5) || (m0 == 5 && m1 == 3) || (m0 == 3 && m1 == 5))) return false;
    if (!symmetric_taps(h0, m0) || !symmetric_taps(h1, m1)) return false;
    if (n2 % 4 || n0 % 2 || n1 % 2 || n0 < 8 || n1 < 8 || n2 < 16) return false;
    if (n0 * n1 * n2 * 4 >= ((int64_t)1 << 31)) return false;          // 32-bit byte offsets inside the volume
    if (mode == 1) return true;
    const int nstrip = dt3m::fwd3m_nstrip((int)n2);
    return (double)(n2 / 4) / (64.0 * nstrip) >= 0.7;                   // lanes that own columns
}
static int launch_fwd3m(const float *X, float *LLL, float *Yh, int n0, int n1, int n2, const double *h0, int m0,
                        const double *h1, int m1, int cus, hipStream_t s) {
    dt3m::Fwd3mParams p{};
    p.X = X; p.LLL = LLL; p.Yh = Yh; p.n0 = n0; p.n1 = n1; p.n2 = n2;
    p.nstrip = dt3m::fwd3m_nstrip(n2); p.nrp = n1 / 2;
    // slices per job: long marches amortise the six warm-up slices (loads + the axis-1 pass only), but a launch should
    // fill the wave slots (four pairs per CU)
    int chunk = 64;
    while (chunk > 8 && (int64_t)p.nstrip * p.nrp * cdiv(n0, chunk) < 4 * (int64_t)cus) chunk /= 2;
    if (const char *e = getenv("DTCWT_HIP_FWD3_CHUNK")) { const int v = atoi(e) / 8 * 8; if (v >= 8) chunk = v; }
    p.chunk = chunk; p.nchunk = cdiv(n0, chunk);
    dt3m::pack_fwd3m(p, h0, m0, h1, m1);
    const int per = (p.nrp + 7) / 8;
    // one wavefront per SIMD (OCC = 1: 256 + 20 registers, no scratch; the two-per-SIMD build spilled 24 of them and measured
    // equal, profiles/r05/ab_fwd3m.txt -- it and DTCWT_HIP_FWD3_OCC are gone)
    dt3m::k_fwd3m_l1<5, 7, 1><<<(unsigned)(8 * per * p.nstrip * p.nchunk), 128, 0, s>>>(p);
    return 0;
}

// Level 1 for the long filters (near_sym_b: 13 / 19 taps) as two launches around four plane volumes (fused3d_long.hpp): the
// in-slice half is the 2-D level-1 march with every slice an image (march2d.hip), then axis 0 + cube2c.
// DTCWT_HIP_LONG3D=0: never (the axis-by-axis generic kernels, as before round 5).
int dtcwt_march_fwd1_planes(const float *X, float *P, int64_t pstride, int B, int R, int C, const double *h0o, int m0,
                            const double *h1o, int m1, int cus, hipStream_t s);
int dtcwt_march_fwd3l_slices(const float *V, int64_t vstride, float *LLL, float *Yh, int n0, int n1, int n2, const double *h0o, int m0,
                             const double *h1o, int m1, int cus, hipStream_t s);
int dtcwt_march_inv3l_slices(const float *LLL, const float *Yh, float *V, int64_t vstride, int n0, int n1, int n2, const double *g0o, int m0,
                             const double *g1o, int m1, int cus, hipStream_t s);
int dtcwt_march_inv1_planes(const float *P, int64_t pstride, float *X, int B, int R, int C, const double *g0o, int m0,
                            const double *g1o, int m1, int cus, hipStream_t s);
static bool long3_ok(int64_t n0, int64_t n1, int64_t n2, int ma, int mb) {
    if (const char *e = getenv("DTCWT_HIP_LONG3D")) { if (e[0] == '0') return false; }
    if (!((ma == 13 && mb == 19) || (ma == 19 && mb == 13))) return false;
    if (n0 % 2 || n1 % 2 || n2 % 4 || n0 < 20 || n1 < 40 || n2 < 40) return false;
    return 4 * n0 * n1 * n2 * 4 < ((int64_t)1 << 40) && n0 * n1 * n2 * 4 < ((int64_t)1 << 31);      // 32-bit byte offsets inside a plane volume
}
static int launch_fwd3l_axis0(const float *P, int64_t pstride, float *LLL, float *Yh, int n0, int n1, int n2, const double *h0,
                              int m0, const double *h1, int m1, int cus, hipStream_t s) {
    dt3l::Fwd3lParams p{};
    p.P = P; p.pstride = pstride; p.LLL = LLL; p.Yh = Yh; p.n0 = n0; p.n1 = n1; p.n2 = n2;
    p.nstrip = cdiv(n2 / 2, 64); p.ncr = n1 / 2;
    // equal chunks of slices, enough of them for two workgroups per CU; every chunk re-reads 18 slices of warm-up
    int nchunk = 1;
    while (nchunk < 8 && n0 / (nchunk + 1) >= 40 && (int64_t)p.nstrip * p.ncr * nchunk < 2 * (int64_t)cus) ++nchunk;
    int chunk = (cdiv(n0, nchunk) + 1) & ~1;
    if (const char *e = getenv("DTCWT_HIP_LONG3D_CHUNK")) { const int v = atoi(e) & ~1; if (v >= 2) chunk = v; }
    p.chunk = chunk; p.nchunk = cdiv(n0, chunk);
    dt3l::pack_fwd3l(p, h0, m0, h1, m1);
    dt3l::k_fwd3l_axis0<13, 19><<<(unsigned)(p.ncr * p.nstrip * p.nchunk), 256, 0, s>>>(p);
    return 0;
}

extern "C" int dtcwt_hip_fwd3_level1(dtcwt_hip_ctx *ctx, const float *X, int64_t n0, int64_t n1, int64_t n2,
                                     const double *h0o, int m0, const double *h1o, int m1, float *LLL,
                                     float *Yh) {
    DT_REQUIRE(ctx && X && h0o && h1o && LLL && Yh, "NULL argument");
    DT_REQUIRE(n0 > 0 && n1 > 0 && n2 > 0 && n0 % 2 == 0 && n1 % 2 == 0 && n2 % 2 == 0,
               "level-1 volume extents must be even (transform3d.py:214-217)");
    DT_REQUIRE(m0 > 0 && m1 > 0 && m0 <= DT_MAXT && m1 <= DT_MAXT, "bad tap counts");
    if (n0 < 8 || n1 < 8 || n2 < 8 || n0 >= (1 << 30) || n1 * n2 >= ((int64_t)1 << 31))
        return dtcwt_set_error(-3, "fused 3-D level 1 needs extents >= 8 and slices < 2^31 samples");
    Fwd3L1Params p{};
    p.X = X; p.LLL = LLL; p.Yh = Yh;
    p.n0 = (int)n0; p.n1 = (int)n1; p.n2 = (int)n2;
    put_taps(p.h0, h0o, m0); put_taps(p.h1, h1o, m1);
    const int m0_in = m0, m1_in = m1;
    // 3-tap filters (legall) run as centred zero-padded 7-tap ones on the 5/7 kernels
    if (m0 == 5 && m1 == 3) { put_centred(p.h1, h1o, 3, 7); m1 = 7; }
    if (m0 == 3 && m1 == 5) { put_centred(p.h0, h0o, 3, 7); m0 = 7; }
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    if (fwd3m_ok(n0, n1, n2, h0o, m0_in, h1o, m1_in)) {
        launch_fwd3m(X, LLL, Yh, (int)n0, (int)n1, (int)n2, h0o, m0_in, h1o, m1_in, ctx->cus, ctx->stream);
        DT_CHECK_HIP(hipGetLastError());
        return 0;
    }
#define X_(A, B)                                                                           \
    if (m0 == A && m1 == B) {                                                              \
        launch_fwd3_l1<Fwd3L1Cfg<A, B>>(p, ctx->cus, ctx->stream);                         \
        DT_CHECK_HIP(hipGetLastError());                                                   \
        return 0;                                                                          \
    }
    DT_FWD3_L1_TABLE(X_)
#undef X_
    // round 6: axis 0 first (the generic marching pair filter: X -> lo0, hi0), then both in-slice axes + cube2c in one launch
    // (k_fwd3l_slices): two intermediate volumes instead of four, 52 instead of 68 B/voxel.  DTCWT_HIP_LONG3D=2: the round-5 cut
    // (in-slice first, four plane volumes, then axis 0) for A/B and as the cross-check of the tests.
    const bool slices_first = [] { const char *e = getenv("DTCWT_HIP_LONG3D"); return !(e && e[0] == '2'); }();
    if (slices_first && long3_ok(n0, n1, n2, m0, m1) && m0 == 13 && symmetric_taps(h0o, m0) && symmetric_taps(h1o, m1)) {
        const int64_t ps = n0 * n1 * n2;
        void *vol = nullptr;
        if (int rc = dtcwt_hip_malloc(ctx, (size_t)(2 * ps) * sizeof(float), &vol)) return rc;
        dtcwt_hip_view v{};
        v.outer = 1; v.n = n0; v.inner = n1 * n2;
        v.xso = ps; v.xsn = n1 * n2; v.xsi = 1; v.yso = ps; v.ysn = n1 * n2; v.ysi = 1;
        int rc = dtcwt_hip_colfilter2(ctx, DTCWT_HIP_F32, X, vol, (float *)vol + ps, &v, h0o, m0, h1o, m1);
        if (!rc) {
            rc = dtcwt_march_fwd3l_slices((const float *)vol, ps, LLL, Yh, (int)n0, (int)n1, (int)n2, h0o, m0, h1o, m1, ctx->cus, ctx->stream);
            if (rc) rc = dtcwt_set_error(-3, "the in-slice march does not take this volume");
        }
        hipError_t e = hipGetLastError();
        dtcwt_hip_free(ctx, vol);          // stream-ordered reuse (common.hpp)
        if (rc) return rc;
        if (e != hipSuccess) return dtcwt_set_error(-2, "3-D level launch failed: %s", hipGetErrorString(e));
        return 0;
    }
    if (long3_ok(n0, n1, n2, m0, m1) && symmetric_taps(h0o, m0) && symmetric_taps(h1o, m1)) {
        const int64_t ps = n0 * n1 * n2;
        void *planes = nullptr;
        if (int rc = dtcwt_hip_malloc(ctx, (size_t)(4 * ps) * sizeof(float), &planes)) return rc;
        int rc = dtcwt_march_fwd1_planes(X, (float *)planes, ps, (int)n0, (int)n1, (int)n2, h0o, m0, h1o, m1, ctx->cus, ctx->stream);
        if (!rc) rc = launch_fwd3l_axis0((const float *)planes, ps, LLL, Yh, (int)n0, (int)n1, (int)n2, h0o, m0, h1o, m1, ctx->cus, ctx->stream);
        hipError_t e = hipGetLastError();
        dtcwt_hip_free(ctx, planes);        // stream-ordered reuse (common.hpp)
        if (rc) return dtcwt_set_error(-3, "the level-1 march does not take this volume");
        if (e != hipSuccess) return dtcwt_set_error(-2, "3-D level launch failed: %s", hipGetErrorString(e));
        return 0;
    }
    return dtcwt_set_error(-3, "no fused 3-D level-1 kernel for %d/%d-tap biort filters", m0, m1);
}

// 0: pad-free multiple of 4; ext_mode 4 pads 1 plane per side, ext_mode 8 pads 2
extern "C" int dtcwt_hip_fwd3_level2(dtcwt_hip_ctx *ctx, const float *X, int64_t n0, int64_t n1, int64_t n2,
                                     int pad0, int pad1, int pad2, const double *h0b, const double *h0a,
                                     const double *h1b, const double *h1a, int m, float *LLL, float *Yh) {
    DT_REQUIRE(ctx && X && h0b && h0a && h1b && h1a && LLL && Yh, "NULL argument");
    DT_REQUIRE(m > 0 && m % 2 == 0 && m <= DT_MAXT, "q-shift filters must have even length <= %d", DT_MAXT);
    DT_REQUIRE(pad0 >= 0 && pad1 >= 0 && pad2 >= 0 && pad0 <= 2 && pad1 <= 2 && pad2 <= 2, "bad padding");
    const int64_t L0 = n0 + 2 * pad0, L1 = n1 + 2 * pad1, L2 = n2 + 2 * pad2;
    DT_REQUIRE(n0 > 0 && n1 > 0 && n2 > 0 && L0 % 4 == 0 && L1 % 4 == 0 && L2 % 4 == 0,
               "padded extents must be multiples of 4 (transform3d.py:322-335)");
    // one-bounce reflection in the tile programs: the window reach (< 2m) must not exceed the plane
    const int minw = 2 * m > 16 ? 2 * m : 16;
    if (L1 < minw || L2 < minw || n0 * L1 * L2 >= ((int64_t)1 << 31))
        return dtcwt_set_error(-3, "fused 3-D level >= 2 needs slices of at least %d x %d", minw, minw);
    const int O0 = (int)(L0 / 2), O1 = (int)(L1 / 2), O2 = (int)(L2 / 2);
    dt2d::Fwd2Params a{};
    a.X = X; a.B = (int)n0; a.inR = (int)n1; a.inC = (int)n2; a.padR = pad1; a.padC = pad2;
    a.LR = (int)L1; a.LC = (int)L2;
    put_taps(a.l_a, h0b, m); put_taps(a.l_b, h0a, m); put_taps(a.h_a, h1b, m); put_taps(a.h_b, h1a, m);
    double dl = 0, dh = 0;
    for (int k = 0; k < m; ++k) { dl += h0b[k] * h0a[k]; dh += h1b[k] * h1a[k]; }
    a.lo_a_first = dl > 0; a.hi_a_first = dh > 0;
    Fwd3L2Params b{};
    b.pstride = n0 * (int64_t)O1 * O2;
    b.LLL = LLL; b.Yh = Yh; b.n0 = (int)n0; b.pad0 = pad0; b.L0 = (int)L0; b.O0 = O0; b.O1 = O1; b.O2 = O2;
    b.lo_a_first = a.lo_a_first; b.hi_a_first = a.hi_a_first;
    put_taps(b.l_a, h0b, m); put_taps(b.l_b, h0a, m); put_taps(b.h_a, h1b, m); put_taps(b.h_b, h1a, m);
    bool have = false;
#define X_(TR, TC, PS, M) if (m == M) have = true;
    DT_FWD2_TABLE(X_)
#undef X_
    if (!have) return dtcwt_set_error(-3, "no fused 3-D level >= 2 kernel for %d-tap q-shift filters", m);
    void *planes = nullptr;
    int rc = dtcwt_hip_malloc(ctx, (size_t)(4 * b.pstride) * sizeof(float), &planes);
    if (rc) return rc;
    b.P = (const float *)planes;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
#define X_(TR, TC, PS, M)                                                                   \
    if (m == M) {                                                                           \
        launch_l2_planes_best<TR, TC, PS, M>(a, (float *)planes, b.pstride, ctx->stream);   \
        launch_l2_axis0<dt2d::Fwd2DCfg<TR, TC, PS, M>>(b, ctx->cus, ctx->stream);                     \
    }
    DT_FWD2_TABLE(X_)
#undef X_
    hipError_t e = hipGetLastError();
    dtcwt_hip_free(ctx, planes);        // stream-ordered reuse (common.hpp)
    if (e != hipSuccess) return dtcwt_set_error(-2, "3-D level launch failed: %s", hipGetErrorString(e));
    return 0;
}

// ======================================================================== inverse
namespace {

// pass A: unpack + axis-0 merge, marching along axis 0 (fused3d_inv_tiles.hpp)
template <class F>
__global__ void __launch_bounds__(DT_NT) k_inv3_axis0(Inv3AParams p, int ntile_xcd) {
    __shared__ __attribute__((aligned(16))) float slab[2][I3_SLAB];
    const int bid = xcd_tile3(ntile_xcd), tid = threadIdx.x;
    if (bid < 0) return;
    const int tk = bid % p.tilesK, tj = (bid / p.tilesK) % p.tilesJ, ch = bid / (p.tilesK * p.tilesJ);
    const int cj0 = tj * I3_CJ, ck0 = tk * I3_CK, c0 = ch * p.chunk;
    const int c1 = min(c0 + p.chunk, p.n0 / 2);
    const int cs = c0 - (2 * F::HP + 1);            // warm-up steps fill the rings
    Inv3AState<F> st;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < F::NS; ++t) { st.ra[q][t] = 0.f; st.rb[q][t] = 0.f; }
    i3a_issue_rec<F>(p, st, tid, cj0, ck0, cs + F::HP + 1);
    i3a_issue_low<F>(p, st, tid, cj0, ck0, cs + F::HP + 1);
    i3a_slab_write<F>(st, slab[0], tid);
    i3a_issue_rec<F>(p, st, tid, cj0, ck0, cs + F::HP + 2);
    __syncthreads();
    for (int c = cs; c < c1; ++c) {
        const int buf = (c - cs) & 1;
        float out[F::NOUT][4];
        if (c >= c0) F::compute(p, st.ra, st.rb, out);
        i3a_push<F>(p, st, slab[buf], tid, c + F::HP + 1);
        i3a_slab_write<F>(st, slab[buf ^ 1], tid);
        if (c + 1 < c1) {
            i3a_issue_rec<F>(p, st, tid, cj0, ck0, c + F::HP + 3);
            i3a_issue_low<F>(p, st, tid, cj0, ck0, c + F::HP + 2);
        }
        if (c >= c0) i3a_store<F>(p, out, tid, cj0, ck0, c);
        __syncthreads();
    }
}

// level 1, first launch: unpack + axis-0 merge + axis-2 merge (fused3d_inv_tiles.hpp, I3Ex)
template <class F, class G>
__global__ void __launch_bounds__(G::NT, 4) k_inv3_l1_axis02(Inv3AParams p, int ntile_xcd) {
    __shared__ __attribute__((aligned(16))) float slab[2][G::SLAB];
    __shared__ __attribute__((aligned(16))) float E[I3Ex<G>::FLOATS];
    const int bid = xcd_tile3(ntile_xcd), tid = threadIdx.x;
    if (bid < 0) return;
    const int tk = bid % p.tilesK, tj = (bid / p.tilesK) % p.tilesJ, ch = bid / (p.tilesK * p.tilesJ);
    const int cj0 = tj * G::CJ, ck0 = tk * (G::CK - 2 * p.hal) - p.hal, c0 = ch * p.chunk;
    const int c1 = min(c0 + p.chunk, p.n0 / 2);
    const int q0 = c0 - F::HP, q1 = c1 - 1 + F::HP;       // virtual records added by this march
    Inv3TState<F, G> st;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < F::NA; ++t) st.acc[q][t] = 0.f;
    i3a_issue_rec<F, G>(p, st, tid, cj0, ck0, q0);
    i3a_issue_low<F, G>(p, st, tid, cj0, ck0, q0);
    i3a_slab_write<F, G>(st, slab[0], tid);
    i3a_issue_rec<F, G>(p, st, tid, cj0, ck0, q0 + 1);
    __syncthreads();
    for (int q = q0; q <= q1; ++q) {
        const int buf = (q - q0) & 1, c = q - F::HP;        // the output pair this step completes
        float out[2][4];
        i3a_accumulate<F, G>(p, st, slab[buf], tid, q, out);
        i3a_slab_write<F, G>(st, slab[buf ^ 1], tid);
        if (q < q1) {
            i3a_issue_rec<F, G>(p, st, tid, cj0, ck0, q + 2);
            i3a_issue_low<F, G>(p, st, tid, cj0, ck0, q + 1);
        }
        if (c >= c0) i3a_exchange<F, G>(p, out, E, tid, ck0);
        __syncthreads();
        if (c >= c0) i3a_merge_k<F, G>(p, E, tid, cj0, ck0, c);
        __syncthreads();
    }
}

// level 1 for long filters (fused3d_long.hpp), first launch: unpack + axis-0 merge into the four plane volumes -- the march
// above without the axis-2 exchange
template <class G>
DT_HD void i3l_store(const Inv3AParams &p, const float (&out)[2][4], int tid, int cj0, int ck0, int c) {
    const int v = G::combo(tid), idx = G::cell(tid);
    const int j = 2 * (cj0 + idx / G::CK), k = 2 * (ck0 + idx % G::CK);
    if (j >= p.n1 || k >= p.n2) return;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int so = 2 * c + e;
        if (so < 0 || so >= p.S) continue;
        float *o = p.P + v * p.pstride + ((int64_t)so * p.n1 + j) * p.n2 + k;
        *reinterpret_cast<f2 *>(o) = f2{out[e][0], out[e][1]};
        *reinterpret_cast<f2 *>(o + p.n2) = f2{out[e][2], out[e][3]};
    }
}
template <class F, class G>
__global__ void __launch_bounds__(G::NT) k_inv3l_axis0(Inv3AParams p, int ntile_xcd) {
    __shared__ __attribute__((aligned(16))) float slab[2][G::SLAB];
    const int bid = xcd_tile3(ntile_xcd), tid = threadIdx.x;
    if (bid < 0) return;
    const int tk = bid % p.tilesK, tj = (bid / p.tilesK) % p.tilesJ, ch = bid / (p.tilesK * p.tilesJ);
    const int cj0 = tj * G::CJ, ck0 = tk * G::CK, c0 = ch * p.chunk;
    const int c1 = min(c0 + p.chunk, p.n0 / 2);
    const int q0 = c0 - F::HP, q1 = c1 - 1 + F::HP;       // virtual records added by this march
    Inv3TState<F, G> st;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < F::NA; ++t) st.acc[q][t] = 0.f;
    i3a_issue_rec<F, G>(p, st, tid, cj0, ck0, q0);
    i3a_issue_low<F, G>(p, st, tid, cj0, ck0, q0);
    i3a_slab_write<F, G>(st, slab[0], tid);
    i3a_issue_rec<F, G>(p, st, tid, cj0, ck0, q0 + 1);
    __syncthreads();
    for (int q = q0; q <= q1; ++q) {
        const int buf = (q - q0) & 1, c = q - F::HP;        // the output pair this step completes
        float out[2][4];
        i3a_accumulate<F, G>(p, st, slab[buf], tid, q, out);
        i3a_slab_write<F, G>(st, slab[buf ^ 1], tid);
        if (q < q1) {
            i3a_issue_rec<F, G>(p, st, tid, cj0, ck0, q + 2);
            i3a_issue_low<F, G>(p, st, tid, cj0, ck0, q + 1);
        }
        if (c >= c0) i3l_store<G>(p, out, tid, cj0, ck0, c);
        __syncthreads();
    }
}

// level 1, second launch: the axis-1 merge
template <class F, int VEC, int RS>
__global__ void __launch_bounds__(DT_NT) k_inv3_l1_axis1(Inv3BParams p) {
    i3b_axis1<F, VEC, RS>(p, (int64_t)blockIdx.x * DT_NT + threadIdx.x);
}

// pass B, level >= 2
template <class C>
__global__ void __launch_bounds__(DT_NT) k_inv3_l2_planes(dt2d::Inv2Params p, const float *planes, int64_t ps, int ntile_xcd) {
    __shared__ __attribute__((aligned(16))) float smem[2 * C::SY];
    const int t = xcd_tile3(ntile_xcd);
    if (t < 0) return;
    int tc, tr, b;
    dt2d::dt_tile_decode(p, t, tc, tr, b);
    float *y1 = smem, *y2 = y1 + C::SY;
    const int r0 = tr * C::TR, c0 = tc * C::TC;
    float wz[C::WS], w1[C::WS], w2[C::WS], w3[C::WS];
    dt2d::inv2r_fetch_from<C, true>(p, planes, wz, threadIdx.x, b, r0, c0);
    dt2d::inv2r_fetch_from<C, true>(p, planes + 2 * ps, w1, threadIdx.x, b, r0, c0);
    dt2d::inv2r_fetch_from<C, true>(p, planes + ps, w2, threadIdx.x, b, r0, c0);
    dt2d::inv2r_fetch_from<C, true>(p, planes + 3 * ps, w3, threadIdx.x, b, r0, c0);
    dt2d::inv2r_fir<C, true>(p, wz, w1, w2, w3, y1, y2, threadIdx.x);
    __syncthreads();
    dt2d::inv2_rows<C>(p, y1, y2, threadIdx.x, b, r0, c0);
}

template <class C>
void launch_inv3_l2_planes(dt2d::Inv2Params &b, const float *planes, int64_t ps, hipStream_t s) {
    b.tilesR = cdiv(b.zr, C::TR); b.tilesC = cdiv(b.zc, C::TC);
    dt2d::dt_set_tile_magic(b);
    const int ntile = b.tilesR * b.tilesC * b.B;
    k_inv3_l2_planes<C><<<xcd3_grid(ntile, XCD3_INV_L2_PLANES), DT_NT, 0, s>>>(b, planes, ps, xcd3_arg(ntile, XCD3_INV_L2_PLANES));
}

template <class F>
void launch_inv3_axis0(Inv3AParams &p, int cus, hipStream_t s) {
    p.tilesJ = cdiv(p.n1 / 2, I3_CJ); p.tilesK = cdiv(p.n2 / 2, I3_CK);
    const int pairs = p.n0 / 2;
    int chunk = 64;             // long marches amortise the 2 HP + 1 warm-up steps
    while (chunk > 8 && (int64_t)p.tilesJ * p.tilesK * cdiv(pairs, chunk) < 4 * (int64_t)cus) chunk /= 2;
    // coarse levels: fewer workgroups than CUs even at 8 pairs per march -- shorter marches (more warm-up steps in
    // all, but the level is a latency chain: 32^3 cells 22.6 -> 14.2 us); with a workgroup per CU or more, 8 stays
    // (64^3 cells: 35 us at 8, 45 us at 4)
    while (chunk > 2 && (int64_t)p.tilesJ * p.tilesK * cdiv(pairs, chunk) < (int64_t)cus) chunk /= 2;
    p.chunk = chunk; p.chunks = cdiv(pairs, chunk);
    const int ntile = p.tilesJ * p.tilesK * p.chunks;
    k_inv3_axis0<F><<<xcd3_grid(ntile, XCD3_INV_AXIS0), DT_NT, 0, s>>>(p, xcd3_arg(ntile, XCD3_INV_AXIS0));
}

// level 1 with the axis-2 merge in the march: geometry by row length
template <class F, class G>
void launch_inv3_l1_axis02_g(Inv3AParams &p, int cus, hipStream_t s) {
    const int e2 = p.n2 / 2;
    p.hal = e2 > G::CK ? 2 : 0;
    p.tilesJ = cdiv(p.n1 / 2, G::CJ); p.tilesK = p.hal ? cdiv(e2, G::CK - 4) : 1;
    const int pairs = p.n0 / 2;
    int chunk = 64;
    while (chunk > 8 && (int64_t)p.tilesJ * p.tilesK * cdiv(pairs, chunk) < 2 * (int64_t)cus) chunk /= 2;
    while (chunk > 2 && (int64_t)p.tilesJ * p.tilesK * cdiv(pairs, chunk) < (int64_t)cus / 2) chunk /= 2;
    p.chunk = chunk; p.chunks = cdiv(pairs, chunk);
    const int ntile = p.tilesJ * p.tilesK * p.chunks;
    k_inv3_l1_axis02<F, G><<<xcd3_grid(ntile, XCD3_INV_AXIS0), G::NT, 0, s>>>(p, xcd3_arg(ntile, XCD3_INV_AXIS0));
}
template <class F>
void launch_inv3_l1_axis02(Inv3AParams &p, int cus, hipStream_t s) {
    const int e2 = p.n2 / 2;
    if (e2 <= 32) launch_inv3_l1_axis02_g<F, I3Geo<512, 32>>(p, cus, s);
    else if (e2 <= 64) launch_inv3_l1_axis02_g<F, I3Geo<512, 64>>(p, cus, s);
    else launch_inv3_l1_axis02_g<F, I3Geo<512, 128>>(p, cus, s);
}
template <class F, class G>
void launch_inv3l_axis0_g(Inv3AParams &p, int cus, hipStream_t s) {
    p.hal = 0;
    p.tilesJ = cdiv(p.n1 / 2, G::CJ); p.tilesK = cdiv(p.n2 / 2, G::CK);
    const int pairs = p.n0 / 2;
    int chunk = 64;             // long marches amortise the 2 HP warm-up records
    while (chunk > 16 && (int64_t)p.tilesJ * p.tilesK * cdiv(pairs, chunk) < 2 * (int64_t)cus) chunk /= 2;
    if (const char *e = getenv("DTCWT_HIP_LONG3D_ICHUNK")) { const int v = atoi(e); if (v >= 4) chunk = v; }
    p.chunk = chunk; p.chunks = cdiv(pairs, chunk);
    const int ntile = p.tilesJ * p.tilesK * p.chunks;
    k_inv3l_axis0<F, G><<<xcd3_grid(ntile, XCD3_INV_AXIS0), G::NT, 0, s>>>(p, xcd3_arg(ntile, XCD3_INV_AXIS0));
}
template <class F>
void launch_inv3l_axis0(Inv3AParams &p, int cus, hipStream_t s) {
    if (p.n2 / 2 <= 32) launch_inv3l_axis0_g<F, I3GeoA>(p, cus, s);
    else launch_inv3l_axis0_g<F, I3Geo<256, 64>>(p, cus, s);
}
template <class F>
void launch_inv3_l1_axis1(Inv3BParams &b, hipStream_t s) {
    constexpr int RS = 8;
    const int vec = (b.n2 & 3) == 0 ? 4 : 2;
    b.kvecs = b.n2 / vec; b.strips = cdiv(b.n1, RS);
    const int64_t tasks = (int64_t)b.kvecs * b.strips * b.S;
    const unsigned grid = (unsigned)((tasks + DT_NT - 1) / DT_NT);
    if (vec == 4) k_inv3_l1_axis1<F, 4, RS><<<grid, DT_NT, 0, s>>>(b);
    else k_inv3_l1_axis1<F, 2, RS><<<grid, DT_NT, 0, s>>>(b);
}

// one-bounce reflection in the tile programs: the window reach (< 2 x taps) must not exceed
// the plane, and the march needs a few records of run-in
int check_inv3_dims(int64_t n0, int64_t n1, int64_t n2, int64_t S, int taps) {
    const int minw = 2 * taps > 16 ? 2 * taps : 16;
    if (n0 < 12 || n1 < minw || n2 < minw || 4 * S * n1 * n2 >= ((int64_t)1 << 31) ||
        n0 * n1 * n2 >= ((int64_t)1 << 31))
        return dtcwt_set_error(-3, "fused 3-D inverse needs n0 >= 12, slices of at least %d x %d and < 2^31 samples",
                               minw, minw);
    return 0;
}

}  // namespace

#define DT_INV3_L1_TABLE(X) X(7, 5) X(7, 9) X(3, 5)
#define DT_INV3_L2_TABLE(X) X(16, 56, 2, 10) X(16, 52, 2, 14)

extern "C" int dtcwt_hip_inv3_level1(dtcwt_hip_ctx *ctx, const float *LLL, const float *Yh, int64_t n0, int64_t n1,
                                     int64_t n2, const double *g0o, int m0, const double *g1o, int m1, float *Z) {
    DT_REQUIRE(ctx && LLL && Yh && g0o && g1o && Z, "NULL argument");
    DT_REQUIRE(n0 > 0 && n1 > 0 && n2 > 0 && n0 % 2 == 0 && n1 % 2 == 0 && n2 % 2 == 0, "extents must be even");
    DT_REQUIRE(m0 > 0 && m1 > 0 && m0 <= DT_MAXT && m1 <= DT_MAXT, "bad tap counts");
    if (int rc = check_inv3_dims(n0, n1, n2, n0, m0 > m1 ? m0 : m1)) return rc;
    bool have = false;
#define X_(A, B) if (m0 == A && m1 == B) have = true;
    DT_INV3_L1_TABLE(X_)
#undef X_
    // round 6: c2cube + both in-slice axes in one launch (k_inv3l_slices: LLL, Yh -> the two volumes the axis-0 synthesis filters take),
    // then the generic marching sum filter along axis 0: 52 instead of 68 B/voxel.  DTCWT_HIP_LONG3D=2: the round-5 cut.
    const bool slices_first = [] { const char *e = getenv("DTCWT_HIP_LONG3D"); return !(e && e[0] == '2'); }();
    if (!have && slices_first && long3_ok(n0, n1, n2, m0, m1) && m0 == 19 && symmetric_taps(g0o, m0) && symmetric_taps(g1o, m1)) {
        const int64_t ps = n0 * n1 * n2;
        void *vol = nullptr;
        if (int rc = dtcwt_hip_malloc(ctx, (size_t)(2 * ps) * sizeof(float), &vol)) return rc;
        DT_CHECK_HIP(hipSetDevice(ctx->device));
        int rc = dtcwt_march_inv3l_slices(LLL, Yh, (float *)vol, ps, (int)n0, (int)n1, (int)n2, g0o, m0, g1o, m1, ctx->cus, ctx->stream);
        if (rc) rc = dtcwt_set_error(-3, "the in-slice inverse march does not take this volume");
        if (!rc) {
            dtcwt_hip_view v{};
            v.outer = 1; v.n = n0; v.inner = n1 * n2;
            v.xso = ps; v.xsn = n1 * n2; v.xsi = 1; v.yso = ps; v.ysn = n1 * n2; v.ysi = 1;
            rc = dtcwt_hip_colfilter_sum2(ctx, DTCWT_HIP_F32, vol, (const float *)vol + ps, Z, &v, g0o, m0, g1o, m1);
        }
        hipError_t er = hipGetLastError();
        dtcwt_hip_free(ctx, vol);
        if (rc) return rc;
        if (er != hipSuccess) return dtcwt_set_error(-2, "3-D inverse launch failed: %s", hipGetErrorString(er));
        return 0;
    }
    if (!have && long3_ok(n0, n1, n2, m0, m1) && m0 == 19 && symmetric_taps(g0o, m0) && symmetric_taps(g1o, m1)) {
        // long filters (fused3d_long.hpp): c2cube + axis 0 into four plane volumes, then the 2-D level-1 march slice by slice
        Inv3AParams a{};
        a.LLL = LLL; a.Yh = Yh; a.n0 = (int)n0; a.n1 = (int)n1; a.n2 = (int)n2; a.S = (int)n0; a.crop0 = 0;
        a.pstride = n0 * n1 * n2;
        put_taps(a.l_a, g0o, m0); put_taps(a.h_a, g1o, m1);
        void *planes = nullptr;
        if (int rc = dtcwt_hip_malloc(ctx, (size_t)(4 * a.pstride) * sizeof(float), &planes)) return rc;
        a.P = (float *)planes;
        DT_CHECK_HIP(hipSetDevice(ctx->device));
        launch_inv3l_axis0<Inv3L1<19, 13>>(a, ctx->cus, ctx->stream);
        const int rc = dtcwt_march_inv1_planes((const float *)planes, a.pstride, Z, (int)n0, (int)n1, (int)n2, g0o, m0, g1o, m1, ctx->cus, ctx->stream);
        hipError_t er = hipGetLastError();
        dtcwt_hip_free(ctx, planes);
        if (rc) return dtcwt_set_error(-3, "the level-1 march does not take this volume");
        if (er != hipSuccess) return dtcwt_set_error(-2, "3-D inverse launch failed: %s", hipGetErrorString(er));
        return 0;
    }
    if (!have) return dtcwt_set_error(-3, "no fused 3-D level-1 inverse for %d/%d-tap biort filters", m0, m1);
    Inv3AParams a{};
    a.LLL = LLL; a.Yh = Yh; a.n0 = (int)n0; a.n1 = (int)n1; a.n2 = (int)n2; a.S = (int)n0; a.crop0 = 0;
    a.pstride = n0 * n1 * n2;
    put_taps(a.l_a, g0o, m0); put_taps(a.h_a, g1o, m1);
    void *planes = nullptr;
    if (int rc = dtcwt_hip_malloc(ctx, (size_t)(2 * a.pstride) * sizeof(float), &planes)) return rc;
    a.P = (float *)planes;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    // (Running the level in slabs along axis 0 -- march and axis-1 merge per slab, Q of one slab in a reused buffer
    // -- paid with the four plane-volumes of the previous scheme; with two it does not: 273 against 305 us at
    // 256^3 with two slabs, 2.04 against 2.11 ms at 512^3 with seven.  profiles/r03/c4_inverse.txt)
    Inv3BParams q{};
    q.Q = (const float *)planes; q.pstride = a.pstride; q.Z = Z; q.S = (int)n0; q.n1 = (int)n1; q.n2 = (int)n2;
    put_taps(q.g0, g0o, m0); put_taps(q.g1, g1o, m1);
#define X_(MA_, MB_)                                                                        \
    if (m0 == MA_ && m1 == MB_) {                                                           \
        launch_inv3_l1_axis02<Inv3L1<MA_, MB_>>(a, ctx->cus, ctx->stream);                  \
        launch_inv3_l1_axis1<Inv3L1<MA_, MB_>>(q, ctx->stream);                             \
    }
    DT_INV3_L1_TABLE(X_)
#undef X_
    hipError_t er = hipGetLastError();
    dtcwt_hip_free(ctx, planes);
    if (er != hipSuccess) return dtcwt_set_error(-2, "3-D inverse launch failed: %s", hipGetErrorString(er));
    return 0;
}

extern "C" int dtcwt_hip_inv3_level2(dtcwt_hip_ctx *ctx, const float *LLL, const float *Yh, int64_t n0, int64_t n1,
                                     int64_t n2, int crop0, int crop1, int crop2, const double *g0b,
                                     const double *g0a, const double *g1b, const double *g1a, int m, float *Z) {
    DT_REQUIRE(ctx && LLL && Yh && g0b && g0a && g1b && g1a && Z, "NULL argument");
    DT_REQUIRE(n0 > 0 && n1 > 0 && n2 > 0 && n0 % 2 == 0 && n1 % 2 == 0 && n2 % 2 == 0, "extents must be even");
    DT_REQUIRE(m > 0 && m % 2 == 0 && m <= DT_MAXT, "q-shift filters must have even length <= %d", DT_MAXT);
    DT_REQUIRE(crop0 >= 0 && crop1 >= 0 && crop2 >= 0 && crop0 <= 2 && crop1 <= 2 && crop2 <= 2, "bad crop");
    const int64_t S = 2 * n0 - 2 * crop0;
    if (int rc = check_inv3_dims(n0, n1, n2, S, m)) return rc;
    bool have = false;
#define X_(TR, TC, JS, M) if (m == M) have = true;
    DT_INV3_L2_TABLE(X_)
#undef X_
    if (!have) return dtcwt_set_error(-3, "no fused 3-D level >= 2 inverse for %d-tap q-shift filters", m);
    double dl = 0, dh = 0;
    for (int k = 0; k < m; ++k) { dl += g0b[k] * g0a[k]; dh += g1b[k] * g1a[k]; }
    Inv3AParams a{};
    a.LLL = LLL; a.Yh = Yh; a.n0 = (int)n0; a.n1 = (int)n1; a.n2 = (int)n2; a.S = (int)S; a.crop0 = crop0;
    a.pstride = S * n1 * n2;
    a.lo_pos = dl > 0; a.hi_pos = dh > 0;
    put_taps(a.l_a, g0b, m); put_taps(a.l_b, g0a, m); put_taps(a.h_a, g1b, m); put_taps(a.h_b, g1a, m);
    dt2d::Inv2Params b{};
    b.Out = Z; b.B = (int)S; b.zr = (int)n1; b.zc = (int)n2; b.cropR = crop1; b.cropC = crop2;
    b.lo_pos = a.lo_pos; b.hi_pos = a.hi_pos;
    put_taps(b.l_a, g0b, m); put_taps(b.l_b, g0a, m); put_taps(b.h_a, g1b, m); put_taps(b.h_b, g1a, m);
    void *planes = nullptr;
    if (int rc = dtcwt_hip_malloc(ctx, (size_t)(4 * a.pstride) * sizeof(float), &planes)) return rc;
    a.P = (float *)planes;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
#define X_(TR_, TC_, JS_, M_)                                                               \
    if (m == M_) {                                                                          \
        launch_inv3_axis0<Inv3L2<M_>>(a, ctx->cus, ctx->stream);                            \
        /* planes a multiple of 64 columns wide take 12 x 64 tiles: the table's 16 x 56 leave a quarter of the lanes of \
         * a 128-column plane idle (56 + 56 + 16) -- 128^2 planes 42.9 -> 34.5 us, 64^2 ones 10.2 -> 8.5 (8 x 64 tiles: \
         * 37.0 / 9.8, 16 x 64 with strips of four: 36.2 / 10.2; profiles/r03/c4_inverse.txt) */ \
        if (M_ == 10 && b.zc % 64 == 0)                                                     \
            launch_inv3_l2_planes<dt2d::Inv2RCfg<12, 64, 2, 10>>(b, (const float *)planes, a.pstride, ctx->stream); \
        else                                                                                \
            launch_inv3_l2_planes<dt2d::Inv2RCfg<TR_, TC_, JS_, M_>>(b, (const float *)planes, a.pstride, ctx->stream); \
    }
    DT_INV3_L2_TABLE(X_)
#undef X_
    hipError_t e = hipGetLastError();
    dtcwt_hip_free(ctx, planes);
    if (e != hipSuccess) return dtcwt_set_error(-2, "3-D inverse launch failed: %s", hipGetErrorString(e));
    return 0;
}

// ---- "would this level run fused?" for the whole-transform plan (plans13.hip): the same conditions the four
// entry points above test before launching, so that a plan is only created where every level of a direction runs
// natively -- a level that declined in the middle of a transform made the host redo the whole transform.
bool dtcwt_fwd3_level1_ok(int64_t n0, int64_t n1, int64_t n2, int m0, int m1, const double *h0o, const double *h1o) {
    if (n0 < 8 || n1 < 8 || n2 < 8 || n0 >= (1 << 30) || n1 * n2 >= ((int64_t)1 << 31)) return false;
    if (m0 == 5 && m1 == 3) m1 = 7;
    if (m0 == 3 && m1 == 5) m0 = 7;
#define X_(A, B) if (m0 == A && m1 == B) return true;
    DT_FWD3_L1_TABLE(X_)
#undef X_
    // the long filters (fused3d_long.hpp) fold their mirror pairs: symmetric taps only (every shipped set)
    return long3_ok(n0, n1, n2, m0, m1) && m0 == 13 && h0o && h1o && symmetric_taps(h0o, m0) && symmetric_taps(h1o, m1);
}
bool dtcwt_fwd3_level2_ok(int64_t n0, int64_t n1, int64_t n2, int pad0, int pad1, int pad2, int m) {
    const int64_t L1 = n1 + 2 * pad1, L2 = n2 + 2 * pad2;
    const int minw = 2 * m > 16 ? 2 * m : 16;
    if (L1 < minw || L2 < minw || n0 * L1 * L2 >= ((int64_t)1 << 31)) return false;
#define X_(TR, TC, PS, M) if (m == M) return true;
    DT_FWD2_TABLE(X_)
#undef X_
    return false;
}
bool dtcwt_inv3_level1_ok(int64_t n0, int64_t n1, int64_t n2, int m0, int m1, const double *g0o, const double *g1o) {
    const int taps = m0 > m1 ? m0 : m1;
    const int minw = 2 * taps > 16 ? 2 * taps : 16;
    if (n0 < 12 || n1 < minw || n2 < minw || 4 * n0 * n1 * n2 >= ((int64_t)1 << 31)) return false;
#define X_(A, B) if (m0 == A && m1 == B) return true;
    DT_INV3_L1_TABLE(X_)
#undef X_
    return long3_ok(n0, n1, n2, m0, m1) && m0 == 19 && g0o && g1o && symmetric_taps(g0o, m0) && symmetric_taps(g1o, m1);
}
bool dtcwt_inv3_level2_ok(int64_t n0, int64_t n1, int64_t n2, int crop0, int m) {
    const int64_t S = 2 * n0 - 2 * crop0;
    const int minw = 2 * m > 16 ? 2 * m : 16;
    if (n0 < 12 || n1 < minw || n2 < minw || 4 * S * n1 * n2 >= ((int64_t)1 << 31)) return false;
#define X_(TR, TC, JS, M) if (m == M) return true;
    DT_INV3_L2_TABLE(X_)
#undef X_
    return false;
}
