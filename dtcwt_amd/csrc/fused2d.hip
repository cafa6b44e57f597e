// Device-resident 2-D DT-CWT level loop: __global__ wrappers of the tile programs in
// fused2d_tiles.hpp and the plan object behind dtcwt_hip_plan2d_* (include/dtcwt_hip.h).
//
// Replaces Transform2d.forward / .inverse of dtcwt/numpy/transform2d.py:40-295 for
// float32 batches: one kernel launch per level and direction, all intermediates of a
// level in LDS, LoLo of each level in HBM (it is the next level's input and the
// `scales` output), Yh written/read as whole 48-byte 6-subband records.
#include <cstdlib>
#include <vector>

#include <string>

#include "common.hpp"
#include "fused2d_tiles.hpp"
#include "fused2d_tiles_v2.hpp"
#include "fused2d_table.hpp"

using namespace dt2d;

// the inverse kernels live in fused2d_inv.hip (built without SLP vectorisation: see the Makefile)
int dtcwt_dispatch_inv1(int m0, int m1, int m2, dt2d::Inv1Params &p, hipStream_t s);
int dtcwt_dispatch_inv2(int m, bool bp, dt2d::Inv2Params &p, hipStream_t s, bool small);
// levels 1 + 2 of the forward transform as one marching launch (march2d.hip)
bool dtcwt_march_fwd12_ok(int batch, int rows, int cols, const std::vector<double> &h0o, const std::vector<double> &h1o,
                          const std::vector<double> &h0a, bool lo_a_first, bool hi_a_first, const DtMarchHint &hint);
int dtcwt_march_fwd12(const float *X, float *Yh0, float *Yh1, float *LoLo2, float *LoLo1, int B, int R, int C,
                      const std::vector<double> &h0o, const std::vector<double> &h1o,
                      const float *l_a, const float *l_b, const float *h_a, const float *h_b, int m,
                      int lo_a_first, int hi_a_first, const DtMarchHint &hint, hipStream_t s);
bool dtcwt_march_inv21_ok(int batch, int rows, int cols, const std::vector<double> &g0o, const std::vector<double> &g1o,
                          const std::vector<double> &g0a, bool lo_pos, bool hi_pos, const DtMarchHint &hint);
int dtcwt_march_inv21(const float *Z2, const float *Yh1, const float *Yh0, float *X, int B, int R, int C,
                      const std::vector<double> &g0o, const std::vector<double> &g1o, const float *l_a, const float *l_b,
                      const float *h_a, const float *h_b, const float *gain1, const float *gain2, const DtMarchHint &hint, hipStream_t s);

// level 1 alone as a march, for the biort sets the fused launches are not built for (march2d_l1.hpp)
bool dtcwt_march_fwd1_ok(int batch, int rows, int cols, const std::vector<double> &h0o, const std::vector<double> &h1o, const DtMarchHint &hint);
int dtcwt_march_fwd1(const float *X, float *LoLo, float *Yh0, int B, int R, int C, const std::vector<double> &h0o,
                     const std::vector<double> &h1o, const DtMarchHint &hint, hipStream_t s);
bool dtcwt_march_inv1_ok(int batch, int rows, int cols, const std::vector<double> &g0o, const std::vector<double> &g1o, const DtMarchHint &hint);
int dtcwt_march_inv1(const float *Z, const float *Yh0, float *X, int B, int R, int C, const std::vector<double> &g0o,
                     const std::vector<double> &g1o, const float *gain1, const DtMarchHint &hint, hipStream_t s);

// levels 1 + 2 of the forward as a marching pair of wavefronts (march2d_pair.hpp)
bool dtcwt_march_fwd12p_ok(int batch, int rows, int cols, const std::vector<double> &h0o, const std::vector<double> &h1o,
                           const std::vector<double> &h0a, bool lo_a_first, bool hi_a_first, const DtMarchHint &hint);
int dtcwt_march_fwd12p(const float *X, float *Yh0, float *Yh1, float *LoLo2, int B, int R, int C,
                       const std::vector<double> &h0o, const std::vector<double> &h1o,
                       const float *l_a, const float *l_b, const float *h_a, const float *h_b, int m,
                       const DtMarchHint &hint, hipStream_t s);
// levels 2 + 1 of the inverse as a marching pair of wavefronts (march2d_ipair.hpp)
bool dtcwt_march_inv21p_ok(int batch, int rows, int cols, const std::vector<double> &g0o, const std::vector<double> &g1o,
                           const std::vector<double> &g0a, bool lo_pos, bool hi_pos, const DtMarchHint &hint);
int dtcwt_march_inv21p(const float *Z2, const float *Yh1, const float *Yh0, float *X, int B, int R, int C,
                       const std::vector<double> &g0o, const std::vector<double> &g1o, const float *l_a, const float *l_b,
                       const float *h_a, const float *h_b, int m, const float *gain1, const float *gain2, const DtMarchHint &hint, hipStream_t s);

// level 2 of the forward alone as a march (march2d_pair.hpp: k_fwd2m)
bool dtcwt_march_fwd2_ok(int batch, int rows, int cols, const std::vector<double> &h0a, bool lo_a_first, bool hi_a_first, const DtMarchHint &hint);
int dtcwt_march_fwd2(const float *LoLo1, float *Yh1, float *LoLo2, int B, int R, int C, const float *l_a, const float *l_b,
                     const float *h_a, const float *h_b, int m, const DtMarchHint &hint, hipStream_t s);
// level 2 of the inverse alone as a march (march2d_ipair.hpp: k_inv2m)
bool dtcwt_march_inv2_ok(int batch, int rows, int cols, const std::vector<double> &g0a, bool lo_pos, bool hi_pos, const DtMarchHint &hint);
int dtcwt_march_inv2(const float *Z2, const float *Yh1, float *Z1, int B, int R, int C, const float *l_a, const float *l_b,
                     const float *h_a, const float *h_b, int m, const float *gain2, const DtMarchHint &hint, hipStream_t s);
namespace {

// record arrays at least this big leave with the non-temporal hint (32 MiB: swept in round 2, profiles/r02; the switch is gone)
inline int64_t stream_records_bytes() { return (int64_t)32 << 20; }

// -------------------------------------------------------------------------- kernels
// Level-1 forward: column pass straight from global memory into two LDS planes, one
// barrier, row pass + q2c with wave-staged coalesced record stores (fused2d_tiles_v2.hpp).
template <class C>
__global__ void __launch_bounds__(DT_NT) k_fwd1(Fwd1Params p) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_FLOATS + 4 * STAGE_FLOATS_PER_WAVE];
    const int ntile = p.tilesR * p.tilesC * p.B;
    int t = tile_of(blockIdx.x, ntile, p.xcd_order);
    if (t >= ntile) return;
    int tc, tr, b;
    dt_tile_decode(p, t, tc, tr, b);
    float *sLo = smem, *sHi = sLo + C::SL, *sBa = sHi + C::SL, *stage = smem + C::LDS_FLOATS;
    int r0 = tr * C::TR, c0 = tc * C::TC;
    fwd1d_cols<C>(p, sLo, sHi, threadIdx.x, b, r0, c0, sBa);
    __syncthreads();
    constexpr int NQ = (C::TR / 2) * (C::TC / 2);
    for (int base = 0; base < NQ; base += DT_NT) {
        fwd1s_rows_compute<C>(p, sLo, sHi, stage, threadIdx.x, base, b, r0, c0, sBa);
        DT_WAVE_LDS_SYNC();
        fwd1s_rows_flush<C>(p, stage, threadIdx.x, base, b, r0, c0);
        DT_WAVE_LDS_SYNC();
    }
}

// Level >= 2 forward: same structure as k_fwd1 (direct column pass, staged record stores).
template <class C>
__global__ void __launch_bounds__(DT_NT) k_fwd2(Fwd2Params p) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_FLOATS + 4 * STAGE_FLOATS_PER_WAVE];
    const int ntile = p.tilesR * p.tilesC * p.B;
    int t = tile_of(blockIdx.x, ntile, p.xcd_order);
    if (t >= ntile) return;
    int tc, tr, b;
    dt_tile_decode(p, t, tc, tr, b);
    float *sLo = smem, *sHi = sLo + C::SL, *sBa = sHi + C::SL, *stage = smem + C::LDS_FLOATS;
    int r0 = tr * C::TR, c0 = tc * C::TC;
    fwd2d_cols<C>(p, sLo, sHi, threadIdx.x, b, r0, c0, sBa);
    __syncthreads();
    for (int base = 0; base < C::TI * C::TJ; base += DT_NT) {
        fwd2s_rows_compute<C>(p, sLo, sHi, stage, threadIdx.x, base, b, r0, c0, sBa);
        DT_WAVE_LDS_SYNC();
        fwd2s_rows_flush<C>(p, stage, threadIdx.x, base, b, r0, c0);
        DT_WAVE_LDS_SYNC();
    }
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// every tile order of tile_of() is a bijection on a grid that is a multiple of 8 x (group size)
inline unsigned grid_for(int ntile, int order = 1) {
    const int q = 8 * (order > 1 ? order : 1);
    return (unsigned)(cdiv(ntile, q) * q);
}

// ------------------------------------------------------------------ launch dispatch
template <class C>
int launch_fwd1(Fwd1Params &p, hipStream_t s) {
    p.tilesR = cdiv(p.LR, C::TR); p.tilesC = cdiv(p.LC, C::TC);
    dt_set_tile_magic(p);
    dt_pack_c01<C::M0, C::M1>(p);
    k_fwd1<C><<<grid_for(p.tilesR * p.tilesC * p.B, p.xcd_order), DT_NT, 0, s>>>(p);
    return 0;
}
template <class C>
int launch_fwd2(Fwd2Params &p, hipStream_t s) {
    p.tilesR = cdiv(p.LR / 2, C::TR); p.tilesC = cdiv(p.LC / 2, C::TC);
    dt_set_tile_magic(p);
    dt_pack_lh(p);
    k_fwd2<C><<<grid_for(p.tilesR * p.tilesC * p.B, p.xcd_order), DT_NT, 0, s>>>(p);
    return 0;
}
// Tile shapes and supported tap lengths live in fused2d_table.hpp (shared with the
// test-only host emulator so both step through identical configurations).
#define DT_CASE_FWD1(TR, TC, RS, A, B) if (m0 == A && m1 == B) return launch_fwd1<Fwd1DCfg<TR, TC, RS, A, B>>(p, s);
#define DT_CASE_FWD2(TR, TC, PS, M) if (m == M) return launch_fwd2<Fwd2DCfg<TR, TC, PS, M>>(p, s);
#define DT_CASE_FWD1_BP(TR, TC, RS, A, B, C2) if (m0 == A && m1 == B && m2 == C2) return launch_fwd1<Fwd1DCfg<TR, TC, RS, A, B, C2>>(p, s);
#define DT_CASE_FWD2_BP(TR, TC, PS, M) if (m == M) return launch_fwd2<Fwd2DCfg<TR, TC, PS, M, true>>(p, s);
// m2 / bp: length of the band-pass filter of a 6-vector biort set / a 12-vector q-shift set (0 / false: none)
int dispatch_fwd1(int m0, int m1, int m2, Fwd1Params &p, hipStream_t s) {
    if (m2) { DT_FWD1_BP_TABLE(DT_CASE_FWD1_BP) return -3; }
    DT_FWD1_TABLE(DT_CASE_FWD1) return -3;
}
int dispatch_fwd2(int m, bool bp, Fwd2Params &p, hipStream_t s) {
    if (bp) { DT_FWD2_BP_TABLE(DT_CASE_FWD2_BP) return -3; }
    DT_FWD2_TABLE(DT_CASE_FWD2) return -3;
}
#define DT_HAS3(TR, TC, RS, A, B, C2) if (m0 == A && m1 == B && m2 == C2) return true;
bool fwd1_bp_supported(int m0, int m1, int m2) { DT_FWD1_BP_TABLE(DT_HAS3) return false; }
bool inv1_bp_supported(int m0, int m1, int m2) { DT_INV1_BP_TABLE(DT_HAS3) return false; }
#define DT_HAS1B(TR, TC, JS, M) if (m == M) return true;
bool q_bp_supported(int m) { DT_INV2_BP_TABLE(DT_HAS1B) return false; }

#define DT_HAS2(TR, TC, A, B) if (m0 == A && m1 == B) return true;
#define DT_HAS2F(TR, TC, RS, A, B) if (m0 == A && m1 == B) return true;
#define DT_HAS1(TR, TC, JS, M) if (m == M) return true;
bool fwd1_supported(int m0, int m1) { DT_FWD1_TABLE(DT_HAS2F) return false; }
bool inv1_supported(int m0, int m1) { DT_INV1_TABLE(DT_HAS2F) return false; }
bool q_supported(int m) { DT_INV2_TABLE(DT_HAS1) return false; }

double dotd(const std::vector<double> &a, const std::vector<double> &b) {
    double s = 0;
    for (size_t k = 0; k < a.size(); ++k) s += a[k] * b[k];
    return s;
}
void put_taps(float *dst, const std::vector<double> &src) {
    for (int k = 0; k < DT_MAXT; ++k) dst[k] = k < (int)src.size() ? (float)src[k] : 0.f;
}

struct Level {
    int inR, inC;       // forward input (real) size
    int padR, padC;     // level >= 2 only
    int LR, LC;         // logical input size
    int loR, loC;       // LoLo output size
    int hR, hC;         // Yh size
};

}  // namespace

struct dtcwt_hip_plan2d {
    dtcwt_hip_ctx *ctx;
    int batch, rows, cols, nlevels;
    int extR, extC;
    std::vector<Level> lv;
    std::vector<double> biort[4];     // h0o g0o h1o g1o
    std::vector<double> qshift[8];    // h0a h0b g0a g0b h1a h1b g1a g1b
    std::vector<double> bp1[2];       // h2o g2o   (band-pass biort, dtcwt_hip_plan2d_set_bandpass)
    std::vector<double> bp2[4];       // h2a h2b g2a g2b
    std::vector<float *> work;        // LoLo / Z per level (level nlevels-1 unused on fwd)
    bool profiling = false;           // record an event pair around every level kernel
    std::vector<hipEvent_t> ev;       // [fwd: 2 per level][inv: 2 per level]
    int xcd_order = -1;               // -1: per-kernel default, 0/1: forced (DTCWT_HIP_XCD_ORDER, read at plan creation)
    DtMarchHint sw;                   // the marching programs' switches as the environment had them when the plan was made
    // tile order of the level-1 forward kernel (tile_of): 0 linear, 1 one contiguous run per XCD, g > 1 groups of g
    // neighbouring tiles per XCD.  Measured at 4096^2 (profiles/r02): g = 8 fetches 74.6 MB for the 67.1 MB image
    // (1.11 x; linear order: 133 MB, 1.99 x, every XCD's L2 fetching its own copy of the shared halo lines) in
    // 60-62 us instead of 64-66; one run per XCD (order 1) is slower (79 us: the write streams thin out).
    int fwd1_order = 8;
    int concurrency = 1;              // independent transforms in flight on the device (dtcwt_hip_plan2d_set_concurrency)
    int program = -1;                 // -1: the library chooses per call, 0: tile programs, 1: marching launches (dtcwt_hip_plan2d_set_program)
    DtMarchHint hint() const { DtMarchHint h = sw; h.cus = ctx->cus; h.nparts = ctx->nparts; h.in_flight = concurrency; h.program = program; return h; }
};

// levels 1 + 2 of the forward / 2 + 1 of the inverse in one marching launch (march2d.hpp): not for the band-pass sets,
// odd-size extension or level-2 padding
static bool plan_march_geometry(const dtcwt_hip_plan2d *p) {
    return p->nlevels >= 2 && p->lv[0].inR == p->lv[0].LR && p->lv[0].inC == p->lv[0].LC && p->lv[1].padR == 0 && p->lv[1].padC == 0;
}
static bool plan_march_fwd12(const dtcwt_hip_plan2d *p) {
    return plan_march_geometry(p) && p->bp1[0].empty() && p->bp2[0].empty() &&
           dtcwt_march_fwd12_ok(p->batch, p->lv[0].LR, p->lv[0].LC, p->biort[0], p->biort[2], p->qshift[0],
                                dotd(p->qshift[1], p->qshift[0]) > 0, dotd(p->qshift[5], p->qshift[4]) > 0, p->hint());
}
// ... or as a marching PAIR of wavefronts, for the sets one wavefront's registers do not hold (near_sym_b; 14- / 18-tap q-shift)
static bool plan_march_fwd12p(const dtcwt_hip_plan2d *p) {
    return plan_march_geometry(p) && p->bp1[0].empty() && p->bp2[0].empty() &&
           dtcwt_march_fwd12p_ok(p->batch, p->lv[0].LR, p->lv[0].LC, p->biort[0], p->biort[2], p->qshift[0],
                                 dotd(p->qshift[1], p->qshift[0]) > 0, dotd(p->qshift[5], p->qshift[4]) > 0, p->hint());
}
static bool plan_march_inv21(const dtcwt_hip_plan2d *p) {
    return plan_march_geometry(p) && p->bp1[1].empty() && p->bp2[2].empty() &&
           dtcwt_march_inv21_ok(p->batch, p->lv[0].LR, p->lv[0].LC, p->biort[1], p->biort[3], p->qshift[2],
                                dotd(p->qshift[3], p->qshift[2]) > 0, dotd(p->qshift[7], p->qshift[6]) > 0, p->hint());
}
static bool plan_march_inv21p(const dtcwt_hip_plan2d *p) {
    return plan_march_geometry(p) && p->bp1[1].empty() && p->bp2[2].empty() &&
           dtcwt_march_inv21p_ok(p->batch, p->lv[0].LR, p->lv[0].LC, p->biort[1], p->biort[3], p->qshift[2],
                                 dotd(p->qshift[3], p->qshift[2]) > 0, dotd(p->qshift[7], p->qshift[6]) > 0, p->hint());
}
// level 2 of the forward alone as a march, where neither the one-wavefront launch nor the pair takes levels 1 + 2
// (`scales`: a forward call with include_scale, which the pair does not serve -- its LoLo1 lives in the LDS exchange only --
// so level 2 of such a call is a march of its own where the q-shift set has one, not a tile program)
static bool plan_march_fwd2(const dtcwt_hip_plan2d *p, bool scales = false) {
    return p->nlevels >= 2 && plan_march_geometry(p) && p->bp2[0].empty() && !plan_march_fwd12(p) && (scales || !plan_march_fwd12p(p)) &&
           dtcwt_march_fwd2_ok(p->batch, p->lv[0].LR, p->lv[0].LC, p->qshift[0], dotd(p->qshift[1], p->qshift[0]) > 0,
                               dotd(p->qshift[5], p->qshift[4]) > 0, p->hint());
}
// level 2 of the inverse alone as a march, where neither the one-wavefront launch nor the pair takes levels 2 + 1
static bool plan_march_inv2(const dtcwt_hip_plan2d *p) {
    return p->nlevels >= 2 && plan_march_geometry(p) && p->bp2[2].empty() && !plan_march_inv21(p) && !plan_march_inv21p(p) &&
           dtcwt_march_inv2_ok(p->batch, p->lv[0].LR, p->lv[0].LC, p->qshift[2], dotd(p->qshift[3], p->qshift[2]) > 0,
                               dotd(p->qshift[7], p->qshift[6]) > 0, p->hint());
}

// level 1 alone as a march (near_sym_b, antonini): no odd-size extension, columns in fours, no band-pass set
static bool plan_march_fwd1(const dtcwt_hip_plan2d *p) {
    const Level &L = p->lv[0];
    return L.inR == L.LR && L.inC == L.LC && p->bp1[0].empty() &&
           dtcwt_march_fwd1_ok(p->batch, L.LR, L.LC, p->biort[0], p->biort[2], p->hint());
}
static bool plan_march_inv1(const dtcwt_hip_plan2d *p) {
    const Level &L = p->lv[0];
    return L.inR == L.LR && L.inC == L.LC && p->bp1[1].empty() &&
           dtcwt_march_inv1_ok(p->batch, L.LR, L.LC, p->biort[1], p->biort[3], p->hint());
}

extern "C" {

int dtcwt_hip_plan2d_level1_march(const dtcwt_hip_plan2d *p, int *fwd1, int *inv1) {
    DT_REQUIRE(p, "NULL plan");
    if (fwd1) *fwd1 = (plan_march_fwd1(p) && !plan_march_fwd12p(p)) ? 1 : 0;
    if (inv1) *inv1 = plan_march_inv1(p) ? 1 : 0;
    return 0;
}

int dtcwt_hip_plan2d_set_concurrency(dtcwt_hip_plan2d *p, int n) {
    DT_REQUIRE(p && n >= 1 && n <= 1024, "transforms in flight: 1 .. 1024");
    p->concurrency = n;
    return 0;
}

int dtcwt_hip_plan2d_set_program(dtcwt_hip_plan2d *p, int program) {
    DT_REQUIRE(p && program >= -1 && program <= 1, "program: DTCWT_HIP_PROGRAM_AUTO (-1), _TILES (0) or _MARCH (1)");
    p->program = program;
    return 0;
}

// Which program runs every level of this plan, and the switches it was made under -- one line of text, e.g.
//   fwd: L1+2 k_fwd12m | L3 k_fwd2 | L4 k_fwd2 ; inv: L4 k_inv2 | L3 k_inv2 | L2+1 k_inv21p ; march=auto band=auto parts=0xff xcd_order=-1 program=auto in_flight=1 cu_shares=1
// `scales` != 0: as a forward with include_scale runs (the marching pair does not serve it).  The same predicates as
// dtcwt_hip_plan2d_forward / _inverse consult: what this says is what runs.
int dtcwt_hip_plan2d_describe(const dtcwt_hip_plan2d *p, int scales, char *buf, size_t len) {
    DT_REQUIRE(p && buf && len >= 64, "NULL argument or a buffer shorter than 64 bytes");
    std::string o = "fwd:";
    const int nl = p->nlevels;
    const bool f12 = plan_march_fwd12(p), f12p = !f12 && !scales && plan_march_fwd12p(p);
    for (int l = 0; l < nl; ++l) {
        char t[96];
        if (l == 0 && (f12 || f12p)) { snprintf(t, sizeof t, " L1+2 %s", f12 ? (scales ? "k_fwd12m+LoLo1" : "k_fwd12m") : "k_fwd12p"); o += t; l = 1; }
        else if (l == 0) { snprintf(t, sizeof t, " L1 %s", plan_march_fwd1(p) ? "k_fwd1m" : "k_fwd1"); o += t; }
        else { snprintf(t, sizeof t, " L%d %s", l + 1, (l == 1 && plan_march_fwd2(p, scales != 0)) ? "k_fwd2m" : "k_fwd2"); o += t; }
        if (l + 1 < nl) o += " |";
    }
    o += " ; inv:";
    const bool i21 = plan_march_inv21(p), i21p = !i21 && plan_march_inv21p(p);
    for (int l = nl - 1; l >= 0; --l) {
        char t[96];
        if (l == 1 && (i21 || i21p)) {
            const DtMarchHint h = p->hint();
            const int nstrip = (p->lv[0].LC + 4 * 58 - 1) / (4 * 58);
            const double useful = (double)p->batch * p->lv[0].LR * p->lv[0].LC * ((double)p->lv[0].LC / (nstrip * 4.0 * 58));
            const bool as_pair = i21 && ((h.parts & DT_PART_INV21_ALWAYS_PAIR) || ((h.parts & DT_PART_INV21_AS_PAIR) && h.in_flight <= 1 && h.nparts <= 1 && useful <= 1.8e7));
            snprintf(t, sizeof t, " L2+1 %s", (i21p || as_pair) ? "k_inv21p" : "k_inv21m"); o += t; break;
        }
        if (l == 0) { snprintf(t, sizeof t, " L1 %s", plan_march_inv1(p) ? "k_inv1m" : "k_inv1"); o += t; }
        else { snprintf(t, sizeof t, " L%d %s |", l + 1, (l == 1 && plan_march_inv2(p)) ? "k_inv2m" : "k_inv2"); o += t; }
    }
    char t[192];
    snprintf(t, sizeof t, " ; march=%s band=%d parts=0x%x xcd_order=%d program=%s in_flight=%d cu_shares=%d",
             p->sw.env_march < 0 ? "auto" : (p->sw.env_march ? "1" : "0"), p->sw.band, p->sw.parts, p->xcd_order,
             p->program < 0 ? "auto" : (p->program ? "march" : "tiles"), p->concurrency, p->ctx->nparts);
    o += t;
    snprintf(buf, len, "%s", o.c_str());
    return 0;
}

int dtcwt_hip_plan2d_launches(const dtcwt_hip_plan2d *p, int *fwd12, int *inv21) {
    DT_REQUIRE(p, "NULL plan");
    if (fwd12) *fwd12 = (plan_march_fwd12(p) || plan_march_fwd12p(p)) ? 1 : 0;
    if (inv21) *inv21 = (plan_march_inv21(p) || plan_march_inv21p(p)) ? 1 : 0;
    return 0;
}

int dtcwt_hip_plan2d_create(dtcwt_hip_ctx *ctx, int batch, int rows, int cols, int nlevels,
                            const double *const *biort_host, const int *biort_len,
                            const double *const *qshift_host, const int *qshift_len,
                            dtcwt_hip_plan2d **out) {
    DT_REQUIRE(ctx && out && biort_host && biort_len && qshift_host && qshift_len, "NULL argument");
    DT_REQUIRE(batch >= 1 && rows >= 1 && cols >= 1 && nlevels >= 1, "bad plan extents");
    for (int i = 0; i < 4; ++i)
        DT_REQUIRE(biort_len[i] >= 1 && biort_len[i] <= DT_MAXT, "biort length out of range");
    for (int i = 0; i < 8; ++i)
        DT_REQUIRE(qshift_len[i] == qshift_len[0] && qshift_len[i] <= DT_MAXT, "qshift lengths differ");
    if (!fwd1_supported(biort_len[0], biort_len[2]) || !inv1_supported(biort_len[1], biort_len[3]) ||
        (nlevels >= 2 && !q_supported(qshift_len[0])))
        return dtcwt_set_error(-3, "no fused kernel for tap lengths biort (%d,%d,%d,%d) qshift %d",
                               biort_len[0], biort_len[1], biort_len[2], biort_len[3], qshift_len[0]);
    dtcwt_hip_plan2d *p = new dtcwt_hip_plan2d();
    p->ctx = ctx; p->batch = batch; p->rows = rows; p->cols = cols; p->nlevels = nlevels;
    for (int i = 0; i < 4; ++i) p->biort[i].assign(biort_host[i], biort_host[i] + biort_len[i]);
    for (int i = 0; i < 8; ++i) p->qshift[i].assign(qshift_host[i], qshift_host[i] + qshift_len[i]);
    { const char *e = getenv("DTCWT_HIP_XCD_ORDER"); p->xcd_order = e ? atoi(e) : -1; }
    p->sw = dt_march_switches();
    p->extR = rows + (rows & 1);
    p->extC = cols + (cols & 1);
    Level l0{rows, cols, 0, 0, p->extR, p->extC, p->extR, p->extC, p->extR / 2, p->extC / 2};
    p->lv.push_back(l0);
    for (int l = 1; l < nlevels; ++l) {
        const Level &pr = p->lv.back();
        Level L;
        L.inR = pr.loR; L.inC = pr.loC;
        L.padR = (L.inR % 4) ? 1 : 0; L.padC = (L.inC % 4) ? 1 : 0;
        L.LR = L.inR + 2 * L.padR; L.LC = L.inC + 2 * L.padC;
        L.loR = L.LR / 2; L.loC = L.LC / 2;
        L.hR = L.LR / 4; L.hC = L.LC / 4;
        p->lv.push_back(L);
    }
    for (const Level &L : p->lv)
        if (L.LR < DT_MIN_FUSED_DIM || L.LC < DT_MIN_FUSED_DIM || L.loR < DT_MIN_FUSED_DIM / 2 || L.loC < DT_MIN_FUSED_DIM / 2) {
            delete p;
            return dtcwt_set_error(-3, "image too small for the fused kernels at some level (< %d samples): use the generic path", DT_MIN_FUSED_DIM);
        }
    p->work.assign(nlevels, nullptr);
    for (int l = 0; l < nlevels; ++l) {
        size_t bytes = (size_t)batch * p->lv[l].loR * p->lv[l].loC * sizeof(float);
        void *d = nullptr;
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess) e = hipMalloc(&d, bytes ? bytes : 16);
        if (e != hipSuccess) {
            for (float *w : p->work) if (w) (void)hipFree(w);
            delete p;
            return dtcwt_set_error(-2, "hipMalloc of plan workspace failed: %s", hipGetErrorString(e));
        }
        p->work[l] = (float *)d;
    }
    *out = p;
    return 0;
}

int dtcwt_hip_plan2d_destroy(dtcwt_hip_plan2d *p) {
    if (!p) return 0;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    for (float *w : p->work) if (w) (void)hipFree(w);
    for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
    delete p;
    return 0;
}

int dtcwt_hip_plan2d_set_bandpass(dtcwt_hip_plan2d *p, const double *h2o, const double *g2o, int m_biort,
                                  const double *h2a, const double *h2b, const double *g2a, const double *g2b,
                                  int m_qshift) {
    DT_REQUIRE(p, "NULL plan");
    const bool b1 = m_biort > 0, b2 = m_qshift > 0;
    DT_REQUIRE(!b1 || (h2o && g2o && m_biort <= DT_MAXT), "band-pass biort pair missing or too long");
    DT_REQUIRE(!b2 || (h2a && h2b && g2a && g2b && m_qshift <= DT_MAXT), "band-pass q-shift vectors missing or too long");
    if (b1 && (!fwd1_bp_supported((int)p->biort[0].size(), (int)p->biort[2].size(), m_biort) ||
               !inv1_bp_supported((int)p->biort[1].size(), (int)p->biort[3].size(), m_biort)))
        return dtcwt_set_error(-3, "no fused band-pass level-1 kernel for biort lengths (%d,%d,%d)",
                               (int)p->biort[0].size(), (int)p->biort[2].size(), m_biort);
    if (b2 && p->nlevels >= 2 && (m_qshift != (int)p->qshift[0].size() || !q_bp_supported(m_qshift)))
        return dtcwt_set_error(-3, "no fused band-pass level >= 2 kernel for %d-tap q-shift filters", m_qshift);
    for (auto &v : p->bp1) v.clear();
    for (auto &v : p->bp2) v.clear();
    if (b1) { p->bp1[0].assign(h2o, h2o + m_biort); p->bp1[1].assign(g2o, g2o + m_biort); }
    if (b2) {
        p->bp2[0].assign(h2a, h2a + m_qshift); p->bp2[1].assign(h2b, h2b + m_qshift);
        p->bp2[2].assign(g2a, g2a + m_qshift); p->bp2[3].assign(g2b, g2b + m_qshift);
    }
    return 0;
}

int dtcwt_hip_plan2d_set_profiling(dtcwt_hip_plan2d *p, int enable) {
    DT_REQUIRE(p, "NULL plan");
    if (enable && p->ev.empty()) {
        DT_CHECK_HIP(hipSetDevice(p->ctx->device));
        p->ev.resize(4 * p->nlevels);
        for (auto &e : p->ev) DT_CHECK_HIP(hipEventCreate(&e));
    }
    p->profiling = enable != 0;
    return 0;
}

int dtcwt_hip_plan2d_kernel_ms(dtcwt_hip_plan2d *p, float *fwd_ms, float *inv_ms) {
    DT_REQUIRE(p && p->profiling && !p->ev.empty(), "profiling is not enabled on this plan");
    DT_CHECK_HIP(hipStreamSynchronize(p->ctx->stream));
    const int nl = p->nlevels;
    for (int l = 0; l < nl; ++l) {
        if (fwd_ms) DT_CHECK_HIP(hipEventElapsedTime(&fwd_ms[l], p->ev[2 * l], p->ev[2 * l + 1]));
        if (inv_ms) DT_CHECK_HIP(hipEventElapsedTime(&inv_ms[l], p->ev[2 * (nl + l)], p->ev[2 * (nl + l) + 1]));
    }
    return 0;
}

int dtcwt_hip_plan2d_shapes(const dtcwt_hip_plan2d *p, int *s) {
    DT_REQUIRE(p && s, "NULL argument");
    s[0] = p->extR; s[1] = p->extC;
    s[2] = p->lv.back().loR; s[3] = p->lv.back().loC;
    for (int l = 0; l < p->nlevels; ++l) {
        s[4 + 4 * l] = p->lv[l].hR; s[5 + 4 * l] = p->lv[l].hC;
        s[6 + 4 * l] = p->lv[l].loR; s[7 + 4 * l] = p->lv[l].loC;
    }
    return 0;
}

int dtcwt_hip_plan2d_forward(dtcwt_hip_plan2d *p, const float *X, float *Yl, void *const *Yh,
                             float *const *Ys) {
    DT_REQUIRE(p && X && Yl && Yh, "NULL argument");
    DT_CHECK_HIP(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    const int nl = p->nlevels;
    const float *in = X;
    // Levels 1 + 2 in one launch (march2d.hpp) when the image needs no odd-size extension or level-2 padding and the
    // filters are ones the marching program is built for; with `scales` the launch stores the level-1 lowpass as well
    const bool march12 = plan_march_fwd12(p);
    for (int l = 0; l < nl; ++l) {
        const Level &L = p->lv[l];
        float *lo = (l == nl - 1 && !Ys) ? Yl : (Ys ? Ys[l] : p->work[l]);
        DT_REQUIRE(lo && Yh[l], "NULL output buffer at level %d", l);
        int rc;
        if (p->profiling) DT_CHECK_HIP(hipEventRecord(p->ev[2 * l], s));
        if (l == 0 && march12) {
            DT_REQUIRE(Yh[1], "NULL output buffer at level 1");
            float *lo2 = Ys ? Ys[1] : (nl == 2 ? Yl : p->work[1]);
            DT_REQUIRE(lo2 && (!Ys || Ys[0]), "NULL scale buffer");
            Fwd2Params q{};
            put_taps(q.l_a, p->qshift[1]); put_taps(q.l_b, p->qshift[0]);
            put_taps(q.h_a, p->qshift[5]); put_taps(q.h_b, p->qshift[4]);
            rc = dtcwt_march_fwd12(in, (float *)Yh[0], (float *)Yh[1], lo2, Ys ? Ys[0] : nullptr, p->batch, L.LR, L.LC, p->biort[0], p->biort[2],
                                   q.l_a, q.l_b, q.h_a, q.h_b, (int)p->qshift[0].size(),
                                   dotd(p->qshift[1], p->qshift[0]) > 0, dotd(p->qshift[5], p->qshift[4]) > 0, p->hint(), s);
            if (rc) return dtcwt_set_error(rc, "no marching forward kernel for levels 1 + 2");
            DT_CHECK_HIP(hipGetLastError());
            if (p->profiling) {     // level 2 has no launch of its own: an empty event pair
                DT_CHECK_HIP(hipEventRecord(p->ev[1], s));
                DT_CHECK_HIP(hipEventRecord(p->ev[2], s));
                DT_CHECK_HIP(hipEventRecord(p->ev[3], s));
            }
            in = lo2;
            l = 1;
            continue;
        }
        if (l == 0 && !Ys && plan_march_fwd12p(p)) {
            DT_REQUIRE(Yh[1], "NULL output buffer at level 1");
            float *lo2 = nl == 2 ? Yl : p->work[1];
            Fwd2Params q{};
            put_taps(q.l_a, p->qshift[1]); put_taps(q.l_b, p->qshift[0]);
            put_taps(q.h_a, p->qshift[5]); put_taps(q.h_b, p->qshift[4]);
            rc = dtcwt_march_fwd12p(in, (float *)Yh[0], (float *)Yh[1], lo2, p->batch, L.LR, L.LC, p->biort[0], p->biort[2],
                                    q.l_a, q.l_b, q.h_a, q.h_b, (int)p->qshift[0].size(), p->hint(), s);
            if (rc) return dtcwt_set_error(rc, "no marching-pair forward kernel for levels 1 + 2");
            DT_CHECK_HIP(hipGetLastError());
            if (p->profiling) {
                DT_CHECK_HIP(hipEventRecord(p->ev[1], s));
                DT_CHECK_HIP(hipEventRecord(p->ev[2], s));
                DT_CHECK_HIP(hipEventRecord(p->ev[3], s));
            }
            in = lo2;
            l = 1;
            continue;
        }
        if (l == 0 && plan_march_fwd1(p)) {
            rc = dtcwt_march_fwd1(in, lo, (float *)Yh[0], p->batch, L.LR, L.LC, p->biort[0], p->biort[2], p->hint(), s);
        } else if (l == 0) {
            Fwd1Params q{};
            q.X = in; q.LoLo = lo; q.Yh = (float *)Yh[l];
            q.B = p->batch; q.inR = L.inR; q.inC = L.inC; q.LR = L.LR; q.LC = L.LC;
            q.xcd_order = p->xcd_order < 0 ? p->fwd1_order : p->xcd_order;      // write-heavy: linear order unless told otherwise
            put_taps(q.h0, p->biort[0]); put_taps(q.h1, p->biort[2]); put_taps(q.h2, p->bp1[0]);
            rc = dispatch_fwd1((int)p->biort[0].size(), (int)p->biort[2].size(), (int)p->bp1[0].size(), q, s);
        } else {
            Fwd2Params q{};
            q.X = in; q.LoLo = lo; q.Yh = (float *)Yh[l];
            q.B = p->batch; q.inR = L.inR; q.inC = L.inC; q.padR = L.padR; q.padC = L.padC;
            q.LR = L.LR; q.LC = L.LC; q.xcd_order = p->xcd_order < 0 ? 1 : p->xcd_order;
            // record arrays of 32 MiB and more cannot wait in the caches for whoever reads them next:
            // stream them past, so that they do not evict the lowpass plane the next level reads
            q.stream_records = (int64_t)p->batch * (L.LR / 4) * (L.LC / 4) * 48 >= stream_records_bytes();
            // coldfilt(X, h0b, h0a) / coldfilt(X, h1b, h1a)   (transform2d.py:143-157)
            put_taps(q.l_a, p->qshift[1]); put_taps(q.l_b, p->qshift[0]);
            put_taps(q.h_a, p->qshift[5]); put_taps(q.h_b, p->qshift[4]);
            q.lo_a_first = dotd(p->qshift[1], p->qshift[0]) > 0;
            q.hi_a_first = dotd(p->qshift[5], p->qshift[4]) > 0;
            const bool bp = !p->bp2[0].empty();     // coldfilt(X, h2b, h2a) for the diagonal subbands
            if (bp) {
                put_taps(q.b_a, p->bp2[1]); put_taps(q.b_b, p->bp2[0]);
                q.bp_a_first = dotd(p->bp2[1], p->bp2[0]) > 0;
            }
            if (l == 1 && plan_march_fwd2(p, Ys != nullptr)) {          // level 2 alone as a march (k_fwd2m): the level-1 lowpass -> Yh[1], LoLo2
                rc = dtcwt_march_fwd2(in, (float *)Yh[1], lo, p->batch, p->lv[0].LR, p->lv[0].LC, q.l_a, q.l_b, q.h_a, q.h_b,
                                      (int)p->qshift[0].size(), p->hint(), s);
                if (rc) return dtcwt_set_error(rc, "no marching level-2 forward kernel");
                DT_CHECK_HIP(hipGetLastError());
                if (p->profiling) DT_CHECK_HIP(hipEventRecord(p->ev[2 * l + 1], s));
                in = lo;
                continue;
            }
            // (8-row tiles for the coarsest levels measured no gain for the forward -- those launches sit at a 3-8 us floor
            // either way: their builds and DTCWT_HIP_SMALL_TILES are gone; the inverse keeps its 8 x 64 tiles below)
            rc = dispatch_fwd2((int)p->qshift[0].size(), bp, q, s);
        }
        if (rc) return dtcwt_set_error(rc, "no fused forward kernel at level %d", l);
        DT_CHECK_HIP(hipGetLastError());
        if (p->profiling) DT_CHECK_HIP(hipEventRecord(p->ev[2 * l + 1], s));
        in = lo;
    }
    if (Ys) {   // Yl is a separate buffer from the last scale
        const Level &L = p->lv[nl - 1];
        DT_CHECK_HIP(hipMemcpyAsync(Yl, Ys[nl - 1], (size_t)p->batch * L.loR * L.loC * sizeof(float),
                                    hipMemcpyDeviceToDevice, s));
    }
    return 0;
}

int dtcwt_hip_plan2d_inverse(dtcwt_hip_plan2d *p, const float *Yl, const void *const *Yh,
                             const double *gain, float *Z) {
    DT_REQUIRE(p && Yl && Yh && Z, "NULL argument");
    DT_CHECK_HIP(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    const int nl = p->nlevels;
    const double rs = 0.70710678118654752440;
    const float *in = Yl;
    // levels 2 + 1 in one launch (march2d.hpp) under the same conditions as the forward's levels 1 + 2
    const bool march21 = plan_march_inv21(p);
    const bool march21p = !march21 && plan_march_inv21p(p);        // ... as a marching pair of wavefronts (the 14- / 18-tap q-shift sets)
    for (int l = nl - 1; l >= 0; --l) {
        const Level &L = p->lv[l];
        DT_REQUIRE(Yh[l], "NULL Yh at level %d", l);
        float g[6];
        for (int d = 0; d < 6; ++d) g[d] = (float)(rs * (gain ? gain[d * nl + l] : 1.0));
        int rc;
        if (p->profiling) DT_CHECK_HIP(hipEventRecord(p->ev[2 * (nl + l)], s));
        if (l == 1 && (march21 || march21p)) {
            DT_REQUIRE(Yh[0], "NULL Yh at level 0");
            float g1[6];
            for (int d = 0; d < 6; ++d) g1[d] = (float)(rs * (gain ? gain[d * nl + 0] : 1.0));
            Inv2Params q{};
            put_taps(q.l_a, p->qshift[3]); put_taps(q.l_b, p->qshift[2]);
            put_taps(q.h_a, p->qshift[7]); put_taps(q.h_b, p->qshift[6]);
            if (march21p)
                rc = dtcwt_march_inv21p(in, (const float *)Yh[1], (const float *)Yh[0], Z, p->batch, p->lv[0].LR, p->lv[0].LC,
                                        p->biort[1], p->biort[3], q.l_a, q.l_b, q.h_a, q.h_b, (int)p->qshift[2].size(), g1, g, p->hint(), s);
            else
            rc = dtcwt_march_inv21(in, (const float *)Yh[1], (const float *)Yh[0], Z, p->batch, p->lv[0].LR, p->lv[0].LC,
                                   p->biort[1], p->biort[3], q.l_a, q.l_b, q.h_a, q.h_b, g1, g, p->hint(), s);
            if (rc) return dtcwt_set_error(rc, "no marching inverse kernel for levels 2 + 1");
            DT_CHECK_HIP(hipGetLastError());
            if (p->profiling) {     // level 1 has no launch of its own: an empty event pair
                DT_CHECK_HIP(hipEventRecord(p->ev[2 * (nl + 1) + 1], s));
                DT_CHECK_HIP(hipEventRecord(p->ev[2 * nl], s));
                DT_CHECK_HIP(hipEventRecord(p->ev[2 * nl + 1], s));
            }
            break;
        }
        if (l == 0 && plan_march_inv1(p)) {
            rc = dtcwt_march_inv1(in, (const float *)Yh[0], Z, p->batch, L.LR, L.LC, p->biort[1], p->biort[3], g, p->hint(), s);
        } else if (l == 0) {
            Inv1Params q{};
            q.Z = in; q.Yh = (const float *)Yh[0]; q.X = Z;
            q.B = p->batch; q.R = L.LR; q.C = L.LC; q.xcd_order = p->xcd_order < 0 ? 1 : p->xcd_order;
            for (int d = 0; d < 6; ++d) q.g[d] = g[d];
            put_taps(q.g0, p->biort[1]); put_taps(q.g1, p->biort[3]); put_taps(q.g2, p->bp1[1]);
            rc = dtcwt_dispatch_inv1((int)p->biort[1].size(), (int)p->biort[3].size(), (int)p->bp1[1].size(), q, s);
        } else {
            Inv2Params q{};
            float *out = p->work[l - 1];          // size of LoLo_{l-1} = this level's input
            q.Z = in; q.Yh = (const float *)Yh[l]; q.Out = out;
            q.B = p->batch; q.zr = L.loR; q.zc = L.loC; q.cropR = L.padR; q.cropC = L.padC;
            q.xcd_order = p->xcd_order < 0 ? 1 : p->xcd_order;
            for (int d = 0; d < 6; ++d) q.g[d] = g[d];
            // colifilt(X, g0b, g0a) / colifilt(X, g1b, g1a)   (transform2d.py:248-260)
            put_taps(q.l_a, p->qshift[3]); put_taps(q.l_b, p->qshift[2]);
            put_taps(q.h_a, p->qshift[7]); put_taps(q.h_b, p->qshift[6]);
            q.lo_pos = dotd(p->qshift[3], p->qshift[2]) > 0;
            q.hi_pos = dotd(p->qshift[7], p->qshift[6]) > 0;
            const bool bp = !p->bp2[2].empty();     // colifilt(., g2b, g2a) on the diagonal plane
            if (bp) {
                put_taps(q.b_a, p->bp2[3]); put_taps(q.b_b, p->bp2[2]);
                q.bp_pos = dotd(p->bp2[3], p->bp2[2]) > 0;
            }
            if (l == 1 && plan_march_inv2(p)) {          // level 2 alone as a march (k_inv2m): Z2 + Yh[1] -> the level-1 lowpass
                rc = dtcwt_march_inv2(in, (const float *)Yh[1], out, p->batch, p->lv[0].LR, p->lv[0].LC, q.l_a, q.l_b, q.h_a, q.h_b,
                                      (int)p->qshift[2].size(), g, p->hint(), s);
                in = out;
                if (rc) return dtcwt_set_error(rc, "no marching level-2 inverse kernel");
                DT_CHECK_HIP(hipGetLastError());
                if (p->profiling) DT_CHECK_HIP(hipEventRecord(p->ev[2 * (nl + l) + 1], s));
                continue;
            }
            // the coarsest levels of a single image (fewer than two 16 x 56 tiles per CU): 8 x 64 tiles, twice the
            // workgroups of half the rows each (fused2d_table.hpp)
            const bool small = (int64_t)cdiv(L.loR, 16) * cdiv(L.loC, 56) * p->batch < DT_INV2_SMALL_BELOW;
            rc = dtcwt_dispatch_inv2((int)p->qshift[0].size(), bp, q, s, small);
            in = out;
        }
        if (rc) return dtcwt_set_error(rc, "no fused inverse kernel at level %d", l);
        DT_CHECK_HIP(hipGetLastError());
        if (p->profiling) DT_CHECK_HIP(hipEventRecord(p->ev[2 * (nl + l) + 1], s));
    }
    return 0;
}

// ---- hipGraph of a forward (+ inverse) into fixed buffers ---------------------------------------
struct dtcwt_hip_graph {
    dtcwt_hip_ctx *ctx;
    hipGraph_t graph;
    hipGraphExec_t exec;
};

int dtcwt_hip_plan2d_capture(dtcwt_hip_plan2d *p, const float *X, float *Yl, void *const *Yh,
                             float *const *Ys, const double *gain_mask_host, float *Z,
                             dtcwt_hip_graph **out) {
    DT_REQUIRE(p && X && Yl && Yh && out, "NULL argument");
    DT_REQUIRE(!p->profiling, "switch per-kernel profiling off before capturing a graph");
    DT_CHECK_HIP(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    DT_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int rc = dtcwt_hip_plan2d_forward(p, X, Yl, Yh, Ys);
    if (!rc && Z) rc = dtcwt_hip_plan2d_inverse(p, Yl, (const void *const *)Yh, gain_mask_host, Z);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(s, &g);
    if (rc) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    DT_CHECK_HIP(e);
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g);
        return dtcwt_set_error(-2, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    }
    *out = new dtcwt_hip_graph{p->ctx, g, ex};
    return 0;
}

int dtcwt_hip_graph_launch(dtcwt_hip_graph *g) {
    DT_REQUIRE(g, "graph is NULL");
    DT_CHECK_HIP(hipSetDevice(g->ctx->device));
    DT_CHECK_HIP(hipGraphLaunch(g->exec, g->ctx->stream));
    return 0;
}

int dtcwt_hip_graph_destroy(dtcwt_hip_graph *g) {
    if (!g) return 0;
    (void)hipGraphExecDestroy(g->exec);
    (void)hipGraphDestroy(g->graph);
    delete g;
    return 0;
}

}  // extern "C"
