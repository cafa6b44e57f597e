// Marching wavefront programs (dtcwt_amd/csrc/march2d.hpp) against the tile programs of the library:
// same input, outputs compared, then timed over four rotating buffer sets with the knock-outs of ko_bench
// (cached loads / stores to a few rows / both = arithmetic only).  Measurement tool, not product.
//
//   make -C tools/kbench march_bench && tools/kbench/march_bench [N=4096] [reps=40]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>

#include "fused2d_tiles.hpp"
#include "fused2d_tiles_v2.hpp"
#include "march2d.hpp"
#include "march_experiments.hpp"

using namespace dt2d;

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline unsigned grid_for(int ntile, int order = 1) {
    const int q = 8 * (order > 1 ? order : 1);
    return (unsigned)(cdiv(ntile, q) * q);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class C>
__global__ void __launch_bounds__(DT_NT) ref_fwd1(Fwd1Params p) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_FLOATS + 4 * STAGE_FLOATS_PER_WAVE];
    const int ntile = p.tilesR * p.tilesC * p.B;
    int t = tile_of(blockIdx.x, ntile, p.xcd_order);
    if (t >= ntile) return;
    int tc, tr, b;
    dt_tile_decode(p, t, tc, tr, b);
    float *sLo = smem, *sHi = sLo + C::SL, *stage = smem + C::LDS_FLOATS;
    fwd1d_cols<C>(p, sLo, sHi, threadIdx.x, b, tr * C::TR, tc * C::TC, nullptr);
    __syncthreads();
    const int r0 = tr * C::TR, c0 = tc * C::TC;
    constexpr int NQ = (C::TR / 2) * (C::TC / 2);
    for (int base = 0; base < NQ; base += DT_NT) {
        fwd1s_rows_compute<C>(p, sLo, sHi, stage, threadIdx.x, base, b, r0, c0, nullptr);
        DT_WAVE_LDS_SYNC();
        fwd1s_rows_flush<C>(p, stage, threadIdx.x, base, b, r0, c0);
        DT_WAVE_LDS_SYNC();
    }
}

template <class C>
__global__ void __launch_bounds__(DT_NT) ref_fwd2(Fwd2Params p) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_FLOATS + 4 * STAGE_FLOATS_PER_WAVE];
    const int ntile = p.tilesR * p.tilesC * p.B;
    int t = tile_of(blockIdx.x, ntile, p.xcd_order);
    if (t >= ntile) return;
    int tc, tr, b;
    dt_tile_decode(p, t, tc, tr, b);
    float *sLo = smem, *sHi = sLo + C::SL, *stage = smem + C::LDS_FLOATS;
    fwd2d_cols<C>(p, sLo, sHi, threadIdx.x, b, tr * C::TR, tc * C::TC, nullptr);
    __syncthreads();
    const int r0 = tr * C::TR, c0 = tc * C::TC;
    for (int base = 0; base < C::TI * C::TJ; base += DT_NT) {
        fwd2s_rows_compute<C>(p, sLo, sHi, stage, threadIdx.x, base, b, r0, c0, nullptr);
        DT_WAVE_LDS_SYNC();
        fwd2s_rows_flush<C>(p, stage, threadIdx.x, base, b, r0, c0);
        DT_WAVE_LDS_SYNC();
    }
}

template <class C>
__global__ void __launch_bounds__(DT_NT) ref_inv1(Inv1Params p) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_ALIASED];
    const int ntile = p.tilesR * p.tilesC * p.B;
    int t = tile_of(blockIdx.x, ntile, p.xcd_order);
    if (t >= ntile) return;
    int tc, tr, b;
    dt_tile_decode(p, t, tc, tr, b);
    float *srec = smem, *y1 = smem, *y2 = y1 + C::SY;
    const int r0 = tr * C::TR, c0 = tc * C::TC;
    const float *Yhb = p.Yh + (int64_t)b * (p.R / 2) * (p.C / 2) * 12;
    float wz[C::WN], w1[C::WN], w2[C::WN], w3[C::WN];
    inv1r_fetch<C>(p, wz, threadIdx.x, b, r0, c0);
    inv_rec_stage<C::QR, C::QC>(Yhb, p.R, p.C, srec, r0 - C::HE, c0 - C::HE, threadIdx.x);
    __syncthreads();
    inv1r_gather<C>(p, srec, w1, w2, w3, threadIdx.x, r0, c0);
    __syncthreads();
    inv1r_fir<C>(p, wz, w1, w2, w3, y1, y2, threadIdx.x, nullptr);
    __syncthreads();
    inv1d_rows<C>(p, y1, y2, threadIdx.x, b, r0, c0, nullptr);
}
template <class C>
__global__ void __launch_bounds__(DT_NT) ref_inv2(Inv2Params p) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_ALIASED];
    const int ntile = p.tilesR * p.tilesC * p.B;
    int t = tile_of(blockIdx.x, ntile, p.xcd_order);
    if (t >= ntile) return;
    int tc, tr, b;
    dt_tile_decode(p, t, tc, tr, b);
    float *srec = smem, *y1 = smem, *y2 = y1 + C::SY;
    const int r0 = tr * C::TR, c0 = tc * C::TC;
    const float *Yhb = p.Yh + (int64_t)b * (p.zr / 2) * (p.zc / 2) * 12;
    float wz[C::WS], w1[C::WS], w2[C::WS], w3[C::WS];
    inv2r_fetch<C>(p, wz, threadIdx.x, b, r0, c0);
    inv_rec_stage<C::QR, C::QC>(Yhb, p.zr, p.zc, srec, r0 + C::ORG, c0 + C::ORG, threadIdx.x);
    __syncthreads();
    inv2r_gather<C>(p, srec, w1, w2, w3, threadIdx.x, r0, c0);
    __syncthreads();
    inv2r_fir<C, false, true>(p, wz, w1, w2, w3, y1, y2, threadIdx.x, nullptr);
    __syncthreads();
    inv2_rows<C, true>(p, y1, y2, threadIdx.x, b, r0, c0, nullptr);
}
static const double G0O[7] = {-0.0107142857142857, -0.0535714285714286, 0.260714285714286, 0.607142857142857,
                              0.260714285714286, -0.0535714285714286, -0.0107142857142857};
static const double G1O[5] = {-0.05, -0.25, 0.6, -0.25, -0.05};

static const double H0A[10] = {0.03516384, 0., -0.08832942, 0.23389032, 0.76027237, 0.5875183, 0., -0.11430184, 0., 0.};
static const double H1A[10] = {0., 0., -0.11430184, 0., 0.5875183, -0.76027237, 0.23389032, 0.08832942, 0., -0.03516384};
static void putr(float *dst, const double *src, int n, bool rev) { for (int k = 0; k < DT_MAXT; ++k) dst[k] = k < n ? (float)src[rev ? n - 1 - k : k] : 0.f; }

static const double H0O[5] = {-0.05, 0.25, 0.6, 0.25, -0.05};
static const double H1O[7] = {-0.0107142857142857, 0.0535714285714286, 0.260714285714286, -0.607142857142857,
                              0.260714285714286, 0.0535714285714286, -0.0107142857142857};
static void put(float *dst, const double *src, int n, int cap) { for (int k = 0; k < cap; ++k) dst[k] = k < n ? (float)src[k] : 0.f; }

constexpr int NSET = 4;
struct Set { float *X, *L1, *Y0, *L1r, *Y0r, *L2, *Y1, *L2r, *Y1r, *Xo, *Xor; };
static Set sets[NSET];
static int N = 4096, REPS = 40;
static hipStream_t st, st_a, st_b;

template <class F>
static double time_it(F launch) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 8; ++i) launch(i % NSET);
    CK(hipEventRecord(a, st));
    for (int i = 0; i < REPS; ++i) launch(i % NSET);
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms * 1e3 / REPS;
}

// two independent images in flight: launches alternate over two streams (bench.py's default protocol)
template <class F>
static double time_two_streams(F launch) {
    for (int i = 0; i < 8; ++i) { st = (i & 1) ? st_b : st_a; launch(i % NSET); }
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < REPS; ++i) { st = (i & 1) ? st_b : st_a; launch(i % NSET); }
    CK(hipDeviceSynchronize());
    const auto t1 = std::chrono::steady_clock::now();
    st = st_a;
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / REPS;
}

using F1 = Fwd1DCfg<32, 64, 8, 5, 7>;
using F2 = Fwd2DCfg<16, 56, 4, 10>;

static void fill_fwd2(Fwd2Params &p) {
    putr(p.l_a, H0A, 10, true); putr(p.l_b, H0A, 10, false); putr(p.h_a, H1A, 10, true); putr(p.h_b, H1A, 10, false);
    p.lo_a_first = 1; p.hi_a_first = 0; dt_pack_lh(p);
}
static void launch_ref2(float *L1, float *L2, float *Y1) {
    Fwd2Params p{}; p.X = L1; p.LoLo = L2; p.Yh = Y1; p.B = 1; p.inR = p.inC = p.LR = p.LC = N;
    p.xcd_order = 1; p.stream_records = 1; fill_fwd2(p);
    p.tilesR = cdiv(N / 2, F2::TR); p.tilesC = cdiv(N / 2, F2::TC);
    dt_set_tile_magic(p);
    ref_fwd2<F2><<<grid_for(p.tilesR * p.tilesC), DT_NT, 0, st>>>(p);
}
template <int P, int KO, int WPS = 2>
static void launch_f12(int s, int band_rows) {
    using G = dtm::Fwd12m<5, 7, 10>;
    dtm::Fwd12mParams p{}; p.X = sets[s].X; p.LoLo1 = nullptr; p.Yh0 = sets[s].Y0; p.Yh1 = sets[s].Y1; p.LoLo2 = sets[s].L2; p.B = 1; p.R = p.C = N;
    const unsigned jobs = dtm::dtm_set_jobs(p.jb, 1, N, cdiv(N, 4 * G::VL), band_rows);
    put(p.h0, H0O, 5, dtm::MAXT1); put(p.h1, H1O, 7, dtm::MAXT1);
    Fwd2Params q{}; fill_fwd2(q);
    dtm::dtm_pack_qshift(p, 10, q.l_a, q.l_b, q.h_a, q.h_b); dtm::dtm_pack_biort(p, 5, 7); dtm::dtm_pack_biort_scaled(p, 5, 7, H0O, H1O); p.lo_a_first = q.lo_a_first; p.hi_a_first = q.hi_a_first;
    dtm::k_fwd12m<5, 7, 10, P, KO, WPS><<<jobs, 64, 0, st>>>(p);
}
using I1 = Inv1RCfg<16, 120, 8, 7, 5>;
using I2 = Inv2RCfg<16, 56, 2, 10>;
static const float GAINS[6] = {0.70710678f, 0.6f, 0.8f, 0.70710678f, 0.5f, 0.9f};
static void fill_inv2(Inv2Params &p) {     // any 10-tap values with the standard phases (as tools/kbench/ko_bench)
    putr(p.l_a, H0A, 10, false); putr(p.l_b, H0A, 10, true); putr(p.h_a, H1A, 10, false); putr(p.h_b, H1A, 10, true);
    p.lo_pos = 1; p.hi_pos = 0;
    for (int d = 0; d < 6; ++d) p.g[d] = GAINS[d];
}
// the library's two tile programs: Z2 = sets[s].L2, Yh1 = Y1, Yh0 = Y0 -> Z1 (L1) -> X
static void launch_ref_inv(int s, float *Z1, float *Xout) {
    Inv2Params q{}; q.Z = sets[s].L2; q.Yh = sets[s].Y1; q.Out = Z1; q.B = 1; q.zr = q.zc = N / 2; q.xcd_order = 1; fill_inv2(q);
    q.tilesR = cdiv(N / 2, I2::TR); q.tilesC = cdiv(N / 2, I2::TC); dt_set_tile_magic(q);
    ref_inv2<I2><<<grid_for(q.tilesR * q.tilesC), DT_NT, 0, st>>>(q);
    Inv1Params p{}; p.Z = Z1; p.Yh = sets[s].Y0; p.X = Xout; p.B = 1; p.R = p.C = N; p.xcd_order = 1;
    for (int d = 0; d < 6; ++d) p.g[d] = GAINS[5 - d];
    put(p.g0, G0O, 7, DT_MAXT); put(p.g1, G1O, 5, DT_MAXT); dt_pack_g01<7, 5>(p);
    p.tilesR = cdiv(N, I1::TR); p.tilesC = cdiv(N, I1::TC); dt_set_tile_magic(p);
    ref_inv1<I1><<<grid_for(p.tilesR * p.tilesC), DT_NT, 0, st>>>(p);
}
template <int KO>
static void launch_i21(int s, int band_rows, float *Xout) {
    using G = dtm::Inv21m<7, 5, 10>;
    dtm::Inv21mParams p{}; p.Z2 = sets[s].L2; p.Yh1 = sets[s].Y1; p.Yh0 = sets[s].Y0; p.X = Xout; p.B = 1; p.R = p.C = N;
    const unsigned jobs = dtm::dtm_set_jobs(p.jb, 1, N, cdiv(N, 4 * G::VL), band_rows);
    Inv2Params q{}; fill_inv2(q);
    for (int k = 0; k < dtm::MAXT2; ++k) { p.l_a[k] = q.l_a[k]; p.l_b[k] = q.l_b[k]; p.h_a[k] = q.h_a[k]; p.h_b[k] = q.h_b[k]; }
    for (int d = 0; d < 6; ++d) { p.g2[d] = GAINS[d]; p.g1[d] = GAINS[5 - d]; }
    put(p.g0o, G0O, 7, dtm::MAXT1); put(p.g1o, G1O, 5, dtm::MAXT1); dtm::dtm_pack_inv_biort(p, 7, 5);
    dtm::k_inv21m<7, 5, 10, KO><<<jobs, 64, 0, st>>>(p);
}
static void run_i21(int band_rows) {
    const double a = time_it([&](int s) { launch_i21<0>(s, band_rows, sets[s].Xo); });
    const double b = time_it([&](int s) { launch_i21<1>(s, band_rows, sets[s].Xo); });
    const double c = time_it([&](int s) { launch_i21<2>(s, band_rows, sets[s].Xo); });
    const double d = time_it([&](int s) { launch_i21<3>(s, band_rows, sets[s].Xo); });
    printf("k_inv21m band_rows=%3d                    %9.2f %9.2f %9.2f %9.2f\n", band_rows, a, b, c, d);
    fflush(stdout);
}
template <int P, int KO>
static void launch_f12w(int s, int band_rows) {
    using G = dtm::Fwd12m<5, 7, 10>;
    dtm::Fwd12mParams p{}; p.X = sets[s].X; p.LoLo1 = nullptr; p.Yh0 = sets[s].Y0; p.Yh1 = sets[s].Y1; p.LoLo2 = sets[s].L2; p.B = 1; p.R = p.C = N;
    const unsigned jobs = dtm::dtm_set_jobs(p.jb, 1, N, cdiv(N, 4 * G::VL), band_rows);
    put(p.h0, H0O, 5, dtm::MAXT1); put(p.h1, H1O, 7, dtm::MAXT1);
    Fwd2Params q{}; fill_fwd2(q);
    dtm::dtm_pack_qshift(p, 10, q.l_a, q.l_b, q.h_a, q.h_b); dtm::dtm_pack_biort(p, 5, 7); dtm::dtm_pack_biort_scaled(p, 5, 7, H0O, H1O); p.lo_a_first = q.lo_a_first; p.hi_a_first = q.hi_a_first;
    dtm::k_fwd12w<5, 7, 10, P, KO><<<jobs, 128, 0, st>>>(p);
}
template <int P>
static void run_f12w(int band_rows) {
    const double a = time_it([&](int s) { launch_f12w<P, 0>(s, band_rows); });
    const double b = time_it([&](int s) { launch_f12w<P, 1>(s, band_rows); });
    const double c = time_it([&](int s) { launch_f12w<P, 2>(s, band_rows); });
    const double d = time_it([&](int s) { launch_f12w<P, 3>(s, band_rows); });
    printf("k_fwd12w P=%d band_rows=%3d (2 roles)    %9.2f %9.2f %9.2f %9.2f\n", P, band_rows, a, b, c, d);
    fflush(stdout);
}
// the pyramid as six planes per level instead of 48-byte records (KO bit 64): no slab on either side
static void run_planar(int band_rows) {
    const double a = time_it([&](int s) { launch_f12<2, 64>(s, band_rows); });
    const double b = time_it([&](int s) { launch_f12<2, 65>(s, band_rows); });
    const double c = time_it([&](int s) { launch_f12<2, 66>(s, band_rows); });
    const double d = time_it([&](int s) { launch_f12<2, 67>(s, band_rows); });
    printf("k_fwd12m PLANAR pyramid, band_rows=%3d    %9.2f %9.2f %9.2f %9.2f\n", band_rows, a, b, c, d);
    const double e = time_it([&](int s) { launch_i21<64>(s, band_rows, sets[s].Xo); });
    const double f = time_it([&](int s) { launch_i21<65>(s, band_rows, sets[s].Xo); });
    const double g = time_it([&](int s) { launch_i21<66>(s, band_rows, sets[s].Xo); });
    const double h = time_it([&](int s) { launch_i21<67>(s, band_rows, sets[s].Xo); });
    printf("k_inv21m PLANAR pyramid, band_rows=%3d    %9.2f %9.2f %9.2f %9.2f\n", band_rows, e, f, g, h);
    fflush(stdout);
}
template <int P, int WPS = 2>
static void run_f12(int band_rows) {
    const double a = time_it([&](int s) { launch_f12<P, 0, WPS>(s, band_rows); });
    const double b = time_it([&](int s) { launch_f12<P, 1, WPS>(s, band_rows); });
    const double c = time_it([&](int s) { launch_f12<P, 2, WPS>(s, band_rows); });
    const double d = time_it([&](int s) { launch_f12<P, 3, WPS>(s, band_rows); });
    printf("k_fwd12m P=%d band_rows=%3d waves/SIMD %d %9.2f %9.2f %9.2f %9.2f\n", P, band_rows, WPS, a, b, c, d);
    fflush(stdout);
}

static void launch_ref(int s, float *L, float *Y) {
    Fwd1Params p{}; p.X = sets[s].X; p.LoLo = L; p.Yh = Y; p.B = 1; p.inR = p.inC = p.LR = p.LC = N;
    p.xcd_order = 8; put(p.h0, H0O, 5, DT_MAXT); put(p.h1, H1O, 7, DT_MAXT); dt_pack_c01<5, 7>(p);
    p.tilesR = cdiv(N, F1::TR); p.tilesC = cdiv(N, F1::TC);
    dt_set_tile_magic(p);
    ref_fwd1<F1><<<grid_for(p.tilesR * p.tilesC, 8), DT_NT, 0, st>>>(p);
}

template <int P, int WPB, int KO>
static void launch_march(int s, int seg_rows) {
    using G = dtm::Fwd1m<5, 7>;
    dtm::Fwd1mParams p{}; p.X = sets[s].X; p.LoLo = sets[s].L1; p.Yh = sets[s].Y0; p.B = 1; p.R = p.C = N;
    p.nstrip = cdiv(N, 4 * G::VL); p.seg_rows = seg_rows; p.nseg = cdiv(N, seg_rows);
    put(p.h0, H0O, 5, dtm::MAXT1); put(p.h1, H1O, 7, dtm::MAXT1);
    const int njob = p.nstrip * p.nseg;
    dtm::k_fwd1m<5, 7, P, WPB, KO><<<cdiv(njob, WPB), 64 * WPB, 0, st>>>(p);
}

static double maxdiff(const float *a, const float *b, size_t n, double *mx) {
    std::vector<float> ha(n), hb(n);
    CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
    double d = 0, m = 0;
    int shown = 0; size_t nbad = 0;
    for (size_t i = 0; i < n; ++i) {
        const double e = fabs((double)ha[i] - hb[i]);
        d = fmax(d, e); m = fmax(m, fabs((double)hb[i]));
        if (e > 1e-5) { ++nbad; if (shown < 12) { printf("   mismatch at %zu (row %zu, col %zu of %d): %g vs %g\n", i, i / N, i % N, N, ha[i], hb[i]); ++shown; } }
    }
    if (nbad) printf("   %zu mismatches\n", nbad);
    *mx = m; return d;
}

template <int P, int WPB>
static void run_all(int seg_rows) {
    const double a = time_it([&](int s) { launch_march<P, WPB, 0>(s, seg_rows); });
    const double b = time_it([&](int s) { launch_march<P, WPB, 1>(s, seg_rows); });
    const double c = time_it([&](int s) { launch_march<P, WPB, 2>(s, seg_rows); });
    const double d = time_it([&](int s) { launch_march<P, WPB, 3>(s, seg_rows); });
    const double e = time_it([&](int s) { launch_march<P, WPB, 4>(s, seg_rows); });
    const double f = time_it([&](int s) { launch_march<P, WPB, 5>(s, seg_rows); });
    const double g = time_it([&](int s) { launch_march<P, WPB, 13>(s, seg_rows); });
    printf("k_fwd1m P=%d WPB=%d seg_rows=%3d  %9.2f %9.2f %9.2f %9.2f | no store instr %6.2f, + cached ld %6.2f, + no DPP %6.2f\n", P, WPB, seg_rows, a, b, c, d, e, f, g);
    fflush(stdout);
}

int main(int argc, char **argv) {
    if (argc > 1) N = atoi(argv[1]);
    if (argc > 2) REPS = atoi(argv[2]);
    CK(hipStreamCreate(&st_a)); CK(hipStreamCreate(&st_b)); st = st_a;
    const size_t px = (size_t)N * N;
    std::vector<float> h(px);
    for (size_t i = 0; i < px; ++i) h[i] = (float)((double)rand() / RAND_MAX - 0.5);
    for (auto &s : sets) {
        CK(hipMalloc(&s.X, px * 4)); CK(hipMalloc(&s.L1, px * 4)); CK(hipMalloc(&s.Y0, px * 12));
        CK(hipMemcpy(s.X, h.data(), px * 4, hipMemcpyHostToDevice));
        CK(hipMemset(s.L1, 0, px * 4)); CK(hipMemset(s.Y0, 0, px * 12));
        CK(hipMalloc(&s.L2, px)); CK(hipMalloc(&s.Y1, px * 3)); CK(hipMemset(s.L2, 0, px)); CK(hipMemset(s.Y1, 0, px * 3));
    }
    CK(hipMalloc(&sets[0].L2r, px)); CK(hipMalloc(&sets[0].Y1r, px * 3));
    for (auto &s : sets) { CK(hipMalloc(&s.Xo, px * 4)); CK(hipMemset(s.Xo, 0, px * 4)); }
    CK(hipMalloc(&sets[0].Xor, px * 4));
    CK(hipMalloc(&sets[0].L1r, px * 4)); CK(hipMalloc(&sets[0].Y0r, px * 12));
    // correctness: march vs tile program on set 0
    launch_ref(0, sets[0].L1r, sets[0].Y0r);
    launch_march<2, 4, 0>(0, 32);
    CK(hipStreamSynchronize(st));
    double m1, m2;
    const double d1 = maxdiff(sets[0].L1, sets[0].L1r, px, &m1), d2 = maxdiff(sets[0].Y0, sets[0].Y0r, px * 3, &m2);
    printf("march vs tile program: LoLo max|diff| %.3g (max %.3g), Yh max|diff| %.3g (max %.3g)\n", d1, m1, d2, m2);
    CK(hipMemset(sets[0].L1, 0, px * 4)); CK(hipMemset(sets[0].Y0, 0, px * 12));
    launch_march<4, 1, 0>(0, 48);
    CK(hipStreamSynchronize(st));
    const double d3 = maxdiff(sets[0].L1, sets[0].L1r, px, &m1), d4 = maxdiff(sets[0].Y0, sets[0].Y0r, px * 3, &m2);
    printf("march (P=4, 1 wave per block, 48-row segments): LoLo %.3g, Yh %.3g\n", d3, d4);

    {   // levels 1 + 2 in one march against the two tile programs
        launch_ref2(sets[0].L1r, sets[0].L2r, sets[0].Y1r);
        for (int br : {64, 128, 32, -40, -24}) {
            CK(hipMemset(sets[0].Y0, 0, px * 12)); CK(hipMemset(sets[0].Y1, 0, px * 3)); CK(hipMemset(sets[0].L2, 0, px));
            if (br > 0) launch_f12<2, 0>(0, br); else launch_f12w<2, 0>(0, -br);
            CK(hipStreamSynchronize(st));
            double ma, mb, mc;
            const double e0 = maxdiff(sets[0].Y0, sets[0].Y0r, px * 3, &ma), e1 = maxdiff(sets[0].Y1, sets[0].Y1r, px * 3 / 4, &mb),
                         e2 = maxdiff(sets[0].L2, sets[0].L2r, px / 4, &mc);
            printf("k_fwd12m (bands of %d rows) vs tile programs: Yh0 %.3g (max %.3g), Yh1 %.3g (max %.3g), LoLo2 %.3g (max %.3g)\n", br, e0, ma, e1, mb, e2, mc);
        }
    }
    {   // levels 2 + 1 of the inverse in one march against the two tile programs (inputs: the pyramid the forward just made)
        launch_ref(0, sets[0].L1r, sets[0].Y0); launch_ref2(sets[0].L1r, sets[0].L2, sets[0].Y1);
        launch_ref_inv(0, sets[0].L1, sets[0].Xor);
        for (int br : {40, 64, 24}) {
            CK(hipMemset(sets[0].Xo, 0, px * 4));
            launch_i21<0>(0, br, sets[0].Xo);
            CK(hipStreamSynchronize(st));
            double ma;
            const double e0 = maxdiff(sets[0].Xo, sets[0].Xor, px, &ma);
            printf("k_inv21m (bands of %d rows) vs tile programs: X %.3g (max %.3g)\n", br, e0, ma);
        }
        for (int s2 = 1; s2 < NSET; ++s2) { launch_ref(s2, sets[s2].L1, sets[s2].Y0); launch_ref2(sets[s2].L1, sets[s2].L2, sets[s2].Y1); }
    }
    {   // the planar-pyramid variants: forward against the records (permuted on the host), inverse from the planes
        float *Y0p, *Y1p, *y0 = sets[0].Y0, *y1 = sets[0].Y1;
        CK(hipMalloc(&Y0p, px * 12)); CK(hipMalloc(&Y1p, px * 3)); CK(hipMemset(Y0p, 0, px * 12)); CK(hipMemset(Y1p, 0, px * 3));
        sets[0].Y0 = Y0p; sets[0].Y1 = Y1p;
        launch_f12<2, 64>(0, 40);
        CK(hipMemset(sets[0].Xo, 0, px * 4));
        launch_i21<64>(0, 40, sets[0].Xo);
        CK(hipStreamSynchronize(st));
        sets[0].Y0 = y0; sets[0].Y1 = y1;
        double e[2] = {0, 0};
        for (int lev = 0; lev < 2; ++lev) {
            const int h = N >> (lev + 1), w = N >> (lev + 1);
            std::vector<float> rec((size_t)h * w * 12), pla((size_t)h * w * 12);
            CK(hipMemcpy(rec.data(), lev ? y1 : y0, rec.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(pla.data(), lev ? Y1p : Y0p, pla.size() * 4, hipMemcpyDeviceToHost));
            for (int sb = 0; sb < 6; ++sb)
                for (int i = 0; i < h; ++i)
                    for (int j = 0; j < w; ++j)
                        for (int c = 0; c < 2; ++c)
                            e[lev] = fmax(e[lev], fabs((double)pla[((size_t)sb * h + i) * w * 2 + 2 * j + c] - rec[((size_t)i * w + j) * 12 + 2 * sb + c]));
        }
        double ma;
        const double ex = maxdiff(sets[0].Xo, sets[0].Xor, px, &ma);
        printf("planar pyramid (KO 64): forward planes vs records Yh0 %.3g, Yh1 %.3g; inverse from planes vs tile programs X %.3g (max %.3g)\n", e[0], e[1], ex, ma);
        CK(hipFree(Y0p)); CK(hipFree(Y1p));
    }
    if (argc > 3) {     // profile mode: a few launches of the marching kernels, whole and with loads / stores knocked out
        for (int s2 = 1; s2 < NSET; ++s2) { launch_ref(s2, sets[s2].L1, sets[s2].Y0); launch_ref2(sets[s2].L1, sets[s2].L2, sets[s2].Y1); }
        for (int i = 0; i < 12; ++i) {
            launch_f12<2, 0>(i % NSET, 40); launch_f12<2, 3>(i % NSET, 40);
            launch_i21<0>(i % NSET, 40, sets[i % NSET].Xo); launch_i21<3>(i % NSET, 40, sets[i % NSET].Xo);
            launch_ref(i % NSET, sets[i % NSET].L1, sets[i % NSET].Y0);
        }
        CK(hipDeviceSynchronize());
        return 0;
    }
    for (int i = 0; i < 30; ++i) launch_ref(i % NSET, sets[i % NSET].L1, sets[i % NSET].Y0);      // settle the clocks
    printf("%dx%d, %d reps over %d buffer sets; us per launch\n", N, N, REPS, NSET);
    printf("%-40s %9s %9s %9s %9s\n", "kernel", "full", "cached-ld", "no-store", "arith-only");
    for (int rep = 0; rep < 2; ++rep) {
        const double r = time_it([&](int s) { launch_ref(s, sets[s].L1, sets[s].Y0); });
        printf("%-40s %9.2f\n", "k_fwd1 tile program (library)", r);
        const double r2 = time_it([&](int s) { launch_ref2(sets[s].L1, sets[s].L2, sets[s].Y1); });
        printf("%-40s %9.2f\n", "k_fwd2 tile program (library)", r2);
        run_f12<2>(40);
        printf("%-40s %9.2f\n", "k_inv2 + k_inv1 tile programs", time_it([&](int s) { launch_ref_inv(s, sets[s].L1, sets[s].Xo); }));
        for (int br : {40, 48, 64}) run_i21(br);
        run_planar(40);
        printf("k_fwd12m bands of 40: plain record stores %.2f, nt loads %.2f, both %.2f\n", time_it([&](int s) { launch_f12<2, 16>(s, 40); }), time_it([&](int s) { launch_f12<2, 32>(s, 40); }), time_it([&](int s) { launch_f12<2, 48>(s, 40); }));
        run_f12w<2>(40);
        printf("two streams, us per image: tile fwd1 + fwd2 %.2f", time_two_streams([&](int s) { launch_ref(s, sets[s].L1, sets[s].Y0); launch_ref2(sets[s].L1, sets[s].L2, sets[s].Y1); }));
        for (int br : {40, 80}) printf(" | k_fwd12m bands of %d: %.2f", br, time_two_streams([&](int s) { launch_f12<2, 0>(s, br); }));
        for (int br : {40}) printf(" | k_fwd12w bands of %d: %.2f", br, time_two_streams([&](int s) { launch_f12w<2, 0>(s, br); }));
        printf("\none stream, us per image: tile fwd1 + fwd2 %.2f\n", time_it([&](int s) { launch_ref(s, sets[s].L1, sets[s].Y0); launch_ref2(sets[s].L1, sets[s].L2, sets[s].Y1); }));
    }
    return 0;
}
