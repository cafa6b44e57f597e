// Knock-out timing of the four per-level 2-D kernels (measurement tool, not product).
//
// Each kernel of libdtcwt_hip.so's float32 2-D plan is rebuilt here from the library's own tile
// programs with two switches:
//   CACHED  every workgroup LOADS the window of one of 16 tiles (so all reads hit the XCD's L2),
//           while its stores still go to its own tile: "what does the kernel cost when its reads
//           are free";
//   NOSTORE the global stores of every workgroup go to one of the same 16 tiles (they stay in L2 and
//           never reach HBM as a stream).
// CACHED + NOSTORE = arithmetic + LDS + barriers only.  The results of the CACHED variants are
// garbage on purpose; the plain variant is checked against nothing here (the test suite does that).
//
//   make -C tools/kbench ko_bench && tools/kbench/ko_bench [N=4096] [reps=40]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fused2d_tiles.hpp"
#include "fused2d_tiles_v2.hpp"

using namespace dt2d;

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline unsigned grid_for(int ntile, int order = 1) {
    const int q = 8 * (order > 1 ? order : 1);
    return (unsigned)(cdiv(ntile, q) * q);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// KO bit 0: CACHED loads, bit 1: NOSTORE
#define KO_TILE()                                                                              \
    const int ntile = p.tilesR * p.tilesC * p.B;                                               \
    int t = tile_of(blockIdx.x, ntile, p.xcd_order);                                           \
    if (t >= ntile) return;                                                                    \
    int tc, tr, b;                                                                             \
    dt_tile_decode(p, t, tc, tr, b);                                                           \
    int trl = tr, tcl = tc;                                                                    \
    if (KO & 1) { trl = 2 + (tr & 3); tcl = 2 + (tc & 3); }                                    \
    if (KO & 2) { tr = 2 + (tr & 3); tc = 2 + (tc & 3); }

template <class C, int KO>
__global__ void __launch_bounds__(DT_NT) ko_fwd1(Fwd1Params p) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_FLOATS + 4 * STAGE_FLOATS_PER_WAVE];
    KO_TILE()
    float *sLo = smem, *sHi = sLo + C::SL, *stage = smem + C::LDS_FLOATS;
    fwd1d_cols<C>(p, sLo, sHi, threadIdx.x, b, trl * C::TR, tcl * C::TC, nullptr);
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

template <class C, int KO>
__global__ void __launch_bounds__(DT_NT) ko_fwd2(Fwd2Params p) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_FLOATS + 4 * STAGE_FLOATS_PER_WAVE];
    KO_TILE()
    float *sLo = smem, *sHi = sLo + C::SL, *stage = smem + C::LDS_FLOATS;
    fwd2d_cols<C>(p, sLo, sHi, threadIdx.x, b, trl * C::TR, tcl * C::TC, nullptr);
    __syncthreads();
    const int r0 = tr * C::TR, c0 = tc * C::TC;
    for (int base = 0; base < C::TI * C::TJ; base += DT_NT) {
        fwd2s_rows_compute<C>(p, sLo, sHi, stage, threadIdx.x, base, b, r0, c0, nullptr);
        DT_WAVE_LDS_SYNC();
        fwd2s_rows_flush<C>(p, stage, threadIdx.x, base, b, r0, c0);
        DT_WAVE_LDS_SYNC();
    }
}

template <class C, int KO>
__global__ void __launch_bounds__(DT_NT) ko_inv1(Inv1Params p) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_ALIASED];
    KO_TILE()
    float *srec = smem, *y1 = smem, *y2 = y1 + C::SY;
    const int r0 = tr * C::TR, c0 = tc * C::TC, r0l = trl * C::TR, c0l = tcl * C::TC;
    const float *Yhb = p.Yh + (int64_t)b * (p.R / 2) * (p.C / 2) * 12;
    float wz[C::WN], w1[C::WN], w2[C::WN], w3[C::WN];
    inv1r_fetch<C>(p, wz, threadIdx.x, b, r0l, c0l);
    inv_rec_stage<C::QR, C::QC>(Yhb, p.R, p.C, srec, r0l - C::HE, c0l - C::HE, threadIdx.x);
    __syncthreads();
    inv1r_gather<C>(p, srec, w1, w2, w3, threadIdx.x, r0l, c0l);
    __syncthreads();
    inv1r_fir<C>(p, wz, w1, w2, w3, y1, y2, threadIdx.x, nullptr);
    __syncthreads();
    inv1d_rows<C>(p, y1, y2, threadIdx.x, b, r0, c0, nullptr);
}

template <class C, int KO>
__global__ void __launch_bounds__(DT_NT) ko_inv2(Inv2Params p) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_ALIASED];
    KO_TILE()
    float *srec = smem, *y1 = smem, *y2 = y1 + C::SY;
    const int r0 = tr * C::TR, c0 = tc * C::TC, r0l = trl * C::TR, c0l = tcl * C::TC;
    const float *Yhb = p.Yh + (int64_t)b * (p.zr / 2) * (p.zc / 2) * 12;
    float wz[C::WS], w1[C::WS], w2[C::WS], w3[C::WS];
    inv2r_fetch<C>(p, wz, threadIdx.x, b, r0l, c0l);
    inv_rec_stage<C::QR, C::QC>(Yhb, p.zr, p.zc, srec, r0l + C::ORG, c0l + C::ORG, threadIdx.x);
    __syncthreads();
    inv2r_gather<C>(p, srec, w1, w2, w3, threadIdx.x, r0l, c0l);
    __syncthreads();
    inv2r_fir<C, false, true>(p, wz, w1, w2, w3, y1, y2, threadIdx.x, nullptr);
    __syncthreads();
    inv2_rows<C, true>(p, y1, y2, threadIdx.x, b, r0, c0, nullptr);
}

// ---- phase stamps: where a workgroup's life goes -------------------------------------------------
// s_memtime (shader clock) at every phase boundary, taken by lane 0 of wavefront 0 of each workgroup; the
// differences go to ts[workgroup][phase] (plain stores; summed on the host).
__device__ inline unsigned long long dt_now() { return __builtin_amdgcn_s_memtime(); }
#define STAMP(k_)                                                                                          \
    do {                                                                                                   \
        if (threadIdx.x == 0) { const unsigned long long n_ = dt_now(); ts[(size_t)blockIdx.x * 8 + k_] = n_ - t_; t_ = n_; } \
    } while (0)

template <class C>
__global__ void __launch_bounds__(DT_NT) st_inv2(Inv2Params p, unsigned long long *ts) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_ALIASED];
    const int KO = 0;
    KO_TILE()
    (void)trl; (void)tcl;
    unsigned long long t_ = dt_now();
    float *srec = smem, *y1 = smem, *y2 = y1 + C::SY;
    const int r0 = tr * C::TR, c0 = tc * C::TC;
    const float *Yhb = p.Yh + (int64_t)b * (p.zr / 2) * (p.zc / 2) * 12;
    float wz[C::WS], w1[C::WS], w2[C::WS], w3[C::WS];
    inv2r_fetch<C>(p, wz, threadIdx.x, b, r0, c0);
    inv_rec_stage<C::QR, C::QC>(Yhb, p.zr, p.zc, srec, r0 + C::ORG, c0 + C::ORG, threadIdx.x);
    STAMP(0);                       // loads issued + waited for + records written to LDS (wave 0)
    __syncthreads();
    STAMP(1);                       // barrier 1
    inv2r_gather<C>(p, srec, w1, w2, w3, threadIdx.x, r0, c0);
    STAMP(2);                       // gather
    __syncthreads();
    STAMP(3);                       // barrier 2
    inv2r_fir<C, false, true>(p, wz, w1, w2, w3, y1, y2, threadIdx.x, nullptr);
    STAMP(4);                       // column FIR + y writes (includes waiting for the lowpass window loads)
    __syncthreads();
    STAMP(5);                       // barrier 3
    inv2_rows<C, true>(p, y1, y2, threadIdx.x, b, r0, c0, nullptr);
    STAMP(6);                       // row pass + stores issued
}

template <class C>
__global__ void __launch_bounds__(DT_NT) st_fwd2(Fwd2Params p, unsigned long long *ts) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_FLOATS + 4 * STAGE_FLOATS_PER_WAVE];
    const int KO = 0;
    KO_TILE()
    (void)trl; (void)tcl;
    unsigned long long t_ = dt_now();
    float *sLo = smem, *sHi = sLo + C::SL, *stage = smem + C::LDS_FLOATS;
    const int r0 = tr * C::TR, c0 = tc * C::TC;
    fwd2d_cols<C>(p, sLo, sHi, threadIdx.x, b, r0, c0, nullptr);
    STAMP(0);                       // column pass: loads, FIR, plane writes
    __syncthreads();
    STAMP(1);                       // barrier
    for (int base = 0; base < C::TI * C::TJ; base += DT_NT) {
        fwd2s_rows_compute<C>(p, sLo, sHi, stage, threadIdx.x, base, b, r0, c0, nullptr);
        DT_WAVE_LDS_SYNC();
        STAMP(2);                   // row pass + q2c + slab
        fwd2s_rows_flush<C>(p, stage, threadIdx.x, base, b, r0, c0);
        DT_WAVE_LDS_SYNC();
        STAMP(3);                   // flush
    }
}

static const double H0O[5] = {-0.05, 0.25, 0.6, 0.25, -0.05};
static const double H1O[7] = {-0.0107142857142857, 0.0535714285714286, 0.260714285714286, -0.607142857142857,
                              0.260714285714286, 0.0535714285714286, -0.0107142857142857};
static const double G0O[7] = {-0.0107142857142857, -0.0535714285714286, 0.260714285714286, 0.607142857142857,
                              0.260714285714286, -0.0535714285714286, -0.0107142857142857};
static const double G1O[5] = {-0.05, -0.25, 0.6, -0.25, -0.05};
static const double H0A[10] = {0.03516384, 0., -0.08832942, 0.23389032, 0.76027237, 0.5875183, 0., -0.11430184, 0., 0.};
static const double H1A[10] = {0., 0., -0.11430184, 0., 0.5875183, -0.76027237, 0.23389032, 0.08832942, 0., -0.03516384};

static void put(float *dst, const double *src, int n, bool rev = false) {
    for (int k = 0; k < DT_MAXT; ++k) dst[k] = k < n ? (float)src[rev ? n - 1 - k : k] : 0.f;
}

constexpr int NSET = 4;
struct Set { float *X, *L1, *L2, *Y0, *Y1, *Z1, *Z0; };
static Set sets[NSET];
static int N = 4096, REPS = 40;
static hipStream_t st;
static int g_magic = 1;        // KO_MAGIC=0: divide in the tile decode (the round-2 prologue)
static size_t g_xlds = 0;      // extra dynamic LDS per workgroup: lowers the occupancy without touching the code

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

using F1 = Fwd1DCfg<32, 64, 8, 5, 7>;
using F2 = Fwd2DCfg<16, 56, 4, 10>;
using I1 = Inv1RCfg<16, 120, 8, 7, 5>;
using I2 = Inv2RCfg<16, 56, 2, 10>;

template <int KO> static double run_fwd1() {
    return time_it([&](int s) {
        Fwd1Params p{}; p.X = sets[s].X; p.LoLo = sets[s].L1; p.Yh = sets[s].Y0; p.B = 1; p.inR = p.inC = p.LR = p.LC = N;
        p.xcd_order = 8; put(p.h0, H0O, 5); put(p.h1, H1O, 7); dt_pack_c01<5, 7>(p);
        p.tilesR = cdiv(N, F1::TR); p.tilesC = cdiv(N, F1::TC);
        if (g_magic) dt_set_tile_magic(p);
        ko_fwd1<F1, KO><<<grid_for(p.tilesR * p.tilesC, 8), DT_NT, g_xlds, st>>>(p);
    });
}
template <int KO> static double run_fwd2() {
    return time_it([&](int s) {
        Fwd2Params p{}; p.X = sets[s].L1; p.LoLo = sets[s].L2; p.Yh = sets[s].Y1; p.B = 1; p.inR = p.inC = p.LR = p.LC = N;
        p.xcd_order = 1; p.stream_records = 1;
        put(p.l_a, H0A, 10, true); put(p.l_b, H0A, 10); put(p.h_a, H1A, 10, true); put(p.h_b, H1A, 10);
        p.lo_a_first = 1; p.hi_a_first = 0; dt_pack_lh(p);
        p.tilesR = cdiv(N / 2, F2::TR); p.tilesC = cdiv(N / 2, F2::TC);
        if (g_magic) dt_set_tile_magic(p);
        ko_fwd2<F2, KO><<<grid_for(p.tilesR * p.tilesC), DT_NT, g_xlds, st>>>(p);
    });
}
template <int KO> static double run_inv1() {
    return time_it([&](int s) {
        Inv1Params p{}; p.Z = sets[s].Z1; p.Yh = sets[s].Y0; p.X = sets[s].Z0; p.B = 1; p.R = p.C = N; p.xcd_order = 1;
        for (int d = 0; d < 6; ++d) p.g[d] = 0.70710678f;
        put(p.g0, G0O, 7); put(p.g1, G1O, 5); dt_pack_g01<7, 5>(p);
        p.tilesR = cdiv(N, I1::TR); p.tilesC = cdiv(N, I1::TC);
        if (g_magic) dt_set_tile_magic(p);
        ko_inv1<I1, KO><<<grid_for(p.tilesR * p.tilesC), DT_NT, g_xlds, st>>>(p);
    });
}
template <int KO> static double run_inv2() {
    return time_it([&](int s) {
        Inv2Params p{}; p.Z = sets[s].L2; p.Yh = sets[s].Y1; p.Out = sets[s].Z1; p.B = 1; p.zr = p.zc = N / 2; p.xcd_order = 1;
        for (int d = 0; d < 6; ++d) p.g[d] = 0.70710678f;
        // g0a = reverse(h0b) ... any 10-tap values do for timing; the phases must be the standard ones
        put(p.l_a, H0A, 10); put(p.l_b, H0A, 10, true); put(p.h_a, H1A, 10); put(p.h_b, H1A, 10, true);
        p.lo_pos = 1; p.hi_pos = 0;
        p.tilesR = cdiv(N / 2, I2::TR); p.tilesC = cdiv(N / 2, I2::TC);
        if (g_magic) dt_set_tile_magic(p);
        ko_inv2<I2, KO><<<grid_for(p.tilesR * p.tilesC), DT_NT, g_xlds, st>>>(p);
    });
}

int main(int argc, char **argv) {
    if (argc > 1) N = atoi(argv[1]);
    if (argc > 2) REPS = atoi(argv[2]);
    if (const char *e = getenv("KO_MAGIC")) g_magic = atoi(e);
    const bool sweep = getenv("KO_SWEEP") != nullptr;
    CK(hipStreamCreate(&st));
    const size_t px = (size_t)N * N;
    std::vector<float> h(px);
    for (size_t i = 0; i < px; ++i) h[i] = (float)((double)rand() / RAND_MAX - 0.5);
    for (auto &s : sets) {
        CK(hipMalloc(&s.X, px * 4)); CK(hipMalloc(&s.L1, px * 4)); CK(hipMalloc(&s.L2, px)); CK(hipMalloc(&s.Y0, px * 12));
        CK(hipMalloc(&s.Y1, px * 3)); CK(hipMalloc(&s.Z1, px * 4)); CK(hipMalloc(&s.Z0, px * 4));
        CK(hipMemcpy(s.X, h.data(), px * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(s.L1, h.data(), px * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(s.Z1, h.data(), px * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(s.L2, h.data(), px, hipMemcpyHostToDevice));
        for (int k = 0; k < 3; ++k) CK(hipMemcpy(s.Y0 + k * px, h.data(), px * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(s.Y1, h.data(), px * 3, hipMemcpyHostToDevice));
    }
    // settle the clocks
    for (int i = 0; i < 3; ++i) run_fwd1<0>();
    printf("%dx%d, %d reps over %d buffer sets; us per launch; tile decode by %s\n", N, N, REPS, NSET, g_magic ? "magic multiply" : "division");
    printf("%-22s %9s %9s %9s %9s\n", "kernel", "full", "cached-ld", "no-store", "arith-only");
    for (int rep = 0; rep < 2; ++rep) {
        printf("%-22s %9.2f %9.2f %9.2f %9.2f\n", "k_fwd1 (level 1)", run_fwd1<0>(), run_fwd1<1>(), run_fwd1<2>(), run_fwd1<3>());
        printf("%-22s %9.2f %9.2f %9.2f %9.2f\n", "k_fwd2 (level 2)", run_fwd2<0>(), run_fwd2<1>(), run_fwd2<2>(), run_fwd2<3>());
        printf("%-22s %9.2f %9.2f %9.2f %9.2f\n", "k_inv2 (level 2)", run_inv2<0>(), run_inv2<1>(), run_inv2<2>(), run_inv2<3>());
        printf("%-22s %9.2f %9.2f %9.2f %9.2f\n", "k_inv1 (level 1)", run_inv1<0>(), run_inv1<1>(), run_inv1<2>(), run_inv1<3>());
        fflush(stdout);
    }
    if (getenv("KO_STAMPS")) {
        const size_t maxwg = 1 << 16;
        unsigned long long *ts; CK(hipMalloc(&ts, maxwg * 8 * 8));
        std::vector<unsigned long long> hts(maxwg * 8);
        auto report = [&](const char *name, int nph, const char *const *lab, double us, size_t nwg) {
            CK(hipMemcpy(hts.data(), ts, nwg * 8 * 8, hipMemcpyDeviceToHost));
            double h[8] = {0}; size_t n = 0;
            for (size_t w = 0; w < nwg; ++w) { if (!hts[w * 8]) continue; ++n; for (int k = 0; k < nph; ++k) h[k] += (double)hts[w * 8 + k]; }
            double tot = 0; for (int k = 0; k < nph; ++k) tot += h[k];
            printf("%s: %.2f us per launch, %zu workgroups stamped, mean life of wavefront 0: %.0f cycles (s_memtime)\n", name, us, n, tot / n);
            for (int k = 0; k < nph; ++k) printf("    %-44s %8.0f cycles  %5.1f %%\n", lab[k], h[k] / n, 100.0 * h[k] / tot);
        };
        {
            CK(hipMemset(ts, 0, maxwg * 8 * 8));
            double us = time_it([&](int s) {
                Inv2Params p{}; p.Z = sets[s].L2; p.Yh = sets[s].Y1; p.Out = sets[s].Z1; p.B = 1; p.zr = p.zc = N / 2; p.xcd_order = 1;
                for (int d = 0; d < 6; ++d) p.g[d] = 0.70710678f;
                put(p.l_a, H0A, 10); put(p.l_b, H0A, 10, true); put(p.h_a, H1A, 10); put(p.h_b, H1A, 10, true);
                p.lo_pos = 1; p.hi_pos = 0;
                p.tilesR = cdiv(N / 2, I2::TR); p.tilesC = cdiv(N / 2, I2::TC); dt_set_tile_magic(p);
                st_inv2<I2><<<grid_for(p.tilesR * p.tilesC), DT_NT, 0, st>>>(p, ts);
            });
            const char *lab[] = {"fetch + stage records (incl. load latency)", "barrier 1", "gather (c2q from LDS records)", "barrier 2",
                                 "column FIR + y writes", "barrier 3", "row pass + stores issued"};
            report("k_inv2 (level 2)", 7, lab, us, grid_for(cdiv(N / 2, I2::TR) * cdiv(N / 2, I2::TC)));
        }
        {
            CK(hipMemset(ts, 0, maxwg * 8 * 8));
            double us = time_it([&](int s) {
                Fwd2Params p{}; p.X = sets[s].L1; p.LoLo = sets[s].L2; p.Yh = sets[s].Y1; p.B = 1; p.inR = p.inC = p.LR = p.LC = N;
                p.xcd_order = 1; p.stream_records = 1;
                put(p.l_a, H0A, 10, true); put(p.l_b, H0A, 10); put(p.h_a, H1A, 10, true); put(p.h_b, H1A, 10);
                p.lo_a_first = 1; p.hi_a_first = 0; dt_pack_lh(p);
                p.tilesR = cdiv(N / 2, F2::TR); p.tilesC = cdiv(N / 2, F2::TC); dt_set_tile_magic(p);
                st_fwd2<F2><<<grid_for(p.tilesR * p.tilesC), DT_NT, 0, st>>>(p, ts);
            });
            const char *lab[] = {"column pass (loads, FIR, plane writes)", "barrier", "row pass + q2c + slab", "flush (record stores issued)"};
            report("k_fwd2 (level 2)", 4, lab, us, grid_for(cdiv(N / 2, F2::TR) * cdiv(N / 2, F2::TC)));
        }
        return 0;
    }
    if (!sweep) return 0;
    printf("\noccupancy sweep (extra dynamic LDS per workgroup), full / arith-only\n");
    const size_t xs[] = {0, 8 << 10, 16 << 10, 24 << 10, 40 << 10};
    for (size_t x : xs) {
        g_xlds = x;
        printf("xlds %2zu KB  fwd1 %6.2f %6.2f   fwd2 %6.2f %6.2f   inv2 %6.2f %6.2f   inv1 %6.2f %6.2f\n", x >> 10,
               run_fwd1<0>(), run_fwd1<3>(), run_fwd2<0>(), run_fwd2<3>(), run_inv2<0>(), run_inv2<3>(), run_inv1<0>(), run_inv1<3>());
        fflush(stdout);
    }
    return 0;
}
