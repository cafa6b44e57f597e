// Micro-benchmark of the fused level-1+2 forward tile program (tools/kbench/fused2d_l12.hpp)
// against the one-launch-per-level kernels it replaces, over several tile shapes.  Measurement
// tool only (not part of libdtcwt_hip.so): it compiles the library's own fused2d.hip into this
// translation unit so that other template configurations can be instantiated here.
//
//   make -C tools/kbench fwd12_bench && tools/kbench/fwd12_bench [N=4096] [reps=40]
//
// Every variant is checked against the two-launch result before it is timed; timings rotate
// over NSET buffer sets so that neither input nor outputs of a repetition are cache-resident.
#include "../../dtcwt_amd/csrc/fused2d.hip"
#include "fwd12_kernel.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static const double H0O[5] = {-0.05, 0.25, 0.6, 0.25, -0.05};
static const double H1O[7] = {-0.0107142857142857, 0.0535714285714286, 0.260714285714286, -0.607142857142857,
                              0.260714285714286, 0.0535714285714286, -0.0107142857142857};
static const double H0A[10] = {0.03516384, 0., -0.08832942, 0.23389032, 0.76027237, 0.5875183, 0., -0.11430184, 0., 0.};
static const double H1A[10] = {0., 0., -0.11430184, 0., 0.5875183, -0.76027237, 0.23389032, 0.08832942, 0., -0.03516384};

struct Set { float *X, *L1, *L2, *Y0, *Y1; };

static void fill_params(Fwd1Params &q, Fwd2Params &q2, const Set &s, int B, int N) {
    q = Fwd1Params{}; q2 = Fwd2Params{};
    q.X = s.X; q.LoLo = nullptr; q.Yh = s.Y0; q.B = B; q.inR = q.inC = q.LR = q.LC = N; q.xcd_order = 0;
    std::vector<double> h0(H0O, H0O + 5), h1(H1O, H1O + 7), a(H0A, H0A + 10), ha(H1A, H1A + 10);
    std::vector<double> b(a.rbegin(), a.rend()), hb(ha.rbegin(), ha.rend());
    put_taps(q.h0, h0); put_taps(q.h1, h1);
    q2.X = s.L1; q2.LoLo = s.L2; q2.Yh = s.Y1; q2.B = B; q2.inR = q2.inC = q2.LR = q2.LC = N;
    q2.xcd_order = 0; q2.stream_records = 1;
    put_taps(q2.l_a, b); put_taps(q2.l_b, a); put_taps(q2.h_a, hb); put_taps(q2.h_b, ha);
    q2.lo_a_first = dotd(b, a) > 0; q2.hi_a_first = dotd(hb, ha) > 0;
}

static double maxdiff(const float *d_a, const float *d_b, size_t n) {
    std::vector<float> a(n), b(n);
    CK(hipMemcpy(a.data(), d_a, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), d_b, n * 4, hipMemcpyDeviceToHost));
    double m = 0;
    for (size_t i = 0; i < n; ++i) { double d = std::fabs((double)a[i] - b[i]); if (!(d <= m)) m = d; }
    return m;
}

constexpr int NSET = 4;
static Set sets[NSET], ref;
static int N = 4096, REPS = 40, B = 1;
static hipStream_t st;

static size_t g_extra_lds = 0;
template <class C, int SKIP = 0, int PERSIST = 0, int MW = 3>
static void run_variant(const char *name, int xcd) {
    Fwd1Params q; Fwd2Params q2;
    const size_t px = (size_t)B * N * N;
    // correctness on set 0 against the two-launch reference
    fill_params(q, q2, sets[0], B, N); q2.xcd_order = xcd;
    CK(hipMemsetAsync(sets[0].L2, 0xff, px, st)); CK(hipMemsetAsync(sets[0].Y0, 0xff, px * 12, st)); CK(hipMemsetAsync(sets[0].Y1, 0xff, px * 3, st));
    if (launch_fwd12<C, SKIP>(q, q2, st, g_extra_lds)) { printf("%-28s launch failed\n", name); return; }
    CK(hipStreamSynchronize(st)); CK(hipGetLastError());
    double e0 = maxdiff(sets[0].Y0, ref.Y0, px * 3), e1 = maxdiff(sets[0].Y1, ref.Y1, px * 3 / 4), e2 = maxdiff(sets[0].L2, ref.L2, px / 4);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 8; ++i) { fill_params(q, q2, sets[i % NSET], B, N); q2.xcd_order = xcd; launch_fwd12<C, SKIP>(q, q2, st, g_extra_lds); }
    CK(hipEventRecord(a, st));
    for (int i = 0; i < REPS; ++i) { fill_params(q, q2, sets[i % NSET], B, N); q2.xcd_order = xcd; launch_fwd12<C, SKIP>(q, q2, st, g_extra_lds); }
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / REPS;
    if (SKIP) printf("skip=%2d ", SKIP);
    if (g_extra_lds) printf("extra LDS %zu KB ", g_extra_lds >> 10);
    printf("%-28s xcd=%d lds=%6zu B  %8.2f us  %6.2f TB/s(20 B/px)  err Yh0 %.2e Yh1 %.2e LoLo2 %.2e\n", name, xcd,
           (size_t)C::LDS_FLOATS * 4, us, 20.0 * px / us / 1e6, e0, e1, e2);
    fflush(stdout);
}

int main(int argc, char **argv) {
    if (argc > 1) N = atoi(argv[1]);
    if (argc > 2) REPS = atoi(argv[2]);
    if (argc > 3) B = atoi(argv[3]);
    const bool only = argc > 4;            // profiling runs: the two launches and the library's configuration only
    CK(hipSetDevice(0)); CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t px = (size_t)B * N * N;
    std::vector<float> h(px);
    srand(7);
    auto mk = [&](Set &s, bool init) {
        CK(hipMalloc(&s.X, px * 4)); CK(hipMalloc(&s.L1, px * 4)); CK(hipMalloc(&s.L2, px)); CK(hipMalloc(&s.Y0, px * 12)); CK(hipMalloc(&s.Y1, px * 3));
        if (init) CK(hipMemcpy(s.X, h.data(), px * 4, hipMemcpyHostToDevice));
    };
    for (size_t i = 0; i < px; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    for (int i = 0; i < NSET; ++i) mk(sets[i], true);
    mk(ref, true);
    // ---- reference: the two launches
    Fwd1Params q; Fwd2Params q2;
    hipEvent_t e[3]; for (auto &x : e) CK(hipEventCreate(&x));
    float t1 = 0, t2 = 0;
    for (int i = -8; i < REPS; ++i) {
        const Set &s = i < 0 ? ref : sets[i % NSET];
        fill_params(q, q2, s, B, N); q.LoLo = s.L1; q2.xcd_order = 1;
        CK(hipEventRecord(e[0], st));
        dispatch_fwd1(5, 7, 0, q, st);
        CK(hipEventRecord(e[1], st));
        dispatch_fwd2(10, false, q2, st, false);
        CK(hipEventRecord(e[2], st)); CK(hipEventSynchronize(e[2]));
        float a, b; CK(hipEventElapsedTime(&a, e[0], e[1])); CK(hipEventElapsedTime(&b, e[1], e[2]));
        if (i >= 0) { t1 += a; t2 += b; }
    }
    fill_params(q, q2, ref, B, N); q.LoLo = ref.L1; q2.xcd_order = 1;
    dispatch_fwd1(5, 7, 0, q, st); dispatch_fwd2(10, false, q2, st, false); CK(hipStreamSynchronize(st));
    printf("N=%d B=%d reps=%d   two launches: k_fwd1 %.2f us + k_fwd2 %.2f us = %.2f us (event pairs, rotating %d buffer sets)\n",
           N, B, REPS, t1 * 1e3 / REPS, t2 * 1e3 / REPS, (t1 + t2) * 1e3 / REPS, NSET);
    // back-to-back (no event in between), as the plan issues them
    {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a, st));
        for (int i = 0; i < REPS; ++i) {
            fill_params(q, q2, sets[i % NSET], B, N); q.LoLo = sets[i % NSET].L1; q2.xcd_order = 1;
            dispatch_fwd1(5, 7, 0, q, st); dispatch_fwd2(10, false, q2, st, false);
        }
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("two launches back to back: %.2f us per image\n", ms * 1e3 / REPS);
    }
#define V(...) run_variant<Fwd12Cfg<__VA_ARGS__>>(#__VA_ARGS__, 0); run_variant<Fwd12Cfg<__VA_ARGS__>>(#__VA_ARGS__, 1);
#define K(S, ...) run_variant<Fwd12Cfg<__VA_ARGS__>, S>(#__VA_ARGS__, 1);
    V(16, 32, 8, 4, 5, 7, 10, 256)
    if (only) return 0;
    // occupancy: 39 KB -> 4 workgroups per CU; + 14 KB -> 3; + 40 KB -> 2; + 100 KB -> 1
    for (size_t extra : {(size_t)14 << 10, (size_t)40 << 10, (size_t)100 << 10}) {
        g_extra_lds = extra;
        run_variant<Fwd12Cfg<16, 32, 8, 4, 5, 7, 10, 256>>("16,32 occupancy", 1);
        run_variant<Fwd12Cfg<16, 32, 8, 4, 5, 7, 10, 256>, 96>("16,32 occupancy", 1);
    }
    g_extra_lds = 0;
    V(32, 32, 8, 4, 5, 7, 10, 512)
    V(16, 64, 8, 4, 5, 7, 10, 512)
    V(8, 32, 8, 4, 5, 7, 10, 256)
    // phase knock-outs (results wrong by construction; time only): 1 = level-1 column pass, 2 = core row pass,
    // 4 = halo LoLo1, 8 = level-2 column pass, 16 = level-2 row pass, 32 = Yh[0] record flush
    K(64, 16, 32, 8, 4, 5, 7, 10, 256)      // 64: all workgroups on 16 cache-resident tiles = the kernel without HBM
    K(96, 16, 32, 8, 4, 5, 7, 10, 256)
    K(1, 16, 32, 8, 4, 5, 7, 10, 256)
    K(2, 16, 32, 8, 4, 5, 7, 10, 256)
    K(4, 16, 32, 8, 4, 5, 7, 10, 256)
    K(8, 16, 32, 8, 4, 5, 7, 10, 256)
    K(16, 16, 32, 8, 4, 5, 7, 10, 256)
    K(24, 16, 32, 8, 4, 5, 7, 10, 256)
    K(32, 16, 32, 8, 4, 5, 7, 10, 256)
    K(30, 16, 32, 8, 4, 5, 7, 10, 256)
    K(31, 16, 32, 8, 4, 5, 7, 10, 256)
    K(29, 16, 32, 8, 4, 5, 7, 10, 256)
    return 0;
}
