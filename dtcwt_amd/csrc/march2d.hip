// Launchers of the marching wavefront programs (march2d.hpp) behind the 2-D plan (fused2d.hip).
//
// Levels 1 + 2 of Transform2d.forward (dtcwt/numpy/transform2d.py:112-160) as ONE launch whose level-1 lowpass never
// leaves the registers; used by dtcwt_hip_plan2d_forward when the geometry allows it, otherwise the plan keeps its
// one tile-program launch per level.  Built without SLP vectorisation (Makefile): packed FMAs have no throughput
// advantage on gfx950 and cost a v_mov per operand.
#include <cmath>
#include <cstdlib>

#include "common.hpp"
#include "march2d.hpp"
#include "march2d_l1.hpp"
#include "march2d_pair.hpp"
#include "march2d_ipair.hpp"
#include "fused3d_long.hpp"

namespace {

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

bool symmetric(const std::vector<double> &h) {
    double mx = 0;
    for (double v : h) mx = std::fmax(mx, std::fabs(v));
    for (size_t k = 0; k < h.size() / 2; ++k)
        if (std::fabs(h[k] - h[h.size() - 1 - k]) > 1e-12 * mx) return false;
    return true;
}

// (`B` counts the images in flight on the device: the batch of the launch x the caller's hint of how many independent
// transforms run beside it -- their jobs fill the same slots.)
// rows per band: one job = one wavefront = (strip, band, image) marches band_rows / 2 + M - 2 steps, of which M - 2 are
// the (cheaper) warm-up steps above and below the band that its level-2 windows reach into.  The kernel needs ~190
// registers, i.e. two wavefronts per SIMD: `slots` wavefronts are resident at a time, and a grid of slots + 1 jobs
// takes as long as one of 2 x slots -- so the band height is the one that minimises rounds x steps.
int pick_band_rows(int B, int R, int nstrip, int M, int cus, int forced) {
    if ((forced & ~3) >= 8) return forced & ~3;          // DTCWT_HIP_MARCH_BAND as the plan read it
    const int64_t slots = (int64_t)cus * 8;
    int best = 64; double best_cost = 1e30;
    for (int br = 24; br <= 512; br += 4) {
        const int64_t jobs = (int64_t)B * nstrip * cdiv(R, br);
        const int64_t rounds = (jobs + slots - 1) / slots;
        const double cost = (double)rounds * (br / 2 + 0.45 * (M - 2));
        if (cost < best_cost * 0.999) { best_cost = cost; best = br; }
    }
    return best;
}
// the level-1 marches (march2d_l1.hpp): bands in multiples of `quant` rows (the forward runs whole periods of its register
// ring: 2 x (HH + 1) rows), `warm` extra steps per band (the inverse reads (HM + 1) / 2 record rows above and below)
int pick_band_rows_l1(int B, int R, int nstrip, int quant, double warm, int cus, int forced) {
    if (forced / quant * quant >= quant) return forced / quant * quant;

    const int64_t slots = (int64_t)cus * 8;
    int best = quant; double best_cost = 1e30;
    for (int br = quant; br <= 640; br += quant) {
        if (br < 16) continue;
        const int64_t jobs = (int64_t)B * nstrip * cdiv(R, br);
        const int64_t rounds = (jobs + slots - 1) / slots;
        const double cost = (double)rounds * (br / 2 + warm);
        if (cost < best_cost * 0.999) { best_cost = cost; best = br; }
    }
    return best;
}

}  // namespace

// The three environment switches of the marching programs (common.hpp), read when a plan is created: a plan keeps its
// programs for life and reports them (dtcwt_hip_plan2d_describe).  Every other decision below is a fixed rule with the
// measurement that set it beside it; the switches that A/B experiments used until round 5 (DTCWT_HIP_MARCH_INV, _MARCH_PAIR,
// _MARCH_IPAIR, _MARCH_FWD2, _MARCH_INV2, _INV21_PAIR -> bits of DTCWT_HIP_MARCH_PARTS; _INV21_PF, _FPAIR_WPS, _IPAIR_WPS ->
// settled, gone with the kernel builds they selected) are in INTEGRATION.md.
DtMarchHint dt_march_switches() {
    DtMarchHint h{0, 1, 1, -1};
    if (const char *e = getenv("DTCWT_HIP_MARCH")) h.env_march = e[0] == '0' ? 0 : 1;
    if (const char *e = getenv("DTCWT_HIP_MARCH_BAND")) h.band = atoi(e) > 0 ? atoi(e) : 0;
    if (const char *e = getenv("DTCWT_HIP_MARCH_PARTS")) h.parts = (unsigned)strtoul(e, nullptr, 0);
    return h;
}
// 0 = never, 1 = wherever the geometry and the filters allow, -1 = where it also pays (below).  A program pinned on the
// plan (dtcwt_hip_plan2d_set_program) wins over the environment.
static int march_mode(const DtMarchHint &h) {
    if (h.program >= 0) return h.program ? 1 : 0;
    return h.env_march;
}

// Does the one-launch form pay?  A marching job is one wavefront running 20-80 dependent steps: a launch takes ~30 us
// however small the image, where the tile programs -- hundreds of short-lived workgroups -- take 5-10 us per level.  The
// crossover is a number of USEFUL pixels per call (pixels x the share of a strip's lanes that own columns), measured per
// situation; each row names the sweep that set it:
struct MarchCrossover { const char *situation; double useful_pixels; const char *measured_in; };
static const MarchCrossover kCrossover[3] = {
    // one transform at a time on the whole device, forward + inverse, march / tiles: 512^2 2.07, 1024^2 1.42, 1536^2 1.13,
    // 1792^2 1.06 | 2048^2 0.95, 4 x 1024^2 0.93, 32 x 512^2 0.93, 4096^2 0.89; 16 x 512^2 (a third of the lanes idle) 1.17
    // re-measured per direction with the round-5 kernels (the inverse up to 4096^2 is the marching pair): 1536^2 1.07 / 1.12,
    // 1792^2 0.97 / 0.97, 2048^2 0.87 / 0.92 (profiles/r05/march_sizes_dir.txt): the crossover is 3.1 M useful pixels now (16 x 512^2, a third of its lanes idle, 3.09 M: stays on the tiles, 1.17)
    {"alone on the whole device", 3.1e6, "profiles/r04/ab_march_sizes.txt, profiles/r05/march_sizes_dir.txt"},
    // four in flight on plain streams (hipGraph replay, us per image, march / tiles): 1024^2 25.7 / 19.6, 1536^2 32.7 / 33.1,
    // 1792^2 38.9 / 42.6 -- the launch latency is partly hidden, the crossover comes down, though not in proportion
    {"others in flight beside it (concurrency hint > 1)", 2.2e6, "profiles/r04/hint_sizes.txt"},
    // four contexts on quarters of the compute units: 1024^2 26.2 / 21.9, 1080 x 1920 30.8 / 35.6, 1536^2 31.8 / 39.0
    {"on a partition context (a share of the compute units)", 1.4e6, "profiles/r04/hint_sizes_part.txt"},
};
static bool march_pays(int batch, int rows, int cols, const DtMarchHint &h) {
    const int nstrip = cdiv(cols, 4 * dtm::Fwd12m<5, 7, 10>::VL);
    const double useful = (double)batch * rows * cols * ((double)cols / (nstrip * 4.0 * dtm::Fwd12m<5, 7, 10>::VL));
    const MarchCrossover &x = kCrossover[h.nparts > 1 ? 2 : (h.in_flight > 1 ? 1 : 0)];
    return useful >= x.useful_pixels;
}
// sizes every marching launch handles: multiples of 4 (no odd-size extension, no level-2 padding), 32-bit row offsets
// inside an image, and a job count the grid can carry (a huge batch of small images falls back to the tile programs)
static bool march_sizes_ok(int batch, int rows, int cols, int VL) {
    if (rows % 4 || cols % 4 || rows < 32 || cols < 32) return false;
    if ((int64_t)rows * cols * 4 >= ((int64_t)1 << 31)) return false;
    if ((int64_t)cdiv(cols, 4 * VL) * cdiv(rows, 8) * batch >= ((int64_t)1 << 30)) return false;
    return true;
}

// the geometry / filters the one-launch levels 1 + 2 handle; everything else stays with the tile programs
bool dtcwt_march_fwd12_ok(int batch, int rows, int cols, const std::vector<double> &h0o, const std::vector<double> &h1o,
                          const std::vector<double> &h0a, bool lo_a_first, bool hi_a_first, const DtMarchHint &hint) {
    const int mm = march_mode(hint);
    if ((!(hint.parts & DT_PART_FWD12) && hint.program < 0) || mm == 0 || (mm < 0 && !march_pays(batch, rows, cols, hint))) return false;
    const int m0 = (int)h0o.size(), m1 = (int)h1o.size(), m = (int)h0a.size();
    if (!((m0 == 5 && m1 == 7) || (m0 == 5 && m1 == 3)) || m != 10) return false;       // near_sym_a, legall + qshift_a / _06
    if (!symmetric(h0o) || !symmetric(h1o)) return false;     // mirrored halo lanes: see march2d.hpp
    if (!lo_a_first || hi_a_first) return false;               // the phases the kernel is compiled for: every shipped set
    return march_sizes_ok(batch, rows, cols, dtm::Fwd12m<5, 7, 10>::VL);
}

// levels 2 + 1 of the inverse in one launch: the same geometry rules as the forward, the synthesis filters of near_sym_a
// (7, 5 taps), 10-tap q-shift filters with the standard phases
bool dtcwt_march_inv21_ok(int batch, int rows, int cols, const std::vector<double> &g0o, const std::vector<double> &g1o,
                          const std::vector<double> &g0a, bool lo_pos, bool hi_pos, const DtMarchHint &hint) {
    // (DTCWT_HIP_MARCH_PARTS without bit 2: never.)  A band re-reads the rows its windows reach into above and below -- for the inverse
    // those are level-1 RECORD rows, 12 of its 16 bytes per pixel -- so with the 40-row bands a single 4096^2 image has
    // to be cut into, the one launch alone is no faster than the two it replaces (97 against 94 us, 1.2 x the
    // records); it still wins where it counts: two images in flight 0.1821 against 0.1853 ms per step (A/B in one
    // call, profiles/r04/ab_inv_march.txt; fewer launches, 64 MB less traffic), and a batch affords tall bands
    // (64 x 2048^2: levels 2 + 1 in 1.27 ms against 1.08 + 0.44).
    const int mm = march_mode(hint);
    if ((!(hint.parts & DT_PART_INV21) && hint.program < 0) || mm == 0 || (mm < 0 && !march_pays(batch, rows, cols, hint))) return false;
    // (7, 5): near_sym_a; (3, 5): legall, whose 3-tap g0o runs as a centred zero-padded 7-tap one
    if (!((g0o.size() == 7 || g0o.size() == 3) && g1o.size() == 5) || g0a.size() != 10 || !lo_pos || hi_pos) return false;
    if (!symmetric(g0o) || !symmetric(g1o)) return false;       // the row filters fold the mirror pairs
    return march_sizes_ok(batch, rows, cols, dtm::Inv21m<7, 5, 10>::VL);
}

static int launch_inv21p_m10(dtm::Inv21mParams &p, const DtMarchHint &hint, hipStream_t s);
int dtcwt_march_inv21(const float *Z2, const float *Yh1, const float *Yh0, float *X, int B, int R, int C,
                      const std::vector<double> &g0o, const std::vector<double> &g1o, const float *l_a, const float *l_b,
                      const float *h_a, const float *h_b, const float *gain1, const float *gain2, const DtMarchHint &hint, hipStream_t s) {
    using G = dtm::Inv21m<7, 5, 10>;
    dtm::Inv21mParams p{};
    p.Z2 = Z2; p.Yh1 = Yh1; p.Yh0 = Yh0; p.X = X; p.B = B; p.R = R; p.C = C;
    const int off0 = (7 - (int)g0o.size()) / 2;             // legall: 3 taps centred in 7
    for (int k = 0; k < dtm::MAXT1; ++k) {
        p.g0o[k] = (k >= off0 && k - off0 < (int)g0o.size()) ? (float)g0o[k - off0] : 0.f;
        p.g1o[k] = k < (int)g1o.size() ? (float)g1o[k] : 0.f;
    }
    for (int k = 0; k < dtm::MAXT2; ++k) { p.l_a[k] = l_a[k]; p.l_b[k] = l_b[k]; p.h_a[k] = h_a[k]; p.h_b[k] = h_b[k]; }
    for (int d = 0; d < 6; ++d) { p.g1[d] = gain1[d]; p.g2[d] = gain2[d]; }
    dtm::dtm_pack_inv_biort(p, 7, 5);
    const int nstrip = cdiv(C, 4 * G::VL);
    if (!march_sizes_ok(B, R, C, G::VL)) return -3;         // (dtcwt_march_inv21_ok said so already)
    // (round 5 also built this kernel for ONE wavefront per SIMD with its requests two macro-steps ahead -- 4 % faster alone, 1.9 x slower
    // in flight, profiles/r05/ab_inv21m_prefetch.txt: not adopted, the builds and their switch are gone)
    // ONE transform at a time on the whole device, up to a 4096^2 image: the same macro-steps as a marching PAIR of wavefronts
    // (k_inv21p<7, 5, 10>, march2d_ipair.hpp; bit-identical output).  Such a launch does not fill the wave slots -- a pair puts two
    // wavefronts on every job and, where the slots are all taken (4096^2), affords bands twice as tall (72 rows instead of 40: 1.15 x
    // instead of 1.28 x the algorithmic bytes).  Inverse launch alone 2048^2 35.9 -> 29.6 us, 3072^2 55 -> 45.7, 4096^2 84 -> 80.6;
    // 5120^2 163 -> 172 (no longer chosen); with four in flight on quarters it loses outright (231 -> 338 us per image-share) and on
    // batches it is level (profiles/r05/pair_headline_inverse.txt).  DTCWT_HIP_MARCH_PARTS bit 128 off / bit 256 on: never / always.
    {
        const double useful = (double)B * R * C * ((double)C / (nstrip * 4.0 * G::VL));
        const bool alone = hint.in_flight <= 1 && hint.nparts <= 1;
        if ((hint.parts & DT_PART_INV21_ALWAYS_PAIR) || ((hint.parts & DT_PART_INV21_AS_PAIR) && alone && useful <= 1.8e7)) return launch_inv21p_m10(p, hint, s);
    }
    const unsigned jobs = dtm::dtm_set_jobs(p.jb, B, R, nstrip, pick_band_rows(B * hint.in_flight, R, nstrip, 10, hint.cus > 0 ? hint.cus : 1, hint.band));
    // (record rows loaded with the non-temporal hint: no difference, 0.1581 against 0.1586 ms per step)
    dtm::k_inv21m<7, 5, 10, 0><<<jobs, 64, 0, s>>>(p);
    return 0;
}

template <int M0, int M1, int M>
static int launch_fwd12(dtm::Fwd12mParams &p, const DtMarchHint &hint, hipStream_t s) {
    using G = dtm::Fwd12m<M0, M1, M>;
    const int nstrip = cdiv(p.C, 4 * G::VL);
    if (!march_sizes_ok(p.B, p.R, p.C, G::VL)) return -3;   // (dtcwt_march_fwd12_ok said so already)
    const unsigned jobs = dtm::dtm_set_jobs(p.jb, p.B, p.R, nstrip, pick_band_rows(p.B * hint.in_flight, p.R, nstrip, M, hint.cus, hint.band));
    // (X rows loaded with the non-temporal hint: 81.4 against 85.4 us alone, no difference inside the transform -- not used)
    // rows requested FOUR steps ahead (P = 4: 249 VGPRs, no scratch) instead of two: k_fwd12m alone 71-74 against 74-77 us, one
    // transform at a time 0.181 against 0.184 ms per step, nothing with four in flight or on the batches; Yh[1] / LoLo2 written with
    // the non-temporal hint (KO 256 / 512) cost the inverse its cache hits (80 -> 84 us alone) and bought nothing
    // (profiles/r05/ab_fwd12_prefetch.txt)
    if (p.LoLo1) dtm::k_fwd12m<M0, M1, M, 4, 128><<<jobs, 64, 0, s>>>(p);       // with `scales`: the level-1 lowpass is stored too
    else dtm::k_fwd12m<M0, M1, M, 4, 0><<<jobs, 64, 0, s>>>(p);
    return 0;
}

// X -> Yh0, Yh1, LoLo2; taps as the plan holds them (l_a .. h_b: coldfilt's first / second argument, Fwd2Params)
int dtcwt_march_fwd12(const float *X, float *Yh0, float *Yh1, float *LoLo2, float *LoLo1, int B, int R, int C,
                      const std::vector<double> &h0o, const std::vector<double> &h1o,
                      const float *l_a, const float *l_b, const float *h_a, const float *h_b, int m,
                      int lo_a_first, int hi_a_first, const DtMarchHint &hint, hipStream_t s) {
    dtm::Fwd12mParams p{};
    p.X = X; p.Yh0 = Yh0; p.Yh1 = Yh1; p.LoLo2 = LoLo2; p.LoLo1 = LoLo1; p.B = B; p.R = R; p.C = C;
    p.lo_a_first = lo_a_first; p.hi_a_first = hi_a_first;
    for (int k = 0; k < dtm::MAXT1; ++k) {
        p.h0[k] = k < (int)h0o.size() ? (float)h0o[k] : 0.f;
        p.h1[k] = k < (int)h1o.size() ? (float)h1o[k] : 0.f;
    }
    dtm::dtm_pack_qshift(p, m, l_a, l_b, h_a, h_b);
    dtm::dtm_pack_biort(p, (int)h0o.size(), (int)h1o.size());
    dtm::dtm_pack_biort_scaled(p, (int)h0o.size(), (int)h1o.size(), h0o.data(), h1o.data());
    const int m0 = (int)h0o.size(), m1 = (int)h1o.size();
    if (m == 10) {
        if (m0 == 5 && m1 == 7) return launch_fwd12<5, 7, 10>(p, hint, s);
        if (m0 == 5 && m1 == 3) return launch_fwd12<5, 3, 10>(p, hint, s);
    }
    return -3;
}

// ---- level 1 alone as a march (march2d_l1.hpp): the biort sets the fused launches above are not built for --------------
// forward: (len h0o, len h1o) = (13, 19) near_sym_b, (9, 7) antonini; inverse: (len g0o, len g1o) = (19, 13), (7, 9)
static bool l1_sizes_ok(int batch, int rows, int cols, int VL) {
    if (rows % 2 || cols % 4 || rows < 32 || cols < 32) return false;
    if ((int64_t)rows * cols * 4 >= ((int64_t)1 << 31)) return false;
    if ((int64_t)cdiv(cols, 4 * VL) * cdiv(rows, 8) * batch >= ((int64_t)1 << 30)) return false;
    return true;
}
static bool l1_pays(int batch, int rows, int cols, int VL, const DtMarchHint &h) {
    const int nstrip = cdiv(cols, 4 * VL);
    const double useful = (double)batch * rows * cols * ((double)cols / (nstrip * 4.0 * VL));
    return useful >= kCrossover[h.nparts > 1 ? 2 : (h.in_flight > 1 ? 1 : 0)].useful_pixels;
}

bool dtcwt_march_fwd1_ok(int batch, int rows, int cols, const std::vector<double> &h0o, const std::vector<double> &h1o,
                         const DtMarchHint &hint) {
    const int m0 = (int)h0o.size(), m1 = (int)h1o.size();
    if (!((m0 == 13 && m1 == 19) || (m0 == 9 && m1 == 7))) return false;
    if (!symmetric(h0o) || !symmetric(h1o)) return false;
    const int VL = m0 == 13 ? dtm::Fwd1m<13, 19>::VL : dtm::Fwd1m<9, 7>::VL;
    if (!l1_sizes_ok(batch, rows, cols, VL)) return false;
    if (!(hint.parts & DT_PART_L1) && hint.program < 0) return false;
    const int mm = march_mode(hint);
    // antonini (9 / 7 taps): the tile programs re-filter 8 halo rows per 32-row tile only, and the march measured no faster
    // (4096^2 forward 67 against 63.5 us alone, the step 0.240 against 0.250 ms with four in flight: profiles/r05/level1_march.txt)
    // -- it runs where the caller pins the marching program, near_sym_b wherever it pays
    return mm > 0 || (mm < 0 && m0 == 13 && l1_pays(batch, rows, cols, VL, hint));
}

template <int M0, int M1, int P>
static int launch_fwd1m(dtm::Fwd1mParams &p, const DtMarchHint &hint, hipStream_t s) {
    using G = dtm::Fwd1m<M0, M1>;
    const int nstrip = cdiv(p.C, 4 * G::VL);
    const unsigned jobs = dtm::dtm_set_jobs(p.jb, p.B, p.R, nstrip, pick_band_rows_l1(p.B * hint.in_flight, p.R, nstrip, 2 * G::PER, 0.5 * G::HH, hint.cus, hint.band));
    dtm::k_fwd1m<M0, M1, P><<<jobs, 64, 0, s>>>(p);
    return 0;
}

int dtcwt_march_fwd1(const float *X, float *LoLo, float *Yh0, int B, int R, int C, const std::vector<double> &h0o,
                     const std::vector<double> &h1o, const DtMarchHint &hint, hipStream_t s) {
    dtm::Fwd1mParams p{};
    p.X = X; p.LoLo = LoLo; p.Yh0 = Yh0; p.B = B; p.R = R; p.C = C;
    const int m0 = (int)h0o.size(), m1 = (int)h1o.size();
    dtm::dtm_pack_fwd1m(p, m0, m1, h0o.data(), h1o.data());
    if (m0 == 13 && m1 == 19) return launch_fwd1m<13, 19, 2>(p, hint, s);
    if (m0 == 9 && m1 == 7) return launch_fwd1m<9, 7, 1>(p, hint, s);
    return -3;
}

bool dtcwt_march_inv1_ok(int batch, int rows, int cols, const std::vector<double> &g0o, const std::vector<double> &g1o,
                         const DtMarchHint &hint) {
    const int m0 = (int)g0o.size(), m1 = (int)g1o.size();
    if (!((m0 == 19 && m1 == 13) || (m0 == 7 && m1 == 9))) return false;
    if (!symmetric(g0o) || !symmetric(g1o)) return false;
    const int VL = m0 == 19 ? dtm::Inv1m<19, 13>::VL : dtm::Inv1m<7, 9>::VL;
    if (!l1_sizes_ok(batch, rows, cols, VL)) return false;
    if (!(hint.parts & DT_PART_L1) && hint.program < 0) return false;
    const int mm = march_mode(hint);
    return mm > 0 || (mm < 0 && m0 == 19 && l1_pays(batch, rows, cols, VL, hint));        // (antonini: see the forward)
}

template <int M0, int M1>
static int launch_inv1m(dtm::Inv1mParams &p, const DtMarchHint &hint, hipStream_t s) {
    using G = dtm::Inv1m<M0, M1>;
    const int nstrip = cdiv(p.C, 4 * G::VL);
    const unsigned jobs = dtm::dtm_set_jobs(p.jb, p.B, p.R, nstrip, pick_band_rows_l1(p.B * hint.in_flight, p.R, nstrip, 4, 2.0 * G::WARM, hint.cus, hint.band));
    dtm::k_inv1m<M0, M1><<<jobs, 64, 0, s>>>(p);
    return 0;
}

int dtcwt_march_inv1(const float *Z, const float *Yh0, float *X, int B, int R, int C, const std::vector<double> &g0o,
                     const std::vector<double> &g1o, const float *gain1, const DtMarchHint &hint, hipStream_t s) {
    dtm::Inv1mParams p{};
    p.Z = Z; p.Yh0 = Yh0; p.X = X; p.B = B; p.R = R; p.C = C;
    for (int d = 0; d < 6; ++d) p.g1[d] = gain1[d];
    const int m0 = (int)g0o.size(), m1 = (int)g1o.size();
    dtm::dtm_pack_inv1m(p, m0, m1, g0o.data(), g1o.data());
    if (m0 == 19 && m1 == 13) return launch_inv1m<19, 13>(p, hint, s);
    if (m0 == 7 && m1 == 9) return launch_inv1m<7, 9>(p, hint, s);
    return -3;
}

// ---- the in-slice half of the 3-D level 1 for long filters (fused3d_long.hpp): the level-1 marches above with the four row-
// filtered planes in place of the lowpass + records; every slice of the volume is an image of the batch.  Called by fused3d.hip.
// DTCWT_HIP_LONG3D_BAND: rows per band of the in-slice launches of the 3-D level 1 (test hook, read per call like the other
// DTCWT_HIP_LONG3D_* hooks: the 3-D level functions have no plan to carry them)
static int long3d_band() { const char *e = getenv("DTCWT_HIP_LONG3D_BAND"); return e && atoi(e) > 0 ? atoi(e) : 0; }
static bool planes_one_strip(int C, int VL) {
    if (const char *e = getenv("DTCWT_HIP_LONG3D_EDGE")) return e[0] != '0' && C <= 256 && C >= 48;       // tests: force / forbid
    return C > 4 * VL && C <= 256;
}
int dtcwt_march_fwd1_planes(const float *X, float *P, int64_t pstride, int B, int R, int C, const double *h0o, int m0,
                            const double *h1o, int m1, int cus, hipStream_t s) {
    if (!(m0 == 13 && m1 == 19) || !l1_sizes_ok(B, R, C, dtm::Fwd1m<13, 19>::VL)) return -3;
    dtm::Fwd1mParams p{};
    p.X = X; p.LoLo = P; p.Yh0 = nullptr; p.pstride = pstride; p.B = B; p.R = R; p.C = C;
    dtm::dtm_pack_fwd1m_planes(p, m0, m1, h0o, h1o);
    using G = dtm::Fwd1m<13, 19>;
    // a row of 236 .. 256 columns: one strip without halo lanes (EDGE build) instead of two strips with 58 owning lanes each
    const bool edge = planes_one_strip(C, G::VL);
    const int nstrip = edge ? 1 : cdiv(C, 4 * G::VL);
    const unsigned jobs = dtm::dtm_set_jobs(p.jb, B, R, nstrip, pick_band_rows_l1(B, R, nstrip, 2 * G::PER, 0.5 * G::HH, cus, long3d_band()));
    if (edge) dtm::k_fwd1m<13, 19, 2, true, true><<<jobs, 64, 0, s>>>(p);
    else dtm::k_fwd1m<13, 19, 2, true><<<jobs, 64, 0, s>>>(p);
    return 0;
}
int dtcwt_march_inv1_planes(const float *P, int64_t pstride, float *X, int B, int R, int C, const double *g0o, int m0,
                            const double *g1o, int m1, int cus, hipStream_t s) {
    if (!(m0 == 19 && m1 == 13) || !l1_sizes_ok(B, R, C, dtm::Inv1m<19, 13>::VL)) return -3;
    dtm::Inv1mParams p{};
    p.Z = P; p.Yh0 = nullptr; p.X = X; p.pstride = pstride; p.B = B; p.R = R; p.C = C;
    dtm::dtm_pack_inv1m(p, m0, m1, g0o, g1o);
    using G = dtm::Inv1m<19, 13>;
    const bool edge = planes_one_strip(C, G::VL);
    const int nstrip = edge ? 1 : cdiv(C, 4 * G::VL);
    const unsigned jobs = dtm::dtm_set_jobs(p.jb, B, R, nstrip, pick_band_rows_l1(B, R, nstrip, 4, 2.0 * G::WARM, cus, long3d_band()));
    if (edge) dtm::k_inv1m<19, 13, true, true><<<jobs, 64, 0, s>>>(p);
    else dtm::k_inv1m<19, 13, true><<<jobs, 64, 0, s>>>(p);
    return 0;
}

// ---- round 6: both in-slice axes + cube2c of the 3-D level 1 for long filters in one launch (fused3d_long.hpp: k_fwd3l_slices),
// after the axis-0 pair filter: V [2][n0][n1][n2] -> LLL, Yh.  A job = four wavefronts = a slice pair x (strip, band of rows).
int dtcwt_march_fwd3l_slices(const float *V, int64_t vstride, float *LLL, float *Yh, int n0, int n1, int n2, const double *h0o, int m0,
                             const double *h1o, int m1, int cus, hipStream_t s) {
    using G = dtm::Fwd1m<13, 19>;
    if (!(m0 == 13 && m1 == 19) || n0 % 2 || !l1_sizes_ok(n0 / 2, n1, n2, G::VL)) return -3;
    dt3l::Fwd3sParams p{};
    p.V = V; p.vstride = vstride; p.LLL = LLL; p.Yh = Yh; p.n0 = n0; p.n1 = n1; p.n2 = n2;
    dt3l::pack_fwd3s(p, h0o, m0, h1o, m1);
    const bool edge = planes_one_strip(n2, G::VL);
    const int nstrip = edge ? 1 : cdiv(n2, 4 * G::VL);
    // two workgroups of four wavefronts per CU where the band picker counts eight one-wavefront jobs
    const int band = pick_band_rows_l1(n0 / 2, n1, nstrip, 2 * G::PER, 0.5 * G::HH, cus / 4 > 0 ? cus / 4 : 1, long3d_band());
    const unsigned jobs = dtm::dtm_set_jobs(p.jb, n0 / 2, n1, nstrip, band);
    if (edge) dt3l::k_fwd3l_slices<13, 19, 2, true><<<jobs, 256, 0, s>>>(p);
    else dt3l::k_fwd3l_slices<13, 19, 2, false><<<jobs, 256, 0, s>>>(p);
    return 0;
}

// ... and the inverse the same way round (k_inv3l_slices): LLL, Yh -> V [2][n0][n1][n2], which the axis-0 sum filter then merges
int dtcwt_march_inv3l_slices(const float *LLL, const float *Yh, float *V, int64_t vstride, int n0, int n1, int n2, const double *g0o, int m0,
                             const double *g1o, int m1, int cus, hipStream_t s) {
    using G = dtm::Inv1m<19, 13>;
    if (!(m0 == 19 && m1 == 13) || n0 % 2 || !l1_sizes_ok(n0 / 2, n1, n2, G::VL)) return -3;
    dt3l::Inv3sParams p{};
    p.LLL = LLL; p.Yh = Yh; p.V = V; p.vstride = vstride; p.n0 = n0; p.n1 = n1; p.n2 = n2;
    dt3l::pack_inv3s(p, m0, m1, g0o, g1o);
    const bool edge = planes_one_strip(n2, G::VL);
    const int nstrip = edge ? 1 : cdiv(n2, 4 * G::VL);
    const int band = pick_band_rows_l1(n0 / 2, n1, nstrip, 4, 2.0 * G::WARM, cus / 4 > 0 ? cus / 4 : 1, long3d_band());
    const unsigned jobs = dtm::dtm_set_jobs(p.jb, n0 / 2, n1, nstrip, band);
    if (edge) dt3l::k_inv3l_slices<19, 13, true><<<jobs, 256, 0, s>>>(p);
    else dt3l::k_inv3l_slices<19, 13, false><<<jobs, 256, 0, s>>>(p);
    return 0;
}

// ---- levels 1 + 2 of the forward as a marching PAIR of wavefronts (march2d_pair.hpp) ------------------------------------------
// near_sym_a / legall-length level-1 filters (5, 7) with the 14- / 18-tap q-shift sets (qshift_b, qshift_d): 112 / 144 registers
// of pending sums that one wavefront cannot hold beside level 1.  Measured (profiles/r05/pair_forward.txt, pair / level-1 tile
// launch + level-2 tile launch): 64 x 2048^2 3.12 / 3.47 ms per step, 64 x 1024^2 0.756 / 0.869, 4096^2 with four in flight 0.214 /
// 0.232 -- and 0.251 / 0.237 for ONE 4096^2 image at a time: the level-1 wavefront of a pair issues all of level 1's
// instructions while its partner waits for rows, and a launch that cannot fill the chip twice over feels that.  So: wherever
// other work shares the device (concurrency hint, partition context) from the usual crossover, alone from 60 M useful pixels.
// near_sym_b was built too and lost everywhere (146 us against 72 + 28 alone, 451 against 245 + 157 in flight: its level-1
// wavefront carries 573 instructions per step, the partner 130): its level 1 stays a march of its own (march2d_l1.hpp).
// DTCWT_HIP_MARCH_PARTS without bit 4: never.
bool dtcwt_march_fwd12p_ok(int batch, int rows, int cols, const std::vector<double> &h0o, const std::vector<double> &h1o,
                           const std::vector<double> &h0a, bool lo_a_first, bool hi_a_first, const DtMarchHint &hint) {
    if (!(hint.parts & DT_PART_FPAIR)) return false;
    const int m0 = (int)h0o.size(), m1 = (int)h1o.size(), m = (int)h0a.size();
    if (!(m0 == 5 && (m1 == 7 || m1 == 3) && (m == 14 || m == 18))) return false;          // near_sym_a, legall
    if (!symmetric(h0o) || !symmetric(h1o) || !lo_a_first || hi_a_first) return false;
    const int VL = m == 14 ? dtm::Fwd12p<5, 7, 14>::VL : dtm::Fwd12p<5, 7, 18>::VL;
    if (!march_sizes_ok(batch, rows, cols, VL)) return false;
    const int mm = march_mode(hint);
    if (mm == 0) return false;
    if (mm > 0) return true;
    const int nstrip = cdiv(cols, 4 * VL);
    const double useful = (double)batch * rows * cols * ((double)cols / (nstrip * 4.0 * VL));
    const bool shared = hint.nparts > 1 || hint.in_flight > 1;
    return useful >= (shared ? kCrossover[hint.nparts > 1 ? 2 : 1].useful_pixels : 6.0e7);
}

template <int M0, int M1, int M>
static int launch_fwd12p(dtm::Fwd12pParams &p, const DtMarchHint &hint, hipStream_t s) {
    using G = dtm::Fwd12p<M0, M1, M>;
    const int nstrip = cdiv(p.C, 4 * G::VL);
    // a job is a PAIR of wavefronts: half as many fit the chip as single-wavefront jobs -- three quarters with the M = 14 build for three
    // wavefronts per SIMD, taken on partition contexts (profiles/r05/ab_fpair_occ.txt)
    const int wps = (M == 14 && hint.nparts > 1) ? 3 : 2;
    const int cus = hint.cus * wps / 4 > 0 ? hint.cus * wps / 4 : 1;
    const unsigned jobs = dtm::dtm_set_jobs(p.jb, p.B, p.R, nstrip, pick_band_rows(p.B * hint.in_flight, p.R, nstrip, M, cus, hint.band));
    if constexpr (M == 14) { if (wps == 3) { dtm::k_fwd12p<M0, M1, M, 2, 3><<<jobs, 128, 0, s>>>(p); return 0; } }
    dtm::k_fwd12p<M0, M1, M, 2><<<jobs, 128, 0, s>>>(p);
    return 0;
}

int dtcwt_march_fwd12p(const float *X, float *Yh0, float *Yh1, float *LoLo2, int B, int R, int C,
                       const std::vector<double> &h0o, const std::vector<double> &h1o,
                       const float *l_a, const float *l_b, const float *h_a, const float *h_b, int m,
                       const DtMarchHint &hint, hipStream_t s) {
    dtm::Fwd12pParams p{};
    p.X = X; p.Yh0 = Yh0; p.Yh1 = Yh1; p.LoLo2 = LoLo2; p.B = B; p.R = R; p.C = C;
    const int m0 = (int)h0o.size(), m1 = (int)h1o.size();
    const double rs = 0.70710678118654752440;
    for (int d = 0; d <= dtm::MAXH1; ++d) {
        const double a = d <= m0 / 2 ? h0o[m0 / 2 - d] : 0.0, b = d <= m1 / 2 ? h1o[m1 / 2 - d] : 0.0;
        p.hp[2 * d] = (float)a; p.hp[2 * d + 1] = (float)b;
        p.hpl[2 * d] = (float)a; p.hpl[2 * d + 1] = (float)(b * rs);
        p.hph[2 * d] = (float)(a * rs); p.hph[2 * d + 1] = (float)(b * rs);
    }
    dtm::dtm_pack_qshift(p, m, l_a, l_b, h_a, h_b);
    if (m0 == 5 && m1 == 7 && m == 14) return launch_fwd12p<5, 7, 14>(p, hint, s);
    if (m0 == 5 && m1 == 7 && m == 18) return launch_fwd12p<5, 7, 18>(p, hint, s);
    if (m0 == 5 && m1 == 3 && m == 14) return launch_fwd12p<5, 3, 14>(p, hint, s);
    if (m0 == 5 && m1 == 3 && m == 18) return launch_fwd12p<5, 3, 18>(p, hint, s);
    return -3;
}

// ---- levels 2 + 1 of the inverse as a marching PAIR of wavefronts (march2d_ipair.hpp) -----------------------------------------
// The synthesis filters of near_sym_a (7, 5) / legall (3, 5) with the 14- / 18-tap q-shift sets, standard phases; chosen like the
// forward pair: wherever other work shares the device from the usual crossover, alone from 60 M useful pixels
// (profiles/r05/pair_inverse.txt).  DTCWT_HIP_MARCH_PARTS without bit 8: never.
bool dtcwt_march_inv21p_ok(int batch, int rows, int cols, const std::vector<double> &g0o, const std::vector<double> &g1o,
                           const std::vector<double> &g0a, bool lo_pos, bool hi_pos, const DtMarchHint &hint) {
    if (!(hint.parts & DT_PART_IPAIR)) return false;
    const int m = (int)g0a.size();
    if (!((g0o.size() == 7 || g0o.size() == 3) && g1o.size() == 5) || !(m == 14 || m == 18) || !lo_pos || hi_pos) return false;
    if (!symmetric(g0o) || !symmetric(g1o)) return false;
    const int VL = m == 14 ? dtm::Inv21p<7, 5, 14>::VL : dtm::Inv21p<7, 5, 18>::VL;
    if (!march_sizes_ok(batch, rows, cols, VL)) return false;
    const int mm = march_mode(hint);
    if (mm == 0) return false;
    if (mm > 0) return true;
    const int nstrip = cdiv(cols, 4 * VL);
    const double useful = (double)batch * rows * cols * ((double)cols / (nstrip * 4.0 * VL));
    const bool shared = hint.nparts > 1 || hint.in_flight > 1;
    return useful >= (shared ? kCrossover[hint.nparts > 1 ? 2 : 1].useful_pixels : 6.0e7);
}

template <int M>
static int launch_inv21p(dtm::Inv21mParams &p, const DtMarchHint &hint, hipStream_t s) {
    using G = dtm::Inv21p<7, 5, M>;
    const int nstrip = cdiv(p.C, 4 * G::VL);
    if (!march_sizes_ok(p.B, p.R, p.C, G::VL)) return -3;
    // A job is a PAIR of wavefronts: half as many are resident as single-wavefront jobs.  At M = 14 the registers (162) allow a third
    // wavefront per SIMD: on a partition context (a share of the CUs, the other shares busy) that measured -8 % per step (four 4096^2 in
    // flight 0.202 -> 0.186 ms, 64 x 2048^2 on quarters -2 %), on the whole device +3.5 % (64 x 1024^2): profiles/r05/ab_ipair_occ.txt
    // (at M = 10 three measured no faster alone)
    const int wps = (M == 14 && hint.nparts > 1) ? 3 : 2;
    const int cus = hint.cus * wps / 4 > 0 ? hint.cus * wps / 4 : 1;
    const unsigned jobs = dtm::dtm_set_jobs(p.jb, p.B, p.R, nstrip, pick_band_rows(p.B * hint.in_flight, p.R, nstrip, M, cus, hint.band));
    if constexpr (M == 14) { if (wps == 3) { dtm::k_inv21p<7, 5, M, 3><<<jobs, 128, 0, s>>>(p); return 0; } }
    dtm::k_inv21p<7, 5, M><<<jobs, 128, 0, s>>>(p);
    return 0;
}

static int launch_inv21p_m10(dtm::Inv21mParams &p, const DtMarchHint &hint, hipStream_t s) { return launch_inv21p<10>(p, hint, s); }

int dtcwt_march_inv21p(const float *Z2, const float *Yh1, const float *Yh0, float *X, int B, int R, int C,
                       const std::vector<double> &g0o, const std::vector<double> &g1o, const float *l_a, const float *l_b,
                       const float *h_a, const float *h_b, int m, const float *gain1, const float *gain2, const DtMarchHint &hint, hipStream_t s) {
    dtm::Inv21mParams p{};
    p.Z2 = Z2; p.Yh1 = Yh1; p.Yh0 = Yh0; p.X = X; p.B = B; p.R = R; p.C = C;
    const int off0 = (7 - (int)g0o.size()) / 2;             // legall: 3 taps centred in 7
    for (int k = 0; k < dtm::MAXT1; ++k) {
        p.g0o[k] = (k >= off0 && k - off0 < (int)g0o.size()) ? (float)g0o[k - off0] : 0.f;
        p.g1o[k] = k < (int)g1o.size() ? (float)g1o[k] : 0.f;
    }
    for (int k = 0; k < dtm::MAXT2; ++k) { p.l_a[k] = l_a[k]; p.l_b[k] = l_b[k]; p.h_a[k] = h_a[k]; p.h_b[k] = h_b[k]; }
    for (int d = 0; d < 6; ++d) { p.g1[d] = gain1[d]; p.g2[d] = gain2[d]; }
    dtm::dtm_pack_inv_biort(p, 7, 5);
    if (m == 14) return launch_inv21p<14>(p, hint, s);
    if (m == 18) return launch_inv21p<18>(p, hint, s);
    return -3;
}

// ---- level 2 of the inverse alone as a march (march2d_ipair.hpp: k_inv2m) ------------------------------------------------------
// For the 14- / 18-tap q-shift sets where no pair takes levels 2 + 1 (near_sym_b: its level 1 is k_inv1m): Z2 + Yh[1] -> Z1.
// Standard phases, sizes in fours; chosen by the crossover of the level-1 marches.  DTCWT_HIP_MARCH_PARTS without bit 32: never.
bool dtcwt_march_inv2_ok(int batch, int rows, int cols, const std::vector<double> &g0a, bool lo_pos, bool hi_pos, const DtMarchHint &hint) {
    if (!(hint.parts & DT_PART_INV2)) return false;
    const int m = (int)g0a.size();
    if (!(m == 14 || m == 18) || !lo_pos || hi_pos) return false;
    const int VL = m == 14 ? dtm::Inv2m<14>::VL : dtm::Inv2m<18>::VL;
    if (!march_sizes_ok(batch, rows, cols, VL)) return false;
    const int mm = march_mode(hint);
    return mm > 0 || (mm < 0 && l1_pays(batch, rows, cols, VL, hint));
}

template <int M, int WPS>
static int launch_inv2m(dtm::Inv21mParams &p, const DtMarchHint &hint, hipStream_t s) {
    using G = dtm::Inv2m<M>;
    const int nstrip = cdiv(p.C, 4 * G::VL);
    const int cus = hint.cus * WPS / 2 > 0 ? hint.cus * WPS / 2 : 1;        // WPS wavefronts per SIMD where the band picker counts two
    const unsigned jobs = dtm::dtm_set_jobs(p.jb, p.B, p.R, nstrip, pick_band_rows(p.B * hint.in_flight, p.R, nstrip, M, cus, hint.band));
    dtm::k_inv2m<M, WPS><<<jobs, 64, 0, s>>>(p);
    return 0;
}

// Z2 [B][R/2][C/2], Yh1 [B][R/4][C/4][12] -> Z1 [B][R][C]; gain2: the level's six subband gains x sqrt(1/2)
int dtcwt_march_inv2(const float *Z2, const float *Yh1, float *Z1, int B, int R, int C, const float *l_a, const float *l_b,
                     const float *h_a, const float *h_b, int m, const float *gain2, const DtMarchHint &hint, hipStream_t s) {
    dtm::Inv21mParams p{};
    p.Z2 = Z2; p.Yh1 = Yh1; p.Yh0 = nullptr; p.X = Z1; p.B = B; p.R = R; p.C = C;
    for (int k = 0; k < dtm::MAXT2; ++k) { p.l_a[k] = l_a[k]; p.l_b[k] = l_b[k]; p.h_a[k] = h_a[k]; p.h_b[k] = h_b[k]; }
    for (int d = 0; d < 6; ++d) p.g2[d] = gain2[d];
    if (m == 14) return launch_inv2m<14, 3>(p, hint, s);
    if (m == 18) return launch_inv2m<18, 2>(p, hint, s);
    return -3;
}

// ---- level 2 of the forward alone as a march (march2d_pair.hpp: k_fwd2m) ------------------------------------------------------
// The forward counterpart of k_inv2m: LoLo1 -> Yh[1], LoLo2 for the 14- / 18-tap q-shift sets where no pair takes levels 1 + 2.
// DTCWT_HIP_MARCH_PARTS without bit 16: never.
bool dtcwt_march_fwd2_ok(int batch, int rows, int cols, const std::vector<double> &h0a, bool lo_a_first, bool hi_a_first, const DtMarchHint &hint) {
    if (!(hint.parts & DT_PART_FWD2)) return false;
    const int m = (int)h0a.size();
    if (!(m == 14 || m == 18) || !lo_a_first || hi_a_first) return false;
    const int VL = m == 14 ? dtm::Fwd2m<14>::VL : dtm::Fwd2m<18>::VL;
    if (!march_sizes_ok(batch, rows, cols, VL)) return false;
    const int mm = march_mode(hint);
    if (mm >= 0) return mm > 0;
    // like the pairs: wherever other work shares the device from the usual crossover, alone from 60 M useful pixels (one 4096^2 image at
    // a time ran 2.6 % slower per step with it, four in flight 3.5 % faster, the batches 2-3 % faster: profiles/r05/fwd2m.txt)
    const int nstrip = cdiv(cols, 4 * VL);
    const double useful = (double)batch * rows * cols * ((double)cols / (nstrip * 4.0 * VL));
    const bool shared = hint.nparts > 1 || hint.in_flight > 1;
    return useful >= (shared ? kCrossover[hint.nparts > 1 ? 2 : 1].useful_pixels : 6.0e7);
}

template <int M, int WPS>
static int launch_fwd2m(dtm::Fwd12pParams &p, const DtMarchHint &hint, hipStream_t s) {
    using G = dtm::Fwd2m<M>;
    const int nstrip = cdiv(p.C, 4 * G::VL);
    const int cus = hint.cus * WPS / 2 > 0 ? hint.cus * WPS / 2 : 1;
    const unsigned jobs = dtm::dtm_set_jobs(p.jb, p.B, p.R, nstrip, pick_band_rows(p.B * hint.in_flight, p.R, nstrip, M, cus, hint.band));
    dtm::k_fwd2m<M, WPS><<<jobs, 64, 0, s>>>(p);
    return 0;
}

// LoLo1 [B][R][C] -> Yh1 [B][R/4][C/4][12], LoLo2 [B][R/2][C/2]
int dtcwt_march_fwd2(const float *LoLo1, float *Yh1, float *LoLo2, int B, int R, int C, const float *l_a, const float *l_b,
                     const float *h_a, const float *h_b, int m, const DtMarchHint &hint, hipStream_t s) {
    dtm::Fwd12pParams p{};
    p.X = LoLo1; p.Yh0 = nullptr; p.Yh1 = Yh1; p.LoLo2 = LoLo2; p.B = B; p.R = R; p.C = C;
    dtm::dtm_pack_qshift(p, m, l_a, l_b, h_a, h_b);
    if (m == 14) return launch_fwd2m<14, 2>(p, hint, s);        // (188 VGPRs: a build forced to three wavefronts per SIMD spills 80 bytes per lane)
    if (m == 18) return launch_fwd2m<18, 2>(p, hint, s);
    return -3;
}
