// What does the device give a TRIVIAL program that moves the algorithmic bytes of a 4096^2 nlevels=4 forward + inverse step
// in the launch structure bench.py times?  Per step and stream, six float4 streaming kernels on rotating buffer sets:
//   forward levels 1 + 2   read  67.1 MB (X), write 268.4 MB (Yh[0] 201.3, Yh[1] 50.3, LoLo2 16.8)        non-temporal stores
//   forward level 3        read  16.8 MB, write 16.8 MB (LoLo3 4.2 + Yh[2] 12.6);  level 4: 4.2 MB -> 4.2 MB
//   inverse level 4, 3     the same bytes the other way round
//   inverse levels 2 + 1   read 268.4 MB, write 67.1 MB
// i.e. 20 B/px per direction + the LoLo2 / LoLo3 round trips the two-launch-per-level tail makes -- what k_fwd12m, k_fwd2 x 2,
// k_inv2 x 2, k_inv21m move when they move nothing twice.  Protocols: one stream on the whole device; four plain streams; four
// streams on quarters of the compute units (hipExtStreamCreateWithCUMask, contiguous mask ranges as dtcwt_hip_ctx_create_partition
// builds them).  Eight buffer sets of 403 MB as in bench.py, so nothing is served by the Infinity Cache from the previous step of
// the same set.  Prints ms per step over 20 and over 200 steps (host clock between device syncs, as bench.py measures).
//   hipcc --offload-arch=gfx950 -O3 tools/kbench/step_probe.hip -o tools/kbench/step_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstring>
#include <cstdlib>
#include <unistd.h>
#include <cstdio>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// n float4 per plane; NR planes in, NW planes out
template <int NR, int NW, bool NT>
__global__ void __launch_bounds__(256) k_mix(const v4 *__restrict__ in, v4 *__restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    v4 a = in[i];
#pragma unroll
    for (int r = 1; r < NR; ++r) a += in[i + r * n];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const v4 b = a * (float)(w + 1);
        if (NT) __builtin_nontemporal_store(b, out + i + w * n); else out[i + w * n] = b;
    }
}

struct Set { v4 *X, *rec, *l2, *l3, *y2, *l4, *y3, *Z; };      // X 67 MB, rec 268 MB (incl. LoLo2 as its last quarter plane), ...
static size_t PX = (size_t)4096 * 4096;     // pixels per step: x K with --scale K (K = 16: a 64 x 2048^2 batch, K = 4: 64 x 1024^2)
static size_t NP = PX / 4;                  // float4 per plane

static hipEvent_t g_ev[7];
static bool g_mark = false;       // record an event before every kernel of the step and after the last
#define MARK(i) do { if (g_mark) CK(hipEventRecord(g_ev[i], st)); } while (0)
static bool g_fused_tail = false;   // --fused-tail: levels 3 + 4 as ONE launch per direction (see main)
static void step(const Set &s, hipStream_t st, bool inv_first = false) {
    const unsigned g1 = (unsigned)((NP + 255) / 256), g3 = (unsigned)((NP / 4 + 255) / 256), g4 = (unsigned)((NP / 16 + 255) / 256);
    if (inv_first) {        // the same six kernels, the inverse half of the step first (it reads what the previous step on this set wrote)
        k_mix<1, 1, false><<<g4, 256, 0, st>>>(s.l4, s.l3, NP / 16);
        k_mix<1, 1, false><<<g3, 256, 0, st>>>(s.l3, s.l2, NP / 4);
        k_mix<4, 1, true><<<g1, 256, 0, st>>>(s.rec, s.Z, NP);
        k_mix<1, 4, true><<<g1, 256, 0, st>>>(s.X, s.rec, NP);
        k_mix<1, 1, false><<<g3, 256, 0, st>>>(s.l2, s.l3, NP / 4);
        k_mix<1, 1, false><<<g4, 256, 0, st>>>(s.l3, s.l4, NP / 16);
        return;
    }
    MARK(0);
    k_mix<1, 4, true><<<g1, 256, 0, st>>>(s.X, s.rec, NP);                 // forward levels 1 + 2
    MARK(1);
    k_mix<1, 1, false><<<g3, 256, 0, st>>>(s.l2, s.l3, NP / 4);            // level 3: LoLo2 -> LoLo3 + Yh[2] (16.8 MB each way)
    MARK(2);
    if (!g_fused_tail) k_mix<1, 1, false><<<g4, 256, 0, st>>>(s.l3, s.l4, NP / 16);           // level 4
    MARK(3);
    if (!g_fused_tail) k_mix<1, 1, false><<<g4, 256, 0, st>>>(s.l4, s.l3, NP / 16);           // inverse level 4
    MARK(4);
    k_mix<1, 1, false><<<g3, 256, 0, st>>>(s.l3, s.l2, NP / 4);            // inverse level 3
    MARK(5);
    k_mix<4, 1, true><<<g1, 256, 0, st>>>(s.rec, s.Z, NP);                 // inverse levels 2 + 1
    MARK(6);
}

static bool g_stagger = false;      // odd streams run the inverse half of their steps first
static double run(const std::vector<Set> &sets, const std::vector<hipStream_t> &sts, int nsteps) {
    const int S = (int)sts.size();
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < nsteps; ++k) step(sets[k % sets.size()], sts[k % S], g_stagger && ((k % S) & 1));
    CK(hipDeviceSynchronize());
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / nsteps;
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const bool quick = argc > 1 && !strcmp(argv[1], "--json");       // one JSON line for bench.py: fewer repetitions
    for (int i = 1; i + 1 < argc; ++i)
        if (!strcmp(argv[i], "--scale")) { PX *= (size_t)atoi(argv[i + 1]); NP = PX / 4; }
    // --fused-tail (round 6): what would ONE launch per direction for levels >= 3 buy at best?  A fused launch reads
    // LoLo2 (16.8 MB) and writes Yh[2] + Yh[3] + LoLo4 (12.6 + 3.1 + 1.0 = 16.8 MB): the bytes of the level-3 kernel above -- the LoLo3
    // round trip (4.2 MB out, 4.2 MB back) is what it saves, exactly what the two level-4 kernels move.  So the step WITHOUT the two level-4
    // launches is the trivial program of a fused tail with no halo, no recomputation and no longer dependency chain: an upper bound.
    for (int i = 1; i < argc; ++i) if (!strcmp(argv[i], "--fused-tail")) { g_fused_tail = true; printf("fused tail: four launches per step (no level-4 launches)\n"); }
    if (PX != (size_t)4096 * 4096) printf("%zu pixels per step (x %zu)\n", PX, PX / ((size_t)4096 * 4096));
    const int NSET = 8;
    std::vector<Set> sets(NSET);
    for (auto &s : sets) {
        CK(hipMalloc(&s.X, NP * 16)); CK(hipMalloc(&s.rec, 4 * NP * 16)); CK(hipMalloc(&s.l3, NP / 4 * 16)); CK(hipMalloc(&s.l4, NP / 16 * 16));
        CK(hipMalloc(&s.Z, NP * 16));
        s.l2 = s.rec + 4 * NP - NP / 4;        // LoLo2: the last 16.8 MB of the record block
        CK(hipMemset(s.X, 0x3c, NP * 16)); CK(hipMemset(s.rec, 0, 4 * NP * 16));
    }
    for (auto &e : g_ev) CK(hipEventCreate(&e));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    for (int proto = 0; proto < (quick ? 3 : 5); ++proto) {
        std::vector<hipStream_t> sts;
        const char *name = proto == 0 ? "one stream, whole device" : (proto == 1 ? "four plain streams" : (proto == 2 ? "four streams on quarters of the CUs" :
                           (proto == 3 ? "four on quarters, odd streams inverse half first" : "four plain streams, odd streams inverse half first")));
        const int S = proto == 0 ? 1 : 4;
        g_stagger = proto >= 3;
        for (int q = 0; q < S; ++q) {
            hipStream_t st;
            if (proto == 2 || proto == 3) {
                const int per = cus / 4;
                std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
                for (int b = q * per; b < (q + 1) * per; ++b) mask[(size_t)b / 32] |= 1u << (b % 32);
                CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
            } else CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            sts.push_back(st);
        }
        for (int w = 0; w < 3; ++w) run(sets, sts, 200);            // clocks up
        double b20 = 1e9, b200 = 1e9, s20 = 0, s200 = 0;
        const int nrep = quick ? 3 : 6;
        for (int rep = 0; rep < nrep; ++rep) {
            const double a = run(sets, sts, 20), b = run(sets, sts, 200);
            b20 = a < b20 ? a : b20; b200 = b < b200 ? b : b200; s20 += a / nrep; s200 += b / nrep;
        }
        const double bytes = 40.0 * PX, moved = bytes + 2 * 2 * (1.0 + 0.25) * PX;
        if (quick) {
            printf("%s\"%s\": {\"ms_per_step_20\": %.5f, \"ms_per_step_200\": %.5f}%s", proto == 0 ? "{" : " ", proto == 0 ? "one_stream" : (proto == 1 ? "four_plain_streams" : "four_streams_on_quarters"),
                   s20, s200, proto == 2 ? "}\n" : ",");
        } else printf("%-40s 20 steps: mean %.4f best %.4f ms   200 steps: mean %.4f best %.4f ms   = %.2f TB/s of algorithmic bytes (%.3f of 8), %.2f TB/s moved\n",
               name, s20, b20, s200, b200, bytes / (s200 * 1e-3) / 1e12, bytes / (s200 * 1e-3) / 8e12, moved / (s200 * 1e-3) / 1e12);
        if (!quick && proto < 3) {
            // the kernels of a step as the protocol runs them: events around the kernels of every fourth step of stream 0, the
            // other streams busy beside it (level 3 -> 4 and 4 -> 3 come as pairs: the event between them is not recorded)
            double acc[6] = {0, 0, 0, 0, 0, 0}; int nacc = 0;
            for (int rep = 0; rep < 40; ++rep) {
                for (int k = 0; k < 3 * S; ++k) step(sets[k % sets.size()], sts[k % S]);
                g_mark = true; step(sets[0], sts[0]); g_mark = false;
                for (int k = 1; k < S; ++k) step(sets[k % sets.size()], sts[k % S]);
                CK(hipDeviceSynchronize());
                if (rep < 5) continue;
                for (int i = 0; i < 6; ++i) { float ms; CK(hipEventElapsedTime(&ms, g_ev[i], g_ev[i + 1])); acc[i] += ms; }
                ++nacc;
            }
            printf("    per kernel on stream 0 (us): fwd 1+2 %.1f | L3 %.1f | L4 %.1f | inv L4 %.1f | inv L3 %.1f | inv 2+1 %.1f\n",
                   acc[0] / nacc * 1e3, acc[1] / nacc * 1e3, acc[2] / nacc * 1e3, acc[3] / nacc * 1e3, acc[4] / nacc * 1e3, acc[5] / nacc * 1e3);
        }
    }
    // no teardown: a process that destroys CU-masked streams and exits normally did not come back on this runtime
    // (the first run of this probe sat in exit until its 600 s limit)
    fflush(stdout);
    _exit(0);
}
