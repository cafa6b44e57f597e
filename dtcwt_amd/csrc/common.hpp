// Shared declarations of libdtcwt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "dtcwt_hip.h"

// The kernels are written for gfx950 (MI355X) only: 160 KiB of LDS per CU (the 3-D level-1 inverse declares 78 KiB
// for one workgroup), wave64, the gfx950 forms of the buffer / LDS instructions.  Other targets are not supported.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libdtcwt_hip is a gfx950 (MI355X) library: build with --offload-arch=gfx950"
#endif

struct dtcwt_hip_ctx {
    int device;
    hipStream_t stream;
    bool owns_stream;
    int cus;                    // compute units the context's stream runs on (its share, for a partition context)
    int nparts = 1;             // > 1: one of `nparts` equal shares of the device (dtcwt_hip_ctx_create_partition)
    // Size-bucketed cache of freed device buffers.  Every buffer of a context is used on
    // the context's single stream, so handing a freed buffer to the next allocation of the
    // same size is safe without synchronising (stream order) -- and it keeps
    // hipMalloc/hipFree (each a device-wide sync of ~100 us) out of the level loops.
    std::unordered_map<size_t, std::vector<void *>> pool;   // size -> free buffers
    std::unordered_map<void *, size_t> live;                // buffer -> size
    size_t pooled_bytes = 0;
    size_t pool_limit = (size_t)16 << 30;                   // raised to HBM/4 at creation; DTCWT_HIP_POOL_MB overrides
    // malloc / free / trim may come from different host threads: the Python layer shares one default context
    // per device, ctypes releases the GIL during calls and garbage collection frees arrays from any thread
    std::mutex pool_mu;
    // things created on behalf of the context that must go before it (cached graphs and their buffers)
    std::vector<std::function<void()>> on_destroy;
    // downloads that overlap the kernels issued after them (dtcwt_hip_memcpy_d2h_overlapped): a second stream that
    // waits, per copy, for an event recorded on `stream`; created on first use
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_event = nullptr;
};

// What the 2-D plan tells the marching launchers (march2d.hip) about the call: the context's compute units, whether that
// is a share of the device, how many independent transforms the caller keeps in flight, and the program the caller
// pinned (dtcwt_hip_plan2d_set_program: -1 = the library chooses, 0 = tile programs, 1 = marching launches wherever
// the geometry and the filters allow).
// ... and the three environment switches of the marching programs, read ONCE when a plan is created (dt_march_switches(),
// march2d.hip) -- a plan runs the same programs for its whole life, and dtcwt_hip_plan2d_describe() reports them:
//   DTCWT_HIP_MARCH        0 = never, 1 = wherever the geometry and the filters allow, unset = where it also pays (-1)
//   DTCWT_HIP_MARCH_BAND   rows per band of every marching launch (0 / unset: chosen per launch from the job count)
//   DTCWT_HIP_MARCH_PARTS  bit mask of the marching programs a plan may use (default: all but the last)
enum {
    DT_PART_FWD12 = 1,             // k_fwd12m   levels 1 + 2 forward, one wavefront per job
    DT_PART_INV21 = 2,             // k_inv21m   levels 2 + 1 inverse
    DT_PART_FPAIR = 4,             // k_fwd12p   levels 1 + 2 forward as a pair of wavefronts (14- / 18-tap q-shift sets)
    DT_PART_IPAIR = 8,             // k_inv21p   levels 2 + 1 inverse as a pair (14- / 18-tap q-shift sets)
    DT_PART_FWD2 = 16,             // k_fwd2m    level 2 forward alone
    DT_PART_INV2 = 32,             // k_inv2m    level 2 inverse alone
    DT_PART_L1 = 64,               // k_fwd1m / k_inv1m   level 1 alone (near_sym_b, antonini)
    DT_PART_INV21_AS_PAIR = 128,   // the headline set's inverse as a pair where ONE transform has the device (bit-identical to k_inv21m)
    DT_PART_INV21_ALWAYS_PAIR = 256,   // ... at every size and in flight too (tests: both forms of the same macro-steps)
    DT_PART_DEFAULT = 255
};
struct DtMarchHint {
    int cus, nparts, in_flight, program;
    int env_march = -1;
    int band = 0;
    unsigned parts = DT_PART_DEFAULT;
};
DtMarchHint dt_march_switches();

struct dtcwt_hip_event {
    hipEvent_t ev;
};

// thread-local error text returned by dtcwt_hip_last_error()
int dtcwt_set_error(int code, const char *fmt, ...);

#define DT_CHECK_HIP(expr)                                                               \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess)                                                           \
            return dtcwt_set_error(-2, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                                   __FILE__, __LINE__);                                  \
    } while (0)

#define DT_REQUIRE(cond, ...)                                                            \
    do {                                                                                 \
        if (!(cond)) return dtcwt_set_error(-1, __VA_ARGS__);                            \
    } while (0)

// Half-sample symmetric reflection of an integer index into [0, n): ... 1 0 | 0 1 ...
// n-1 | n-1 n-2 ...  (what dtcwt/utils.py:136-153 `reflect(x, -0.5, n-0.5)` computes for
// integer x).  Multi-bounce safe.
__host__ __device__ inline int64_t dt_reflect(int64_t u, int64_t n) {
    int64_t p = 2 * n;
    int64_t j = u % p;
    if (j < 0) j += p;
    return j < n ? j : p - 1 - j;
}

// Cheap variant for |overshoot| < n (single bounce), used by the fused tiles.
__device__ inline int dt_reflect1(int u, int n) {
    u = u < 0 ? -1 - u : u;
    return u >= n ? 2 * n - 1 - u : u;
}

__host__ __device__ inline int64_t dt_clamp(int64_t u, int64_t lo, int64_t hi) {
    return u < lo ? lo : (u > hi ? hi : u);
}

// Dense single-axis fast paths of generic2d.hip on [outer][n][inner] arrays (inner >= 32: marching
// kernels, inner == 1: LDS-row kernels).  kind 0 = colfilter algebra (odd lengths), kind 1 =
// coldfilt (pair) / colifilt (sum).  Return 1 = launched, 0 = not applicable, < 0 = error.
int dtcwt_g2_pair(dtcwt_hip_ctx *ctx, int dtype, int kind, const void *X, void *Y0, void *Y1,
                  int64_t outer, int64_t n, int64_t inner, int pad_lo, int pad_hi,
                  const double *lo_a, const double *lo_b, const double *hi_a, const double *hi_b,
                  int m_lo, int m_hi, int pack_hi);
int dtcwt_g2_sum(dtcwt_hip_ctx *ctx, int dtype, int kind, const void *X0, const void *X1, void *Y,
                 int64_t outer, int64_t n, int64_t inner, int crop, const double *lo_a,
                 const double *lo_b, const double *hi_a, const double *hi_b, int m_lo, int m_hi,
                 int packed_x1, double gain1);

// would the fused 3-D level kernels take this level? (fused3d.hip; used by the whole-transform plan)
bool dtcwt_fwd3_level1_ok(int64_t n0, int64_t n1, int64_t n2, int m0, int m1, const double *h0o = nullptr, const double *h1o = nullptr);
bool dtcwt_fwd3_level2_ok(int64_t n0, int64_t n1, int64_t n2, int pad0, int pad1, int pad2, int m);
bool dtcwt_inv3_level1_ok(int64_t n0, int64_t n1, int64_t n2, int m0, int m1, const double *g0o = nullptr, const double *g1o = nullptr);
bool dtcwt_inv3_level2_ok(int64_t n0, int64_t n1, int64_t n2, int crop0, int m);
