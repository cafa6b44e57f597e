// Generic (any length, any axis, f32/f64) column filters and the quad/cube <-> complex
// packings of the DT-CWT.  These are the building blocks behind the public low-level
// functions (colfilter / coldfilt / colifilt), the 1-D and 3-D transforms and the
// band-pass ("_bp") 2-D wavelets; the float32 2-D level loop has fused kernels of its own
// (fused2d.hip).
//
// One thread produces one output *group* along the filter axis: 1 sample (colfilter),
// the (A, B) pair 2i, 2i+1 (coldfilt) or the four phases 4j..4j+3 (colifilt).  Lanes run
// along whichever of the two other view dimensions is contiguous, so global loads of a
// wavefront are coalesced rows; tap re-reads are served by L1/L2.
//
// Index algebra: SURVEY.md Appendix A (verified against the reference by the oracle).
#include "common.hpp"

namespace {

template <typename T>
struct Taps {
    T a[DTCWT_HIP_MAX_TAPS];
    T b[DTCWT_HIP_MAX_TAPS];
};

struct Geo {
    int64_t outer, n, inner;
    int64_t xso, xsn, xsi;
    int64_t yso, ysn, ysi;
    int64_t L;        // logical input length n + pad_lo + pad_hi
    int64_t nout;     // logical output length
    int64_t ngroups;  // thread groups along the filter axis
    int32_t pad_lo, crop_lo;
    int64_t nwrite;   // nout - crop_lo - crop_hi
    int32_t inner_fast;
    int32_t accumulate;
    int32_t idx32;    // every extent and the thread count fit in 31 bits: 32-bit index decode
};

__device__ inline bool decode(const Geo &g, int64_t &o, int64_t &grp, int64_t &i) {
    int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = g.outer * g.ngroups * g.inner;
    if (id >= total) return false;
    if (g.idx32) {      // 64-bit integer division costs hundreds of instructions on gfx950
        uint32_t u = (uint32_t)id, inner = (uint32_t)g.inner, ng = (uint32_t)g.ngroups;
        if (g.inner_fast) {
            uint32_t t = u / inner;
            i = u - t * inner;
            uint32_t oo = t / ng;
            grp = t - oo * ng;
            o = oo;
        } else {
            uint32_t t = u / ng;
            grp = u - t * ng;
            uint32_t oo = t / inner;
            i = t - oo * inner;
            o = oo;
        }
        return true;
    }
    if (g.inner_fast) {
        i = id % g.inner;
        int64_t t = id / g.inner;
        grp = t % g.ngroups;
        o = t / g.ngroups;
    } else {
        grp = id % g.ngroups;
        int64_t t = id / g.ngroups;
        i = t % g.inner;
        o = t / g.inner;
    }
    return true;
}

// logical sample u (any integer) -> real input sample.  Reflection by repeated bouncing:
// one iteration for every ordinary signal (|overshoot| < L), a few for signals shorter than
// the filter; never an integer modulo.
__device__ inline int64_t src_index(const Geo &g, int64_t u) {
    int r = (int)u;
    const int L = (int)g.L;
    while ((unsigned)r >= (unsigned)L) r = r < 0 ? -1 - r : 2 * L - 1 - r;
    r -= g.pad_lo;
    const int n1 = (int)g.n - 1;
    return r < 0 ? 0 : (r > n1 ? n1 : r);
}

template <typename T>
__device__ inline void put(const Geo &g, T *Yb, int64_t lo, T v) {
    int64_t w = lo - g.crop_lo;
    if (w < 0 || w >= g.nwrite) return;
    T *p = Yb + w * g.ysn;
    *p = g.accumulate ? (*p + v) : v;
}

// Y[i] = sum_k h[k] X[rho(i + m-1-k - m/2)]          (dtcwt/numpy/lowlevel.py:47-80)
template <typename T>
__global__ void __launch_bounds__(256) k_colfilter(const T *__restrict__ X, T *__restrict__ Y,
                                                   Geo g, Taps<T> taps, int m) {
    int64_t o, i, grp;
    if (!decode(g, o, grp, i)) return;
    const T *Xb = X + o * g.xso + i * g.xsi;
    T *Yb = Y + o * g.yso + i * g.ysi;
    int64_t lo = grp + g.crop_lo;     // only the written range is launched
    int64_t base = lo + (m - 1) - (m / 2);
    T acc = 0;
    for (int k = 0; k < m; ++k) acc += taps.a[k] * Xb[src_index(g, base - k) * g.xsn];
    put(g, Yb, lo, acc);
}

// ---- marching variants ----------------------------------------------------------------
// When the lanes run along a contiguous inner dimension and the filter axis is strided
// (the orientation of the public colfilter / coldfilt: "down the columns"), one thread
// produces G consecutive outputs from a sliding register window: (G + MB - 1) / G loads
// per output instead of m, every load a coalesced row.  MB is the compile-time tap bucket
// (taps zero padded to it); the reflection is the branch-free one-bounce form, valid
// because the launcher only takes this path when the window reach is shorter than the
// signal (anything else: the one-output-per-thread kernels above).
__device__ inline int64_t src_index1(const Geo &g, int u) {
    const int L = (int)g.L;
    u = u < 0 ? -1 - u : u;
    u = u >= L ? 2 * L - 1 - u : u;
    u -= g.pad_lo;
    const int n1 = (int)g.n - 1;
    return u < 0 ? 0 : (u > n1 ? n1 : u);
}

// colfilter: Y[lo] = sum_k h[k] X[rho(lo + c - k)], c = (m-1) - m/2
template <typename T, int G, int MB>
__global__ void __launch_bounds__(256) k_colfilter_march(const T *__restrict__ X, T *__restrict__ Y, Geo g,
                                                         Taps<T> taps, int m) {
    int64_t o, i, grp;
    if (!decode(g, o, grp, i)) return;
    const T *Xb = X + o * g.xso + i * g.xsi;
    T *Yb = Y + o * g.yso + i * g.ysi;
    const int lo0 = (int)grp * G + g.crop_lo;
    const int u0 = lo0 + (m - 1) - (m / 2) - (MB - 1);
    T w[G + MB - 1];
#pragma unroll
    for (int j = 0; j < G + MB - 1; ++j) w[j] = Xb[src_index1(g, u0 + j) * g.xsn];
#pragma unroll
    for (int q = 0; q < G; ++q) {
        T acc = 0;
#pragma unroll
        for (int k = 0; k < MB; ++k) acc += taps.a[k] * w[q + MB - 1 - k];
        put(g, Yb, lo0 + q, acc);
    }
}

// coldfilt: pair i reads the window starting at 4i - m + 2; taps.a/b hold ha/hb FRONT padded
// to MB (the launcher shifts them), which keeps the register indices compile-time:
//   A = sum_k a[2k] w[4q + 2MB-2-4k] + a[2k+1] w[4q + 2MB-4-4k],  B: +1 on both indices
template <typename T, int GP, int MB>
__global__ void __launch_bounds__(256) k_coldfilt_march(const T *__restrict__ X, T *__restrict__ Y, Geo g,
                                                        Taps<T> taps, int m, int a_first) {
    int64_t o, i, grp;
    if (!decode(g, o, grp, i)) return;
    const T *Xb = X + o * g.xso + i * g.xsi;
    T *Yb = Y + o * g.yso + i * g.ysi;
    const int i0 = (int)grp * GP;
    const int u0 = 4 * i0 - m + 2;
    constexpr int WN = 4 * (GP - 1) + 2 * MB;
    T w[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) w[j] = Xb[src_index1(g, u0 + j) * g.xsn];
#pragma unroll
    for (int q = 0; q < GP; ++q) {
        T A = 0, B = 0;
#pragma unroll
        for (int k = 0; k < MB / 2; ++k) {
            A += taps.a[2 * k] * w[4 * q + 2 * MB - 2 - 4 * k];
            A += taps.a[2 * k + 1] * w[4 * q + 2 * MB - 4 - 4 * k];
            B += taps.b[2 * k] * w[4 * q + 2 * MB - 1 - 4 * k];
            B += taps.b[2 * k + 1] * w[4 * q + 2 * MB - 3 - 4 * k];
        }
        put(g, Yb, 2 * (i0 + q), a_first ? A : B);
        put(g, Yb, 2 * (i0 + q) + 1, a_first ? B : A);
    }
}

// colifilt: input pair j (samples 2j, 2j+1) gives outputs 4j .. 4j+3 from the window starting
// at 2j + ORG (ORG = 1 - m/2 for odd m/2, -m/2 for even).  Zero padding ha / hb by the same
// EVEN number of taps on both sides leaves every output unchanged (all indices of the
// polyphase sums shift together), so m is padded to the bucket MB of its m/2 parity and the
// register indices stay compile-time.  GJ input pairs per thread.
template <typename T, int GJ, int MB>
__global__ void __launch_bounds__(256) k_colifilt_march(const T *__restrict__ X, T *__restrict__ Y, Geo g,
                                                        Taps<T> taps, int pos) {
    constexpr int M2 = MB / 2;
    constexpr bool ODD = (M2 % 2) == 1;
    constexpr int WN = ODD ? MB : MB + 2;
    constexpr int ORG = ODD ? 1 - M2 : -M2;
    int64_t o, i, grp;
    if (!decode(g, o, grp, i)) return;
    const T *Xb = X + o * g.xso + i * g.xsi;
    T *Yb = Y + o * g.yso + i * g.ysi;
    const int j0 = (int)grp * GJ;
    T w[WN + 2 * (GJ - 1)];
#pragma unroll
    for (int j = 0; j < WN + 2 * (GJ - 1); ++j) w[j] = Xb[src_index1(g, 2 * j0 + ORG + j) * g.xsn];
#pragma unroll
    for (int q = 0; q < GJ; ++q) {
        const T *v = w + 2 * q;
        T y0 = 0, y1 = 0, y2 = 0, y3 = 0;
        if (ODD) {
#pragma unroll
            for (int k = 0; k < M2; ++k) {
                const T hi = v[MB - 1 - 2 * k], lo = v[MB - 2 - 2 * k];
                const T xa = pos ? hi : lo, xb = pos ? lo : hi;
                y0 += taps.a[2 * k] * xb; y1 += taps.b[2 * k] * xa;
                y2 += taps.a[2 * k + 1] * xb; y3 += taps.b[2 * k + 1] * xa;
            }
        } else {
#pragma unroll
            for (int k = 0; k < M2; ++k) {
                const T t0 = v[MB + 1 - 2 * k], t1 = v[MB - 2 * k], t2 = v[MB - 1 - 2 * k], t3 = v[MB - 2 - 2 * k];
                const T xa = pos ? t0 : t1, xb = pos ? t1 : t0, xa2 = pos ? t2 : t3, xb2 = pos ? t3 : t2;
                y0 += taps.a[2 * k + 1] * xb2; y1 += taps.b[2 * k + 1] * xa2;
                y2 += taps.a[2 * k] * xb; y3 += taps.b[2 * k] * xa;
            }
        }
        put(g, Yb, 4 * (j0 + q), y0); put(g, Yb, 4 * (j0 + q) + 1, y1);
        put(g, Yb, 4 * (j0 + q) + 2, y2); put(g, Yb, 4 * (j0 + q) + 3, y3);
    }
}

// ---- LDS-row variants: the filter axis IS the contiguous one --------------------------------------------------------
// "Along the rows" (axis_colfilter(X, h, axis=1); the reference transposes, filters down the columns and transposes back:
// dtcwt/numpy/transform2d.py:112-130).  With one output group per thread and every tap a load of its own the three filters
// ran at 1.2-2.6 TB/s on a 4096^2 image against 3.8-4.4 down the columns (profiles/r01/lowlevel_d.txt): every input sample
// was requested m times.  Here a workgroup stages the segment of the row its 256 threads need into LDS once -- coalesced
// loads, the symmetric extension applied on the way in -- and a thread takes the window of its FOUR consecutive outputs
// (eight for colifilt) from there as 16-byte LDS reads, the register indices compile-time exactly as in the marching
// kernels above (same tap buckets, same padding rules), and writes them as one 16-byte store (VEC: rows and crops in
// fours, no accumulation).  blockIdx = (row, segment of the row).
template <typename T, bool VEC>
__device__ inline void put4(const Geo &g, T *Yb, int lo, const T (&v)[4]) {
    const int w = lo - g.crop_lo;
    if (VEC) {
        if (w >= 0 && w + 3 < g.nwrite) {
            if (sizeof(T) == 4) {
                *reinterpret_cast<float4 *>(Yb + w) = float4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
            } else {
                *reinterpret_cast<double2 *>(Yb + w) = double2{(double)v[0], (double)v[1]};
                *reinterpret_cast<double2 *>(Yb + w + 2) = double2{(double)v[2], (double)v[3]};
            }
            return;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) put(g, Yb, (int64_t)lo + q, v[q]);
}

template <typename T, int MB, bool VEC>
__global__ void __launch_bounds__(256) k_colfilter_rows(const T *__restrict__ X, T *__restrict__ Y, Geo g,
                                                        Taps<T> taps, int m, int nseg) {
    constexpr int G = 4, SEG = 256 * G, W = SEG + MB - 1;
    __shared__ __attribute__((aligned(16))) T s[(W + 3) & ~3];
    const int seg = blockIdx.x % nseg;
    const int64_t row = blockIdx.x / nseg;
    const T *Xb = X + row * g.xso;
    T *Yb = Y + row * g.yso;
    const int lo0 = seg * SEG + g.crop_lo;
    const int u0 = lo0 + (m - 1) - (m / 2) - (MB - 1);
    for (int j = threadIdx.x; j < W; j += 256) s[j] = Xb[src_index1(g, u0 + j)];
    __syncthreads();
    T w[G + MB - 1];
#pragma unroll
    for (int j = 0; j < G + MB - 1; ++j) w[j] = s[G * threadIdx.x + j];
    T v[G];
#pragma unroll
    for (int q = 0; q < G; ++q) {
        T acc = 0;
#pragma unroll
        for (int k = 0; k < MB; ++k) acc += taps.a[k] * w[q + MB - 1 - k];
        v[q] = acc;
    }
    put4<T, VEC>(g, Yb, lo0 + G * (int)threadIdx.x, v);
}

// taps front padded to MB (k_coldfilt_march); two (A, B) pairs = four outputs per thread
template <typename T, int MB, bool VEC>
__global__ void __launch_bounds__(256) k_coldfilt_rows(const T *__restrict__ X, T *__restrict__ Y, Geo g,
                                                       Taps<T> taps, int m, int a_first, int nseg) {
    constexpr int GP = 2, SEGP = 256 * GP, W = 4 * (SEGP - 1) + 2 * MB;
    __shared__ __attribute__((aligned(16))) T s[(W + 3) & ~3];
    const int seg = blockIdx.x % nseg;
    const int64_t row = blockIdx.x / nseg;
    const T *Xb = X + row * g.xso;
    T *Yb = Y + row * g.yso;
    const int i0w = seg * SEGP;
    const int u0 = 4 * i0w - m + 2;
    for (int j = threadIdx.x; j < W; j += 256) s[j] = Xb[src_index1(g, u0 + j)];
    __syncthreads();
    constexpr int WN = 4 * (GP - 1) + 2 * MB;
    T w[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) w[j] = s[4 * GP * threadIdx.x + j];
    T v[4];
#pragma unroll
    for (int q = 0; q < GP; ++q) {
        T A = 0, B = 0;
#pragma unroll
        for (int k = 0; k < MB / 2; ++k) {
            A += taps.a[2 * k] * w[4 * q + 2 * MB - 2 - 4 * k];
            A += taps.a[2 * k + 1] * w[4 * q + 2 * MB - 4 - 4 * k];
            B += taps.b[2 * k] * w[4 * q + 2 * MB - 1 - 4 * k];
            B += taps.b[2 * k + 1] * w[4 * q + 2 * MB - 3 - 4 * k];
        }
        v[2 * q] = a_first ? A : B; v[2 * q + 1] = a_first ? B : A;
    }
    const int i0 = i0w + GP * (int)threadIdx.x;
    if (2 * i0 < g.nout) put4<T, VEC>(g, Yb, 2 * i0, v);
}

// taps padded to the bucket of their m / 2 parity (k_colifilt_march); two input pairs = eight outputs per thread
template <typename T, int MB, bool VEC>
__global__ void __launch_bounds__(256) k_colifilt_rows(const T *__restrict__ X, T *__restrict__ Y, Geo g,
                                                       Taps<T> taps, int pos, int nseg) {
    constexpr int M2 = MB / 2;
    constexpr bool ODD = (M2 % 2) == 1;
    constexpr int WN = ODD ? MB : MB + 2;
    constexpr int ORG = ODD ? 1 - M2 : -M2;
    constexpr int GJ = 2, SEGJ = 256 * GJ, W = WN + 2 * (SEGJ - 1);
    __shared__ __attribute__((aligned(16))) T s[(W + 3) & ~3];
    const int seg = blockIdx.x % nseg;
    const int64_t row = blockIdx.x / nseg;
    const T *Xb = X + row * g.xso;
    T *Yb = Y + row * g.yso;
    const int j0w = seg * SEGJ;
    for (int j = threadIdx.x; j < W; j += 256) s[j] = Xb[src_index1(g, 2 * j0w + ORG + j)];
    __syncthreads();
    T w[WN + 2 * (GJ - 1)];
#pragma unroll
    for (int j = 0; j < WN + 2 * (GJ - 1); ++j) w[j] = s[2 * GJ * threadIdx.x + j];
    const int j0 = j0w + GJ * (int)threadIdx.x;
#pragma unroll
    for (int q = 0; q < GJ; ++q) {
        const T *v = w + 2 * q;
        T y[4] = {0, 0, 0, 0};
        if (ODD) {
#pragma unroll
            for (int k = 0; k < M2; ++k) {
                const T hi = v[MB - 1 - 2 * k], lo = v[MB - 2 - 2 * k];
                const T xa = pos ? hi : lo, xb = pos ? lo : hi;
                y[0] += taps.a[2 * k] * xb; y[1] += taps.b[2 * k] * xa;
                y[2] += taps.a[2 * k + 1] * xb; y[3] += taps.b[2 * k + 1] * xa;
            }
        } else {
#pragma unroll
            for (int k = 0; k < M2; ++k) {
                const T t0 = v[MB + 1 - 2 * k], t1 = v[MB - 2 * k], t2 = v[MB - 1 - 2 * k], t3 = v[MB - 2 - 2 * k];
                const T xa = pos ? t0 : t1, xb = pos ? t1 : t0, xa2 = pos ? t2 : t3, xb2 = pos ? t3 : t2;
                y[0] += taps.a[2 * k + 1] * xb2; y[1] += taps.b[2 * k + 1] * xa2;
                y[2] += taps.a[2 * k] * xb; y[3] += taps.b[2 * k] * xa;
            }
        }
        if (4 * (j0 + q) < g.nout) put4<T, VEC>(g, Yb, 4 * (j0 + q), y);
    }
}

// Two undecimated filters of the same length parity applied to the same input in one pass
// (the 3-D level loops always filter a volume with the lo AND the hi filter,
// dtcwt/numpy/transform3d.py:256-273): every input sample is loaded once and feeds both.
template <typename T>
__global__ void __launch_bounds__(256) k_colfilter2(const T *__restrict__ X, T *__restrict__ Y0,
                                                    T *__restrict__ Y1, Geo g, Taps<T> taps, int m0, int m1) {
    int64_t o, i, grp;
    if (!decode(g, o, grp, i)) return;
    const T *Xb = X + o * g.xso + i * g.xsi;
    int64_t lo = grp + g.crop_lo;
    // Y[i] = sum_k h[k] X[i + (m-1-m/2) - k]: offsets d = (m-1-m/2) - k
    const int c0 = (m0 - 1) - m0 / 2, c1 = (m1 - 1) - m1 / 2;
    const int dmin = (c0 - (m0 - 1)) < (c1 - (m1 - 1)) ? (c0 - (m0 - 1)) : (c1 - (m1 - 1));
    const int dmax = c0 > c1 ? c0 : c1;
    T a0 = 0, a1 = 0;
    for (int d = dmax; d >= dmin; --d) {
        T x = Xb[src_index(g, lo + d) * g.xsn];
        int k0 = c0 - d, k1 = c1 - d;
        if (k0 >= 0 && k0 < m0) a0 += taps.a[k0] * x;
        if (k1 >= 0 && k1 < m1) a1 += taps.b[k1] * x;
    }
    put(g, Y0 + o * g.yso + i * g.ysi, lo, a0);
    put(g, Y1 + o * g.yso + i * g.ysi, lo, a1);
}

// y = colfilter(X0, h0) + colfilter(X1, h1) in one pass (the 3-D inverse merges,
// transform3d.py:425-435): no read-modify-write of the output.
template <typename T>
__global__ void __launch_bounds__(256) k_colfilter_sum2(const T *__restrict__ X0, const T *__restrict__ X1,
                                                        T *__restrict__ Y, Geo g, Taps<T> taps, int m0, int m1) {
    int64_t o, i, grp;
    if (!decode(g, o, grp, i)) return;
    const T *P0 = X0 + o * g.xso + i * g.xsi, *P1 = X1 + o * g.xso + i * g.xsi;
    int64_t lo = grp + g.crop_lo;
    T acc = 0;
    int64_t b0 = lo + (m0 - 1) - m0 / 2, b1 = lo + (m1 - 1) - m1 / 2;
    for (int k = 0; k < m0; ++k) acc += taps.a[k] * P0[src_index(g, b0 - k) * g.xsn];
    for (int k = 0; k < m1; ++k) acc += taps.b[k] * P1[src_index(g, b1 - k) * g.xsn];
    put(g, Y + o * g.yso + i * g.ysi, lo, acc);
}

// Two dual-tree decimating filter pairs on the same input in one pass (taps.a/b = pair 0,
// taps2.a/b = pair 1), cf. k_coldfilt.
template <typename T>
__global__ void __launch_bounds__(256) k_coldfilt2(const T *__restrict__ X, T *__restrict__ Y0,
                                                   T *__restrict__ Y1, Geo g, Taps<T> taps, Taps<T> taps2,
                                                   int m, int a_first0, int a_first1) {
    int64_t o, i, grp;
    if (!decode(g, o, grp, i)) return;
    const T *Xb = X + o * g.xso + i * g.xsi;
    int p = m / 2;
    T A0 = 0, B0 = 0, A1 = 0, B1 = 0;
    for (int k = 0; k < p; ++k) {
        int64_t b = 4 * (grp + p - 1 - k) - m;
        T x4 = Xb[src_index(g, b + 4) * g.xsn], x2 = Xb[src_index(g, b + 2) * g.xsn];
        T x5 = Xb[src_index(g, b + 5) * g.xsn], x3 = Xb[src_index(g, b + 3) * g.xsn];
        A0 += taps.a[2 * k] * x4; A0 += taps.a[2 * k + 1] * x2;
        B0 += taps.b[2 * k] * x5; B0 += taps.b[2 * k + 1] * x3;
        A1 += taps2.a[2 * k] * x4; A1 += taps2.a[2 * k + 1] * x2;
        B1 += taps2.b[2 * k] * x5; B1 += taps2.b[2 * k + 1] * x3;
    }
    T *P0 = Y0 + o * g.yso + i * g.ysi, *P1 = Y1 + o * g.yso + i * g.ysi;
    put(g, P0, 2 * grp, a_first0 ? A0 : B0); put(g, P0, 2 * grp + 1, a_first0 ? B0 : A0);
    put(g, P1, 2 * grp, a_first1 ? A1 : B1); put(g, P1, 2 * grp + 1, a_first1 ? B1 : A1);
}

// y = colifilt(X0, pair 0) + colifilt(X1, pair 1) in one pass (transform3d.py:485-495)
template <typename T>
__global__ void __launch_bounds__(256) k_colifilt_sum2(const T *__restrict__ X0, const T *__restrict__ X1,
                                                       T *__restrict__ Y, Geo g, Taps<T> taps, Taps<T> taps2,
                                                       int m, int pos0, int pos1) {
    int64_t o, i, grp;
    if (!decode(g, o, grp, i)) return;
    int m2 = m / 2, n = m2;
    T y0 = 0, y1 = 0, y2 = 0, y3 = 0;
    for (int s = 0; s < 2; ++s) {
        const T *Xb = (s ? X1 : X0) + o * g.xso + i * g.xsi;
        const Taps<T> &tp = s ? taps2 : taps;
        const int pos = s ? pos1 : pos0;
        if ((m2 & 1) == 0) {
            for (int k = 0; k < n; ++k) {
                int64_t t = 3 + 2 * (grp + n - 1 - k);
                int64_t ta = pos ? t : t - 1, tb = pos ? t - 1 : t;
                y0 += tp.a[2 * k + 1] * Xb[src_index(g, tb - 2 - m2) * g.xsn];
                y1 += tp.b[2 * k + 1] * Xb[src_index(g, ta - 2 - m2) * g.xsn];
                y2 += tp.a[2 * k] * Xb[src_index(g, tb - m2) * g.xsn];
                y3 += tp.b[2 * k] * Xb[src_index(g, ta - m2) * g.xsn];
            }
        } else {
            for (int k = 0; k < n; ++k) {
                int64_t t = 2 + 2 * (grp + n - 1 - k);
                int64_t ta = pos ? t : t - 1, tb = pos ? t - 1 : t;
                T xb = Xb[src_index(g, tb - m2) * g.xsn];
                T xa = Xb[src_index(g, ta - m2) * g.xsn];
                y0 += tp.a[2 * k] * xb; y1 += tp.b[2 * k] * xa;
                y2 += tp.a[2 * k + 1] * xb; y3 += tp.b[2 * k + 1] * xa;
            }
        }
    }
    T *Yb = Y + o * g.yso + i * g.ysi;
    put(g, Yb, 4 * grp, y0); put(g, Yb, 4 * grp + 1, y1);
    put(g, Yb, 4 * grp + 2, y2); put(g, Yb, 4 * grp + 3, y3);
}

// coldfilt (dtcwt/numpy/lowlevel.py:82-154): with p = m/2, b_k = 4(i+p-1-k) - m
//   A[i] = sum_k ha[2k] X[rho(b_k+4)] + ha[2k+1] X[rho(b_k+2)]
//   B[i] = sum_k hb[2k] X[rho(b_k+5)] + hb[2k+1] X[rho(b_k+3)]
// (Y[2i], Y[2i+1]) = (A, B) if sum(ha*hb) > 0 else (B, A).
template <typename T>
__global__ void __launch_bounds__(256) k_coldfilt(const T *__restrict__ X, T *__restrict__ Y,
                                                  Geo g, Taps<T> taps, int m, int a_first) {
    int64_t o, i, grp;
    if (!decode(g, o, grp, i)) return;
    const T *Xb = X + o * g.xso + i * g.xsi;
    T *Yb = Y + o * g.yso + i * g.ysi;
    int p = m / 2;
    T A = 0, B = 0;
    for (int k = 0; k < p; ++k) {
        int64_t b = 4 * (grp + p - 1 - k) - m;
        A += taps.a[2 * k] * Xb[src_index(g, b + 4) * g.xsn];
        A += taps.a[2 * k + 1] * Xb[src_index(g, b + 2) * g.xsn];
        B += taps.b[2 * k] * Xb[src_index(g, b + 5) * g.xsn];
        B += taps.b[2 * k + 1] * Xb[src_index(g, b + 3) * g.xsn];
    }
    put(g, Yb, 2 * grp, a_first ? A : B);
    put(g, Yb, 2 * grp + 1, a_first ? B : A);
}

// colifilt (dtcwt/numpy/lowlevel.py:156-260), n = m/2 taps per polyphase branch,
// jj = j+n-1-k:
//   m/2 even: t = 3+2jj, (ta,tb) = pos ? (t,t-1) : (t-1,t)
//      Y[4j]   += hae[k] X[rho(tb-2-m2)]   Y[4j+1] += hbe[k] X[rho(ta-2-m2)]
//      Y[4j+2] += hao[k] X[rho(tb-m2)]     Y[4j+3] += hbo[k] X[rho(ta-m2)]
//   m/2 odd:  t = 2+2jj
//      Y[4j]   += hao[k] X[rho(tb-m2)]     Y[4j+1] += hbo[k] X[rho(ta-m2)]
//      Y[4j+2] += hae[k] X[rho(tb-m2)]     Y[4j+3] += hbe[k] X[rho(ta-m2)]
template <typename T>
__global__ void __launch_bounds__(256) k_colifilt(const T *__restrict__ X, T *__restrict__ Y,
                                                  Geo g, Taps<T> taps, int m, int pos) {
    int64_t o, i, grp;
    if (!decode(g, o, grp, i)) return;
    const T *Xb = X + o * g.xso + i * g.xsi;
    T *Yb = Y + o * g.yso + i * g.ysi;
    int m2 = m / 2, n = m2;
    T y0 = 0, y1 = 0, y2 = 0, y3 = 0;
    if ((m2 & 1) == 0) {
        for (int k = 0; k < n; ++k) {
            int64_t t = 3 + 2 * (grp + n - 1 - k);
            int64_t ta = pos ? t : t - 1, tb = pos ? t - 1 : t;
            T hao = taps.a[2 * k], hae = taps.a[2 * k + 1];
            T hbo = taps.b[2 * k], hbe = taps.b[2 * k + 1];
            y0 += hae * Xb[src_index(g, tb - 2 - m2) * g.xsn];
            y1 += hbe * Xb[src_index(g, ta - 2 - m2) * g.xsn];
            y2 += hao * Xb[src_index(g, tb - m2) * g.xsn];
            y3 += hbo * Xb[src_index(g, ta - m2) * g.xsn];
        }
    } else {
        for (int k = 0; k < n; ++k) {
            int64_t t = 2 + 2 * (grp + n - 1 - k);
            int64_t ta = pos ? t : t - 1, tb = pos ? t - 1 : t;
            T xb = Xb[src_index(g, tb - m2) * g.xsn];
            T xa = Xb[src_index(g, ta - m2) * g.xsn];
            y0 += taps.a[2 * k] * xb;
            y1 += taps.b[2 * k] * xa;
            y2 += taps.a[2 * k + 1] * xb;
            y3 += taps.b[2 * k + 1] * xa;
        }
    }
    put(g, Yb, 4 * grp, y0);
    put(g, Yb, 4 * grp + 1, y1);
    put(g, Yb, 4 * grp + 2, y2);
    put(g, Yb, 4 * grp + 3, y3);
}

// q2c (dtcwt/numpy/transform2d.py:301-322): quad (a b / c d):
//   z0 = s((a-d) + j(b+c)),  z1 = s((a+d) + j(b-c)),  s = sqrt(1/2)
template <typename T>
__global__ void __launch_bounds__(256) k_q2c(const T *__restrict__ y, int64_t batch, int64_t R2,
                                             int64_t C2, int64_t sb, int64_t sr,
                                             T *__restrict__ Yh, int slot0, int slot1) {
    int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= batch * R2 * C2) return;
    int64_t v = id % C2, u = (id / C2) % R2, b = id / (C2 * R2);
    const T *q = y + b * sb + (2 * u) * sr + 2 * v;
    T a = q[0], bb = q[1], c = q[sr], d = q[sr + 1];
    const T s = (T)0.70710678118654752440;
    T *rec = Yh + id * 12;
    rec[2 * slot0] = s * (a - d);
    rec[2 * slot0 + 1] = s * (bb + c);
    rec[2 * slot1] = s * (a + d);
    rec[2 * slot1 + 1] = s * (bb - c);
}

// c2q (dtcwt/numpy/transform2d.py:324-350): P = s(g0 w0 + g1 w1), Q = s(g0 w0 - g1 w1);
//   a = Re P, b = Im P, c = Im Q, d = -Re Q.
template <typename T>
__global__ void __launch_bounds__(256) k_c2q(const T *__restrict__ Yh, int64_t batch, int64_t R,
                                             int64_t C, int slot0, int slot1, T g0, T g1,
                                             T *__restrict__ x, int64_t sb, int64_t sr) {
    int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= batch * R * C) return;
    int64_t v = id % C, u = (id / C) % R, b = id / (C * R);
    const T *rec = Yh + id * 12;
    T w0r = rec[2 * slot0] * g0, w0i = rec[2 * slot0 + 1] * g0;
    T w1r = rec[2 * slot1] * g1, w1i = rec[2 * slot1 + 1] * g1;
    T *q = x + b * sb + (2 * u) * sr + 2 * v;
    q[0] = w0r + w1r;
    q[1] = w0i + w1i;
    q[sr] = w0i - w1i;
    q[sr + 1] = -(w0r - w1r);
}

// cube2c (dtcwt/numpy/transform3d.py:532-579)
template <typename T>
__global__ void __launch_bounds__(256) k_cube2c(const T *__restrict__ y, int64_t e0, int64_t e1,
                                                int64_t e2, int64_t s0, int64_t s1,
                                                T *__restrict__ Yh, int octant) {
    int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= e0 * e1 * e2) return;
    int64_t w = id % e2, v = (id / e2) % e1, u = id / (e2 * e1);
    const T *q = y + (2 * u) * s0 + (2 * v) * s1 + 2 * w;
    T A = q[0], B = q[s1], C = q[s0], D = q[s0 + s1];
    T E = q[1], F = q[s1 + 1], G = q[s0 + 1], H = q[s0 + s1 + 1];
    const T h = (T)0.5;
    T *rec = Yh + id * 56 + octant * 8;
    rec[0] = (A - G - D - F) * h;  rec[1] = (B - H + C + E) * h;
    rec[2] = (A - G + D + F) * h;  rec[3] = (-B + H + C + E) * h;
    rec[4] = (A + G + D - F) * h;  rec[5] = (B + H - C + E) * h;
    rec[6] = (A + G - D + F) * h;  rec[7] = (-B - H - C + E) * h;
}

// c2cube (dtcwt/numpy/transform3d.py:581-619)
template <typename T>
__global__ void __launch_bounds__(256) k_c2cube(const T *__restrict__ Yh, int64_t e0, int64_t e1,
                                                int64_t e2, int octant, T *__restrict__ y,
                                                int64_t s0, int64_t s1) {
    int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= e0 * e1 * e2) return;
    int64_t w = id % e2, v = (id / e2) % e1, u = id / (e2 * e1);
    const T *rec = Yh + id * 56 + octant * 8;
    T pr = rec[0], pi = rec[1], qr = rec[2], qi = rec[3];
    T rr = rec[4], ri = rec[5], sr = rec[6], si = rec[7];
    const T h = (T)0.5;
    T *q = y + (2 * u) * s0 + (2 * v) * s1 + 2 * w;
    q[0] = (pr + qr + rr + sr) * h;               // A  (0,0,0)
    q[s0 + 1] = (-pr - qr + rr + sr) * h;         // G  (1,0,1)
    q[s0 + s1] = (-pr + qr + rr - sr) * h;        // D  (1,1,0)
    q[s1 + 1] = (-pr + qr - rr + sr) * h;         // F  (0,1,1)
    q[s1] = (pi - qi + ri - si) * h;              // B  (0,1,0)
    q[s0 + s1 + 1] = (-pi + qi + ri - si) * h;    // H  (1,1,1)
    q[s0] = (pi + qi - ri - si) * h;              // C  (1,0,0)
    q[1] = (pi + qi + ri + si) * h;               // E  (0,0,1)
}

template <typename T>
__global__ void __launch_bounds__(256) k_scale(T *x, int64_t n, T g) {
    int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id < n) x[id] *= g;
}

// 1-D packing: Yh[j] = Hi[2j] + i Hi[2j+1]   (dtcwt/numpy/transform1d.py:88,100)
template <typename T>
__global__ void __launch_bounds__(256) k_pack1d(const T *__restrict__ hi, int64_t J, int64_t k,
                                                T *__restrict__ Yh) {
    int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= J * k) return;
    int64_t c = id % k, j = id / k;
    Yh[2 * id] = hi[(2 * j) * k + c];
    Yh[2 * id + 1] = hi[(2 * j + 1) * k + c];
}

// c2q1d with gain: Hi[2j] = g Re Yh[j], Hi[2j+1] = g Im Yh[j]   (transform1d.py:153,171,186-196)
template <typename T>
__global__ void __launch_bounds__(256) k_unpack1d(const T *__restrict__ Yh, int64_t J, int64_t k,
                                                  T g, T *__restrict__ hi) {
    int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= J * k) return;
    int64_t c = id % k, j = id / k;
    hi[(2 * j) * k + c] = g * Yh[2 * id];
    hi[(2 * j + 1) * k + c] = g * Yh[2 * id + 1];
}

int make_geo(const dtcwt_hip_view *v, int64_t nout_of_L(int64_t, int), int m, int group,
             int flags, Geo &g) {
    DT_REQUIRE(v, "view is NULL");
    DT_REQUIRE(v->outer >= 0 && v->n >= 1 && v->inner >= 0, "bad view extents");
    DT_REQUIRE(v->pad_lo >= 0 && v->pad_hi >= 0 && v->crop_lo >= 0 && v->crop_hi >= 0,
               "negative pad/crop");
    g.outer = v->outer; g.n = v->n; g.inner = v->inner;
    g.xso = v->xso; g.xsn = v->xsn; g.xsi = v->xsi;
    g.yso = v->yso; g.ysn = v->ysn; g.ysi = v->ysi;
    g.L = v->n + v->pad_lo + v->pad_hi;
    g.nout = nout_of_L(g.L, m);
    g.pad_lo = v->pad_lo;
    g.crop_lo = v->crop_lo;
    g.nwrite = g.nout - v->crop_lo - v->crop_hi;
    DT_REQUIRE(g.nwrite >= 0, "crop exceeds output length");
    g.ngroups = group == 1 ? g.nwrite : g.nout / group;
    g.inner_fast = (v->xsi == 1 && v->ysi == 1) || v->inner == 1 ? 1 : 0;
    if (v->inner > 1 && v->xsn == 1 && v->ysn == 1) g.inner_fast = 0;
    g.accumulate = (flags & DTCWT_HIP_ACCUMULATE) ? 1 : 0;
    DT_REQUIRE(g.L < ((int64_t)1 << 30), "filter axis too long");
    int64_t total = g.outer * g.ngroups * g.inner;
    g.idx32 = (total < ((int64_t)1 << 31) && g.inner < ((int64_t)1 << 31) && g.ngroups < ((int64_t)1 << 31)) ? 1 : 0;
    return 0;
}

template <typename T>
void load_taps(Taps<T> &t, const double *a, const double *b, int m) {
    for (int k = 0; k < DTCWT_HIP_MAX_TAPS; ++k) {
        t.a[k] = (a && k < m) ? (T)a[k] : (T)0;
        t.b[k] = (b && k < m) ? (T)b[k] : (T)0;
    }
}

inline unsigned blocks_for(int64_t total) { return (unsigned)((total + 255) / 256); }

// The marching kernels: lanes along a contiguous inner dimension of at least a wavefront,
// 32-bit decode, and a signal longer than the window reach (one-bounce reflection).
bool march_ok(const Geo &g, int reach) {
    return g.inner_fast && g.inner >= 64 && g.idx32 && g.L >= reach && g.nwrite >= 16;
}
// Lanes along the filter axis itself (a contiguous axis): one output group per thread as in the
// generic kernels, but with the compile-time tap bucket -- unrolled taps, all loads of a
// thread issued up front, branch-free reflection.
bool unrolled_ok(const Geo &g, int reach) {
    return !(g.inner_fast && g.inner > 1) && g.idx32 && g.L >= reach && g.nwrite >= 16;
}

// The LDS-row kernels: the filter axis is the contiguous one on both sides (so inner == 1: rows = outer), one-bounce reflection,
// a row long enough to be worth a workgroup; `vec`: 16-byte stores (rows, crop and base address in fours / 16 bytes)
bool rows_ok(const Geo &g, int reach) {
    // (one workgroup per (row, segment of >= 512 outputs): the grid must fit 31 bits)
    return g.inner == 1 && g.xsn == 1 && g.ysn == 1 && g.idx32 && g.L >= reach && g.nwrite >= 256 &&
           (g.nout / 512 + 2) * g.outer < ((int64_t)1 << 31);
}
bool rows_vec(const Geo &g, const void *Y, size_t esz) {
    return !g.accumulate && g.crop_lo % 4 == 0 && g.yso % 4 == 0 && ((uintptr_t)Y % 16) == 0 && (esz == 4 || esz == 8);
}

// taps FRONT padded to mb entries (k_coldfilt_march)
template <typename T>
void load_taps_front(Taps<T> &t, const double *a, const double *b, int m, int mb) {
    for (int k = 0; k < DTCWT_HIP_MAX_TAPS; ++k) {
        const int s = k - (mb - m);
        t.a[k] = (s >= 0 && s < m) ? (T)a[s] : (T)0;
        t.b[k] = (s >= 0 && s < m) ? (T)b[s] : (T)0;
    }
}

// taps padded to mb entries, (mb - m) / 2 zeros on each side (k_colifilt_march)
template <typename T>
void load_taps_centred(Taps<T> &t, const double *a, const double *b, int m, int mb) {
    for (int k = 0; k < DTCWT_HIP_MAX_TAPS; ++k) {
        const int s = k - (mb - m) / 2;
        t.a[k] = (s >= 0 && s < m) ? (T)a[s] : (T)0;
        t.b[k] = (s >= 0 && s < m) ? (T)b[s] : (T)0;
    }
}

int64_t nout_colfilter(int64_t L, int m) { return (m & 1) ? L : L + 1; }
int64_t nout_coldfilt(int64_t L, int) { return L / 2; }
int64_t nout_colifilt(int64_t L, int) { return 2 * L; }

double dot(const double *a, const double *b, int m) {
    double s = 0;
    for (int k = 0; k < m; ++k) s += a[k] * b[k];
    return s;
}

}  // namespace

#define DT_LAUNCH_CHECK() DT_CHECK_HIP(hipGetLastError())

// A view that is a dense [outer][n][inner] array on both sides (what the transforms pass): the
// pair / sum entry points try the marching and LDS-row kernels of generic2d.hip first.
static bool dense_view(const dtcwt_hip_view *v, int64_t nwrite) {
    return v->xsi == 1 && v->xsn == v->inner && v->xso == v->n * v->inner && v->ysi == 1 &&
           v->ysn == v->inner && v->yso == nwrite * v->inner;
}

extern "C" {

int dtcwt_hip_colfilter(dtcwt_hip_ctx *ctx, int dtype, const void *X, void *Y,
                        const dtcwt_hip_view *v, const double *h, int m, int flags) {
    DT_REQUIRE(ctx && X && Y && h, "NULL argument");
    DT_REQUIRE(m >= 1 && m <= DTCWT_HIP_MAX_TAPS, "filter length %d not in [1, %d]", m,
               DTCWT_HIP_MAX_TAPS);
    DT_REQUIRE(dtype == DTCWT_HIP_F32 || dtype == DTCWT_HIP_F64, "bad dtype %d", dtype);
    Geo g;
    int rc = make_geo(v, nout_colfilter, m, 1, flags, g);
    if (rc) return rc;
    int64_t total = g.outer * g.ngroups * g.inner;
    if (total == 0) return 0;
    DT_REQUIRE(total < ((int64_t)1 << 39), "problem too large");
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    if (march_ok(g, 8 + 20) && m <= 20) {
        // lanes along the contiguous inner dimension: 8 outputs per thread from a register window
        g.ngroups = (g.nwrite + 7) / 8;
        const int64_t tot = g.outer * g.ngroups * g.inner;
        if (dtype == DTCWT_HIP_F32) {
            Taps<float> t; load_taps(t, h, nullptr, m);
            if (m <= 8) k_colfilter_march<float, 8, 8><<<blocks_for(tot), 256, 0, ctx->stream>>>((const float *)X, (float *)Y, g, t, m);
            else k_colfilter_march<float, 8, 20><<<blocks_for(tot), 256, 0, ctx->stream>>>((const float *)X, (float *)Y, g, t, m);
        } else {
            Taps<double> t; load_taps(t, h, nullptr, m);
            if (m <= 8) k_colfilter_march<double, 8, 8><<<blocks_for(tot), 256, 0, ctx->stream>>>((const double *)X, (double *)Y, g, t, m);
            else k_colfilter_march<double, 8, 20><<<blocks_for(tot), 256, 0, ctx->stream>>>((const double *)X, (double *)Y, g, t, m);
        }
        DT_LAUNCH_CHECK();
        return 0;
    }
    if (rows_ok(g, 20) && m <= 20) {
        const int nseg = (int)((g.nwrite + 1023) / 1024);
        const unsigned nb = (unsigned)(nseg * g.outer);
#define DT_ROWS(T_, MB_)                                                                                                        \
        do {                                                                                                                    \
            Taps<T_> t; load_taps(t, h, nullptr, m);                                                                            \
            if (rows_vec(g, Y, sizeof(T_))) k_colfilter_rows<T_, MB_, true><<<nb, 256, 0, ctx->stream>>>((const T_ *)X, (T_ *)Y, g, t, m, nseg); \
            else k_colfilter_rows<T_, MB_, false><<<nb, 256, 0, ctx->stream>>>((const T_ *)X, (T_ *)Y, g, t, m, nseg);          \
        } while (0)
        if (dtype == DTCWT_HIP_F32) { if (m <= 8) DT_ROWS(float, 8); else DT_ROWS(float, 20); }
        else { if (m <= 8) DT_ROWS(double, 8); else DT_ROWS(double, 20); }
#undef DT_ROWS
        DT_LAUNCH_CHECK();
        return 0;
    }
    if (unrolled_ok(g, 20) && m <= 20) {
        if (dtype == DTCWT_HIP_F32) {
            Taps<float> t; load_taps(t, h, nullptr, m);
            if (m <= 8) k_colfilter_march<float, 1, 8><<<blocks_for(total), 256, 0, ctx->stream>>>((const float *)X, (float *)Y, g, t, m);
            else k_colfilter_march<float, 1, 20><<<blocks_for(total), 256, 0, ctx->stream>>>((const float *)X, (float *)Y, g, t, m);
        } else {
            Taps<double> t; load_taps(t, h, nullptr, m);
            if (m <= 8) k_colfilter_march<double, 1, 8><<<blocks_for(total), 256, 0, ctx->stream>>>((const double *)X, (double *)Y, g, t, m);
            else k_colfilter_march<double, 1, 20><<<blocks_for(total), 256, 0, ctx->stream>>>((const double *)X, (double *)Y, g, t, m);
        }
        DT_LAUNCH_CHECK();
        return 0;
    }
    if (dtype == DTCWT_HIP_F32) {
        Taps<float> t; load_taps(t, h, nullptr, m);
        k_colfilter<float><<<blocks_for(total), 256, 0, ctx->stream>>>((const float *)X, (float *)Y, g, t, m);
    } else {
        Taps<double> t; load_taps(t, h, nullptr, m);
        k_colfilter<double><<<blocks_for(total), 256, 0, ctx->stream>>>((const double *)X, (double *)Y, g, t, m);
    }
    DT_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_coldfilt(dtcwt_hip_ctx *ctx, int dtype, const void *X, void *Y,
                       const dtcwt_hip_view *v, const double *ha, const double *hb, int m,
                       int flags) {
    DT_REQUIRE(ctx && X && Y && ha && hb, "NULL argument");
    DT_REQUIRE(m >= 2 && m <= DTCWT_HIP_MAX_TAPS && (m % 2) == 0,
               "Lengths of ha and hb must be even (and <= %d)", DTCWT_HIP_MAX_TAPS);
    DT_REQUIRE(dtype == DTCWT_HIP_F32 || dtype == DTCWT_HIP_F64, "bad dtype %d", dtype);
    Geo g;
    int rc = make_geo(v, nout_coldfilt, m, 2, flags, g);
    if (rc) return rc;
    DT_REQUIRE(g.L % 4 == 0, "No. of rows in X must be a multiple of 4");
    int64_t total = g.outer * g.ngroups * g.inner;
    if (total == 0) return 0;
    DT_REQUIRE(total < ((int64_t)1 << 39), "problem too large");
    int a_first = dot(ha, hb, m) > 0 ? 1 : 0;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    if (march_ok(g, 12 + 40) && m <= 20) {
        // 4 (A, B) pairs per thread from a register window, taps front padded to the bucket
        const int mb = m <= 10 ? 10 : 20;
        g.ngroups = (g.ngroups + 3) / 4;
        const int64_t tot = g.outer * g.ngroups * g.inner;
        if (dtype == DTCWT_HIP_F32) {
            Taps<float> t; load_taps_front(t, ha, hb, m, mb);
            if (mb == 10) k_coldfilt_march<float, 4, 10><<<blocks_for(tot), 256, 0, ctx->stream>>>((const float *)X, (float *)Y, g, t, m, a_first);
            else k_coldfilt_march<float, 4, 20><<<blocks_for(tot), 256, 0, ctx->stream>>>((const float *)X, (float *)Y, g, t, m, a_first);
        } else {
            Taps<double> t; load_taps_front(t, ha, hb, m, mb);
            if (mb == 10) k_coldfilt_march<double, 4, 10><<<blocks_for(tot), 256, 0, ctx->stream>>>((const double *)X, (double *)Y, g, t, m, a_first);
            else k_coldfilt_march<double, 4, 20><<<blocks_for(tot), 256, 0, ctx->stream>>>((const double *)X, (double *)Y, g, t, m, a_first);
        }
        DT_LAUNCH_CHECK();
        return 0;
    }
    if (rows_ok(g, 40) && m <= 20) {
        const int mb = m <= 10 ? 10 : 20;
        const int nseg = (int)((g.ngroups + 511) / 512);
        const unsigned nb = (unsigned)(nseg * g.outer);
#define DT_ROWS(T_, MB_)                                                                                                        \
        do {                                                                                                                    \
            Taps<T_> t; load_taps_front(t, ha, hb, m, mb);                                                                      \
            if (rows_vec(g, Y, sizeof(T_))) k_coldfilt_rows<T_, MB_, true><<<nb, 256, 0, ctx->stream>>>((const T_ *)X, (T_ *)Y, g, t, m, a_first, nseg); \
            else k_coldfilt_rows<T_, MB_, false><<<nb, 256, 0, ctx->stream>>>((const T_ *)X, (T_ *)Y, g, t, m, a_first, nseg);  \
        } while (0)
        if (dtype == DTCWT_HIP_F32) { if (mb == 10) DT_ROWS(float, 10); else DT_ROWS(float, 20); }
        else { if (mb == 10) DT_ROWS(double, 10); else DT_ROWS(double, 20); }
#undef DT_ROWS
        DT_LAUNCH_CHECK();
        return 0;
    }
    if (unrolled_ok(g, 40) && m <= 20) {
        const int mb = m <= 10 ? 10 : 20;
        if (dtype == DTCWT_HIP_F32) {
            Taps<float> t; load_taps_front(t, ha, hb, m, mb);
            if (mb == 10) k_coldfilt_march<float, 1, 10><<<blocks_for(total), 256, 0, ctx->stream>>>((const float *)X, (float *)Y, g, t, m, a_first);
            else k_coldfilt_march<float, 1, 20><<<blocks_for(total), 256, 0, ctx->stream>>>((const float *)X, (float *)Y, g, t, m, a_first);
        } else {
            Taps<double> t; load_taps_front(t, ha, hb, m, mb);
            if (mb == 10) k_coldfilt_march<double, 1, 10><<<blocks_for(total), 256, 0, ctx->stream>>>((const double *)X, (double *)Y, g, t, m, a_first);
            else k_coldfilt_march<double, 1, 20><<<blocks_for(total), 256, 0, ctx->stream>>>((const double *)X, (double *)Y, g, t, m, a_first);
        }
        DT_LAUNCH_CHECK();
        return 0;
    }
    if (dtype == DTCWT_HIP_F32) {
        Taps<float> t; load_taps(t, ha, hb, m);
        k_coldfilt<float><<<blocks_for(total), 256, 0, ctx->stream>>>((const float *)X, (float *)Y, g, t, m, a_first);
    } else {
        Taps<double> t; load_taps(t, ha, hb, m);
        k_coldfilt<double><<<blocks_for(total), 256, 0, ctx->stream>>>((const double *)X, (double *)Y, g, t, m, a_first);
    }
    DT_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_colifilt(dtcwt_hip_ctx *ctx, int dtype, const void *X, void *Y,
                       const dtcwt_hip_view *v, const double *ha, const double *hb, int m,
                       int flags) {
    DT_REQUIRE(ctx && X && Y && ha && hb, "NULL argument");
    DT_REQUIRE(m >= 2 && m <= DTCWT_HIP_MAX_TAPS && (m % 2) == 0,
               "Lengths of ha and hb must be even (and <= %d)", DTCWT_HIP_MAX_TAPS);
    DT_REQUIRE(dtype == DTCWT_HIP_F32 || dtype == DTCWT_HIP_F64, "bad dtype %d", dtype);
    Geo g;
    int rc = make_geo(v, nout_colifilt, m, 4, flags, g);
    if (rc) return rc;
    DT_REQUIRE(g.L % 2 == 0, "No. of rows in X must be a multiple of 2");
    int64_t total = g.outer * g.ngroups * g.inner;
    if (total == 0) return 0;
    DT_REQUIRE(total < ((int64_t)1 << 39), "problem too large");
    int pos = dot(ha, hb, m) > 0 ? 1 : 0;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    // bucket of the same m/2 parity (zero padding must be even per side): 10 / 18 odd, 8 / 16 even
    const int mb = ((m / 2) & 1) ? (m <= 10 ? 10 : (m <= 18 ? 18 : 0)) : (m <= 8 ? 8 : (m <= 16 ? 16 : 0));
    if (mb && march_ok(g, mb + 4)) {
        g.ngroups = (g.ngroups + 1) / 2;            // two input pairs (eight outputs) per thread
        const int64_t tot = g.outer * g.ngroups * g.inner;
#define DT_IFILT_MARCH(T_)                                                                              \
        do {                                                                                            \
            Taps<T_> t; load_taps_centred(t, ha, hb, m, mb);                                            \
            if (mb == 10) k_colifilt_march<T_, 2, 10><<<blocks_for(tot), 256, 0, ctx->stream>>>((const T_ *)X, (T_ *)Y, g, t, pos); \
            else if (mb == 18) k_colifilt_march<T_, 2, 18><<<blocks_for(tot), 256, 0, ctx->stream>>>((const T_ *)X, (T_ *)Y, g, t, pos); \
            else if (mb == 8) k_colifilt_march<T_, 2, 8><<<blocks_for(tot), 256, 0, ctx->stream>>>((const T_ *)X, (T_ *)Y, g, t, pos); \
            else k_colifilt_march<T_, 2, 16><<<blocks_for(tot), 256, 0, ctx->stream>>>((const T_ *)X, (T_ *)Y, g, t, pos); \
        } while (0)
        if (dtype == DTCWT_HIP_F32) DT_IFILT_MARCH(float); else DT_IFILT_MARCH(double);
#undef DT_IFILT_MARCH
        DT_LAUNCH_CHECK();
        return 0;
    }
    if (mb && rows_ok(g, mb + 4)) {
        const int nseg = (int)((g.ngroups + 511) / 512);
        const unsigned nb = (unsigned)(nseg * g.outer);
#define DT_ROWS(T_, MB_)                                                                                                        \
        do {                                                                                                                    \
            Taps<T_> t; load_taps_centred(t, ha, hb, m, mb);                                                                    \
            if (rows_vec(g, Y, sizeof(T_))) k_colifilt_rows<T_, MB_, true><<<nb, 256, 0, ctx->stream>>>((const T_ *)X, (T_ *)Y, g, t, pos, nseg); \
            else k_colifilt_rows<T_, MB_, false><<<nb, 256, 0, ctx->stream>>>((const T_ *)X, (T_ *)Y, g, t, pos, nseg);         \
        } while (0)
#define DT_ROWS_T(T_) do { if (mb == 10) DT_ROWS(T_, 10); else if (mb == 18) DT_ROWS(T_, 18); else if (mb == 8) DT_ROWS(T_, 8); else DT_ROWS(T_, 16); } while (0)
        if (dtype == DTCWT_HIP_F32) DT_ROWS_T(float); else DT_ROWS_T(double);
#undef DT_ROWS_T
#undef DT_ROWS
        DT_LAUNCH_CHECK();
        return 0;
    }
    if (mb && unrolled_ok(g, mb + 2)) {
#define DT_IFILT_UNROLLED(T_)                                                                           \
        do {                                                                                            \
            Taps<T_> t; load_taps_centred(t, ha, hb, m, mb);                                            \
            if (mb == 10) k_colifilt_march<T_, 1, 10><<<blocks_for(total), 256, 0, ctx->stream>>>((const T_ *)X, (T_ *)Y, g, t, pos); \
            else if (mb == 18) k_colifilt_march<T_, 1, 18><<<blocks_for(total), 256, 0, ctx->stream>>>((const T_ *)X, (T_ *)Y, g, t, pos); \
            else if (mb == 8) k_colifilt_march<T_, 1, 8><<<blocks_for(total), 256, 0, ctx->stream>>>((const T_ *)X, (T_ *)Y, g, t, pos); \
            else k_colifilt_march<T_, 1, 16><<<blocks_for(total), 256, 0, ctx->stream>>>((const T_ *)X, (T_ *)Y, g, t, pos); \
        } while (0)
        if (dtype == DTCWT_HIP_F32) DT_IFILT_UNROLLED(float); else DT_IFILT_UNROLLED(double);
#undef DT_IFILT_UNROLLED
        DT_LAUNCH_CHECK();
        return 0;
    }
    if (dtype == DTCWT_HIP_F32) {
        Taps<float> t; load_taps(t, ha, hb, m);
        k_colifilt<float><<<blocks_for(total), 256, 0, ctx->stream>>>((const float *)X, (float *)Y, g, t, m, pos);
    } else {
        Taps<double> t; load_taps(t, ha, hb, m);
        k_colifilt<double><<<blocks_for(total), 256, 0, ctx->stream>>>((const double *)X, (double *)Y, g, t, m, pos);
    }
    DT_LAUNCH_CHECK();
    return 0;
}

#define DT_DISPATCH_T(CALLF, CALLD)                                                     \
    do {                                                                               \
        if (dtype == DTCWT_HIP_F32) { CALLF; } else { CALLD; }                          \
        DT_LAUNCH_CHECK();                                                             \
    } while (0)

int dtcwt_hip_colfilter2(dtcwt_hip_ctx *ctx, int dtype, const void *X, void *Y0, void *Y1,
                         const dtcwt_hip_view *v, const double *h0, int m0, const double *h1, int m1) {
    DT_REQUIRE(ctx && X && Y0 && Y1 && h0 && h1, "NULL argument");
    DT_REQUIRE(m0 >= 1 && m0 <= DTCWT_HIP_MAX_TAPS && m1 >= 1 && m1 <= DTCWT_HIP_MAX_TAPS, "filter length out of range");
    DT_REQUIRE((m0 & 1) == (m1 & 1), "both filters must have odd or both even length");
    DT_REQUIRE(dtype == DTCWT_HIP_F32 || dtype == DTCWT_HIP_F64, "bad dtype %d", dtype);
    Geo g;
    int rc = make_geo(v, nout_colfilter, m0, 1, 0, g);
    if (rc) return rc;
    int64_t total = g.outer * g.ngroups * g.inner;
    if (total == 0) return 0;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    if (!v->crop_lo && !v->crop_hi && dense_view(v, g.nwrite)) {
        rc = dtcwt_g2_pair(ctx, dtype, 0, X, Y0, Y1, v->outer, v->n, v->inner, v->pad_lo, v->pad_hi, h0, nullptr,
                           h1, nullptr, m0, m1, 0);
        if (rc) return rc < 0 ? rc : 0;
    }
    Taps<float> tf; Taps<double> td;
    load_taps(tf, h0, nullptr, m0); load_taps(td, h0, nullptr, m0);
    for (int k = 0; k < DTCWT_HIP_MAX_TAPS; ++k) { tf.b[k] = k < m1 ? (float)h1[k] : 0.f; td.b[k] = k < m1 ? h1[k] : 0.0; }
    DT_DISPATCH_T((k_colfilter2<float><<<blocks_for(total), 256, 0, ctx->stream>>>((const float *)X, (float *)Y0, (float *)Y1, g, tf, m0, m1)),
                  (k_colfilter2<double><<<blocks_for(total), 256, 0, ctx->stream>>>((const double *)X, (double *)Y0, (double *)Y1, g, td, m0, m1)));
    return 0;
}

int dtcwt_hip_colfilter_sum2(dtcwt_hip_ctx *ctx, int dtype, const void *X0, const void *X1, void *Y,
                             const dtcwt_hip_view *v, const double *h0, int m0, const double *h1, int m1) {
    DT_REQUIRE(ctx && X0 && X1 && Y && h0 && h1, "NULL argument");
    DT_REQUIRE(m0 >= 1 && m0 <= DTCWT_HIP_MAX_TAPS && m1 >= 1 && m1 <= DTCWT_HIP_MAX_TAPS, "filter length out of range");
    DT_REQUIRE((m0 & 1) == (m1 & 1), "both filters must have odd or both even length");
    DT_REQUIRE(dtype == DTCWT_HIP_F32 || dtype == DTCWT_HIP_F64, "bad dtype %d", dtype);
    Geo g;
    int rc = make_geo(v, nout_colfilter, m0, 1, 0, g);
    if (rc) return rc;
    int64_t total = g.outer * g.ngroups * g.inner;
    if (total == 0) return 0;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    if (!v->pad_lo && !v->pad_hi && !v->crop_lo && !v->crop_hi && dense_view(v, g.nwrite)) {
        rc = dtcwt_g2_sum(ctx, dtype, 0, X0, X1, Y, v->outer, v->n, v->inner, 0, h0, nullptr, h1, nullptr, m0, m1,
                          0, 1.0);
        if (rc) return rc < 0 ? rc : 0;
    }
    Taps<float> tf; Taps<double> td;
    load_taps(tf, h0, nullptr, m0); load_taps(td, h0, nullptr, m0);
    for (int k = 0; k < DTCWT_HIP_MAX_TAPS; ++k) { tf.b[k] = k < m1 ? (float)h1[k] : 0.f; td.b[k] = k < m1 ? h1[k] : 0.0; }
    DT_DISPATCH_T((k_colfilter_sum2<float><<<blocks_for(total), 256, 0, ctx->stream>>>((const float *)X0, (const float *)X1, (float *)Y, g, tf, m0, m1)),
                  (k_colfilter_sum2<double><<<blocks_for(total), 256, 0, ctx->stream>>>((const double *)X0, (const double *)X1, (double *)Y, g, td, m0, m1)));
    return 0;
}

int dtcwt_hip_coldfilt2(dtcwt_hip_ctx *ctx, int dtype, const void *X, void *Y0, void *Y1,
                        const dtcwt_hip_view *v, const double *ha0, const double *hb0, const double *ha1,
                        const double *hb1, int m) {
    DT_REQUIRE(ctx && X && Y0 && Y1 && ha0 && hb0 && ha1 && hb1, "NULL argument");
    DT_REQUIRE(m >= 2 && m <= DTCWT_HIP_MAX_TAPS && (m % 2) == 0, "Lengths of ha and hb must be even (and <= %d)", DTCWT_HIP_MAX_TAPS);
    DT_REQUIRE(dtype == DTCWT_HIP_F32 || dtype == DTCWT_HIP_F64, "bad dtype %d", dtype);
    Geo g;
    int rc = make_geo(v, nout_coldfilt, m, 2, 0, g);
    if (rc) return rc;
    DT_REQUIRE(g.L % 4 == 0, "No. of rows in X must be a multiple of 4");
    int64_t total = g.outer * g.ngroups * g.inner;
    if (total == 0) return 0;
    int f0 = dot(ha0, hb0, m) > 0, f1 = dot(ha1, hb1, m) > 0;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    if (!v->crop_lo && !v->crop_hi && dense_view(v, g.nwrite)) {
        rc = dtcwt_g2_pair(ctx, dtype, 1, X, Y0, Y1, v->outer, v->n, v->inner, v->pad_lo, v->pad_hi, ha0, hb0,
                           ha1, hb1, m, m, 0);
        if (rc) return rc < 0 ? rc : 0;
    }
    if (dtype == DTCWT_HIP_F32) {
        Taps<float> t0, t1; load_taps(t0, ha0, hb0, m); load_taps(t1, ha1, hb1, m);
        k_coldfilt2<float><<<blocks_for(total), 256, 0, ctx->stream>>>((const float *)X, (float *)Y0, (float *)Y1, g, t0, t1, m, f0, f1);
    } else {
        Taps<double> t0, t1; load_taps(t0, ha0, hb0, m); load_taps(t1, ha1, hb1, m);
        k_coldfilt2<double><<<blocks_for(total), 256, 0, ctx->stream>>>((const double *)X, (double *)Y0, (double *)Y1, g, t0, t1, m, f0, f1);
    }
    DT_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_colifilt_sum2(dtcwt_hip_ctx *ctx, int dtype, const void *X0, const void *X1, void *Y,
                            const dtcwt_hip_view *v, const double *ha0, const double *hb0, const double *ha1,
                            const double *hb1, int m) {
    DT_REQUIRE(ctx && X0 && X1 && Y && ha0 && hb0 && ha1 && hb1, "NULL argument");
    DT_REQUIRE(m >= 2 && m <= DTCWT_HIP_MAX_TAPS && (m % 2) == 0, "Lengths of ha and hb must be even (and <= %d)", DTCWT_HIP_MAX_TAPS);
    DT_REQUIRE(dtype == DTCWT_HIP_F32 || dtype == DTCWT_HIP_F64, "bad dtype %d", dtype);
    Geo g;
    int rc = make_geo(v, nout_colifilt, m, 4, 0, g);
    if (rc) return rc;
    DT_REQUIRE(g.L % 2 == 0, "No. of rows in X must be a multiple of 2");
    int64_t total = g.outer * g.ngroups * g.inner;
    if (total == 0) return 0;
    int p0 = dot(ha0, hb0, m) > 0, p1 = dot(ha1, hb1, m) > 0;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    if (!v->pad_lo && !v->pad_hi && v->crop_lo == v->crop_hi && dense_view(v, g.nwrite)) {
        rc = dtcwt_g2_sum(ctx, dtype, 1, X0, X1, Y, v->outer, v->n, v->inner, v->crop_lo, ha0, hb0, ha1, hb1, m, m,
                          0, 1.0);
        if (rc) return rc < 0 ? rc : 0;
    }
    if (dtype == DTCWT_HIP_F32) {
        Taps<float> t0, t1; load_taps(t0, ha0, hb0, m); load_taps(t1, ha1, hb1, m);
        k_colifilt_sum2<float><<<blocks_for(total), 256, 0, ctx->stream>>>((const float *)X0, (const float *)X1, (float *)Y, g, t0, t1, m, p0, p1);
    } else {
        Taps<double> t0, t1; load_taps(t0, ha0, hb0, m); load_taps(t1, ha1, hb1, m);
        k_colifilt_sum2<double><<<blocks_for(total), 256, 0, ctx->stream>>>((const double *)X0, (const double *)X1, (double *)Y, g, t0, t1, m, p0, p1);
    }
    DT_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_q2c(dtcwt_hip_ctx *ctx, int dtype, const void *y, int64_t batch, int64_t rows,
                  int64_t cols, int64_t y_sb, int64_t y_sr, void *Yh, int slot0, int slot1) {
    DT_REQUIRE(ctx && y && Yh, "NULL argument");
    DT_REQUIRE(rows % 2 == 0 && cols % 2 == 0, "q2c needs even rows/cols");
    DT_REQUIRE(slot0 >= 0 && slot0 < 6 && slot1 >= 0 && slot1 < 6, "bad subband slot");
    int64_t total = batch * (rows / 2) * (cols / 2);
    if (total == 0) return 0;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    if (dtype == DTCWT_HIP_F32)
        k_q2c<float><<<blocks_for(total), 256, 0, ctx->stream>>>((const float *)y, batch, rows / 2, cols / 2, y_sb, y_sr, (float *)Yh, slot0, slot1);
    else if (dtype == DTCWT_HIP_F64)
        k_q2c<double><<<blocks_for(total), 256, 0, ctx->stream>>>((const double *)y, batch, rows / 2, cols / 2, y_sb, y_sr, (double *)Yh, slot0, slot1);
    else
        return dtcwt_set_error(-1, "bad dtype %d", dtype);
    DT_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_c2q(dtcwt_hip_ctx *ctx, int dtype, const void *Yh, int64_t batch, int64_t rows,
                  int64_t cols, int slot0, int slot1, double gain0, double gain1, void *x,
                  int64_t x_sb, int64_t x_sr) {
    DT_REQUIRE(ctx && x && Yh, "NULL argument");
    DT_REQUIRE(slot0 >= 0 && slot0 < 6 && slot1 >= 0 && slot1 < 6, "bad subband slot");
    int64_t total = batch * rows * cols;
    if (total == 0) return 0;
    const double s = 0.70710678118654752440;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    if (dtype == DTCWT_HIP_F32)
        k_c2q<float><<<blocks_for(total), 256, 0, ctx->stream>>>((const float *)Yh, batch, rows, cols, slot0, slot1, (float)(s * gain0), (float)(s * gain1), (float *)x, x_sb, x_sr);
    else if (dtype == DTCWT_HIP_F64)
        k_c2q<double><<<blocks_for(total), 256, 0, ctx->stream>>>((const double *)Yh, batch, rows, cols, slot0, slot1, s * gain0, s * gain1, (double *)x, x_sb, x_sr);
    else
        return dtcwt_set_error(-1, "bad dtype %d", dtype);
    DT_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_cube2c(dtcwt_hip_ctx *ctx, int dtype, const void *y, int64_t d0, int64_t d1,
                     int64_t d2, int64_t s0, int64_t s1, void *Yh, int octant) {
    DT_REQUIRE(ctx && y && Yh, "NULL argument");
    DT_REQUIRE(d0 % 2 == 0 && d1 % 2 == 0 && d2 % 2 == 0, "cube2c needs even extents");
    DT_REQUIRE(octant >= 0 && octant < 7, "bad octant");
    int64_t total = (d0 / 2) * (d1 / 2) * (d2 / 2);
    if (total == 0) return 0;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    if (dtype == DTCWT_HIP_F32)
        k_cube2c<float><<<blocks_for(total), 256, 0, ctx->stream>>>((const float *)y, d0 / 2, d1 / 2, d2 / 2, s0, s1, (float *)Yh, octant);
    else if (dtype == DTCWT_HIP_F64)
        k_cube2c<double><<<blocks_for(total), 256, 0, ctx->stream>>>((const double *)y, d0 / 2, d1 / 2, d2 / 2, s0, s1, (double *)Yh, octant);
    else
        return dtcwt_set_error(-1, "bad dtype %d", dtype);
    DT_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_c2cube(dtcwt_hip_ctx *ctx, int dtype, const void *Yh, int64_t e0, int64_t e1,
                     int64_t e2, int octant, void *y, int64_t s0, int64_t s1) {
    DT_REQUIRE(ctx && y && Yh, "NULL argument");
    DT_REQUIRE(octant >= 0 && octant < 7, "bad octant");
    int64_t total = e0 * e1 * e2;
    if (total == 0) return 0;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    if (dtype == DTCWT_HIP_F32)
        k_c2cube<float><<<blocks_for(total), 256, 0, ctx->stream>>>((const float *)Yh, e0, e1, e2, octant, (float *)y, s0, s1);
    else if (dtype == DTCWT_HIP_F64)
        k_c2cube<double><<<blocks_for(total), 256, 0, ctx->stream>>>((const double *)Yh, e0, e1, e2, octant, (double *)y, s0, s1);
    else
        return dtcwt_set_error(-1, "bad dtype %d", dtype);
    DT_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_pack1d(dtcwt_hip_ctx *ctx, int dtype, const void *hi, int64_t J, int64_t k, void *Yh) {
    DT_REQUIRE(ctx && hi && Yh, "NULL argument");
    if (J * k == 0) return 0;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    if (dtype == DTCWT_HIP_F32)
        k_pack1d<float><<<blocks_for(J * k), 256, 0, ctx->stream>>>((const float *)hi, J, k, (float *)Yh);
    else if (dtype == DTCWT_HIP_F64)
        k_pack1d<double><<<blocks_for(J * k), 256, 0, ctx->stream>>>((const double *)hi, J, k, (double *)Yh);
    else
        return dtcwt_set_error(-1, "bad dtype %d", dtype);
    DT_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_unpack1d(dtcwt_hip_ctx *ctx, int dtype, const void *Yh, int64_t J, int64_t k, double gain,
                       void *hi) {
    DT_REQUIRE(ctx && hi && Yh, "NULL argument");
    if (J * k == 0) return 0;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    if (dtype == DTCWT_HIP_F32)
        k_unpack1d<float><<<blocks_for(J * k), 256, 0, ctx->stream>>>((const float *)Yh, J, k, (float)gain, (float *)hi);
    else if (dtype == DTCWT_HIP_F64)
        k_unpack1d<double><<<blocks_for(J * k), 256, 0, ctx->stream>>>((const double *)Yh, J, k, gain, (double *)hi);
    else
        return dtcwt_set_error(-1, "bad dtype %d", dtype);
    DT_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_scale(dtcwt_hip_ctx *ctx, int dtype, void *x, int64_t count, double gain) {
    DT_REQUIRE(ctx && x, "NULL argument");
    if (count == 0) return 0;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    if (dtype == DTCWT_HIP_F32)
        k_scale<float><<<blocks_for(count), 256, 0, ctx->stream>>>((float *)x, count, (float)gain);
    else if (dtype == DTCWT_HIP_F64)
        k_scale<double><<<blocks_for(count), 256, 0, ctx->stream>>>((double *)x, count, gain);
    else
        return dtcwt_set_error(-1, "bad dtype %d", dtype);
    DT_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
