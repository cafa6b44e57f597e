// One 2-D DT-CWT level for ANY wavelet length and float32 / float64: the path of every (dtype,
// wavelet) combination that has no fused tile program in fused2d.hip -- above all float64, which
// is what the reference computes in for non-float32 input (dtcwt/utils.py:104-134 asfarray).  It
// replaces the seven-to-nine launches of the filter-by-filter level (colfilter2 / coldfilt2 x3 +
// q2c x3, c2q x3 + *_sum2 x3) with ONE (k_g2_fwd_fused / k_g2_inv_fused: both passes, the level
// intermediate in LDS) or, as building blocks and with DTCWT_HIP_TWO_PASS=1, two:
//
//   forward   pass 1  (Lo, Hi)  = filter pair down the image columns        (marching)
//             pass 2  LoLo + the six q2c-packed subbands from (Lo, Hi)      (LDS rows)
//   inverse   pass 1  (y1, y2)  = column filters of LoLo + c2q(subbands)    (marching)
//             pass 2  Z         = row filters of y1 + y2                    (LDS rows)
//
// so the quad <-> complex packings (dtcwt/numpy/transform2d.py:301-350) never make a trip
// through HBM of their own.  KIND 0 is level 1 (odd-length biorthogonal pair, colfilter
// algebra, transform2d.py:112-130 / :275-293), KIND 1 levels >= 2 (q-shift pairs, coldfilt /
// colifilt algebra, :132-160 / :242-273).  The same passes serve the 1-D levels and, on dense
// arrays, the axis passes of the filter-by-filter 3-D levels.
//
// "Marching" kernels: lanes along the contiguous image axis, every thread slides a register
// window down the rows.  "LDS rows" kernels filter ALONG the contiguous axis: a block stages
// line segments (with the reflected halo) into LDS with coalesced loads, threads read their
// windows back as 16-byte vectors and write 16-byte lane-contiguous results; the interleaved
// subband records of forward pass 2 are staged through LDS once more so that they leave as
// contiguous 16-byte stores rather than 16 bytes every 48 / 96.
//
// Taps arrive zero padded to a compile-time bucket MB so that every register index is a
// constant (same device as the marching kernels of filters.hip).  Index algebra: SURVEY.md
// Appendix A.
#include "common.hpp"

#include <type_traits>
#include <cstdlib>

namespace {

constexpr int G2_MAXB = 20;

template <typename T>
struct QTaps {          // KIND 0: a = lo filter, b = hi filter.  KIND 1: (a, b) lo pair, (c, d) hi pair
    T a[G2_MAXB], b[G2_MAXB], c[G2_MAXB], d[G2_MAXB];
};

// logical sample u -> real sample of a line of n samples replicated by (pad_lo, L-n-pad_lo)
// and reflected about its ends; u further out than one period is clamped first (such window
// entries only feed outputs that are never written).
__device__ inline int g2_src(int u, int L, int pad_lo, int n) {
    u = u < -L ? -L : (u > 2 * L - 1 ? 2 * L - 1 : u);
    u = u < 0 ? -1 - u : u;
    u = u >= L ? 2 * L - 1 - u : u;
    u -= pad_lo;
    return u < 0 ? 0 : (u > n - 1 ? n - 1 : u);
}

template <typename T> struct Vec16;
template <> struct Vec16<float> { using type = float4; static constexpr int N = 4; };
template <> struct Vec16<double> { using type = double2; static constexpr int N = 2; };

// 16-byte accesses to write-once / read-once data (subband records): non-temporal, so that the
// lowpass planes the next launch reads back are what stays in L2 / the Infinity Cache
template <typename V> __device__ inline void stream_store(V *dst, const V &v);
template <> __device__ inline void stream_store<float4>(float4 *dst, const float4 &v) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(*reinterpret_cast<const v4 *>(&v), reinterpret_cast<v4 *>(dst));
}
template <> __device__ inline void stream_store<double2>(double2 *dst, const double2 &v) {
    typedef double v2 __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(*reinterpret_cast<const v2 *>(&v), reinterpret_cast<v2 *>(dst));
}
template <> __device__ inline void stream_store<float2>(float2 *dst, const float2 &v) {
    typedef float v2 __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(*reinterpret_cast<const v2 *>(&v), reinterpret_cast<v2 *>(dst));
}
template <typename V> __device__ inline V stream_load(const V *src);
template <> __device__ inline float4 stream_load<float4>(const float4 *src) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const v4 t = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(src));
    return make_float4(t.x, t.y, t.z, t.w);
}
template <> __device__ inline double2 stream_load<double2>(const double2 *src) {
    typedef double v2 __attribute__((ext_vector_type(2)));
    const v2 t = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(src));
    return make_double2(t.x, t.y);
}

template <typename T, int N>
__device__ inline void lds_window(const T *p, T (&w)[N]) {        // p 16-byte aligned
    using V = typename Vec16<T>::type;
    constexpr int VN = Vec16<T>::N;
    static_assert(N % VN == 0, "window not a whole number of vectors");
#pragma unroll
    for (int j = 0; j < N / VN; ++j) {
        V v = reinterpret_cast<const V *>(p)[j];
        const T *e = reinterpret_cast<const T *>(&v);
#pragma unroll
        for (int t = 0; t < VN; ++t) w[j * VN + t] = e[t];
    }
}

// ---- window algebra ------------------------------------------------------------------------
// colfilter, taps end padded to MB: out[q] = sum_k h[k] w[q + MB-1-k]
template <typename T, int G, int MB, int WN>
__device__ inline void fir_colfilter(const T (&w)[WN], const T *h, T (&acc)[G]) {
#pragma unroll
    for (int q = 0; q < G; ++q) {
#pragma unroll
        for (int k = 0; k < MB; ++k) acc[q] += h[k] * w[q + MB - 1 - k];
    }
}

// coldfilt, taps FRONT padded to MB (cf. k_coldfilt_march): pair q -> outputs 2q, 2q+1
template <typename T, int GP, int MB, int WN>
__device__ inline void fir_coldfilt(const T (&w)[WN], const T *ha, const T *hb, int a_first, T (&out)[2 * GP]) {
#pragma unroll
    for (int q = 0; q < GP; ++q) {
        T A = 0, B = 0;
#pragma unroll
        for (int k = 0; k < MB / 2; ++k) {
            A += ha[2 * k] * w[4 * q + 2 * MB - 2 - 4 * k];
            A += ha[2 * k + 1] * w[4 * q + 2 * MB - 4 - 4 * k];
            B += hb[2 * k] * w[4 * q + 2 * MB - 1 - 4 * k];
            B += hb[2 * k + 1] * w[4 * q + 2 * MB - 3 - 4 * k];
        }
        out[2 * q] = a_first ? A : B;
        out[2 * q + 1] = a_first ? B : A;
    }
}

// coldfilt in streaming form for long filters: window sample j (logical row u0 + j) is consumed as it
// arrives instead of being kept -- it feeds pair q with exactly one tap of ha (j - 4q even) or hb (odd),
// tap index static after unrolling.  Two filter pairs (lo: a, b; hi: c, d) at once; no WN-register window
// (float64 x 20 taps: 52 doubles), only the 4 GP accumulators per pair.
template <typename T, int GP, int MB, int J>
__device__ inline void scatter_coldfilt(T x, const T *a, const T *b, const T *c, const T *d, T (&A0)[GP], T (&B0)[GP],
                                        T (&A1)[GP], T (&B1)[GP]) {
#pragma unroll
    for (int q = 0; q < GP; ++q) {
        constexpr int dummy = 0; (void)dummy;
        const int dd = J - 4 * q;
        if (dd < 0 || dd >= 2 * MB) continue;
        if ((dd & 1) == 0) {
            // A: w[4q + 2MB-2-4k] with ha[2k], w[4q + 2MB-4-4k] with ha[2k+1]
            const int r = 2 * MB - 2 - dd;
            const int t = (r % 4 == 0) ? 2 * (r / 4) : 2 * ((r - 2) / 4) + 1;
            A0[q] += a[t] * x; A1[q] += c[t] * x;
        } else {
            // B: w[4q + 2MB-1-4k] with hb[2k], w[4q + 2MB-3-4k] with hb[2k+1]
            const int r = 2 * MB - 1 - dd;
            const int t = (r % 4 == 0) ? 2 * (r / 4) : 2 * ((r - 2) / 4) + 1;
            B0[q] += b[t] * x; B1[q] += d[t] * x;
        }
    }
}

template <typename T, int GP, int MB, int J0, int I, int CH>
__device__ inline void stream_coldfilt_apply(const T (&x)[CH], const T *a, const T *b, const T *c, const T *d,
                                             T (&A0)[GP], T (&B0)[GP], T (&A1)[GP], T (&B1)[GP]) {
    if constexpr (I < CH) {
        scatter_coldfilt<T, GP, MB, J0 + I>(x[I], a, b, c, d, A0, B0, A1, B1);
        stream_coldfilt_apply<T, GP, MB, J0, I + 1, CH>(x, a, b, c, d, A0, B0, A1, B1);
    }
}

template <typename T, int GP, int MB, int J0, int N, typename LOAD>
__device__ inline void stream_coldfilt_chunk(LOAD &&load, const T *a, const T *b, const T *c, const T *d,
                                             T (&A0)[GP], T (&B0)[GP], T (&A1)[GP], T (&B1)[GP]) {
    constexpr int WN = 4 * (GP - 1) + 2 * MB;
    if constexpr (J0 < WN) {
        constexpr int CH = J0 + N <= WN ? N : WN - J0;
        T x[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) x[i] = load(J0 + i);
        stream_coldfilt_apply<T, GP, MB, J0, 0, CH>(x, a, b, c, d, A0, B0, A1, B1);
        // this chunk is consumed before the next one is requested beyond what the chunk size allows
#pragma unroll
        for (int q = 0; q < GP; ++q) asm volatile("" : "+v"(A0[q]), "+v"(B0[q]), "+v"(A1[q]), "+v"(B1[q]));
        stream_coldfilt_chunk<T, GP, MB, J0 + N, N>(load, a, b, c, d, A0, B0, A1, B1);
    }
}

// colifilt, taps centre padded to MB (cf. k_colifilt_march): input pair q -> outputs 4q..4q+3,
// ACCUMULATED into acc
template <int MB> struct IfiltGeo {
    static constexpr int M2 = MB / 2;
    static constexpr bool ODD = (M2 % 2) == 1;
    static constexpr int WN = ODD ? MB : MB + 2;
    static constexpr int ORG = ODD ? 1 - M2 : -M2;
};
template <typename T, int GJ, int MB, int WN>
__device__ inline void fir_colifilt(const T (&w)[WN], const T *ha, const T *hb, int pos, T (&acc)[4 * GJ]) {
    using IG = IfiltGeo<MB>;
    constexpr int M2 = IG::M2;
#pragma unroll
    for (int q = 0; q < GJ; ++q) {
        T y0 = 0, y1 = 0, y2 = 0, y3 = 0;
        if constexpr (IG::ODD) {
#pragma unroll
            for (int k = 0; k < M2; ++k) {
                const T hi = w[2 * q + MB - 1 - 2 * k], lo = w[2 * q + MB - 2 - 2 * k];
                const T xa = pos ? hi : lo, xb = pos ? lo : hi;
                y0 += ha[2 * k] * xb; y1 += hb[2 * k] * xa;
                y2 += ha[2 * k + 1] * xb; y3 += hb[2 * k + 1] * xa;
            }
        } else {
#pragma unroll
            for (int k = 0; k < M2; ++k) {
                const T t0 = w[2 * q + MB + 1 - 2 * k], t1 = w[2 * q + MB - 2 * k];
                const T t2 = w[2 * q + MB - 1 - 2 * k], t3 = w[2 * q + MB - 2 - 2 * k];
                const T xa = pos ? t0 : t1, xb = pos ? t1 : t0, xa2 = pos ? t2 : t3, xb2 = pos ? t3 : t2;
                y0 += ha[2 * k + 1] * xb2; y1 += hb[2 * k + 1] * xa2;
                y2 += ha[2 * k] * xb; y3 += hb[2 * k] * xa;
            }
        }
        acc[4 * q] += y0; acc[4 * q + 1] += y1; acc[4 * q + 2] += y2; acc[4 * q + 3] += y3;
    }
}

// ---- forward pass 1: (Lo, Hi) down the columns -------------------------------------------
struct P1Geo {
    int B, R, C;        // input [B][R][C]
    int pad_lo, L;      // logical rows L = R + pad_lo + pad_hi
    int nout;           // output rows
    int ngroups;        // thread groups along the rows
    int u_shift;        // window start = group origin + u_shift
    int af0, af1;       // KIND 1: (A, B) order of the lo / hi pair
    int pack;           // Hi rows (2j, 2j+1) stored interleaved as complex [nout/2][C] (1-D highpass)
};

// rows row0 .. row0+N-1 (row0 even, N even) of the lo and hi outputs of one thread
template <typename T, int N>
__device__ inline void store_p1(T *Lb, T *Hb, T *Hi, const T (&l)[N], const T (&h)[N], int row0,
                                unsigned b, unsigned c, const P1Geo &g) {
    using V2 = typename std::conditional<sizeof(T) == 4, float2, double2>::type;
#pragma unroll
    for (int q = 0; q < N; ++q)
        if (row0 + q < g.nout) Lb[(size_t)(row0 + q) * g.C] = l[q];
    if (g.pack) {       // Yh[j][c] = Hi[2j][c] + i Hi[2j+1][c]   (dtcwt/numpy/transform1d.py:88,100)
        T *Pb = Hi + ((size_t)b * (g.nout >> 1) * g.C + c) * 2;
#pragma unroll
        for (int q = 0; q < N; q += 2)
            if (row0 + q < g.nout) {
                V2 v; v.x = h[q]; v.y = h[q + 1];
                stream_store<V2>(reinterpret_cast<V2 *>(Pb + (size_t)((row0 + q) >> 1) * g.C * 2), v);
            }
    } else {
#pragma unroll
        for (int q = 0; q < N; ++q)
            if (row0 + q < g.nout) Hb[(size_t)(row0 + q) * g.C] = h[q];
    }
}

template <typename T, int KIND, int MB>
__global__ void __launch_bounds__(256) k_g2_fwd_p1(const T *__restrict__ X, T *__restrict__ Lo,
                                                   T *__restrict__ Hi, P1Geo g, QTaps<T> tp) {
    const unsigned id = blockIdx.x * 256u + threadIdx.x;
    const unsigned total = (unsigned)g.B * g.ngroups * g.C;
    if (id >= total) return;
    const unsigned t = id / g.C, c = id - t * g.C;
    const unsigned b = t / g.ngroups, grp = t - b * g.ngroups;
    const T *Xb = X + (size_t)b * g.R * g.C + c;
    T *Lb = Lo + (size_t)b * g.nout * g.C + c, *Hb = Hi + (size_t)b * g.nout * g.C + c;
    if constexpr (KIND == 0) {
        constexpr int G = 8, WN = G + MB - 1;
        const int lo0 = grp * G, u0 = lo0 + g.u_shift;
        T w[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j) w[j] = Xb[(size_t)g2_src(u0 + j, g.L, g.pad_lo, g.R) * g.C];
        T l[G], h[G];
#pragma unroll
        for (int q = 0; q < G; ++q) l[q] = h[q] = 0;
        fir_colfilter<T, G, MB>(w, tp.a, l);
        fir_colfilter<T, G, MB>(w, tp.b, h);
        store_p1<T, G>(Lb, Hb, Hi, l, h, lo0, b, c, g);
    } else {
        constexpr int GP = 4, WN = 4 * (GP - 1) + 2 * MB;
        const int i0 = grp * GP, u0 = 4 * i0 + g.u_shift;
        T w[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j) w[j] = Xb[(size_t)g2_src(u0 + j, g.L, g.pad_lo, g.R) * g.C];
        T l[2 * GP], h[2 * GP];
        fir_coldfilt<T, GP, MB>(w, tp.a, tp.b, g.af0, l);
        fir_coldfilt<T, GP, MB>(w, tp.c, tp.d, g.af1, h);
        store_p1<T, 2 * GP>(Lb, Hb, Hi, l, h, 2 * i0, b, c, g);
    }
}

// ---- LDS row kernels: common geometry -----------------------------------------------------
// A block of 256 threads = (256 / tpl) lines (or row pairs) x tpl threads per line; a thread
// owns IN_STEP input samples of its line and the WIN-sample window that starts there.
template <int KIND, int MB, bool INVERSE> struct RowGeo;
template <int MB> struct RowGeo<0, MB, false> {      // colfilter, 4 outputs per thread
    static constexpr int IN_STEP = 4, WIN = (4 + MB - 1 + 3) / 4 * 4, OUTS = 4;
};
template <int MB> struct RowGeo<1, MB, false> {      // coldfilt, 2 (A, B) pairs per thread
    static constexpr int IN_STEP = 8, WIN = 4 + 2 * MB, OUTS = 4;
};
template <int MB> struct RowGeo<0, MB, true> {       // colfilter
    static constexpr int IN_STEP = 4, WIN = (4 + MB - 1 + 3) / 4 * 4, OUTS = 4;
};
template <int MB> struct RowGeo<1, MB, true> {       // colifilt, 2 input pairs -> 8 outputs
    static constexpr int IN_STEP = 4, WIN = IfiltGeo<MB>::WN + 2, OUTS = 8;
};

struct P2Geo {
    int nlines;         // lines (inverse) or row pairs (forward) in the whole problem
    int R1;             // forward: rows of Lo / Hi per image
    int Cin;            // samples per input line
    int pad_lo, L;      // logical line length L = Cin + pads
    int Cout;           // written samples per output line
    int crop;           // inverse: logical output samples dropped at the start
    int tpl, tpl_log2;  // threads per line (power of two, 16..256)
    int nseg;           // blocks per line
    int span, stride;   // staged samples per line, LDS line stride (elements)
    int region;         // LDS elements per line / row pair
    int u_shift;
    int f0, f1;         // a_first (forward KIND 1) or pos (inverse KIND 1) of the lo / hi pair
};

// Stage NL lines into LDS: all loads of a thread are issued before the first LDS write (a loop
// with a run-time trip count compiles to load -> wait -> write per sample, one memory latency
// each).  NLD = compile-time bound on samples per thread and line (tpl >= 16); every address is
// in bounds whatever s is, so the loads need no guard.
template <int IN_STEP, int WIN> struct StageGeo {
    static constexpr int NLD = IN_STEP + (WIN - IN_STEP + 15) / 16;
};
template <typename T, int NLD, int NL>
__device__ inline void stage_lines(T *sm, int stride, const T *const (&line)[NL], int q, int tpl,
                                   int span, int ublk, int L, int pad_lo, int n) {
    T v[NL][NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int src = g2_src(ublk + q + i * tpl, L, pad_lo, n);
#pragma unroll
        for (int l = 0; l < NL; ++l) v[l][i] = line[l][src];
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int s = q + i * tpl;
        if (s < span) {
#pragma unroll
            for (int l = 0; l < NL; ++l) sm[l * stride + s] = v[l][i];
        }
    }
}

// 4 consecutive outputs of one row for the lo and the hi filter (forward pass 2)
template <typename T, int KIND, int MB, int WIN>
__device__ inline void fwd_row_fir(const T *wp, const QTaps<T> &tp, const P2Geo &g, T (&lo)[4], T (&hi)[4]) {
    T w[WIN];
    lds_window<T, WIN>(wp, w);
    if constexpr (KIND == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) lo[q] = hi[q] = 0;
        fir_colfilter<T, 4, MB>(w, tp.a, lo);
        fir_colfilter<T, 4, MB>(w, tp.b, hi);
    } else {
        fir_coldfilt<T, 2, MB>(w, tp.a, tp.b, g.f0, lo);
        fir_coldfilt<T, 2, MB>(w, tp.c, tp.d, g.f1, hi);
    }
}

// the same for long q-shift filters, streaming from LDS (no WIN-register window, cf. scatter_coldfilt)
template <typename T, int MB>
__device__ inline void fwd_row_fir_stream(const T *wp, const QTaps<T> &tp, int f0, int f1, T (&lo)[4], T (&hi)[4]) {
    T A0[2] = {0, 0}, B0[2] = {0, 0}, A1[2] = {0, 0}, B1[2] = {0, 0};
    auto load = [&](int j) { return wp[j]; };
    stream_coldfilt_chunk<T, 2, MB, 0, 8>(load, tp.a, tp.b, tp.c, tp.d, A0, B0, A1, B1);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        lo[2 * q] = f0 ? A0[q] : B0[q]; lo[2 * q + 1] = f0 ? B0[q] : A0[q];
        hi[2 * q] = f1 ? A1[q] : B1[q]; hi[2 * q + 1] = f1 ? B1[q] : A1[q];
    }
}

// q2c of the two quads held as rows r0 / r1 of four columns into slots (s0, s1) of the two
// pixel records rec[0], rec[1]  (dtcwt/numpy/transform2d.py:301-322)
template <typename T>
__device__ inline void q2c_pair(const T (&r0)[4], const T (&r1)[4], T (&rec)[2][12], int s0, int s1) {
    const T s = (T)0.70710678118654752440;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const T a = r0[2 * t], b = r0[2 * t + 1], c = r1[2 * t], d = r1[2 * t + 1];
        rec[t][2 * s0] = s * (a - d);
        rec[t][2 * s0 + 1] = s * (b + c);
        rec[t][2 * s1] = s * (a + d);
        rec[t][2 * s1 + 1] = s * (b - c);
    }
}

template <typename T>
__device__ inline void store4(T *p, const T (&v)[4], int valid, bool vec_ok) {
    using V = typename Vec16<T>::type;
    constexpr int VN = Vec16<T>::N;
    if (valid >= 4 && vec_ok) {
#pragma unroll
        for (int j = 0; j < 4 / VN; ++j) {
            V x;
            T *e = reinterpret_cast<T *>(&x);
#pragma unroll
            for (int t = 0; t < VN; ++t) e[t] = v[j * VN + t];
            reinterpret_cast<V *>(p)[j] = x;
        }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (t < valid) p[t] = v[t];
    }
}

// forward pass 2: Lo, Hi [B][R1][Cin] -> LoLo [B][R1][Cout], Yh [B][R1/2][Cout/2][6] complex
template <typename T, int KIND, int MB>
__global__ void __launch_bounds__(256) k_g2_fwd_p2(const T *__restrict__ Lo, const T *__restrict__ Hi,
                                                   T *__restrict__ LoLo, T *__restrict__ Yh, P2Geo g,
                                                   QTaps<T> tp) {
    using RG = RowGeo<KIND, MB, false>;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, q = tid & (g.tpl - 1), rp = tid >> g.tpl_log2;
    const int seg = blockIdx.x % g.nseg, rpg = (blockIdx.x / g.nseg) * (256 >> g.tpl_log2) + rp;
    const bool live = rpg < g.nlines;
    T *sm = reinterpret_cast<T *>(smem_raw) + (size_t)rp * g.region;
    const int ublk = seg * g.tpl * RG::IN_STEP + g.u_shift;
    // rows 2 rpg, 2 rpg + 1 (all images stacked); idle threads stage the last row pair again
    const size_t in0 = (size_t)(live ? rpg : g.nlines - 1) * 2 * g.Cin;
    const int col0 = (seg * g.tpl + q) * 4;
    const int valid = g.Cout - col0;                          // outputs of this thread that exist
    const T *wp = sm + q * RG::IN_STEP;
    const int C2 = g.Cout >> 1;
    constexpr int NLD = StageGeo<RG::IN_STEP, RG::WIN>::NLD;
    T rec[2][12];

    // KIND 0 stages the Lo and the Hi rows together (they fit the LDS the records need anyway);
    // KIND 1 (twice the input per thread) one image after the other.
    constexpr bool BOTH = KIND == 0;
    if constexpr (BOTH) {
        const T *const lines[4] = {Lo + in0, Lo + in0 + g.Cin, Hi + in0, Hi + in0 + g.Cin};
        stage_lines<T, NLD, 4>(sm, g.stride, lines, q, g.tpl, g.span, ublk, g.L, g.pad_lo, g.Cin);
    } else {
        const T *const lines[2] = {Lo + in0, Lo + in0 + g.Cin};
        stage_lines<T, NLD, 2>(sm, g.stride, lines, q, g.tpl, g.span, ublk, g.L, g.pad_lo, g.Cin);
    }
    __syncthreads();
    if (live && valid > 0) {
        T ll0[4], lh0[4], ll1[4], lh1[4];
        fwd_row_fir<T, KIND, MB, RG::WIN>(wp, tp, g, ll0, lh0);
        fwd_row_fir<T, KIND, MB, RG::WIN>(wp + g.stride, tp, g, ll1, lh1);
        T *o = LoLo + (size_t)rpg * 2 * g.Cout + col0;
        const bool vec_ok = ((g.Cout * sizeof(T)) & 15) == 0;
        store4(o, ll0, valid, vec_ok);
        store4(o + g.Cout, ll1, valid, vec_ok);
        q2c_pair(lh0, lh1, rec, 2, 3);
    }
    if constexpr (!BOTH) {
        __syncthreads();
        const T *const lines[2] = {Hi + in0, Hi + in0 + g.Cin};
        stage_lines<T, NLD, 2>(sm, g.stride, lines, q, g.tpl, g.span, ublk, g.L, g.pad_lo, g.Cin);
        __syncthreads();
    }
    if (live && valid > 0) {
        const T *hp = wp + (BOTH ? 2 * g.stride : 0);
        T hl0[4], hh0[4], hl1[4], hh1[4];
        fwd_row_fir<T, KIND, MB, RG::WIN>(hp, tp, g, hl0, hh0);
        fwd_row_fir<T, KIND, MB, RG::WIN>(hp + g.stride, tp, g, hl1, hh1);
        q2c_pair(hl0, hl1, rec, 0, 5);
        q2c_pair(hh0, hh1, rec, 1, 4);
    }
    __syncthreads();
    // records -> LDS -> contiguous 16-byte stores
    using V = typename Vec16<T>::type;
    constexpr int VN = Vec16<T>::N;
    if (live && valid > 0) {
        V *dst = reinterpret_cast<V *>(sm + q * 24);
        const T *src = &rec[0][0];
#pragma unroll
        for (int j = 0; j < 24 / VN; ++j) {
            V x;
            T *e = reinterpret_cast<T *>(&x);
#pragma unroll
            for (int t = 0; t < VN; ++t) e[t] = src[j * VN + t];
            dst[j] = x;
        }
    }
    __syncthreads();
    if (live) {
        const int pix0 = seg * g.tpl * 2;
        int npix = C2 - pix0;
        npix = npix > 2 * g.tpl ? 2 * g.tpl : npix;
        const int nvec = npix > 0 ? npix * 12 / VN : 0;
        V *dst = reinterpret_cast<V *>(Yh + ((size_t)rpg * C2 + pix0) * 12);
        const V *src = reinterpret_cast<const V *>(sm);
        for (int e = q; e < nvec; e += g.tpl) stream_store<V>(dst + e, src[e]);
    }
}

// ---- forward level in ONE launch: pass 1 -> LDS -> pass 2 -------------------------------------
// A workgroup owns 8 rows (4 row pairs) x 256 logical columns of the level intermediate: thread c
// filters its column down the rows (register window, as k_g2_fwd_p1) and leaves lo / hi in LDS;
// after the barrier 4 x NQ threads run the row pass of k_g2_fwd_p2 from those LDS rows.  Lo and Hi
// (twice the image, written and read back by the two-pass form) never reach HBM; the price is the
// vertical halo (MB-1 of G+MB-1 window rows re-read, mostly from L2: vertically adjacent tiles are
// launched next to each other) and idle threads in the row pass.
struct FGeo {
    int B, R, C;            // input [B][R][C]
    int pad_r_lo, LR, R1;   // logical rows, rows of the intermediate (= of LoLo)
    int pad_c_lo, LC, C1;   // logical columns, columns of LoLo
    int nrt, nct;           // tiles along rows / columns
    int u_shift;            // window start = group origin + u_shift (both passes)
    int f0, f1;             // KIND 1: (A, B) order of the lo / hi pair
};

template <typename T, int KIND, int MB>
struct FusedFwd {
    using RG = RowGeo<KIND, MB, false>;
    static constexpr int TW = 256, STR = TW + 4;
    static constexpr int NQ = (TW - (RG::WIN - RG::IN_STEP)) / RG::IN_STEP;     // row-pass threads per row pair
    static constexpr int PLANES = 2 * 8 * STR, RECS = 4 * NQ * 24;
    static constexpr int LDS_ELEMS = PLANES > RECS ? PLANES : RECS;
};

template <typename T, int KIND, int MB>
__global__ void __launch_bounds__(256) k_g2_fwd_fused(const T *__restrict__ X, T *__restrict__ LoLo,
                                                      T *__restrict__ Yh, FGeo g, QTaps<T> tp) {
    using F = FusedFwd<T, KIND, MB>;
    using RG = typename F::RG;
    __shared__ __align__(16) T sm[F::LDS_ELEMS];
    T *sLo = sm, *sHi = sm + 8 * F::STR;
    const int tid = threadIdx.x;
    const int ct = blockIdx.x % g.nct, t1 = blockIdx.x / g.nct;
    const int rt = t1 % g.nrt, b = t1 / g.nrt;
    // ---- pass 1: column tid of the tile, 8 output rows
    {
        const int ucol = ct * F::NQ * RG::IN_STEP + g.u_shift + tid;      // logical column
        const T *Xb = X + (size_t)b * g.R * g.C + g2_src(ucol, g.LC, g.pad_c_lo, g.C);
        T l[8], h[8];
        if constexpr (KIND == 0) {
            constexpr int WN = 8 + MB - 1;
            const int u0 = rt * 8 + g.u_shift;
            T w[WN];
#pragma unroll
            for (int j = 0; j < WN; ++j) w[j] = Xb[(size_t)g2_src(u0 + j, g.LR, g.pad_r_lo, g.R) * g.C];
#pragma unroll
            for (int q = 0; q < 8; ++q) l[q] = h[q] = 0;
            fir_colfilter<T, 8, MB>(w, tp.a, l);
            fir_colfilter<T, 8, MB>(w, tp.b, h);
        } else {
            constexpr int GP = 4, WN = 4 * (GP - 1) + 2 * MB;
            const int u0 = 4 * (rt * GP) + g.u_shift;
            if constexpr (MB > 10) {        // long q-shift filters: streaming form, 8 rows in flight
                T A0[GP], B0[GP], A1[GP], B1[GP];
#pragma unroll
                for (int q = 0; q < GP; ++q) A0[q] = B0[q] = A1[q] = B1[q] = 0;
                auto load = [&](int j) { return Xb[(size_t)g2_src(u0 + j, g.LR, g.pad_r_lo, g.R) * g.C]; };
                stream_coldfilt_chunk<T, GP, MB, 0, 8>(load, tp.a, tp.b, tp.c, tp.d, A0, B0, A1, B1);
#pragma unroll
                for (int q = 0; q < GP; ++q) {
                    l[2 * q] = g.f0 ? A0[q] : B0[q]; l[2 * q + 1] = g.f0 ? B0[q] : A0[q];
                    h[2 * q] = g.f1 ? A1[q] : B1[q]; h[2 * q + 1] = g.f1 ? B1[q] : A1[q];
                }
            } else {
                T w[WN];
#pragma unroll
                for (int j = 0; j < WN; ++j) w[j] = Xb[(size_t)g2_src(u0 + j, g.LR, g.pad_r_lo, g.R) * g.C];
                fir_coldfilt<T, GP, MB>(w, tp.a, tp.b, g.f0, l);
                fir_coldfilt<T, GP, MB>(w, tp.c, tp.d, g.f1, h);
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            sLo[q * F::STR + tid] = l[q];
            sHi[q * F::STR + tid] = h[q];
        }
    }
    __syncthreads();
    // ---- pass 2: 4 row pairs x NQ threads, 4 output columns each
    const int rp = tid / F::NQ, q = tid - rp * F::NQ;
    const int i = rt * 4 + rp;                                  // row pair within the image
    const int col0 = (ct * F::NQ + q) * 4;
    const int valid = g.C1 - col0;
    const bool work = tid < 4 * F::NQ && 2 * i < g.R1 && valid > 0;
    const int C2 = g.C1 >> 1;
    T rec[2][12];
    if (work) {
        P2Geo pg;
        pg.f0 = g.f0; pg.f1 = g.f1;
        const T *wl = sLo + (2 * rp) * F::STR + q * RG::IN_STEP, *wh = sHi + (2 * rp) * F::STR + q * RG::IN_STEP;
        T ll0[4], lh0[4], ll1[4], lh1[4];
        constexpr bool STREAM = KIND == 1 && MB > 10;
        if constexpr (STREAM) {
            fwd_row_fir_stream<T, MB>(wl, tp, g.f0, g.f1, ll0, lh0);
            fwd_row_fir_stream<T, MB>(wl + F::STR, tp, g.f0, g.f1, ll1, lh1);
        } else {
            fwd_row_fir<T, KIND, MB, RG::WIN>(wl, tp, pg, ll0, lh0);
            fwd_row_fir<T, KIND, MB, RG::WIN>(wl + F::STR, tp, pg, ll1, lh1);
        }
        T *o = LoLo + ((size_t)b * g.R1 + 2 * i) * g.C1 + col0;
        const bool vec_ok = ((g.C1 * sizeof(T)) & 15) == 0;
        store4(o, ll0, valid, vec_ok);
        store4(o + g.C1, ll1, valid, vec_ok);
        q2c_pair(lh0, lh1, rec, 2, 3);
        T hl0[4], hh0[4], hl1[4], hh1[4];
        if constexpr (STREAM) {
            fwd_row_fir_stream<T, MB>(wh, tp, g.f0, g.f1, hl0, hh0);
            fwd_row_fir_stream<T, MB>(wh + F::STR, tp, g.f0, g.f1, hl1, hh1);
        } else {
            fwd_row_fir<T, KIND, MB, RG::WIN>(wh, tp, pg, hl0, hh0);
            fwd_row_fir<T, KIND, MB, RG::WIN>(wh + F::STR, tp, pg, hl1, hh1);
        }
        q2c_pair(hl0, hl1, rec, 0, 5);
        q2c_pair(hh0, hh1, rec, 1, 4);
    }
    __syncthreads();
    using V = typename Vec16<T>::type;
    constexpr int VN = Vec16<T>::N;
    if (work) {
        V *dst = reinterpret_cast<V *>(sm + tid * 24);
        const T *src = &rec[0][0];
#pragma unroll
        for (int j = 0; j < 24 / VN; ++j) {
            V x;
            T *e = reinterpret_cast<T *>(&x);
#pragma unroll
            for (int t = 0; t < VN; ++t) e[t] = src[j * VN + t];
            dst[j] = x;
        }
    }
    __syncthreads();
    // records of row pair r: 2 NQ pixels x 12, contiguous in LDS and in Yh; one wavefront per row pair
    {
        const int r = tid >> 6, lane = tid & 63;
        const int ir = rt * 4 + r;
        const int pix0 = ct * F::NQ * 2;
        int npix = C2 - pix0;
        npix = npix > 2 * F::NQ ? 2 * F::NQ : npix;
        if (2 * ir < g.R1 && npix > 0) {
            const int nvec = npix * 12 / VN;
            V *dst = reinterpret_cast<V *>(Yh + (((size_t)b * (g.R1 >> 1) + ir) * C2 + pix0) * 12);
            const V *src = reinterpret_cast<const V *>(sm + r * F::NQ * 24);
            for (int e = lane; e < nvec; e += 64) stream_store<V>(dst + e, src[e]);
        }
    }
}

// inverse pass 2: Z = filter(y1, lo) + filter(y2, hi) along the contiguous axis
template <typename T, int KIND, int MB>
__global__ void __launch_bounds__(256) k_g2_inv_p2(const T *__restrict__ Y1, const T *__restrict__ Y2,
                                                   T *__restrict__ Z, P2Geo g, QTaps<T> tp) {
    using RG = RowGeo<KIND, MB, true>;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, q = tid & (g.tpl - 1), ln = tid >> g.tpl_log2;
    const int seg = blockIdx.x % g.nseg, line = (blockIdx.x / g.nseg) * (256 >> g.tpl_log2) + ln;
    const bool live = line < g.nlines;
    T *sm = reinterpret_cast<T *>(smem_raw) + (size_t)ln * g.region;
    const int ublk = seg * g.tpl * RG::IN_STEP + g.u_shift;
    {
        const size_t in0 = (size_t)(live ? line : g.nlines - 1) * g.Cin;
        const T *const lines[2] = {Y1 + in0, Y2 + in0};
        stage_lines<T, StageGeo<RG::IN_STEP, RG::WIN>::NLD, 2>(sm, g.stride, lines, q, g.tpl, g.span, ublk, g.L, 0, g.Cin);
    }
    __syncthreads();
    if (!live) return;
    const T *wp = sm + q * RG::IN_STEP;
    T acc[RG::OUTS];
#pragma unroll
    for (int k = 0; k < RG::OUTS; ++k) acc[k] = 0;
    {
        T w[RG::WIN];
        lds_window<T, RG::WIN>(wp, w);
        if constexpr (KIND == 0) fir_colfilter<T, 4, MB>(w, tp.a, acc);
        else fir_colifilt<T, 2, MB>(w, tp.a, tp.b, g.f0, acc);
    }
    {
        T w[RG::WIN];
        lds_window<T, RG::WIN>(wp + g.stride, w);
        if constexpr (KIND == 0) fir_colfilter<T, 4, MB>(w, tp.b, acc);
        else fir_colifilt<T, 2, MB>(w, tp.c, tp.d, g.f1, acc);
    }
    const int lo0 = (seg * g.tpl + q) * RG::OUTS - g.crop;    // first written index of this thread
    T *o = Z + (size_t)line * g.Cout;
    const bool vec_ok = g.crop == 0 && ((g.Cout * sizeof(T)) & 15) == 0;
#pragma unroll
    for (int h = 0; h < RG::OUTS / 4; ++h) {
        const int w0 = lo0 + 4 * h;
        T v[4] = {acc[4 * h], acc[4 * h + 1], acc[4 * h + 2], acc[4 * h + 3]};
        if (w0 >= 0 && w0 + 4 <= g.Cout && vec_ok) store4(o + w0, v, 4, true);
        else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (w0 + t >= 0 && w0 + t < g.Cout) o[w0 + t] = v[t];
        }
    }
}

// ---- inverse pass 1: (y1, y2) down the columns, c2q on load --------------------------------
// A block is 256 adjacent column pairs x one group of 8 output rows.  The window is consumed
// one subband-record row (= two image rows) at a time: the block copies the 256 records of that
// row into LDS with contiguous 16-byte loads (a record is 48 / 96 bytes, so per-lane record
// loads would touch every cache line six times), every thread takes its own record back, undoes
// the q2c packing of all three planes at once and SCATTERS the two rows into the output
// accumulators (tap index = output row - input row, a compile-time constant once the loops
// are unrolled) -- no register window, every record read once per row group.  The next row's
// records and lowpass samples are in flight while the current one is consumed.
struct I1Geo {
    int B, Rl, C;       // lowpass input [B][Rl][C]; subbands [B][Rl/2][C/2][6]
    int Rout;           // written rows of y1 / y2
    int crop;           // logical output rows dropped at the start
    int ngroups;        // groups of InvP1Outs output rows
    int bpr;            // blocks per row of column pairs
    int u_shift;
    int f0, f1;         // KIND 1: pos of the lo / hi pair
};

// colfilter, rows (2P, 2P+1) of the window: acc[q] += h[q + MB-1 - j] w[j]
template <typename T, int MB, int P, int OUTS>
__device__ inline void scatter_colfilter(const T *h, T e, T o, T (&acc)[OUTS]) {
#pragma unroll
    for (int q = 0; q < OUTS; ++q) {
        constexpr int base = MB - 1 - 2 * P;
        const int k = q + base;
        if (k >= 0 && k < MB) acc[q] += h[k] * e;
        if (k - 1 >= 0 && k - 1 < MB) acc[q] += h[k - 1] * o;
    }
}

// colifilt, input pair P of the window (cf. fir_colifilt): the pair feeds output pair q with
// tap pair k = q + M2-1-P (and, for even M2, k = q + M2-P on the other two phases)
template <typename T, int MB, int P, int OUTS>
__device__ inline void scatter_colifilt(const T *ha, const T *hb, int pos, T e, T o, T (&acc)[OUTS]) {
    using IG = IfiltGeo<MB>;
    constexpr int M2 = IG::M2;
    const T xa = pos ? o : e, xb = pos ? e : o;
#pragma unroll
    for (int q = 0; q < OUTS / 4; ++q) {
        const int k = q + M2 - 1 - P;
        if constexpr (IG::ODD) {
            if (k >= 0 && k < M2) {
                acc[4 * q] += ha[2 * k] * xb; acc[4 * q + 1] += hb[2 * k] * xa;
                acc[4 * q + 2] += ha[2 * k + 1] * xb; acc[4 * q + 3] += hb[2 * k + 1] * xa;
            }
        } else {
            if (k >= 0 && k < M2) {
                acc[4 * q] += ha[2 * k + 1] * xb; acc[4 * q + 1] += hb[2 * k + 1] * xa;
            }
            if (k + 1 >= 0 && k + 1 < M2) {
                acc[4 * q + 2] += ha[2 * (k + 1)] * xb; acc[4 * q + 3] += hb[2 * (k + 1)] * xa;
            }
        }
    }
}

template <typename T>
struct Gains { T g[6]; };

// OUTS output rows per workgroup: the window is OUTS + MB (KIND 0) or OUTS / 2 + MB or so (KIND 1)
// rows, so more rows per group = fewer re-reads, more accumulators.
template <typename T, int KIND> struct InvP1Outs { static constexpr int N = KIND == 0 ? 8 : 16; };

template <typename T, int KIND, int MB>
struct InvP1 {
    static constexpr int OUTS = InvP1Outs<T, KIND>::N;
    using V = typename Vec16<T>::type;
    using V2 = typename std::conditional<sizeof(T) == 4, float2, double2>::type;
    static constexpr int VN = Vec16<T>::N, NV = 12 / VN;
    static constexpr int NP = (KIND == 0 ? OUTS + MB : IfiltGeo<MB>::WN + 2 * (OUTS / 4 - 1)) / 2;   // record rows per window

    const T *Zb, *Yrow0;
    T *O1, *O2;
    int tid, nvec, u0, Rl, C;
    size_t rstride;
    bool live, swn;
    T y1a[OUTS], y1b[OUTS], y2a[OUTS], y2b[OUTS];       // (column 0, column 1) of y1, y2
    V rg[NV];
    V2 z0, z1;

    __device__ inline void issue(int p) {
        int u = u0 + 2 * p;
        asm volatile("" : "+v"(u));         // addresses of row p are computed here, not all up front
        const int r = g2_src(u, Rl, 0, Rl);
        swn = r & 1;
        const V *src = reinterpret_cast<const V *>(Yrow0 + (size_t)(r >> 1) * rstride);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + 256 * i;
            if (idx < nvec) rg[i] = src[idx];      // temporal: the neighbouring row group wants these rows too
        }
        if (live) {
            z0 = *reinterpret_cast<const V2 *>(Zb + (size_t)r * C);
            z1 = *reinterpret_cast<const V2 *>(Zb + (size_t)g2_src(u + 1, Rl, 0, Rl) * C);
        }
    }

    // S = position in the sequence; the record row consumed is P = S (top-down) or NP-1-S
    // (REV, bottom-up).  Odd row groups run bottom-up so that the rows two vertically adjacent
    // workgroups share are wanted by both at the same time (one of them finds them in L2):
    // without this the 2x window overlap went to the fabric twice (FETCH 1.92x algorithmic).
    template <int S, bool REV>
    __device__ inline void step(V *buf, const QTaps<T> &tp, const Gains<T> &gn, int f0, int f1) {
        constexpr int P = REV ? NP - 1 - S : S;
        constexpr int PN = REV ? P - 1 : P + 1;
        V *bw = buf + (S & 1) * 256 * NV;
        asm volatile("" ::: "memory");      // keep the loads of later rows from being hoisted up here
#pragma unroll
        for (int i = 0; i < NV; ++i) bw[tid + 256 * i] = rg[i];
        const V2 c0 = z0, c1 = z1;
        const bool sw = swn;
        if (S + 1 < NP) issue(PN);
        __syncthreads();
        T rec[12];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            V v = bw[tid * NV + i];
            const T *e = reinterpret_cast<const T *>(&v);
#pragma unroll
            for (int t = 0; t < VN; ++t) rec[i * VN + t] = e[t];
        }
        // lowpass rows
        acc<P, 0>(tp, f0, c0.x, c1.x, y1a);
        acc<P, 0>(tp, f0, c0.y, c1.y, y1b);
        // the three planes of c2q (dtcwt/numpy/transform2d.py:324-350)
        plane<P, 1>(rec, 0, 5, gn, sw, tp, f1, y1a, y1b);
        plane<P, 0>(rec, 2, 3, gn, sw, tp, f0, y2a, y2b);
        plane<P, 1>(rec, 1, 4, gn, sw, tp, f1, y2a, y2b);
        // the row is consumed HERE: without this the scheduler runs every load / barrier of the
        // window first and keeps all the records live until one big block of arithmetic
#pragma unroll
        for (int k = 0; k < OUTS; ++k)
            asm volatile("" : "+v"(y1a[k]), "+v"(y1b[k]), "+v"(y2a[k]), "+v"(y2b[k]));
    }

    template <int P, int HI>
    __device__ inline void acc(const QTaps<T> &tp, int pos, T e, T o, T (&a)[OUTS]) {
        if constexpr (KIND == 0) scatter_colfilter<T, MB, P, OUTS>(HI ? tp.b : tp.a, e, o, a);
        else scatter_colifilt<T, MB, P, OUTS>(HI ? tp.c : tp.a, HI ? tp.d : tp.b, pos, e, o, a);
    }

    template <int P, int HI>
    __device__ inline void plane(const T (&rec)[12], int s0, int s1, const Gains<T> &gn, bool sw,
                                 const QTaps<T> &tp, int pos, T (&ya)[OUTS], T (&yb)[OUTS]) {
        const T w0r = rec[2 * s0] * gn.g[s0], w0i = rec[2 * s0 + 1] * gn.g[s0];
        const T w1r = rec[2 * s1] * gn.g[s1], w1i = rec[2 * s1 + 1] * gn.g[s1];
        const T a = w0r + w1r, b = w0i + w1i, c = w0i - w1i, d = -(w0r - w1r);
        // image rows (a b) / (c d); a reflected row pair is the same record, rows exchanged
        acc<P, HI>(tp, pos, sw ? c : a, sw ? a : c, ya);
        acc<P, HI>(tp, pos, sw ? d : b, sw ? b : d, yb);
    }

    template <int S, bool REV>
    __device__ inline void run(V *buf, const QTaps<T> &tp, const Gains<T> &gn, int f0, int f1) {
        if constexpr (S < NP) {
            step<S, REV>(buf, tp, gn, f0, f1);
            run<S + 1, REV>(buf, tp, gn, f0, f1);
        }
    }
};

template <typename T, int KIND, int MB>
__global__ void __launch_bounds__(256) k_g2_inv_p1(const T *__restrict__ Zl, const T *__restrict__ Yh,
                                                   T *__restrict__ Y1, T *__restrict__ Y2, I1Geo g,
                                                   QTaps<T> tp, Gains<T> gn) {
    using S = InvP1<T, KIND, MB>;
    __shared__ typename S::V buf[2 * 256 * S::NV];
    const int C2 = g.C >> 1;
    const int bx = blockIdx.x % g.bpr, t = blockIdx.x / g.bpr;
    const int grp = t % g.ngroups, b = t / g.ngroups;
    const int jc0 = bx * 256;
    S st;
    st.tid = threadIdx.x;
    st.live = jc0 + st.tid < C2;
    const int nrec = C2 - jc0 < 256 ? C2 - jc0 : 256;
    st.nvec = nrec * S::NV;
    st.u0 = grp * (KIND == 0 ? S::OUTS : S::OUTS / 2) + g.u_shift;
    st.Rl = g.Rl; st.C = g.C;
    st.rstride = (size_t)C2 * 12;
    st.Zb = Zl + (size_t)b * g.Rl * g.C + 2 * (jc0 + st.tid);
    st.Yrow0 = Yh + ((size_t)b * (g.Rl >> 1) * C2 + jc0) * 12;
#pragma unroll
    for (int k = 0; k < S::OUTS; ++k) st.y1a[k] = st.y1b[k] = st.y2a[k] = st.y2b[k] = 0;
#pragma unroll
    for (int i = 0; i < S::NV; ++i) st.rg[i] = typename S::V{};
    st.z0 = st.z1 = typename S::V2{};
    if (grp & 1) {
        st.issue(S::NP - 1);
        st.template run<0, true>(buf, tp, gn, g.f0, g.f1);
    } else {
        st.issue(0);
        st.template run<0, false>(buf, tp, gn, g.f0, g.f1);
    }
    if (!st.live) return;
    const size_t ob = (size_t)b * g.Rout * g.C + 2 * (jc0 + st.tid);
    const int lo0 = grp * S::OUTS - g.crop;
#pragma unroll
    for (int k = 0; k < S::OUTS; ++k) {
        const int r = lo0 + k;
        if (r >= 0 && r < g.Rout) {
            typename S::V2 v1, v2;
            v1.x = st.y1a[k]; v1.y = st.y1b[k];
            v2.x = st.y2a[k]; v2.y = st.y2b[k];
            *reinterpret_cast<typename S::V2 *>(Y1 + ob + (size_t)r * g.C) = v1;
            *reinterpret_cast<typename S::V2 *>(Y2 + ob + (size_t)r * g.C) = v2;
        }
    }
}

// ---- inverse level in ONE launch: pass 1 -> LDS -> pass 2 -------------------------------------
// A workgroup owns 8 output rows x 256 logical columns of the level intermediate (y1, y2).  Pass 1 is
// the streaming form of k_g2_inv_p1 with ONE column per thread (two threads share a record): the
// record rows of the window pass through a double-buffered LDS copy, every thread undoes q2c for
// its column and scatters into y1[8], y2[8]; these go to LDS (over the record buffers) and
// 8 x NQ thread-tasks run the row pass of k_g2_inv_p2 from there.  y1 / y2 never reach HBM.
struct IGeo {
    int B, Rl, Cl;          // lowpass input [B][Rl][Cl]; subbands [B][Rl/2][Cl/2][6]
    int Rout, Cout;         // written rows / columns of Z
    int crop_r, crop_c;
    int nrt, nct;
    int u_shift;
    int f0, f1;             // KIND 1: pos of the lo / hi pair
};

template <typename T, int KIND, int MB>
struct FusedInv {
    using RG = RowGeo<KIND, MB, true>;
    using V = typename Vec16<T>::type;
    static constexpr int TW = 256, STR = TW + 4, NREC = 130;
    static constexpr int NQ = (TW - (RG::WIN - RG::IN_STEP)) / RG::IN_STEP;
    static constexpr int VN = Vec16<T>::N, NV = 12 / VN;
    static constexpr int NVEC = NREC * NV, NVT = (NVEC + 255) / 256;
    static constexpr int NP = (KIND == 0 ? 8 + MB : IfiltGeo<MB>::WN + 2) / 2;
    static constexpr int RECBUF = NREC * 12;
    static constexpr int LDS_ELEMS = 2 * RECBUF > 2 * 8 * STR ? 2 * RECBUF : 2 * 8 * STR;

    const T *Zc, *Yrow0;    // lowpass column of this thread; records of the tile, row 0
    int tid, nvec, u0, Rl, Cl, ridx, par;
    size_t rstride;
    bool swn;
    T y1[8], y2[8];
    V rg[NVT];
    T z0, z1;

    __device__ inline void issue(int p) {
        int u = u0 + 2 * p;
        asm volatile("" : "+v"(u));
        const int r = g2_src(u, Rl, 0, Rl);
        swn = r & 1;
        const V *src = reinterpret_cast<const V *>(Yrow0 + (size_t)(r >> 1) * rstride);
#pragma unroll
        for (int i = 0; i < NVT; ++i) {
            const int idx = tid + 256 * i;
            if (idx < nvec) rg[i] = src[idx];
        }
        z0 = Zc[(size_t)r * Cl];
        z1 = Zc[(size_t)g2_src(u + 1, Rl, 0, Rl) * Cl];
    }

    template <int P, int HI>
    __device__ inline void acc(const QTaps<T> &tp, int pos, T e, T o, T (&a)[8]) {
        if constexpr (KIND == 0) scatter_colfilter<T, MB, P, 8>(HI ? tp.b : tp.a, e, o, a);
        else scatter_colifilt<T, MB, P, 8>(HI ? tp.c : tp.a, HI ? tp.d : tp.b, pos, e, o, a);
    }

    template <int P, int HI>
    __device__ inline void plane(const T (&rec)[12], int s0, int s1, const Gains<T> &gn, bool sw,
                                 const QTaps<T> &tp, int pos, T (&y)[8]) {
        const T w0r = rec[2 * s0] * gn.g[s0], w0i = rec[2 * s0 + 1] * gn.g[s0];
        const T w1r = rec[2 * s1] * gn.g[s1], w1i = rec[2 * s1 + 1] * gn.g[s1];
        const T a = w0r + w1r, b = w0i + w1i, c = w0i - w1i, d = -(w0r - w1r);
        const T top = par ? b : a, bot = par ? d : c;       // image rows (a b) / (c d), this thread's column
        acc<P, HI>(tp, pos, sw ? bot : top, sw ? top : bot, y);
    }

    template <int S, bool REV>
    __device__ inline void step(T *sm, const QTaps<T> &tp, const Gains<T> &gn, int f0, int f1) {
        constexpr int P = REV ? NP - 1 - S : S;
        constexpr int PN = REV ? P - 1 : P + 1;
        V *bw = reinterpret_cast<V *>(sm + (S & 1) * RECBUF);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < NVT; ++i) {
            const int idx = tid + 256 * i;
            if (idx < NVEC) bw[idx] = rg[i];
        }
        const T c0 = z0, c1 = z1;
        const bool sw = swn;
        if (S + 1 < NP) issue(PN);
        __syncthreads();
        T rec[12];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            V v = bw[ridx * NV + i];
            const T *e = reinterpret_cast<const T *>(&v);
#pragma unroll
            for (int t = 0; t < VN; ++t) rec[i * VN + t] = e[t];
        }
        acc<P, 0>(tp, f0, c0, c1, y1);
        plane<P, 1>(rec, 0, 5, gn, sw, tp, f1, y1);
        plane<P, 0>(rec, 2, 3, gn, sw, tp, f0, y2);
        plane<P, 1>(rec, 1, 4, gn, sw, tp, f1, y2);
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(y1[k]), "+v"(y2[k]));
    }

    template <int S, bool REV>
    __device__ inline void run(T *sm, const QTaps<T> &tp, const Gains<T> &gn, int f0, int f1) {
        if constexpr (S < NP) {
            step<S, REV>(sm, tp, gn, f0, f1);
            run<S + 1, REV>(sm, tp, gn, f0, f1);
        }
    }
};

template <typename T, int KIND, int MB>
__global__ void __launch_bounds__(256) k_g2_inv_fused(const T *__restrict__ Zl, const T *__restrict__ Yh,
                                                      T *__restrict__ Z, IGeo g, QTaps<T> tp, Gains<T> gn) {
    using F = FusedInv<T, KIND, MB>;
    using RG = typename F::RG;
    __shared__ __align__(16) T sm[F::LDS_ELEMS];
    const int tid = threadIdx.x;
    const int ct = blockIdx.x % g.nct, t1 = blockIdx.x / g.nct;
    const int rt = t1 % g.nrt, b = t1 / g.nrt;
    const int C2 = g.Cl >> 1;
    // ---- pass 1
    F st;
    {
        const int ucol0 = ct * F::NQ * 4 + g.u_shift;           // logical column of LDS column 0
        const int sc = g2_src(ucol0 + tid, g.Cl, 0, g.Cl);
        // smallest source column of the tile (the reflection folds the edge tiles back)
        int smin = ucol0 < 0 ? 0 : ucol0;
        if (ucol0 + F::TW - 1 >= g.Cl) {
            const int m = 2 * g.Cl - 1 - (ucol0 + F::TW - 1);
            smin = m < smin ? (m < 0 ? 0 : m) : smin;
        }
        if (smin > g.Cl - 1) smin = g.Cl - 1;
        const int rec0 = smin >> 1;
        st.tid = tid;
        st.ridx = (sc >> 1) - rec0;
        st.par = sc & 1;
        const int nrec = C2 - rec0 < F::NREC ? C2 - rec0 : F::NREC;
        st.nvec = nrec * F::NV;
        st.u0 = (KIND == 0 ? rt * 8 : rt * 4) + g.u_shift;
        st.Rl = g.Rl; st.Cl = g.Cl;
        st.rstride = (size_t)C2 * 12;
        st.Zc = Zl + (size_t)b * g.Rl * g.Cl + sc;
        st.Yrow0 = Yh + ((size_t)b * (g.Rl >> 1) * C2 + rec0) * 12;
#pragma unroll
        for (int k = 0; k < 8; ++k) st.y1[k] = st.y2[k] = 0;
#pragma unroll
        for (int i = 0; i < F::NVT; ++i) st.rg[i] = typename F::V{};
        if (rt & 1) {
            st.issue(F::NP - 1);
            st.template run<0, true>(sm, tp, gn, g.f0, g.f1);
        } else {
            st.issue(0);
            st.template run<0, false>(sm, tp, gn, g.f0, g.f1);
        }
    }
    __syncthreads();            // every record read is done: y1 / y2 take the buffers over
    T *sY1 = sm, *sY2 = sm + 8 * F::STR;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sY1[k * F::STR + tid] = st.y1[k];
        sY2[k * F::STR + tid] = st.y2[k];
    }
    __syncthreads();
    // ---- pass 2: 8 rows x NQ threads
    const bool vec_ok = g.crop_c == 0 && ((g.Cout * sizeof(T)) & 15) == 0;
#pragma unroll
    for (int it = 0; it < (8 * F::NQ + 255) / 256; ++it) {
        const int task = it * 256 + tid;
        const int row = task / F::NQ, q = task - row * F::NQ;
        const int orow = rt * 8 + row - g.crop_r;
        if (row >= 8 || orow < 0 || orow >= g.Rout) continue;
        T acc[RG::OUTS];
#pragma unroll
        for (int k = 0; k < RG::OUTS; ++k) acc[k] = 0;
        {
            T w[RG::WIN];
            lds_window<T, RG::WIN>(sY1 + row * F::STR + q * 4, w);
            if constexpr (KIND == 0) fir_colfilter<T, 4, MB>(w, tp.a, acc);
            else fir_colifilt<T, 2, MB>(w, tp.a, tp.b, g.f0, acc);
        }
        {
            T w[RG::WIN];
            lds_window<T, RG::WIN>(sY2 + row * F::STR + q * 4, w);
            if constexpr (KIND == 0) fir_colfilter<T, 4, MB>(w, tp.b, acc);
            else fir_colifilt<T, 2, MB>(w, tp.c, tp.d, g.f1, acc);
        }
        const int lo0 = (ct * F::NQ + q) * RG::OUTS - g.crop_c;
        T *o = Z + ((size_t)b * g.Rout + orow) * g.Cout;
#pragma unroll
        for (int h = 0; h < RG::OUTS / 4; ++h) {
            const int w0 = lo0 + 4 * h;
            T v[4] = {acc[4 * h], acc[4 * h + 1], acc[4 * h + 2], acc[4 * h + 3]};
            if (w0 >= 0 && w0 + 4 <= g.Cout && vec_ok) store4(o + w0, v, 4, true);
            else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (w0 + t >= 0 && w0 + t < g.Cout) o[w0 + t] = v[t];
            }
        }
    }
}

// ---- single-axis building blocks on dense [B][n][C] arrays (1-D transform, 3-D axis passes) ----
// lo / hi filter pair ALONG the contiguous axis of [nlines][Cin] lines -> Y0, Y1 [nlines][Cout]
template <typename T, int KIND, int MB>
__global__ void __launch_bounds__(256) k_g2_rows_pair(const T *__restrict__ X, T *__restrict__ Y0,
                                                      T *__restrict__ Y1, P2Geo g, QTaps<T> tp) {
    using RG = RowGeo<KIND, MB, false>;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, q = tid & (g.tpl - 1), ln = tid >> g.tpl_log2;
    const int seg = blockIdx.x % g.nseg, line = (blockIdx.x / g.nseg) * (256 >> g.tpl_log2) + ln;
    const bool live = line < g.nlines;
    T *sm = reinterpret_cast<T *>(smem_raw) + (size_t)ln * g.region;
    const int ublk = seg * g.tpl * RG::IN_STEP + g.u_shift;
    {
        const T *const lines[1] = {X + (size_t)(live ? line : g.nlines - 1) * g.Cin};
        stage_lines<T, StageGeo<RG::IN_STEP, RG::WIN>::NLD, 1>(sm, g.stride, lines, q, g.tpl, g.span, ublk, g.L, g.pad_lo, g.Cin);
    }
    __syncthreads();
    const int col0 = (seg * g.tpl + q) * 4;
    const int valid = g.Cout - col0;
    if (!live || valid <= 0) return;
    T lo[4], hi[4];
    fwd_row_fir<T, KIND, MB, RG::WIN>(sm + q * RG::IN_STEP, tp, g, lo, hi);
    const bool vec_ok = ((g.Cout * sizeof(T)) & 15) == 0;
    store4(Y0 + (size_t)line * g.Cout + col0, lo, valid, vec_ok);
    if (g.crop && valid >= 4 && vec_ok) {       // crop doubles as "Y1 is a highpass: write-once" here
        using V = typename Vec16<T>::type;
        constexpr int VN = Vec16<T>::N;
#pragma unroll
        for (int j = 0; j < 4 / VN; ++j) {
            V x;
            T *e = reinterpret_cast<T *>(&x);
#pragma unroll
            for (int t = 0; t < VN; ++t) e[t] = hi[j * VN + t];
            stream_store<V>(reinterpret_cast<V *>(Y1 + (size_t)line * g.Cout + col0) + j, x);
        }
    } else {
        store4(Y1 + (size_t)line * g.Cout + col0, hi, valid, vec_ok);
    }
}

// Y = filter(X0, lo) + filter(X1, hi) down the rows of [B][R][C] arrays (lanes along C); X1 may be
// the interleaved complex form [B][R/2][C] of forward pass 1 (PACK).  8 output rows per thread.
struct S1Geo {
    int B, R, C;
    int Rout, crop, ngroups, u_shift, f0, f1, packed;
};

template <typename T, int KIND, int MB>
__global__ void __launch_bounds__(256) k_g2_sum_march(const T *__restrict__ X0, const T *__restrict__ X1,
                                                      T *__restrict__ Y, S1Geo g, QTaps<T> tp) {
    constexpr int WN = KIND == 0 ? 8 + MB - 1 : IfiltGeo<MB>::WN + 2;
    const unsigned id = blockIdx.x * 256u + threadIdx.x;
    const unsigned total = (unsigned)g.B * g.ngroups * g.C;
    if (id >= total) return;
    const unsigned t = id / g.C, c = id - t * g.C;
    const unsigned b = t / g.ngroups, grp = t - b * g.ngroups;
    const int u0 = (KIND == 0 ? grp * 8 : grp * 4) + g.u_shift;
    const T *P0 = X0 + (size_t)b * g.R * g.C + c;
    T acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0;
    {
        T w[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j) w[j] = P0[(size_t)g2_src(u0 + j, g.R, 0, g.R) * g.C];
        if constexpr (KIND == 0) fir_colfilter<T, 8, MB>(w, tp.a, acc);
        else fir_colifilt<T, 2, MB>(w, tp.a, tp.b, g.f0, acc);
    }
    {
        T w[WN];
        if (g.packed) {
            const T *P1 = X1 + ((size_t)b * (g.R >> 1) * g.C + c) * 2;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int r = g2_src(u0 + j, g.R, 0, g.R);
                w[j] = P1[(size_t)(r >> 1) * g.C * 2 + (r & 1)];
            }
        } else {
            const T *P1 = X1 + (size_t)b * g.R * g.C + c;
#pragma unroll
            for (int j = 0; j < WN; ++j) w[j] = P1[(size_t)g2_src(u0 + j, g.R, 0, g.R) * g.C];
        }
        if constexpr (KIND == 0) fir_colfilter<T, 8, MB>(w, tp.b, acc);
        else fir_colifilt<T, 2, MB>(w, tp.c, tp.d, g.f1, acc);
    }
    T *Ob = Y + (size_t)b * g.Rout * g.C + c;
    const int lo0 = (int)grp * 8 - g.crop;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (lo0 + k >= 0 && lo0 + k < g.Rout) Ob[(size_t)(lo0 + k) * g.C] = acc[k];
}

// ---- 3-D level 1, filter-by-filter form: the packings fused into the neighbouring axis pass ----
// (float64 and every other case without a fused 3-D tile program.)  Forward: the LAST axis pass
// (down axis 0) of one of the four (a1, a2) volumes computes its lo and hi octant for a 2 x 2
// block of columns and 4 rows = two 2x2x2 cells, and stores them packed (cube2c,
// dtcwt/numpy/transform3d.py:532-579) -- or plain, for the lowpass octant.  Inverse: the FIRST
// merge pass (along axis 1) unpacks (c2cube, :581-619) its two inputs on load.  Seven volumes less
// to write and read back per direction.
struct G3Geo {
    int n0, n1, n2;
    int ngroups;
    int u_shift;
    int o_lo, o_hi;     // octant slot (0..6) of the lo / hi side, -1 = plain volume
};

template <typename T>
__device__ inline void cube2c_store(T *rec, T A, T B, T C, T D, T E, T F, T G, T H) {
    using V = typename Vec16<T>::type;
    const T h = (T)0.5;
    T r[8] = {(A - G - D - F) * h, (B - H + C + E) * h, (A - G + D + F) * h, (-B + H + C + E) * h,
              (A + G + D - F) * h, (B + H - C + E) * h, (A + G - D + F) * h, (-B - H - C + E) * h};
    constexpr int VN = Vec16<T>::N;
#pragma unroll
    for (int j = 0; j < 8 / VN; ++j) {
        V v;
        T *e = reinterpret_cast<T *>(&v);
#pragma unroll
        for (int t = 0; t < VN; ++t) e[t] = r[j * VN + t];
        reinterpret_cast<V *>(rec)[j] = v;
    }
}

template <typename T, int MB>
__global__ void __launch_bounds__(256) k_g3_fwd_axis0_cube(const T *__restrict__ V, T *__restrict__ Plain,
                                                           T *__restrict__ Yh, G3Geo g, QTaps<T> tp) {
    using V2 = typename std::conditional<sizeof(T) == 4, float2, double2>::type;
    constexpr int G = 4, WN = G + MB - 1;
    const unsigned h1 = g.n1 >> 1, h2 = g.n2 >> 1;
    const unsigned id = blockIdx.x * 256u + threadIdx.x;
    if (id >= (unsigned)g.ngroups * h1 * h2) return;
    const unsigned t = id / h2, j2 = id - t * h2;
    const unsigned grp = t / h1, j1 = t - grp * h1;
    const size_t s0 = (size_t)g.n1 * g.n2;
    const int lo0 = grp * G, u0 = lo0 + g.u_shift;
    T lo[G][2][2], hi[G][2][2];             // [row][i1][i2]
#pragma unroll
    for (int i1 = 0; i1 < 2; ++i1) {
        const T *col = V + (size_t)(2 * j1 + i1) * g.n2 + 2 * j2;
        T w0[WN], w1[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const V2 v = *reinterpret_cast<const V2 *>(col + (size_t)g2_src(u0 + j, g.n0, 0, g.n0) * s0);
            w0[j] = v.x; w1[j] = v.y;
        }
        T a[G], b[G], c[G], d[G];
#pragma unroll
        for (int q = 0; q < G; ++q) a[q] = b[q] = c[q] = d[q] = 0;
        fir_colfilter<T, G, MB>(w0, tp.a, a);
        fir_colfilter<T, G, MB>(w0, tp.b, b);
        fir_colfilter<T, G, MB>(w1, tp.a, c);
        fir_colfilter<T, G, MB>(w1, tp.b, d);
#pragma unroll
        for (int q = 0; q < G; ++q) {
            lo[q][i1][0] = a[q]; hi[q][i1][0] = b[q];
            lo[q][i1][1] = c[q]; hi[q][i1][1] = d[q];
        }
    }
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const int o = side ? g.o_hi : g.o_lo;
        if (o < 0) {
#pragma unroll
            for (int q = 0; q < G; ++q)
                if (lo0 + q < g.n0) {
#pragma unroll
                    for (int i1 = 0; i1 < 2; ++i1) {
                        V2 v;
                        v.x = side ? hi[q][i1][0] : lo[q][i1][0];
                        v.y = side ? hi[q][i1][1] : lo[q][i1][1];
                        *reinterpret_cast<V2 *>(Plain + (size_t)(lo0 + q) * s0 + (size_t)(2 * j1 + i1) * g.n2 + 2 * j2) = v;
                    }
                }
        } else {
#pragma unroll
            for (int c = 0; c < G / 2; ++c) {
                const int u = (lo0 >> 1) + c;
                if (2 * u < g.n0) {
                    T *rec = Yh + (((size_t)u * h1 + j1) * h2 + j2) * 56 + o * 8;
#define X_(r_, i1_, i2_) (side ? hi[2 * c + r_][i1_][i2_] : lo[2 * c + r_][i1_][i2_])
                    cube2c_store<T>(rec, X_(0, 0, 0), X_(0, 1, 0), X_(1, 0, 0), X_(1, 1, 0), X_(0, 0, 1), X_(0, 1, 1),
                                    X_(1, 0, 1), X_(1, 1, 1));
#undef X_
                }
            }
        }
    }
}

// one side of the inverse merge along axis 1: the two image rows (2v, 2v+1) of the 2 x 2 column
// block (i0, i2) of record (u, v, w), plain volume or octant o of Yh
template <typename T>
__device__ inline void g3_load_rows(const T *__restrict__ P, const T *__restrict__ Yh, int o, const G3Geo &g,
                                    unsigned u, unsigned w, int r, T (&e)[4], T (&od)[4]) {
    using V2 = typename std::conditional<sizeof(T) == 4, float2, double2>::type;
    using V = typename Vec16<T>::type;
    constexpr int VN = Vec16<T>::N;
    const bool sw = r & 1;
    const int v = r >> 1;
    if (o < 0) {
        const size_t s0 = (size_t)g.n1 * g.n2;
        const T *b = P + (size_t)(2 * u) * s0 + (size_t)(2 * v) * g.n2 + 2 * w;
        const V2 x00 = *reinterpret_cast<const V2 *>(b), x01 = *reinterpret_cast<const V2 *>(b + g.n2);
        const V2 x10 = *reinterpret_cast<const V2 *>(b + s0), x11 = *reinterpret_cast<const V2 *>(b + s0 + g.n2);
        // columns (i0, i2): 0 = (0,0), 1 = (0,1), 2 = (1,0), 3 = (1,1)
        const T ev[4] = {x00.x, x00.y, x10.x, x10.y}, ov[4] = {x01.x, x01.y, x11.x, x11.y};
#pragma unroll
        for (int k = 0; k < 4; ++k) { e[k] = sw ? ov[k] : ev[k]; od[k] = sw ? ev[k] : ov[k]; }
        return;
    }
    const T *rec = Yh + ((((size_t)u * (g.n1 >> 1)) + v) * (g.n2 >> 1) + w) * 56 + o * 8;
    T z[8];
#pragma unroll
    for (int j = 0; j < 8 / VN; ++j) {
        const V x = reinterpret_cast<const V *>(rec)[j];
        const T *q = reinterpret_cast<const T *>(&x);
#pragma unroll
        for (int t = 0; t < VN; ++t) z[j * VN + t] = q[t];
    }
    const T h = (T)0.5;
    const T pr = z[0], pi = z[1], qr = z[2], qi = z[3], rr = z[4], ri = z[5], sr = z[6], si = z[7];
    const T A = (pr + qr + rr + sr) * h, G = (-pr - qr + rr + sr) * h, D = (-pr + qr + rr - sr) * h;
    const T F = (-pr + qr - rr + sr) * h, B = (pi - qi + ri - si) * h, H = (-pi + qi + ri - si) * h;
    const T C = (pi + qi - ri - si) * h, E = (pi + qi + ri + si) * h;
    // row 2v: A (0,.,0) E (0,.,1) C (1,.,0) G (1,.,1);  row 2v+1: B F D H
    const T ev[4] = {A, E, C, G}, ov[4] = {B, F, D, H};
#pragma unroll
    for (int k = 0; k < 4; ++k) { e[k] = sw ? ov[k] : ev[k]; od[k] = sw ? ev[k] : ov[k]; }
}

template <typename T, int MB>
struct G3Inv {
    static constexpr int NP = (8 + MB) / 2;
    T acc[4][8];
    template <int P>
    __device__ inline void run(const T *__restrict__ P0, const T *__restrict__ Yh, const G3Geo &g, unsigned u,
                               unsigned w, int u0, const QTaps<T> &tp) {
        if constexpr (P < NP) {
            int uu = u0 + 2 * P;
            asm volatile("" : "+v"(uu));
            const int r = g2_src(uu, g.n1, 0, g.n1);
            T e[4], o[4];
            g3_load_rows<T>(P0, Yh, g.o_lo, g, u, w, r, e, o);
#pragma unroll
            for (int k = 0; k < 4; ++k) scatter_colfilter<T, MB, P, 8>(tp.a, e[k], o[k], acc[k]);
            g3_load_rows<T>(nullptr, Yh, g.o_hi, g, u, w, r, e, o);
#pragma unroll
            for (int k = 0; k < 4; ++k) scatter_colfilter<T, MB, P, 8>(tp.b, e[k], o[k], acc[k]);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("" : "+v"(acc[0][k]), "+v"(acc[1][k]), "+v"(acc[2][k]), "+v"(acc[3][k]));
            run<P + 1>(P0, Yh, g, u, w, u0, tp);
        }
    }
};

template <typename T, int MB>
__global__ void __launch_bounds__(256) k_g3_inv_axis1_cube(const T *__restrict__ P0, const T *__restrict__ Yh,
                                                           T *__restrict__ Out, G3Geo g, QTaps<T> tp) {
    using V2 = typename std::conditional<sizeof(T) == 4, float2, double2>::type;
    const unsigned h0 = g.n0 >> 1, h2 = g.n2 >> 1;
    const unsigned id = blockIdx.x * 256u + threadIdx.x;
    if (id >= h0 * (unsigned)g.ngroups * h2) return;
    const unsigned t = id / h2, w = id - t * h2;
    const unsigned u = t / g.ngroups, grp = t - u * g.ngroups;
    G3Inv<T, MB> st;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 8; ++q) st.acc[k][q] = 0;
    st.template run<0>(P0, Yh, g, u, w, (int)grp * 8 + g.u_shift, tp);
    const size_t s0 = (size_t)g.n1 * g.n2;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = (int)grp * 8 + q;
        if (r < g.n1) {
#pragma unroll
            for (int i0 = 0; i0 < 2; ++i0) {
                V2 v; v.x = st.acc[2 * i0][q]; v.y = st.acc[2 * i0 + 1][q];
                *reinterpret_cast<V2 *>(Out + (size_t)(2 * u + i0) * s0 + (size_t)r * g.n2 + 2 * w) = v;
            }
        }
    }
}

// ---- host side -----------------------------------------------------------------------------
struct TapPrep {
    double a[G2_MAXB], b[G2_MAXB], c[G2_MAXB], d[G2_MAXB];
    int mb = 0, u_shift = 0, f0 = 0, f1 = 0;
};

double dotp(const double *x, const double *y, int m) {
    double s = 0;
    for (int k = 0; k < m; ++k) s += x[k] * y[k];
    return s;
}

// level-1 pair of odd lengths: both symmetric padded to the longer, `front` zeros in front,
// end padded to the bucket.  Window start of the output group at lo0: lo0 + u_shift.
bool prep_level1(const double *h0, int m0, const double *h1, int m1, bool even_start, TapPrep &p) {
    if (!(m0 & 1) || !(m1 & 1)) return false;
    const int mm = m0 > m1 ? m0 : m1, cc = (mm - 1) / 2;
    // smallest bucket (8 / 12 / 20) that holds the filters plus, where the window must start on an
    // even row, one zero in front
    int mb = 0, front = 0;
    for (int cand : {8, 12, 20}) {
        front = (even_start && ((cc - (cand - 1)) & 1)) ? 1 : 0;
        if (mm + front <= cand) { mb = cand; break; }
    }
    if (!mb) return false;
    for (int k = 0; k < G2_MAXB; ++k) p.a[k] = p.b[k] = p.c[k] = p.d[k] = 0;
    for (int k = 0; k < m0; ++k) p.a[front + (mm - m0) / 2 + k] = h0[k];
    for (int k = 0; k < m1; ++k) p.b[front + (mm - m1) / 2 + k] = h1[k];
    p.mb = mb;
    p.u_shift = cc + front - (mb - 1);
    return true;
}

// q-shift pairs for coldfilt: front padded to 10 / 14 / 18 / 20
bool prep_dfilt(const double *ha0, const double *hb0, const double *ha1, const double *hb1, int m, TapPrep &p) {
    if (m < 2 || (m & 1) || m > 20) return false;
    const int mb = m <= 10 ? 10 : (m <= 14 ? 14 : (m <= 18 ? 18 : 20));
    for (int k = 0; k < G2_MAXB; ++k) p.a[k] = p.b[k] = p.c[k] = p.d[k] = 0;
    for (int k = 0; k < m; ++k) {
        p.a[mb - m + k] = ha0[k]; p.b[mb - m + k] = hb0[k];
        p.c[mb - m + k] = ha1[k]; p.d[mb - m + k] = hb1[k];
    }
    p.mb = mb;
    p.u_shift = -m + 2;
    p.f0 = dotp(ha0, hb0, m) > 0 ? 1 : 0;
    p.f1 = dotp(ha1, hb1, m) > 0 ? 1 : 0;
    return true;
}

// q-shift pairs for colifilt: centre padded (an even number of zeros per side) to the bucket of
// the same m/2 parity: 10 / 18 (odd), 8 / 16 (even)
bool prep_ifilt(const double *ha0, const double *hb0, const double *ha1, const double *hb1, int m, TapPrep &p) {
    if (m < 2 || (m & 1)) return false;
    const int mb = ((m / 2) & 1) ? (m <= 10 ? 10 : (m <= 18 ? 18 : 0)) : (m <= 8 ? 8 : (m <= 16 ? 16 : 0));
    if (!mb) return false;
    const int s = (mb - m) / 2;
    for (int k = 0; k < G2_MAXB; ++k) p.a[k] = p.b[k] = p.c[k] = p.d[k] = 0;
    for (int k = 0; k < m; ++k) {
        p.a[s + k] = ha0[k]; p.b[s + k] = hb0[k];
        p.c[s + k] = ha1[k]; p.d[s + k] = hb1[k];
    }
    p.mb = mb;
    const int m2 = mb / 2;
    p.u_shift = (m2 & 1) ? 1 - m2 : -m2;
    p.f0 = dotp(ha0, hb0, m) > 0 ? 1 : 0;
    p.f1 = dotp(ha1, hb1, m) > 0 ? 1 : 0;
    return true;
}

template <typename T>
QTaps<T> to_device_taps(const TapPrep &p) {
    QTaps<T> t;
    for (int k = 0; k < G2_MAXB; ++k) {
        t.a[k] = (T)p.a[k]; t.b[k] = (T)p.b[k]; t.c[k] = (T)p.c[k]; t.d[k] = (T)p.d[k];
    }
    return t;
}

// threads per line: the power of two in [16, 256] that covers `need` threads with least waste
void choose_tpl(int need, int &tpl, int &lg) {
    lg = 4;
    while (lg < 8 && (1 << lg) < need) ++lg;
    tpl = 1 << lg;
}

// fill the LDS-row geometry; returns the dynamic LDS bytes
template <typename T>
size_t finish_rows(P2Geo &g, int in_step, int win, int threads_per_line, int rec_elems_per_thread, int lines = 2) {
    choose_tpl(threads_per_line, g.tpl, g.tpl_log2);
    g.nseg = (threads_per_line + g.tpl - 1) / g.tpl;
    g.span = g.tpl * in_step + win - in_step;
    g.stride = g.span + 4;
    const int a = lines * g.stride, b = g.tpl * rec_elems_per_thread;
    g.region = a > b ? a : b;
    return (size_t)(256 / g.tpl) * g.region * sizeof(T);
}

constexpr int G2_NA = -3;       // "use the filter-by-filter path"

}  // namespace

#define G2_LAUNCH_CHECK() DT_CHECK_HIP(hipGetLastError())

// KIND 0 buckets 8 / 12 / 20; coldfilt buckets 10 / 14 / 18 / 20; colifilt buckets 8 / 10 / 16 / 18
#define G2_SWITCH_FWD(KERNEL, T_, ...)                                                     \
    do {                                                                                   \
        if (kind == 0) {                                                                   \
            if (p.mb == 8) KERNEL<T_, 0, 8> __VA_ARGS__;                                    \
            else if (p.mb == 12) KERNEL<T_, 0, 12> __VA_ARGS__;                             \
            else KERNEL<T_, 0, 20> __VA_ARGS__;                                             \
        } else {                                                                           \
            if (p.mb == 10) KERNEL<T_, 1, 10> __VA_ARGS__;                                  \
            else if (p.mb == 14) KERNEL<T_, 1, 14> __VA_ARGS__;                             \
            else if (p.mb == 18) KERNEL<T_, 1, 18> __VA_ARGS__;                             \
            else KERNEL<T_, 1, 20> __VA_ARGS__;                                             \
        }                                                                                  \
    } while (0)

#define G2_SWITCH_INV(KERNEL, T_, ...)                                                     \
    do {                                                                                   \
        if (kind == 0) {                                                                   \
            if (p.mb == 8) KERNEL<T_, 0, 8> __VA_ARGS__;                                    \
            else if (p.mb == 12) KERNEL<T_, 0, 12> __VA_ARGS__;                             \
            else KERNEL<T_, 0, 20> __VA_ARGS__;                                             \
        } else {                                                                           \
            if (p.mb == 8) KERNEL<T_, 1, 8> __VA_ARGS__;                                    \
            else if (p.mb == 10) KERNEL<T_, 1, 10> __VA_ARGS__;                             \
            else if (p.mb == 16) KERNEL<T_, 1, 16> __VA_ARGS__;                             \
            else KERNEL<T_, 1, 18> __VA_ARGS__;                                             \
        }                                                                                  \
    } while (0)

// ---- dense single-axis pair / sum (declared in common.hpp) -----------------------------------
// Return 1 when a kernel was launched, 0 when this geometry / these filters have none (caller
// uses the one-output-per-thread kernels of filters.hip), < 0 on error.
int dtcwt_g2_pair(dtcwt_hip_ctx *ctx, int dtype, int kind, const void *X, void *Y0, void *Y1,
                  int64_t outer, int64_t n, int64_t inner, int pad_lo, int pad_hi,
                  const double *lo_a, const double *lo_b, const double *hi_a, const double *hi_b,
                  int m_lo, int m_hi, int pack_hi) {
    TapPrep p;
    if (kind == 0) {
        if (!prep_level1(lo_a, m_lo, hi_a, m_hi, false, p)) return 0;
    } else {
        if (m_lo != m_hi || !prep_dfilt(lo_a, lo_b, hi_a, hi_b, m_lo, p)) return 0;
    }
    const int64_t L = n + pad_lo + pad_hi, nout = kind == 0 ? L : L / 2;
    if (kind == 1 && (L & 3)) return 0;
    if (L < p.mb + 4 || outer < 1) return 0;
    if (pack_hi && (nout & 1)) return 0;
    if (outer * L * inner >= ((int64_t)1 << 31)) return 0;
    if (inner >= 32) {
        P1Geo g;
        g.B = (int)outer; g.R = (int)n; g.C = (int)inner; g.pad_lo = pad_lo; g.L = (int)L; g.nout = (int)nout;
        g.ngroups = kind == 0 ? (int)((nout + 7) / 8) : (int)((nout / 2 + 3) / 4);
        g.u_shift = p.u_shift; g.af0 = p.f0; g.af1 = p.f1; g.pack = pack_hi;
        const unsigned blocks = (unsigned)(((int64_t)g.B * g.ngroups * g.C + 255) / 256);
        if (dtype == DTCWT_HIP_F32) {
            QTaps<float> t = to_device_taps<float>(p);
            G2_SWITCH_FWD(k_g2_fwd_p1, float, <<<blocks, 256, 0, ctx->stream>>>((const float *)X, (float *)Y0, (float *)Y1, g, t));
        } else {
            QTaps<double> t = to_device_taps<double>(p);
            G2_SWITCH_FWD(k_g2_fwd_p1, double, <<<blocks, 256, 0, ctx->stream>>>((const double *)X, (double *)Y0, (double *)Y1, g, t));
        }
        G2_LAUNCH_CHECK();
        return 1;
    }
    if (inner == 1) {       // the filter axis is the contiguous one: LDS rows (interleaved hi = hi itself)
        P2Geo g;
        g.nlines = (int)outer; g.R1 = 0; g.Cin = (int)n; g.pad_lo = pad_lo; g.L = (int)L; g.Cout = (int)nout;
        g.crop = pack_hi; g.u_shift = p.u_shift; g.f0 = p.f0; g.f1 = p.f1;
        const int in_step = kind == 0 ? 4 : 8;
        const int win = kind == 0 ? (4 + p.mb - 1 + 3) / 4 * 4 : 4 + 2 * p.mb;
        const size_t lds = dtype == DTCWT_HIP_F32 ? finish_rows<float>(g, in_step, win, (int)((nout + 3) / 4), 0, 1)
                                                  : finish_rows<double>(g, in_step, win, (int)((nout + 3) / 4), 0, 1);
        const int lpb = 256 / g.tpl;
        const unsigned blocks = (unsigned)(((g.nlines + lpb - 1) / lpb) * g.nseg);
        if (dtype == DTCWT_HIP_F32) {
            QTaps<float> t = to_device_taps<float>(p);
            G2_SWITCH_FWD(k_g2_rows_pair, float, <<<blocks, 256, lds, ctx->stream>>>((const float *)X, (float *)Y0, (float *)Y1, g, t));
        } else {
            QTaps<double> t = to_device_taps<double>(p);
            G2_SWITCH_FWD(k_g2_rows_pair, double, <<<blocks, 256, lds, ctx->stream>>>((const double *)X, (double *)Y0, (double *)Y1, g, t));
        }
        G2_LAUNCH_CHECK();
        return 1;
    }
    return 0;
}

int dtcwt_g2_sum(dtcwt_hip_ctx *ctx, int dtype, int kind, const void *X0, const void *X1, void *Y,
                 int64_t outer, int64_t n, int64_t inner, int crop, const double *lo_a,
                 const double *lo_b, const double *hi_a, const double *hi_b, int m_lo, int m_hi,
                 int packed_x1, double gain1) {
    TapPrep p;
    if (kind == 0) {
        if (crop || !prep_level1(lo_a, m_lo, hi_a, m_hi, false, p)) return 0;
        for (int k = 0; k < G2_MAXB; ++k) p.b[k] *= gain1;
    } else {
        if (m_lo != m_hi || !prep_ifilt(lo_a, lo_b, hi_a, hi_b, m_lo, p)) return 0;
        for (int k = 0; k < G2_MAXB; ++k) { p.c[k] *= gain1; p.d[k] *= gain1; }
    }
    if ((n & 1) || n < p.mb + 4 || outer < 1) return 0;
    const int64_t nout = (kind == 0 ? n : 2 * n) - 2 * crop;
    if (nout < 1 || outer * nout * inner >= ((int64_t)1 << 31)) return 0;
    if (inner >= 32) {
        S1Geo g;
        g.B = (int)outer; g.R = (int)n; g.C = (int)inner; g.Rout = (int)nout; g.crop = crop;
        g.ngroups = kind == 0 ? (int)((n + 7) / 8) : (int)((n / 2 + 1) / 2);
        g.u_shift = p.u_shift; g.f0 = p.f0; g.f1 = p.f1; g.packed = packed_x1;
        const unsigned blocks = (unsigned)(((int64_t)g.B * g.ngroups * g.C + 255) / 256);
        if (dtype == DTCWT_HIP_F32) {
            QTaps<float> t = to_device_taps<float>(p);
            G2_SWITCH_INV(k_g2_sum_march, float, <<<blocks, 256, 0, ctx->stream>>>((const float *)X0, (const float *)X1, (float *)Y, g, t));
        } else {
            QTaps<double> t = to_device_taps<double>(p);
            G2_SWITCH_INV(k_g2_sum_march, double, <<<blocks, 256, 0, ctx->stream>>>((const double *)X0, (const double *)X1, (double *)Y, g, t));
        }
        G2_LAUNCH_CHECK();
        return 1;
    }
    if (inner == 1) {
        P2Geo g;
        g.nlines = (int)outer; g.R1 = 0; g.Cin = (int)n; g.pad_lo = 0; g.L = (int)n; g.Cout = (int)nout;
        g.crop = crop; g.u_shift = p.u_shift; g.f0 = p.f0; g.f1 = p.f1;
        const int win = kind == 0 ? (4 + p.mb - 1 + 3) / 4 * 4 : ((((p.mb / 2) & 1) ? p.mb : p.mb + 2) + 2);
        const int need = kind == 0 ? (int)((n + 3) / 4) : (int)((n / 2 + 1) / 2);
        const size_t lds = dtype == DTCWT_HIP_F32 ? finish_rows<float>(g, 4, win, need, 0)
                                                  : finish_rows<double>(g, 4, win, need, 0);
        const int lpb = 256 / g.tpl;
        const unsigned blocks = (unsigned)(((g.nlines + lpb - 1) / lpb) * g.nseg);
        if (dtype == DTCWT_HIP_F32) {
            QTaps<float> t = to_device_taps<float>(p);
            G2_SWITCH_INV(k_g2_inv_p2, float, <<<blocks, 256, lds, ctx->stream>>>((const float *)X0, (const float *)X1, (float *)Y, g, t));
        } else {
            QTaps<double> t = to_device_taps<double>(p);
            G2_SWITCH_INV(k_g2_inv_p2, double, <<<blocks, 256, lds, ctx->stream>>>((const double *)X0, (const double *)X1, (double *)Y, g, t));
        }
        G2_LAUNCH_CHECK();
        return 1;
    }
    return 0;
}

extern "C" {

int dtcwt_hip_level2d_forward(dtcwt_hip_ctx *ctx, int dtype, int kind, const void *X, int64_t B,
                              int64_t R, int64_t C, int pad_r_lo, int pad_r_hi, int pad_c_lo,
                              int pad_c_hi, const double *lo_a, const double *lo_b,
                              const double *hi_a, const double *hi_b, int m_lo, int m_hi,
                              void *Lo, void *Hi, void *LoLo, void *Yh) {
    DT_REQUIRE(ctx && X && Lo && Hi && LoLo && Yh && lo_a && hi_a, "NULL argument");
    DT_REQUIRE(dtype == DTCWT_HIP_F32 || dtype == DTCWT_HIP_F64, "bad dtype %d", dtype);
    DT_REQUIRE(kind == 0 || kind == 1, "bad kind %d", kind);
    DT_REQUIRE(B >= 1 && R >= 1 && C >= 1, "bad extents");
    TapPrep p;
    if (kind == 0) {
        if (!prep_level1(lo_a, m_lo, hi_a, m_hi, false, p)) return G2_NA;
    } else {
        DT_REQUIRE(lo_b && hi_b, "NULL argument");
        if (m_lo != m_hi || !prep_dfilt(lo_a, lo_b, hi_a, hi_b, m_lo, p)) return G2_NA;
    }
    const int64_t LR = R + pad_r_lo + pad_r_hi, LC = C + pad_c_lo + pad_c_hi;
    const int64_t R1 = kind == 0 ? LR : LR / 2, C1 = kind == 0 ? LC : LC / 2;
    if (kind == 1 && ((LR & 3) || (LC & 3))) return G2_NA;
    if ((R1 & 1) || (C1 & 1)) return G2_NA;
    const int reach = p.mb + 4;      // every sample a written output needs is one bounce away
    if (LR < reach || LC < reach) return G2_NA;
    if (B * LR * LC >= ((int64_t)1 << 31) || B * R1 * C1 * 6 >= ((int64_t)1 << 31)) return G2_NA;
    DT_CHECK_HIP(hipSetDevice(ctx->device));

    if (!getenv("DTCWT_HIP_TWO_PASS")) {        // pass 1 -> LDS -> pass 2 in one launch; Lo / Hi stay unused
        FGeo f;
        f.B = (int)B; f.R = (int)R; f.C = (int)C; f.pad_r_lo = pad_r_lo; f.LR = (int)LR; f.R1 = (int)R1;
        f.pad_c_lo = pad_c_lo; f.LC = (int)LC; f.C1 = (int)C1; f.u_shift = p.u_shift; f.f0 = p.f0; f.f1 = p.f1;
        f.nrt = (int)((R1 + 7) / 8);
#define G2_FUSED_LAUNCH(T_, KIND_, MB_)                                                                   \
        do {                                                                                              \
            f.nct = (int)((C1 + 4 * FusedFwd<T_, KIND_, MB_>::NQ - 1) / (4 * FusedFwd<T_, KIND_, MB_>::NQ));    \
            k_g2_fwd_fused<T_, KIND_, MB_><<<(unsigned)((int64_t)f.B * f.nrt * f.nct), 256, 0, ctx->stream>>>( \
                (const T_ *)X, (T_ *)LoLo, (T_ *)Yh, f, to_device_taps<T_>(p));                           \
        } while (0)
        if (dtype == DTCWT_HIP_F32) {
            if (kind == 0) { if (p.mb == 8) G2_FUSED_LAUNCH(float, 0, 8); else if (p.mb == 12) G2_FUSED_LAUNCH(float, 0, 12); else G2_FUSED_LAUNCH(float, 0, 20); }
            else { if (p.mb == 10) G2_FUSED_LAUNCH(float, 1, 10); else if (p.mb == 14) G2_FUSED_LAUNCH(float, 1, 14); else if (p.mb == 18) G2_FUSED_LAUNCH(float, 1, 18); else G2_FUSED_LAUNCH(float, 1, 20); }
        } else {
            if (kind == 0) { if (p.mb == 8) G2_FUSED_LAUNCH(double, 0, 8); else if (p.mb == 12) G2_FUSED_LAUNCH(double, 0, 12); else G2_FUSED_LAUNCH(double, 0, 20); }
            else { if (p.mb == 10) G2_FUSED_LAUNCH(double, 1, 10); else if (p.mb == 14) G2_FUSED_LAUNCH(double, 1, 14); else if (p.mb == 18) G2_FUSED_LAUNCH(double, 1, 18); else G2_FUSED_LAUNCH(double, 1, 20); }
        }
#undef G2_FUSED_LAUNCH
        G2_LAUNCH_CHECK();
        return 0;
    }

    P1Geo g1;
    g1.B = (int)B; g1.R = (int)R; g1.C = (int)C;
    g1.pad_lo = pad_r_lo; g1.L = (int)LR; g1.nout = (int)R1;
    g1.ngroups = kind == 0 ? (int)((R1 + 7) / 8) : (int)((R1 / 2 + 3) / 4);
    g1.u_shift = p.u_shift; g1.af0 = p.f0; g1.af1 = p.f1; g1.pack = 0;
    const unsigned blocks1 = (unsigned)(((int64_t)g1.B * g1.ngroups * g1.C + 255) / 256);

    P2Geo g2;
    g2.nlines = (int)(B * R1 / 2); g2.R1 = (int)R1; g2.Cin = (int)C;
    g2.pad_lo = pad_c_lo; g2.L = (int)LC; g2.Cout = (int)C1; g2.crop = 0;
    g2.u_shift = p.u_shift; g2.f0 = p.f0; g2.f1 = p.f1;
    const int in_step = kind == 0 ? 4 : 8;
    const int win = kind == 0 ? (4 + p.mb - 1 + 3) / 4 * 4 : 4 + 2 * p.mb;
    size_t lds;
    const int lines = kind == 0 ? 4 : 2;        // rows staged at once (k_g2_fwd_p2 BOTH)
    if (dtype == DTCWT_HIP_F32) lds = finish_rows<float>(g2, in_step, win, (int)((C1 + 3) / 4), 24, lines);
    else lds = finish_rows<double>(g2, in_step, win, (int)((C1 + 3) / 4), 24, lines);
    const int rpb = 256 / g2.tpl;
    const unsigned blocks2 = (unsigned)(((g2.nlines + rpb - 1) / rpb) * g2.nseg);

    if (dtype == DTCWT_HIP_F32) {
        QTaps<float> t = to_device_taps<float>(p);
        G2_SWITCH_FWD(k_g2_fwd_p1, float, <<<blocks1, 256, 0, ctx->stream>>>((const float *)X, (float *)Lo, (float *)Hi, g1, t));
        G2_LAUNCH_CHECK();
        G2_SWITCH_FWD(k_g2_fwd_p2, float, <<<blocks2, 256, lds, ctx->stream>>>((const float *)Lo, (const float *)Hi, (float *)LoLo, (float *)Yh, g2, t));
    } else {
        QTaps<double> t = to_device_taps<double>(p);
        G2_SWITCH_FWD(k_g2_fwd_p1, double, <<<blocks1, 256, 0, ctx->stream>>>((const double *)X, (double *)Lo, (double *)Hi, g1, t));
        G2_LAUNCH_CHECK();
        G2_SWITCH_FWD(k_g2_fwd_p2, double, <<<blocks2, 256, lds, ctx->stream>>>((const double *)Lo, (const double *)Hi, (double *)LoLo, (double *)Yh, g2, t));
    }
    G2_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_level2d_inverse(dtcwt_hip_ctx *ctx, int dtype, int kind, const void *Zl, const void *Yh,
                              int64_t B, int64_t Rl, int64_t Cl, const double *gains6,
                              int crop_r, int crop_c, const double *lo_a, const double *lo_b,
                              const double *hi_a, const double *hi_b, int m_lo, int m_hi,
                              void *Y1, void *Y2, void *Z) {
    DT_REQUIRE(ctx && Zl && Yh && Y1 && Y2 && Z && lo_a && hi_a && gains6, "NULL argument");
    DT_REQUIRE(dtype == DTCWT_HIP_F32 || dtype == DTCWT_HIP_F64, "bad dtype %d", dtype);
    DT_REQUIRE(kind == 0 || kind == 1, "bad kind %d", kind);
    DT_REQUIRE(B >= 1 && Rl >= 2 && Cl >= 2 && crop_r >= 0 && crop_c >= 0, "bad extents");
    if ((Rl & 1) || (Cl & 1)) return G2_NA;
    TapPrep p;
    if (kind == 0) {
        if (crop_r || crop_c) return G2_NA;
        if (!prep_level1(lo_a, m_lo, hi_a, m_hi, true, p)) return G2_NA;
    } else {
        DT_REQUIRE(lo_b && hi_b, "NULL argument");
        if (m_lo != m_hi || !prep_ifilt(lo_a, lo_b, hi_a, hi_b, m_lo, p)) return G2_NA;
    }
    const int reach = p.mb + 4;      // every sample a written output needs is one bounce away
    if (Rl < reach || Cl < reach) return G2_NA;
    const int64_t Rout = (kind == 0 ? Rl : 2 * Rl) - 2 * crop_r, Cout = (kind == 0 ? Cl : 2 * Cl) - 2 * crop_c;
    if (B * Rout * Cout >= ((int64_t)1 << 31) || B * Rl * Cl * 3 >= ((int64_t)1 << 31)) return G2_NA;
    DT_CHECK_HIP(hipSetDevice(ctx->device));

    if (!getenv("DTCWT_HIP_TWO_PASS")) {        // pass 1 -> LDS -> pass 2 in one launch; Y1 / Y2 stay unused
        IGeo f;
        f.B = (int)B; f.Rl = (int)Rl; f.Cl = (int)Cl; f.Rout = (int)Rout; f.Cout = (int)Cout;
        f.crop_r = crop_r; f.crop_c = crop_c; f.u_shift = p.u_shift; f.f0 = p.f0; f.f1 = p.f1;
        f.nrt = kind == 0 ? (int)((Rl + 7) / 8) : (int)((Rl + 3) / 4);
        const int need = kind == 0 ? (int)((Cl + 3) / 4) : (int)((Cl / 2 + 1) / 2);      // row-pass threads per row
        const double s2 = 0.70710678118654752440;
#define G2_FUSED_LAUNCH(T_, KIND_, MB_)                                                                   \
        do {                                                                                              \
            Gains<T_> gn;                                                                                 \
            for (int k = 0; k < 6; ++k) gn.g[k] = (T_)(s2 * gains6[k]);                                   \
            f.nct = (need + FusedInv<T_, KIND_, MB_>::NQ - 1) / FusedInv<T_, KIND_, MB_>::NQ;              \
            k_g2_inv_fused<T_, KIND_, MB_><<<(unsigned)((int64_t)f.B * f.nrt * f.nct), 256, 0, ctx->stream>>>( \
                (const T_ *)Zl, (const T_ *)Yh, (T_ *)Z, f, to_device_taps<T_>(p), gn);                   \
        } while (0)
#define G2_FUSED_KINDS(T_)                                                                                \
        do {                                                                                              \
            if (kind == 0) { if (p.mb == 8) G2_FUSED_LAUNCH(T_, 0, 8); else if (p.mb == 12) G2_FUSED_LAUNCH(T_, 0, 12); else G2_FUSED_LAUNCH(T_, 0, 20); }  \
            else if (p.mb == 8) G2_FUSED_LAUNCH(T_, 1, 8);                                                \
            else if (p.mb == 10) G2_FUSED_LAUNCH(T_, 1, 10);                                              \
            else if (p.mb == 16) G2_FUSED_LAUNCH(T_, 1, 16);                                              \
            else G2_FUSED_LAUNCH(T_, 1, 18);                                                              \
        } while (0)
        if (dtype == DTCWT_HIP_F32) G2_FUSED_KINDS(float); else G2_FUSED_KINDS(double);
#undef G2_FUSED_KINDS
#undef G2_FUSED_LAUNCH
        G2_LAUNCH_CHECK();
        return 0;
    }

    I1Geo g1;
    g1.B = (int)B; g1.Rl = (int)Rl; g1.C = (int)Cl;
    g1.Rout = (int)Rout; g1.crop = crop_r;
    g1.ngroups = kind == 0 ? (int)((Rl + 7) / 8) : (int)((2 * Rl + 15) / 16);      // InvP1Outs rows per group
    g1.u_shift = p.u_shift; g1.f0 = p.f0; g1.f1 = p.f1;
    g1.bpr = (int)((Cl / 2 + 255) / 256);
    const unsigned blocks1 = (unsigned)((int64_t)g1.B * g1.ngroups * g1.bpr);

    P2Geo g2;
    g2.nlines = (int)(B * Rout); g2.R1 = 0; g2.Cin = (int)Cl; g2.pad_lo = 0; g2.L = (int)Cl;
    g2.Cout = (int)Cout; g2.crop = crop_c;
    // pass 2 has no parity constraint on its window start: level 1 reuses the pass-1 padding
    g2.u_shift = p.u_shift; g2.f0 = p.f0; g2.f1 = p.f1;
    const int win = kind == 0 ? (4 + p.mb - 1 + 3) / 4 * 4 : ((((p.mb / 2) & 1) ? p.mb : p.mb + 2) + 2);
    const int tpl_need = kind == 0 ? (int)((Cl + 3) / 4) : (int)((Cl / 2 + 1) / 2);
    size_t lds;
    if (dtype == DTCWT_HIP_F32) lds = finish_rows<float>(g2, 4, win, tpl_need, 0);
    else lds = finish_rows<double>(g2, 4, win, tpl_need, 0);
    const int lpb = 256 / g2.tpl;
    const unsigned blocks2 = (unsigned)(((g2.nlines + lpb - 1) / lpb) * g2.nseg);
    const double s = 0.70710678118654752440;

    if (dtype == DTCWT_HIP_F32) {
        QTaps<float> t = to_device_taps<float>(p);
        Gains<float> gn;
        for (int k = 0; k < 6; ++k) gn.g[k] = (float)(s * gains6[k]);
        G2_SWITCH_INV(k_g2_inv_p1, float, <<<blocks1, 256, 0, ctx->stream>>>((const float *)Zl, (const float *)Yh, (float *)Y1, (float *)Y2, g1, t, gn));
        G2_LAUNCH_CHECK();
        G2_SWITCH_INV(k_g2_inv_p2, float, <<<blocks2, 256, lds, ctx->stream>>>((const float *)Y1, (const float *)Y2, (float *)Z, g2, t));
    } else {
        QTaps<double> t = to_device_taps<double>(p);
        Gains<double> gn;
        for (int k = 0; k < 6; ++k) gn.g[k] = s * gains6[k];
        G2_SWITCH_INV(k_g2_inv_p1, double, <<<blocks1, 256, 0, ctx->stream>>>((const double *)Zl, (const double *)Yh, (double *)Y1, (double *)Y2, g1, t, gn));
        G2_LAUNCH_CHECK();
        G2_SWITCH_INV(k_g2_inv_p2, double, <<<blocks2, 256, lds, ctx->stream>>>((const double *)Y1, (const double *)Y2, (double *)Z, g2, t));
    }
    G2_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_level1d_forward(dtcwt_hip_ctx *ctx, int dtype, int kind, const void *X, int64_t n,
                              int64_t k, int pad_lo, int pad_hi, const double *lo_a, const double *lo_b,
                              const double *hi_a, const double *hi_b, int m_lo, int m_hi, void *Lo,
                              void *Yh) {
    DT_REQUIRE(ctx && X && Lo && Yh && lo_a && hi_a, "NULL argument");
    DT_REQUIRE(dtype == DTCWT_HIP_F32 || dtype == DTCWT_HIP_F64, "bad dtype %d", dtype);
    DT_REQUIRE(kind == 0 || (kind == 1 && lo_b && hi_b), "bad kind %d", kind);
    DT_REQUIRE(n >= 1 && k >= 1 && pad_lo >= 0 && pad_hi >= 0, "bad extents");
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    const int rc = dtcwt_g2_pair(ctx, dtype, kind, X, Lo, Yh, 1, n, k, pad_lo, pad_hi, lo_a, lo_b, hi_a, hi_b,
                                 m_lo, m_hi, 1);
    return rc < 0 ? rc : (rc == 1 ? 0 : G2_NA);
}

int dtcwt_hip_level1d_inverse(dtcwt_hip_ctx *ctx, int dtype, int kind, const void *Lo, const void *Yh,
                              int64_t n, int64_t k, double gain, int crop, const double *lo_a,
                              const double *lo_b, const double *hi_a, const double *hi_b, int m_lo,
                              int m_hi, void *Z) {
    DT_REQUIRE(ctx && Lo && Yh && Z && lo_a && hi_a, "NULL argument");
    DT_REQUIRE(dtype == DTCWT_HIP_F32 || dtype == DTCWT_HIP_F64, "bad dtype %d", dtype);
    DT_REQUIRE(kind == 0 || (kind == 1 && lo_b && hi_b), "bad kind %d", kind);
    DT_REQUIRE(n >= 2 && k >= 1 && crop >= 0, "bad extents");
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    const int rc = dtcwt_g2_sum(ctx, dtype, kind, Lo, Yh, Z, 1, n, k, crop, lo_a, lo_b, hi_a, hi_b, m_lo, m_hi,
                                1, gain);
    return rc < 0 ? rc : (rc == 1 ? 0 : G2_NA);
}

int dtcwt_hip_fwd3_axis0_cube2c(dtcwt_hip_ctx *ctx, int dtype, const void *V, int64_t n0, int64_t n1,
                                int64_t n2, const double *h0, int m0, const double *h1, int m1,
                                int octant_lo, int octant_hi, void *plain, void *Yh) {
    DT_REQUIRE(ctx && V && Yh && h0 && h1, "NULL argument");
    DT_REQUIRE(dtype == DTCWT_HIP_F32 || dtype == DTCWT_HIP_F64, "bad dtype %d", dtype);
    DT_REQUIRE(octant_lo >= -1 && octant_lo < 7 && octant_hi >= -1 && octant_hi < 7, "bad octant");
    DT_REQUIRE((octant_lo >= 0 && octant_hi >= 0) || plain, "a plain output needs a buffer");
    TapPrep p;
    if (!prep_level1(h0, m0, h1, m1, false, p)) return G2_NA;
    if ((n0 & 1) || (n1 & 1) || (n2 & 1) || n0 < p.mb + 4 || n1 < 2 || n2 < 2) return G2_NA;
    if (n0 * n1 * n2 >= ((int64_t)1 << 31)) return G2_NA;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    G3Geo g;
    g.n0 = (int)n0; g.n1 = (int)n1; g.n2 = (int)n2; g.ngroups = (int)((n0 + 3) / 4); g.u_shift = p.u_shift;
    g.o_lo = octant_lo; g.o_hi = octant_hi;
    const unsigned blocks = (unsigned)(((int64_t)g.ngroups * (n1 / 2) * (n2 / 2) + 255) / 256);
    if (dtype == DTCWT_HIP_F32) {
        QTaps<float> t = to_device_taps<float>(p);
        if (p.mb == 8) k_g3_fwd_axis0_cube<float, 8><<<blocks, 256, 0, ctx->stream>>>((const float *)V, (float *)plain, (float *)Yh, g, t);
        else if (p.mb == 12) k_g3_fwd_axis0_cube<float, 12><<<blocks, 256, 0, ctx->stream>>>((const float *)V, (float *)plain, (float *)Yh, g, t);
        else k_g3_fwd_axis0_cube<float, 20><<<blocks, 256, 0, ctx->stream>>>((const float *)V, (float *)plain, (float *)Yh, g, t);
    } else {
        QTaps<double> t = to_device_taps<double>(p);
        if (p.mb == 8) k_g3_fwd_axis0_cube<double, 8><<<blocks, 256, 0, ctx->stream>>>((const double *)V, (double *)plain, (double *)Yh, g, t);
        else if (p.mb == 12) k_g3_fwd_axis0_cube<double, 12><<<blocks, 256, 0, ctx->stream>>>((const double *)V, (double *)plain, (double *)Yh, g, t);
        else k_g3_fwd_axis0_cube<double, 20><<<blocks, 256, 0, ctx->stream>>>((const double *)V, (double *)plain, (double *)Yh, g, t);
    }
    G2_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_inv3_axis1_c2cube(dtcwt_hip_ctx *ctx, int dtype, const void *plain, const void *Yh,
                                int64_t n0, int64_t n1, int64_t n2, const double *g0, int m0,
                                const double *g1, int m1, int octant_lo, int octant_hi, void *out) {
    DT_REQUIRE(ctx && Yh && out && g0 && g1, "NULL argument");
    DT_REQUIRE(dtype == DTCWT_HIP_F32 || dtype == DTCWT_HIP_F64, "bad dtype %d", dtype);
    DT_REQUIRE(octant_lo >= -1 && octant_lo < 7 && octant_hi >= 0 && octant_hi < 7, "bad octant");
    DT_REQUIRE(octant_lo >= 0 || plain, "a plain input needs a buffer");
    TapPrep p;
    if (!prep_level1(g0, m0, g1, m1, true, p)) return G2_NA;
    if ((n0 & 1) || (n1 & 1) || (n2 & 1) || n1 < p.mb + 4 || n0 < 2 || n2 < 2) return G2_NA;
    if (n0 * n1 * n2 >= ((int64_t)1 << 31)) return G2_NA;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    G3Geo g;
    g.n0 = (int)n0; g.n1 = (int)n1; g.n2 = (int)n2; g.ngroups = (int)((n1 + 7) / 8); g.u_shift = p.u_shift;
    g.o_lo = octant_lo; g.o_hi = octant_hi;
    const unsigned blocks = (unsigned)(((int64_t)(n0 / 2) * g.ngroups * (n2 / 2) + 255) / 256);
    if (dtype == DTCWT_HIP_F32) {
        QTaps<float> t = to_device_taps<float>(p);
        if (p.mb == 8) k_g3_inv_axis1_cube<float, 8><<<blocks, 256, 0, ctx->stream>>>((const float *)plain, (const float *)Yh, (float *)out, g, t);
        else if (p.mb == 12) k_g3_inv_axis1_cube<float, 12><<<blocks, 256, 0, ctx->stream>>>((const float *)plain, (const float *)Yh, (float *)out, g, t);
        else k_g3_inv_axis1_cube<float, 20><<<blocks, 256, 0, ctx->stream>>>((const float *)plain, (const float *)Yh, (float *)out, g, t);
    } else {
        QTaps<double> t = to_device_taps<double>(p);
        if (p.mb == 8) k_g3_inv_axis1_cube<double, 8><<<blocks, 256, 0, ctx->stream>>>((const double *)plain, (const double *)Yh, (double *)out, g, t);
        else if (p.mb == 12) k_g3_inv_axis1_cube<double, 12><<<blocks, 256, 0, ctx->stream>>>((const double *)plain, (const double *)Yh, (double *)out, g, t);
        else k_g3_inv_axis1_cube<double, 20><<<blocks, 256, 0, ctx->stream>>>((const double *)plain, (const double *)Yh, (double *)out, g, t);
    }
    G2_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
