// Re-sampling of lowpass images and complex highpass subbands on the device: the kernels
// behind dtcwt_hip_sample / _rescale / _upsample2 / _phase_roll_* (include/dtcwt_hip.h).
//
// Replaces dtcwt/sampling.py of the reference (the step directly
// downstream of the pyramids in registration and keypoint detection), so that pyramids can
// be re-sampled without leaving HBM.  Everything is one gather-and-weight form
//
//   out[p][c] = sum_{a,b} wy_b(frac y_p) wx_a(frac x_p) im[rho(floor y_p + b)][rho(floor x_p + a)][c]
//
// with rho the half-sample symmetric reflection (what `reflect(., -0.5, n-0.5).astype(int)`
// of sampling.py:36-40 computes for integer-valued input) and per method
//   nearest (:42-43): one tap at round-half-even;  bilinear (:45-66): 2x2, w = (1-f, f);
//   lanczos (:68-103): 6x6, w = L(f - a), a = -2..3, L(t) = sinc(t) sinc(t/3).
// An image is [H][W][ncomp] of the real type T (channels, times two for complex data): one
// thread per (point, component), components fastest, so a wavefront reads whole pixels.
//
// Coordinates and phases are double precision whatever T is: a float32 coordinate of a
// 4096-wide image has an ulp of 2.4e-4 pixels, and the phase ramps of the highpass variants
// (:167-190) reach tens of thousands of radians.
#include "common.hpp"

namespace {

enum { NEAREST = 0, BILINEAR = 1, LANCZOS = 2 };

// Reflection of a tap index.  Samples within one image size of the image (every tap of a
// rescale, practically every tap of a sample) take the branch-free 32-bit bounce; anything
// further out the 64-bit modulo form (64-bit integer division costs hundreds of instructions
// on gfx950, and there are up to 12 reflections per output).
__device__ inline int refl(int64_t u, int n) {
    if ((uint64_t)(u + n) < (uint64_t)(3 * (int64_t)n)) {
        int v = (int)u;
        v = v < 0 ? -1 - v : v;
        return v >= n ? 2 * n - 1 - v : v;
    }
    return (int)dt_reflect(u, n);
}

__device__ inline float sinpi_t(float x) { return sinpif(x); }
__device__ inline double sinpi_t(double x) { return sinpi(x); }
__device__ inline void sincospi_t(float x, float *s, float *c) { sincospif(x, s, c); }
__device__ inline void sincospi_t(double x, double *s, double *c) { sincospi(x, s, c); }

// L(t) = sinc(t) sinc(t/3) (numpy.sinc convention) evaluated directly; |t| < 1e-4 uses the
// series 1 - (pi^2 / 6)(1 + 1/9) t^2
template <typename T>
__device__ inline T lanczos_direct(double td) {
    const T t = (T)td;
    if (fabs(t) < (T)1e-4) return (T)1 - (T)1.8277437593757612 * t * t;
    const T pit = (T)3.14159265358979323846 * t;
    return (sinpi_t(t) / pit) * (sinpi_t(t / (T)3) / (pit / (T)3));
}

// w[a] = L(f - (a - 2)), a = 0..5, f in [0, 1).  The two taps next to the sample (|t| < 1)
// are evaluated directly: there the weight is O(1) and t is small, so numerator and
// denominator must see the same rounded t.  The four outer taps (1 <= |t| <= 3) share one
// sinpi and one sincospi through sin(pi (f - d)) = (-1)^d sin(pi f) and
// sin(pi (f - d) / 3) = sin(pi f / 3) cos(d pi / 3) - cos(pi f / 3) sin(d pi / 3).
template <typename T>
__device__ inline void lanczos6(double fd, T (&w)[6]) {
    const T f = (T)fd;
    const T s1 = sinpi_t(f);
    T s3, c3;
    sincospi_t(f / (T)3, &s3, &c3);
    const T h = (T)0.5, r = (T)0.86602540378443864676;            // cos, sin of pi/3
    const T cd[6] = {-h, h, (T)1, h, -h, (T)-1};                   // cos(d pi / 3), d = -2..3
    const T sd[6] = {-r, -r, (T)0, r, r, (T)0};                    // sin(d pi / 3)
    const T k = (T)(3.0 / (3.14159265358979323846 * 3.14159265358979323846));
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        if (a == 2 || a == 3) {
            w[a] = lanczos_direct<T>(fd - (double)(a - 2));
        } else {
            const T t = (T)(fd - (double)(a - 2));
            const T sgn = (a & 1) ? (T)-1 : (T)1;                  // (-1)^d, d = a - 2
            w[a] = k * sgn * s1 * (s3 * cd[a] - c3 * sd[a]) / (t * t);
        }
    }
}

// one interpolated value at (x, y) for component c
template <typename T, int METHOD>
__device__ inline T sample_at(const T *__restrict__ im, int H, int W, int K, double x, double y, int c) {
    const int64_t rs = (int64_t)W * K;
    if (METHOD == NEAREST) {
        int xi = refl((int64_t)rint(x), W), yi = refl((int64_t)rint(y), H);
        return im[yi * rs + xi * K + c];
    }
    const double fx0 = floor(x), fy0 = floor(y);
    const int64_t x0 = (int64_t)fx0, y0 = (int64_t)fy0;
    if (METHOD == BILINEAR) {
        // the reference's association (sampling.py:63-66): x first, then y
        const T fx = (T)(x - fx0), fy = (T)(y - fy0);
        const int xa = refl(x0, W) * K + c, xb = refl(x0 + 1, W) * K + c;
        const T *ra = im + refl(y0, H) * rs, *rb = im + refl(y0 + 1, H) * rs;
        const T lower = ((T)1 - fx) * ra[xa] + fx * ra[xb];
        const T upper = ((T)1 - fx) * rb[xa] + fx * rb[xb];
        return ((T)1 - fy) * lower + fy * upper;
    }
    T wx[6], wy[6];
    lanczos6<T>(x - fx0, wx);
    lanczos6<T>(y - fy0, wy);
    int xi[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) xi[a] = refl(x0 + a - 2, W) * K + c;
    T acc = (T)0;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const T *row = im + refl(y0 + b - 2, H) * rs;
        T r = (T)0;
#pragma unroll
        for (int a = 0; a < 6; ++a) r += wx[a] * row[xi[a]];
        acc += wy[b] * r;
    }
    return acc;
}

template <typename T, int METHOD>
__global__ void __launch_bounds__(256) k_sample(const T *__restrict__ im, int H, int W, int K,
                                                const double *__restrict__ xs, const double *__restrict__ ys,
                                                int64_t npts, T *__restrict__ out) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= npts * K) return;
    const int64_t p = id / K;
    const int c = (int)(id - p * K);
    out[id] = sample_at<T, METHOD>(im, H, W, K, xs[p], ys[p], c);
}

// destination pixel (dy, dx) of an oh x ow array samples the source at
// (xscale (dx + 1/2) - 1/2, yscale (dy + 1/2) - 1/2)     (sampling.py:141-163).
// A workgroup owns 256 consecutive (dx, c) of a band of RB output rows: the x-side taps and
// weights of a thread are computed once per band, the y-side ones once per row (by one
// thread each, shared through LDS) -- the separable structure of a regular grid.
constexpr int RESCALE_RB = 16;

template <typename T, int METHOD>
__global__ void __launch_bounds__(256) k_rescale(const T *__restrict__ im, int H, int W, int K, int oh, int ow,
                                                 double xscale, double yscale, T *__restrict__ out) {
    constexpr int NTAP = METHOD == LANCZOS ? 6 : (METHOD == BILINEAR ? 2 : 1);
    __shared__ T s_wy[RESCALE_RB][6];
    __shared__ int s_yi[RESCALE_RB][6];
    const int64_t rs = (int64_t)W * K;
    for (int band = blockIdx.y; band * RESCALE_RB < oh; band += gridDim.y) {
        const int dy0 = band * RESCALE_RB;
        if (threadIdx.x < RESCALE_RB && dy0 + (int)threadIdx.x < oh) {
            const double y = yscale * ((double)(dy0 + (int)threadIdx.x) + 0.5) - 0.5;
            T wy[6];
            int64_t y0;
            if (METHOD == NEAREST) { y0 = (int64_t)rint(y); wy[0] = (T)1; }
            else {
                const double fy0 = floor(y);
                y0 = (int64_t)fy0;
                if (METHOD == BILINEAR) { wy[1] = (T)(y - fy0); wy[0] = (T)1 - wy[1]; }
                else { lanczos6<T>(y - fy0, wy); y0 -= 2; }
            }
            for (int b = 0; b < NTAP; ++b) {
                s_wy[threadIdx.x][b] = wy[b];
                s_yi[threadIdx.x][b] = refl(y0 + b, H);
            }
        }
        __syncthreads();
        const unsigned q = blockIdx.x * 256u + threadIdx.x;        // (dx, c) within the row
        if (q < (unsigned)ow * (unsigned)K) {
            const unsigned dx = q / (unsigned)K;
            const int c = (int)(q - dx * (unsigned)K);
            const double x = xscale * ((double)dx + 0.5) - 0.5;
            T wx[6];
            int xi[6];
            int64_t x0;
            if (METHOD == NEAREST) { x0 = (int64_t)rint(x); wx[0] = (T)1; }
            else {
                const double fx0 = floor(x);
                x0 = (int64_t)fx0;
                if (METHOD == BILINEAR) { wx[1] = (T)(x - fx0); wx[0] = (T)1 - wx[1]; }
                else { lanczos6<T>(x - fx0, wx); x0 -= 2; }
            }
#pragma unroll
            for (int a = 0; a < NTAP; ++a) xi[a] = refl(x0 + a, W) * K + c;
            const int nrow = oh - dy0 < RESCALE_RB ? oh - dy0 : RESCALE_RB;
            for (int r = 0; r < nrow; ++r) {
                T acc = (T)0;
                if (METHOD == BILINEAR) {
                    // the reference's association (sampling.py:63-66): x first, then y
                    const T *ra = im + s_yi[r][0] * rs, *rb = im + s_yi[r][1] * rs;
                    const T lower = wx[0] * ra[xi[0]] + wx[1] * ra[xi[1]];
                    const T upper = wx[0] * rb[xi[0]] + wx[1] * rb[xi[1]];
                    acc = s_wy[r][0] * lower + s_wy[r][1] * upper;
                } else {
#pragma unroll
                    for (int b = 0; b < NTAP; ++b) {
                        const T *row = im + s_yi[r][b] * rs;
                        T t = (T)0;
#pragma unroll
                        for (int a = 0; a < NTAP; ++a) t += wx[a] * row[xi[a]];
                        acc += s_wy[r][b] * t;
                    }
                }
                out[(int64_t)(dy0 + r) * ow * K + q] = acc;
            }
        }
        __syncthreads();
    }
}

struct UpTaps {
    int off[8];
    double wa[8], wb[8];   // even outputs (i - 1/4), odd outputs (i + 1/4)   (sampling.py:280-336)
};

// NT taps per axis (1 nearest, 3 bilinear, 7 lanczos); grid.y = output row
template <typename T, int NT>
__global__ void __launch_bounds__(256) k_upsample2(const T *__restrict__ im, int H, int W, int K, UpTaps t,
                                                   T *__restrict__ out) {
    const unsigned q = blockIdx.x * 256u + threadIdx.x;        // (ox, c) within the output row
    if (q >= 2u * (unsigned)W * (unsigned)K) return;
    const unsigned ox = q / (unsigned)K;
    const int c = (int)(q - ox * (unsigned)K);
    const int ix = (int)(ox >> 1);
    const bool px = ox & 1;
    int xi[NT];
    T wx[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        xi[a] = refl((int64_t)ix + t.off[a], W) * K + c;
        wx[a] = (T)(px ? t.wb[a] : t.wa[a]);
    }
    // a band of RESCALE_RB output rows per workgroup: the x-side taps are set up once per band
    for (int oy = blockIdx.y * RESCALE_RB; oy < 2 * H; oy += gridDim.y * RESCALE_RB)
    for (int rr = 0; rr < RESCALE_RB && oy + rr < 2 * H; ++rr) {
        const int oyy = oy + rr;
        const int iy = oyy >> 1;
        const bool py = oyy & 1;
        T acc = (T)0;
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const T *row = im + (int64_t)refl((int64_t)iy + t.off[b], H) * W * K;
            T r = (T)0;
#pragma unroll
            for (int a = 0; a < NT; ++a) r += wx[a] * row[xi[a]];
            acc += (T)(py ? t.wb[b] : t.wa[b]) * r;
        }
        out[(int64_t)oyy * 2 * W * K + q] = acc;
    }
}

struct Roll {
    int nch;               // output channels (complex)
    int src[6];            // source channel of each output channel
    double dx[6], dy[6];   // phase advance per unit x / y of each output channel
};

// out[p][ch] = in[p][src[ch]] * exp(sign j (dx[ch] x_p + dy[ch] y_p)),   complex as (re, im) pairs of T
template <typename T, bool GRID>
__global__ void __launch_bounds__(256) k_phase_roll(const T *__restrict__ in, int64_t nin, T *__restrict__ out,
                                                    int64_t npts, int64_t ow, double xscale, double yscale,
                                                    const double *__restrict__ xs, const double *__restrict__ ys,
                                                    Roll r, double sign) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= npts * r.nch) return;
    const int64_t p = id / r.nch;
    const int ch = (int)(id - p * r.nch);
    double x, y;
    if (GRID) {
        const int64_t py = p / ow, px = p - py * ow;
        x = xscale * ((double)px + 0.5) - 0.5;
        y = yscale * ((double)py + 0.5) - 0.5;
    } else {
        x = xs[p]; y = ys[p];
    }
    double s, c;
    sincos(sign * (r.dx[ch] * x + r.dy[ch] * y), &s, &c);
    const T re = in[(p * nin + r.src[ch]) * 2], im = in[(p * nin + r.src[ch]) * 2 + 1];
    out[id * 2] = (T)((double)re * c - (double)im * s);
    out[id * 2 + 1] = (T)((double)re * s + (double)im * c);
}

// Grid form, separable: exp(j (a x + b y)) = exp(j a x) exp(j b y), so the trigonometry is
// done once per column and once per row (tables of W + H entries per channel) and the roll
// itself is two complex multiplications per element.
//   tab[i][ch] = exp(sign j d[ch] (scale (i + 1/2) - 1/2))
__global__ void __launch_bounds__(256) k_phase_table(int n, double scale, double sign, Roll r, int use_dx,
                                                     double2 *__restrict__ tab) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= n * r.nch) return;
    const int i = id / r.nch, ch = id - i * r.nch;
    const double u = scale * ((double)i + 0.5) - 0.5;
    double s, c;
    sincos(sign * (use_dx ? r.dx[ch] : r.dy[ch]) * u, &s, &c);
    tab[id] = make_double2(c, s);
}

template <typename T>
__global__ void __launch_bounds__(256) k_phase_roll_tab(const T *__restrict__ in, int nin, T *__restrict__ out,
                                                        int W, Roll r, const double2 *__restrict__ tx,
                                                        const double2 *__restrict__ ty) {
    const int py = blockIdx.y;
    const unsigned q = blockIdx.x * 256u + threadIdx.x;        // (px, ch) within the row
    if (q >= (unsigned)W * (unsigned)r.nch) return;
    const unsigned px = q / (unsigned)r.nch;
    const int ch = (int)(q - px * (unsigned)r.nch);
    const double2 a = tx[q], b = ty[py * r.nch + ch];
    const double c = a.x * b.x - a.y * b.y, s = a.x * b.y + a.y * b.x;
    const int64_t p = (int64_t)py * W + px;
    const T re = in[(p * nin + r.src[ch]) * 2], im = in[(p * nin + r.src[ch]) * 2 + 1];
    const int64_t o = (p * r.nch + ch) * 2;
    out[o] = (T)((double)re * c - (double)im * s);
    out[o + 1] = (T)((double)re * s + (double)im * c);
}

inline unsigned blocks_for(int64_t total) { return (unsigned)((total + 255) / 256); }

template <typename T>
int launch_sample(dtcwt_hip_ctx *ctx, const void *im, int64_t H, int64_t W, int64_t K, const double *xs,
                  const double *ys, int64_t n, int method, void *out) {
    const unsigned nb = blocks_for(n * K);
    if (method == NEAREST)
        k_sample<T, NEAREST><<<nb, 256, 0, ctx->stream>>>((const T *)im, (int)H, (int)W, (int)K, xs, ys, n, (T *)out);
    else if (method == BILINEAR)
        k_sample<T, BILINEAR><<<nb, 256, 0, ctx->stream>>>((const T *)im, (int)H, (int)W, (int)K, xs, ys, n, (T *)out);
    else
        k_sample<T, LANCZOS><<<nb, 256, 0, ctx->stream>>>((const T *)im, (int)H, (int)W, (int)K, xs, ys, n, (T *)out);
    return 0;
}

template <typename T>
int launch_rescale(dtcwt_hip_ctx *ctx, const void *im, int64_t H, int64_t W, int64_t K, int64_t oh, int64_t ow,
                   int method, void *out) {
    const int64_t bands = (oh + RESCALE_RB - 1) / RESCALE_RB;
    const dim3 nb(blocks_for(ow * K), (unsigned)(bands < 65535 ? bands : 65535));
    const double xs = (double)W / (double)ow, ys = (double)H / (double)oh;
    if (method == NEAREST)
        k_rescale<T, NEAREST><<<nb, 256, 0, ctx->stream>>>((const T *)im, (int)H, (int)W, (int)K, (int)oh, (int)ow, xs, ys, (T *)out);
    else if (method == BILINEAR)
        k_rescale<T, BILINEAR><<<nb, 256, 0, ctx->stream>>>((const T *)im, (int)H, (int)W, (int)K, (int)oh, (int)ow, xs, ys, (T *)out);
    else
        k_rescale<T, LANCZOS><<<nb, 256, 0, ctx->stream>>>((const T *)im, (int)H, (int)W, (int)K, (int)oh, (int)ow, xs, ys, (T *)out);
    return 0;
}

template <typename T>
int launch_upsample2(dtcwt_hip_ctx *ctx, const void *im, int64_t H, int64_t W, int64_t K, int ntaps, const UpTaps &t,
                     void *out) {
    const int64_t bands = (2 * H + RESCALE_RB - 1) / RESCALE_RB;
    const dim3 nb(blocks_for(2 * W * K), (unsigned)(bands < 65535 ? bands : 65535));
    if (ntaps == 1) k_upsample2<T, 1><<<nb, 256, 0, ctx->stream>>>((const T *)im, (int)H, (int)W, (int)K, t, (T *)out);
    else if (ntaps == 3) k_upsample2<T, 3><<<nb, 256, 0, ctx->stream>>>((const T *)im, (int)H, (int)W, (int)K, t, (T *)out);
    else if (ntaps == 7) k_upsample2<T, 7><<<nb, 256, 0, ctx->stream>>>((const T *)im, (int)H, (int)W, (int)K, t, (T *)out);
    else return -1;
    return 0;
}

}  // namespace

#define DT_LAUNCH_CHECK() DT_CHECK_HIP(hipGetLastError())
#define DT_BAD_DTYPE() return dtcwt_set_error(-1, "bad dtype %d", dtype)

extern "C" {

int dtcwt_hip_sample(dtcwt_hip_ctx *ctx, int dtype, const void *im, int64_t H, int64_t W, int64_t ncomp,
                     const double *xs, const double *ys, int64_t npts, int method, void *out) {
    DT_REQUIRE(ctx && im && xs && ys && out, "NULL argument");
    DT_REQUIRE(H > 0 && W > 0 && ncomp > 0 && npts >= 0, "bad extents");
    DT_REQUIRE(H < (1 << 30) && W * ncomp < (1 << 30), "image too large");
    DT_REQUIRE(method >= 0 && method <= 2, "method must be 0 (nearest), 1 (bilinear) or 2 (lanczos)");
    if (npts == 0) return 0;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    if (dtype == DTCWT_HIP_F32) launch_sample<float>(ctx, im, H, W, ncomp, xs, ys, npts, method, out);
    else if (dtype == DTCWT_HIP_F64) launch_sample<double>(ctx, im, H, W, ncomp, xs, ys, npts, method, out);
    else DT_BAD_DTYPE();
    DT_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_rescale(dtcwt_hip_ctx *ctx, int dtype, const void *im, int64_t H, int64_t W, int64_t ncomp,
                      int64_t out_h, int64_t out_w, int method, void *out) {
    DT_REQUIRE(ctx && im && out, "NULL argument");
    DT_REQUIRE(H > 0 && W > 0 && ncomp > 0 && out_h >= 0 && out_w >= 0, "bad extents");
    DT_REQUIRE(H < (1 << 30) && W * ncomp < (1 << 30) && out_h < (1LL << 31) && out_w * ncomp < (1LL << 31),
               "image too large");
    DT_REQUIRE(method >= 0 && method <= 2, "method must be 0 (nearest), 1 (bilinear) or 2 (lanczos)");
    if (out_h * out_w == 0) return 0;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    if (dtype == DTCWT_HIP_F32) launch_rescale<float>(ctx, im, H, W, ncomp, out_h, out_w, method, out);
    else if (dtype == DTCWT_HIP_F64) launch_rescale<double>(ctx, im, H, W, ncomp, out_h, out_w, method, out);
    else DT_BAD_DTYPE();
    DT_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_upsample2(dtcwt_hip_ctx *ctx, int dtype, const void *im, int64_t H, int64_t W, int64_t ncomp,
                        int ntaps, const int *offsets, const double *w_even, const double *w_odd, void *out) {
    DT_REQUIRE(ctx && im && offsets && w_even && w_odd && out, "NULL argument");
    DT_REQUIRE(H > 0 && W > 0 && ncomp > 0, "bad extents");
    DT_REQUIRE(ntaps == 1 || ntaps == 3 || ntaps == 7, "1, 3 or 7 taps per axis (nearest, bilinear, lanczos)");
    DT_REQUIRE(H < (1 << 30) && W * ncomp < (1 << 30), "image too large");
    UpTaps t{};
    for (int k = 0; k < ntaps; ++k) { t.off[k] = offsets[k]; t.wa[k] = w_even[k]; t.wb[k] = w_odd[k]; }
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    if (dtype == DTCWT_HIP_F32) launch_upsample2<float>(ctx, im, H, W, ncomp, ntaps, t, out);
    else if (dtype == DTCWT_HIP_F64) launch_upsample2<double>(ctx, im, H, W, ncomp, ntaps, t, out);
    else DT_BAD_DTYPE();
    DT_LAUNCH_CHECK();
    return 0;
}

static int fill_roll(Roll &r, int nch, const int *src, const double *dx, const double *dy, int64_t nin) {
    DT_REQUIRE(nch >= 1 && nch <= 6 && src && dx && dy, "1..6 output channels");
    r.nch = nch;
    for (int k = 0; k < nch; ++k) {
        DT_REQUIRE(src[k] >= 0 && src[k] < nin, "source channel out of range");
        r.src[k] = src[k]; r.dx[k] = dx[k]; r.dy[k] = dy[k];
    }
    return 0;
}

int dtcwt_hip_phase_roll_grid(dtcwt_hip_ctx *ctx, int dtype, const void *in, int64_t H, int64_t W, int64_t nin,
                              int nch, const int *src, const double *dtheta_dx, const double *dtheta_dy,
                              double xscale, double yscale, double sign, void *out) {
    DT_REQUIRE(ctx && in && out, "NULL argument");
    DT_REQUIRE(H >= 0 && W >= 0 && nin > 0, "bad extents");
    Roll r{};
    if (int rc = fill_roll(r, nch, src, dtheta_dx, dtheta_dy, nin)) return rc;
    if (H * W == 0) return 0;
    DT_REQUIRE(H < 65536 && W * nch < (1LL << 31), "grid too large");
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    void *tab = nullptr;
    if (int rc = dtcwt_hip_malloc(ctx, (size_t)(H + W) * nch * sizeof(double2), &tab)) return rc;
    double2 *tx = (double2 *)tab, *ty = tx + W * nch;
    k_phase_table<<<blocks_for(W * nch), 256, 0, ctx->stream>>>((int)W, xscale, sign, r, 1, tx);
    k_phase_table<<<blocks_for(H * nch), 256, 0, ctx->stream>>>((int)H, yscale, sign, r, 0, ty);
    const dim3 nb(blocks_for(W * nch), (unsigned)H);
    hipError_t e = hipSuccess;
    if (dtype == DTCWT_HIP_F32)
        k_phase_roll_tab<float><<<nb, 256, 0, ctx->stream>>>((const float *)in, (int)nin, (float *)out, (int)W, r, tx, ty);
    else if (dtype == DTCWT_HIP_F64)
        k_phase_roll_tab<double><<<nb, 256, 0, ctx->stream>>>((const double *)in, (int)nin, (double *)out, (int)W, r, tx, ty);
    else
        e = hipErrorInvalidValue;
    if (e == hipSuccess) e = hipGetLastError();
    dtcwt_hip_free(ctx, tab);           // stream-ordered reuse (common.hpp)
    if (e != hipSuccess) return dtcwt_set_error(-2, "phase roll failed: %s", hipGetErrorString(e));
    return 0;
}

int dtcwt_hip_phase_roll_points(dtcwt_hip_ctx *ctx, int dtype, const void *in, int64_t npts, int64_t nin, int nch,
                                const int *src, const double *dtheta_dx, const double *dtheta_dy,
                                const double *xs, const double *ys, double sign, void *out) {
    DT_REQUIRE(ctx && in && out && xs && ys, "NULL argument");
    DT_REQUIRE(npts >= 0 && nin > 0, "bad extents");
    Roll r{};
    if (int rc = fill_roll(r, nch, src, dtheta_dx, dtheta_dy, nin)) return rc;
    if (npts == 0) return 0;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    const unsigned nb = blocks_for(npts * nch);
    if (dtype == DTCWT_HIP_F32)
        k_phase_roll<float, false><<<nb, 256, 0, ctx->stream>>>((const float *)in, nin, (float *)out, npts, 1, 1.0, 1.0, xs, ys, r, sign);
    else if (dtype == DTCWT_HIP_F64)
        k_phase_roll<double, false><<<nb, 256, 0, ctx->stream>>>((const double *)in, nin, (double *)out, npts, 1, 1.0, 1.0, xs, ys, r, sign);
    else DT_BAD_DTYPE();
    DT_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
