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

__device__ inline int64_t refl(int64_t u, int64_t n) { return dt_reflect(u, n); }

__device__ inline float sinpi_t(float x) { return sinpif(x); }
__device__ inline double sinpi_t(double x) { return sinpi(x); }

template <typename T>
__device__ inline T lanczos_w(double t) {       // sinc(t) sinc(t/3), numpy.sinc convention
    if (t == 0.0) return (T)1;
    const T x = (T)t;
    const T pix = (T)3.14159265358979323846 * x;
    return (sinpi_t(x) / pix) * (sinpi_t(x / (T)3) / (pix / (T)3));
}

// one interpolated value at (x, y) for component c
template <typename T, int METHOD>
__device__ inline T sample_at(const T *__restrict__ im, int64_t H, int64_t W, int64_t K, double x, double y,
                              int64_t c) {
    if (METHOD == NEAREST) {
        int64_t xi = refl((int64_t)rint(x), W), yi = refl((int64_t)rint(y), H);
        return im[(yi * W + xi) * K + c];
    }
    const double fx0 = floor(x), fy0 = floor(y);
    const int64_t x0 = (int64_t)fx0, y0 = (int64_t)fy0;
    if (METHOD == BILINEAR) {
        // the reference's association (sampling.py:63-66): x first, then y
        const T fx = (T)(x - fx0), fy = (T)(y - fy0);
        const int64_t xa = refl(x0, W), xb = refl(x0 + 1, W), ya = refl(y0, H), yb = refl(y0 + 1, H);
        const T lower = ((T)1 - fx) * im[(ya * W + xa) * K + c] + fx * im[(ya * W + xb) * K + c];
        const T upper = ((T)1 - fx) * im[(yb * W + xa) * K + c] + fx * im[(yb * W + xb) * K + c];
        return ((T)1 - fy) * lower + fy * upper;
    }
    T wx[6], wy[6];
    int64_t xi[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        wx[a] = lanczos_w<T>((x - fx0) - (double)(a - 2));
        wy[a] = lanczos_w<T>((y - fy0) - (double)(a - 2));
        xi[a] = refl(x0 + a - 2, W);
    }
    T acc = (T)0;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const T *row = im + refl(y0 + b - 2, H) * W * K + c;
        T r = (T)0;
#pragma unroll
        for (int a = 0; a < 6; ++a) r += wx[a] * row[xi[a] * K];
        acc += wy[b] * r;
    }
    return acc;
}

template <typename T, int METHOD>
__global__ void __launch_bounds__(256) k_sample(const T *__restrict__ im, int64_t H, int64_t W, int64_t K,
                                                const double *__restrict__ xs, const double *__restrict__ ys,
                                                int64_t npts, T *__restrict__ out) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= npts * K) return;
    const int64_t p = id / K, c = id - p * K;
    out[id] = sample_at<T, METHOD>(im, H, W, K, xs[p], ys[p], c);
}

// destination pixel (dy, dx) of an oh x ow array samples the source at
// (xscale (dx + 1/2) - 1/2, yscale (dy + 1/2) - 1/2)     (sampling.py:141-163)
template <typename T, int METHOD>
__global__ void __launch_bounds__(256) k_rescale(const T *__restrict__ im, int64_t H, int64_t W, int64_t K,
                                                 int64_t oh, int64_t ow, double xscale, double yscale,
                                                 T *__restrict__ out) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= oh * ow * K) return;
    const int64_t p = id / K, c = id - p * K;
    const int64_t dy = p / ow, dx = p - dy * ow;
    out[id] = sample_at<T, METHOD>(im, H, W, K, xscale * ((double)dx + 0.5) - 0.5,
                                   yscale * ((double)dy + 0.5) - 0.5, c);
}

struct UpTaps {
    int n;                 // taps per axis (1, 3 or 7)
    int off[8];
    double wa[8], wb[8];   // even outputs (i - 1/4), odd outputs (i + 1/4)   (sampling.py:280-336)
};

template <typename T>
__global__ void __launch_bounds__(256) k_upsample2(const T *__restrict__ im, int64_t H, int64_t W, int64_t K,
                                                   UpTaps t, T *__restrict__ out) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= 4 * H * W * K) return;
    const int64_t p = id / K, c = id - p * K;
    const int64_t oy = p / (2 * W), ox = p - oy * (2 * W);
    const int64_t iy = oy >> 1, ix = ox >> 1;
    const bool py = oy & 1, px = ox & 1;
    T acc = (T)0;
    for (int b = 0; b < t.n; ++b) {
        const T *row = im + refl(iy + t.off[b], H) * W * K + c;
        T r = (T)0;
        for (int a = 0; a < t.n; ++a) r += (T)(px ? t.wb[a] : t.wa[a]) * row[refl(ix + t.off[a], W) * K];
        acc += (T)(py ? t.wb[b] : t.wa[b]) * r;
    }
    out[id] = acc;
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

inline unsigned blocks_for(int64_t total) { return (unsigned)((total + 255) / 256); }

template <typename T>
int launch_sample(dtcwt_hip_ctx *ctx, const void *im, int64_t H, int64_t W, int64_t K, const double *xs,
                  const double *ys, int64_t n, int method, void *out) {
    const unsigned nb = blocks_for(n * K);
    if (method == NEAREST)
        k_sample<T, NEAREST><<<nb, 256, 0, ctx->stream>>>((const T *)im, H, W, K, xs, ys, n, (T *)out);
    else if (method == BILINEAR)
        k_sample<T, BILINEAR><<<nb, 256, 0, ctx->stream>>>((const T *)im, H, W, K, xs, ys, n, (T *)out);
    else
        k_sample<T, LANCZOS><<<nb, 256, 0, ctx->stream>>>((const T *)im, H, W, K, xs, ys, n, (T *)out);
    return 0;
}

template <typename T>
int launch_rescale(dtcwt_hip_ctx *ctx, const void *im, int64_t H, int64_t W, int64_t K, int64_t oh, int64_t ow,
                   int method, void *out) {
    const unsigned nb = blocks_for(oh * ow * K);
    const double xs = (double)W / (double)ow, ys = (double)H / (double)oh;
    if (method == NEAREST)
        k_rescale<T, NEAREST><<<nb, 256, 0, ctx->stream>>>((const T *)im, H, W, K, oh, ow, xs, ys, (T *)out);
    else if (method == BILINEAR)
        k_rescale<T, BILINEAR><<<nb, 256, 0, ctx->stream>>>((const T *)im, H, W, K, oh, ow, xs, ys, (T *)out);
    else
        k_rescale<T, LANCZOS><<<nb, 256, 0, ctx->stream>>>((const T *)im, H, W, K, oh, ow, xs, ys, (T *)out);
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
    DT_REQUIRE(ntaps >= 1 && ntaps <= 8, "1..8 taps per axis");
    UpTaps t{};
    t.n = ntaps;
    for (int k = 0; k < ntaps; ++k) { t.off[k] = offsets[k]; t.wa[k] = w_even[k]; t.wb[k] = w_odd[k]; }
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    const unsigned nb = blocks_for(4 * H * W * ncomp);
    if (dtype == DTCWT_HIP_F32) k_upsample2<float><<<nb, 256, 0, ctx->stream>>>((const float *)im, H, W, ncomp, t, (float *)out);
    else if (dtype == DTCWT_HIP_F64) k_upsample2<double><<<nb, 256, 0, ctx->stream>>>((const double *)im, H, W, ncomp, t, (double *)out);
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
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    const unsigned nb = blocks_for(H * W * nch);
    if (dtype == DTCWT_HIP_F32)
        k_phase_roll<float, true><<<nb, 256, 0, ctx->stream>>>((const float *)in, nin, (float *)out, H * W, W, xscale, yscale, nullptr, nullptr, r, sign);
    else if (dtype == DTCWT_HIP_F64)
        k_phase_roll<double, true><<<nb, 256, 0, ctx->stream>>>((const double *)in, nin, (double *)out, H * W, W, xscale, yscale, nullptr, nullptr, r, sign);
    else DT_BAD_DTYPE();
    DT_LAUNCH_CHECK();
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
