// Inner loops of the DT-CWT image registration on the device: the kernels behind
// dtcwt_hip_qtilde / _solve6 / _boxfilter / _colsum / _affine_velocity / _warp_coords /
// _axpy / _fill_rows (include/dtcwt_hip.h).
//
// Replaces the per-pixel work of dtcwt/registration.py of the reference (SURVEY.md section
// 8(f) row 2): `qtildematrices` with its `confidence` and `phasegradient` (:31-214),
// `solvetransform` (:216-250), `_boxfilter` (:417-446) and the coordinate arithmetic of
// `velocityfield` / `warp` / `warphighpass` (:374-415).  Pyramids stay in HBM: the kernels
// read the complex subband records [H][W][6] the transform kernels wrote.  All arithmetic is
// float64 whatever the pyramid's precision (the reference accumulates Q-tilde in float64,
// :190); the data are tiny next to the transforms (levels >= 2 of a pyramid).
#include <vector>

#include "common.hpp"

namespace {

struct c2 { double re, im; };
__device__ inline c2 mk(double r, double i) { c2 z; z.re = r; z.im = i; return z; }
__device__ inline c2 operator+(c2 a, c2 b) { return mk(a.re + b.re, a.im + b.im); }
__device__ inline c2 mul(c2 a, c2 b) { return mk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
__device__ inline c2 mulconj(c2 a, c2 b) { return mk(a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im); }   // a conj(b)
__device__ inline double ang(c2 z) { return atan2(z.im, z.re); }

template <typename T>
__device__ inline c2 ld(const T *__restrict__ Yh, int W, int y, int x, int sb) {
    const T *p = Yh + (((int64_t)y * W + x) * 6 + sb) * 2;
    return mk((double)p[0], (double)p[1]);
}

// registration.py:29
__constant__ double c_shift[6][2] = {{-1, -3}, {-3, -3}, {-3, -1}, {-3, 1}, {-3, 3}, {-1, 3}};

// One level of `qtildematrices` (registration.py:166-212):
//   out[y][x][0..26] = sum over the six subbands of C^2 * (t_r t_c over triu(6x6), then t_r t_6)
// One thread per (pixel, subband): a subband's confidence and phase gradients are a chain of float64 hypot /
// atan2 / sincos calls, and with one thread doing the six subbands of a pixel in turn the kernel took 15-21 us
// whatever the level size (half of estimatereg).  The six contributions of a pixel meet in LDS and are summed
// in subband order, i.e. exactly as the serial loop accumulated them.
constexpr int QT_PIX = 32;                 // pixels per workgroup (x 6 subbands = 192 threads)
constexpr int QT_TW = 8, QT_TH = 4;        // the tile of the batched kernels

// subband `sb` of pixel (y, x) from records in memory
template <typename T>
struct LdGlobal {
    const T *p; int W, sb;
    __device__ c2 operator()(int y, int x) const { return ld(p, W, y, x, sb); }
};

// q[0..26] = the contribution of subband sb at (y, x); A(y, x) / B(y, x) load that subband of the two pyramids
template <typename LA, typename LB>
__device__ inline void qtilde_contrib(const LA &A, const LB &B, int H, int W, int y, int x, int sb, double eps,
                                      double *__restrict__ q) {
    const double xs = x * (1.0 / W), ys = y * (1.0 / H);          // np.arange(0, 1, 1/W)  (:168-169)
    const double wx = c_shift[sb][0] * (3.14159265358979323846 / 2.15);
    const double wy = c_shift[sb][1] * (3.14159265358979323846 / 2.15);
    // confidence (:83-137): the four diagonal neighbours, edges replicated
    c2 num = mk(0, 0);
    double den = eps;
#pragma unroll
    for (int oy = -1; oy <= 1; oy += 2)
#pragma unroll
        for (int ox = -1; ox <= 1; ox += 2) {
            const int yy = min(max(y + oy, 0), H - 1), xx = min(max(x + ox, 0), W - 1);
            const c2 u = A(yy, xx), v = B(yy, xx);
            num = num + mulconj(v, u);                         // conj(u) v
            const double au = hypot(u.re, u.im), av = hypot(v.re, v.im);
            den += au * au * au + av * av * av;
        }
    const double an = hypot(num.re, num.im);
    const double C = an * an / den;
    // phase gradients (:31-75)
    const c2 a0 = A(y, x), b0 = B(y, x);
    const c2 ex = mk(cos(wx), -sin(wx)), ey = mk(cos(wy), -sin(wy));
    // S(i) = (a[i+1] conj a[i] + b[i+1] conj b[i]) exp(-j w) between samples i and i+1
    c2 Sx;
    {
        const int xl = x > 0 ? x - 1 : 0, xr = x < W - 1 ? x : W - 2;    // pairs (xl, xl+1), (xr, xr+1)
        const c2 s1 = mul(mulconj(A(y, xl + 1), A(y, xl)) + mulconj(B(y, xl + 1), B(y, xl)), ex);
        const c2 s2 = mul(mulconj(A(y, xr + 1), A(y, xr)) + mulconj(B(y, xr + 1), B(y, xr)), ex);
        Sx = (x == 0) ? s1 : (x == W - 1 ? s2 : mk(0.5 * (s1.re + s2.re), 0.5 * (s1.im + s2.im)));
    }
    c2 Sy;
    {
        const int yl = y > 0 ? y - 1 : 0, yr = y < H - 1 ? y : H - 2;
        const c2 s1 = mul(mulconj(A(yl + 1, x), A(yl, x)) + mulconj(B(yl + 1, x), B(yl, x)), ey);
        const c2 s2 = mul(mulconj(A(yr + 1, x), A(yr, x)) + mulconj(B(yr + 1, x), B(yr, x)), ey);
        Sy = (y == 0) ? s1 : (y == H - 1 ? s2 : mk(0.5 * (s1.re + s2.re), 0.5 * (s1.im + s2.im)));
    }
    const double dx = (ang(Sx) + wx) * W, dy = (ang(Sy) + wy) * H;
    const double dt = ang(mulconj(b0, a0));                     // angle(b conj a)
    const double t[7] = {dx, dy, xs * dx, xs * dy, ys * dx, ys * dy, -dt};
    const double c2w = C * C;
    int e = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c) q[e++] = c2w * (t[r] * t[c]);
#pragma unroll
    for (int r = 0; r < 6; ++r) q[21 + r] = c2w * (t[r] * t[6]);
}

template <typename T>
__global__ void __launch_bounds__(QT_PIX * 6) k_qtilde(const T *__restrict__ A, const T *__restrict__ B, int H, int W,
                                                       double eps, double *__restrict__ out) {
    __shared__ double part[QT_PIX][6][27];
    const int pl = threadIdx.x / 6, sb = threadIdx.x - 6 * pl;
    const int64_t id = (int64_t)blockIdx.x * QT_PIX + pl;
    if (id < (int64_t)H * W) {
        const int y = (int)(id / W), x = (int)(id - (int64_t)y * W);
        qtilde_contrib(LdGlobal<T>{A, W, sb}, LdGlobal<T>{B, W, sb}, H, W, y, x, sb, eps, part[pl][sb]);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < QT_PIX * 27; k += QT_PIX * 6) {
        const int p2 = k / 27, e = k - 27 * p2;
        const int64_t id2 = (int64_t)blockIdx.x * QT_PIX + p2;
        if (id2 >= (int64_t)H * W) continue;
        double acc = 0.0;
#pragma unroll
        for (int s6 = 0; s6 < 6; ++s6) acc += part[p2][s6][e];
        out[id2 * 27 + e] = acc;
    }
}

// a = -Q^{-1} q with the Q the reference builds -- upper triangle only (registration.py:231-232),
// so the solve is a back substitution.  Qt: [n][27], a: [n][6].
__device__ inline void solve6(const double *q, double *s) {
    double U[6][6];
    int e = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c) U[r][c] = q[e++];
#pragma unroll
    for (int r = 5; r >= 0; --r) {
        double acc = -q[21 + r];
#pragma unroll
        for (int c = r + 1; c < 6; ++c) acc -= U[r][c] * s[c];
        s[r] = acc / U[r][r];
    }
}

__global__ void __launch_bounds__(256) k_solve6(const double *__restrict__ Qt, int64_t n, double *__restrict__ a) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    double s[6];
    solve6(Qt + id * 27, s);
#pragma unroll
    for (int r = 0; r < 6; ++r) a[id * 6 + r] = s[r];
}

__device__ inline int refl_i(int u, int n) { return (int)dt_reflect(u, n); }

// box filter of odd size k over the first two axes with symmetric extension (registration.py:417-446)
__global__ void __launch_bounds__(256) k_boxfilter(const double *__restrict__ in, int H, int W, int K, int half,
                                                   double *__restrict__ out) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (int64_t)H * W * K) return;
    const int c = (int)(id % K);
    const int64_t p = id / K;
    const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
    const double inv = 1.0 / (2 * half + 1);
    double acc = 0.0;
    for (int dy = -half; dy <= half; ++dy) {
        const double *row = in + (int64_t)refl_i(y + dy, H) * W * K + c;
        double r = 0.0;
        for (int dx = -half; dx <= half; ++dx) r += row[(int64_t)refl_i(x + dx, W) * K];
        acc += r * inv;
    }
    out[id] = acc * inv;
}

// out[c] += sum_p in[p][c]   (out zeroed by the caller)
__global__ void __launch_bounds__(256) k_colsum(const double *__restrict__ in, int64_t n, int K, double *__restrict__ out) {
    __shared__ double part[256];
    const int c = blockIdx.y;
    double acc = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (int64_t)gridDim.x * 256) acc += in[p * K + c];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out + c, part[0]);
}

// velocity of the affine parameters at their own grid (registration.py:385-390): pixel
// coordinates are float32 quotients, as np.arange(dtype=float32) / w gives them
__global__ void __launch_bounds__(256) k_affine_velocity(const double *__restrict__ av, int h, int w,
                                                         double *__restrict__ vx, double *__restrict__ vy) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (int64_t)h * w) return;
    const int y = (int)(id / w), x = (int)(id - (int64_t)y * w);
    const double px = (double)((float)x / (float)w), py = (double)((float)y / (float)h);
    const double *a = av + id * 6;
    vx[id] = a[0] + a[2] * px + a[4] * py;
    vy[id] = a[1] + a[3] * px + a[5] * py;
}

// sample positions of a warp (registration.py:401-415): ((x/W)_f32 + vx) W, ((y/H)_f32 + vy) H
__global__ void __launch_bounds__(256) k_warp_coords(const double *__restrict__ vx, const double *__restrict__ vy,
                                                     int H, int W, double *__restrict__ xs, double *__restrict__ ys) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (int64_t)H * W) return;
    const int y = (int)(id / W), x = (int)(id - (int64_t)y * W);
    xs[id] = ((double)((float)x / (float)W) + vx[id]) * W;
    ys[id] = ((double)((float)y / (float)H) + vy[id]) * H;
}

__global__ void __launch_bounds__(256) k_axpy(int64_t n, double alpha, const double *__restrict__ x, double *__restrict__ y) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id < n) y[id] += alpha * x[id];
}

struct Row8 { double v[8]; };
__global__ void __launch_bounds__(256) k_fill_rows(int64_t n, int K, Row8 row, double *__restrict__ out) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id < n * K) out[id] = row.v[id % K];
}

// out[p][k] = row[k] with the row on the device (no host round trip inside estimatereg)
__global__ void __launch_bounds__(256) k_broadcast_rows(int64_t n, int K, const double *__restrict__ row,
                                                        double *__restrict__ out) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id < n * K) out[id] = row[id % K];
}

// ---- fused steps of the refinement loop of estimatereg ------------------------------------------------
// One launch for `warphighpass(Yh_src[l], avecs, 'bilinear')` (registration.py:397-415 over sampling.py:
// 192-278): velocity field of the affine parameters at their own grid (k_affine_velocity) -> bilinear rescale of
// vx and vy to the level (k_rescale<double>) -> sample positions (k_warp_coords) -> phase unroll of the subbands
// at the integer grid (k_phase_roll_tab) -> bilinear sample at the positions (k_sample<T>) -> phase re-roll at the
// positions (k_phase_roll<T, false>), each of which used to be a launch of its own on arrays of a few thousand
// elements.  Same arithmetic, same association and the same rounding to T between the steps.
__device__ inline int refl_near(int64_t u, int n) { return (int)dt_reflect(u, n); }

template <typename T>
__device__ inline void warp_sample(const T *__restrict__ Yh, int H, int W, const double *__restrict__ av, int rh, int rw,
                                   int py, int px, int ch, T *__restrict__ o_re, T *__restrict__ o_im) {
    const double W0 = -3 * 3.14159265358979323846 / 2.15, W1 = -3.14159265358979323846 / 2.15;
    const double tdx[6] = {W1, W0, W0, W0, W0, W1}, tdy[6] = {W0, W0, W1, -W1, -W0, -W0};   // sampling.py:26-33
    // velocity at this pixel: (vx, vy) of the reg grid, bilinear (dtcwt_hip_rescale, float64)
    double vxs, vys;
    {
        const double x = ((double)rw / (double)W) * ((double)px + 0.5) - 0.5;
        const double y = ((double)rh / (double)H) * ((double)py + 0.5) - 0.5;
        const double fx0 = floor(x), fy0 = floor(y);
        const double wx1 = x - fx0, wx0 = 1.0 - wx1, wy1 = y - fy0, wy0 = 1.0 - wy1;
        const int xa = refl_near((int64_t)fx0, rw), xb = refl_near((int64_t)fx0 + 1, rw);
        const int ya = refl_near((int64_t)fy0, rh), yb = refl_near((int64_t)fy0 + 1, rh);
        double v[2][2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int yy = j ? yb : ya, xx = i ? xb : xa;
                const double pxn = (double)((float)xx / (float)rw), pyn = (double)((float)yy / (float)rh);
                const double *a = av + ((int64_t)yy * rw + xx) * 6;
                v[j][i][0] = a[0] + a[2] * pxn + a[4] * pyn;
                v[j][i][1] = a[1] + a[3] * pxn + a[5] * pyn;
            }
        vxs = wy0 * (wx0 * v[0][0][0] + wx1 * v[0][1][0]) + wy1 * (wx0 * v[1][0][0] + wx1 * v[1][1][0]);
        vys = wy0 * (wx0 * v[0][0][1] + wx1 * v[0][1][1]) + wy1 * (wx0 * v[1][0][1] + wx1 * v[1][1][1]);
    }
    const double xs = ((double)((float)px / (float)W) + vxs) * W;
    const double ys = ((double)((float)py / (float)H) + vys) * H;
    // bilinear sample of the unrolled subband at (xs, ys); the reference's association: x first, then y
    const double fx0 = floor(xs), fy0 = floor(ys);
    const T fx = (T)(xs - fx0), fy = (T)(ys - fy0);
    const int xi[2] = {refl_near((int64_t)fx0, W), refl_near((int64_t)fx0 + 1, W)};
    const int yi[2] = {refl_near((int64_t)fy0, H), refl_near((int64_t)fy0 + 1, H)};
    T ur[2][2], ui[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const T *q = Yh + (((int64_t)yi[j] * W + xi[i]) * 6 + ch) * 2;
            double sn, cs;
            sincos(-(tdx[ch] * (double)xi[i] + tdy[ch] * (double)yi[j]), &sn, &cs);
            ur[j][i] = (T)((double)q[0] * cs - (double)q[1] * sn);
            ui[j][i] = (T)((double)q[0] * sn + (double)q[1] * cs);
        }
    const T lr = ((T)1 - fx) * ur[0][0] + fx * ur[0][1], hr = ((T)1 - fx) * ur[1][0] + fx * ur[1][1];
    const T li = ((T)1 - fx) * ui[0][0] + fx * ui[0][1], hi = ((T)1 - fx) * ui[1][0] + fx * ui[1][1];
    const T sr = ((T)1 - fy) * lr + fy * hr, si = ((T)1 - fy) * li + fy * hi;
    double sn, cs;
    sincos(tdx[ch] * xs + tdy[ch] * ys, &sn, &cs);
    *o_re = (T)((double)sr * cs - (double)si * sn);
    *o_im = (T)((double)sr * sn + (double)si * cs);
}

template <typename T>
__global__ void __launch_bounds__(256) k_warp_level(const T *__restrict__ Yh, int H, int W, const double *__restrict__ av,
                                                    int rh, int rw, T *__restrict__ out) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (int64_t)H * W * 6) return;
    const int64_t p = id / 6;
    const int ch = (int)(id - p * 6);
    const int py = (int)(p / W), px = (int)(p - (int64_t)py * W);
    warp_sample(Yh, H, W, av, rh, rw, py, px, ch, out + id * 2, out + id * 2 + 1);
}

// One launch for `_boxfilter(qtilde, 3)` (registration.py:417-446) -> bilinear rescale onto the reg grid
// (sampling.py:131-165) -> accumulation over the levels of a group (:362-368):
//   acc[r][c] (+)= rescale(boxfilter(q))[r][c],   q: [H][W][27], acc: [rh][rw][27], float64
__device__ inline double box_rescale(const double *__restrict__ q, int H, int W, int rh, int rw, int ry, int rx, int c) {
    const double x = ((double)W / (double)rw) * ((double)rx + 0.5) - 0.5;
    const double y = ((double)H / (double)rh) * ((double)ry + 0.5) - 0.5;
    const double fx0 = floor(x), fy0 = floor(y);
    const double wx1 = x - fx0, wx0 = 1.0 - wx1, wy1 = y - fy0, wy0 = 1.0 - wy1;
    const int xt[2] = {refl_near((int64_t)fx0, W), refl_near((int64_t)fx0 + 1, W)};
    const int yt[2] = {refl_near((int64_t)fy0, H), refl_near((int64_t)fy0 + 1, H)};
    const double inv = 1.0 / 3.0;
    double b[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            double s = 0.0;
            for (int dy = -1; dy <= 1; ++dy) {
                const double *row = q + (int64_t)refl_i(yt[j] + dy, H) * W * 27 + c;
                double t = 0.0;
                for (int dx = -1; dx <= 1; ++dx) t += row[(int64_t)refl_i(xt[i] + dx, W) * 27];
                s += t * inv;
            }
            b[j][i] = s * inv;
        }
    return wy0 * (wx0 * b[0][0] + wx1 * b[0][1]) + wy1 * (wx0 * b[1][0] + wx1 * b[1][1]);
}

__global__ void __launch_bounds__(256) k_box_rescale_acc(const double *__restrict__ q, int H, int W, int rh, int rw,
                                                         int accumulate, double *__restrict__ acc) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (int64_t)rh * rw * 27) return;
    const int c = (int)(id % 27);
    const int64_t r = id / 27;
    const int ry = (int)(r / rw), rx = (int)(r - (int64_t)ry * rw);
    const double v = box_rescale(q, H, W, rh, rw, ry, rx, c);
    acc[id] = accumulate ? acc[id] + v : v;
}

// ---- the levels of one group of estimatereg in one launch per step ----------------------------------------
// A refinement pass used to be warp -> qtilde -> box-filter/resample/accumulate per level, then solve and axpy:
// 3 x levels + 2 dependent launches of a few thousand threads, each costing the ~4 us a dependent dispatch
// takes however little it does.  The levels of a group only depend on the parameters of the previous group, so
//   k_warp_qtilde_levels   warps a tile (+1 halo) of a level into LDS and forms its Q-tilde from there, for all
//                          levels of the group at once (blocks are assigned to levels by `blk0`);
//   k_box_solve_levels     box-filters / resamples / accumulates the levels in order, solves the 6 x 6 system
//                          of each block of the grid and adds the update to avecs.
// Same arithmetic and the same order of accumulation as the per-level kernels above.
constexpr int REG_MAXL = 12;
struct RegLevel {
    const void *src, *ref;     // records [H][W][6] complex
    double *q;                 // [H][W][27]
    int H, W, blk0, tw;        // first block of this level in the launch, tiles per row
};
struct RegBatch {
    RegLevel lv[REG_MAXL];
    int n, nblk;
};

// A warped in LDS: tile (QT_TH + 2) x (QT_TW + 2) around (y0, x0), records of T
template <typename T>
struct LdTile {
    const T *t; int y0, x0, sb;
    __device__ c2 operator()(int y, int x) const {
        const T *p = t + (((y - y0 + 1) * (QT_TW + 2) + (x - x0 + 1)) * 6 + sb) * 2;
        return mk((double)p[0], (double)p[1]);
    }
};

// MODE 0: A = src warped by the parameters `av`, q written per pixel
// MODE 1: A = src as it is, the block's pixels summed (in pixel order) into partial[block][27]
template <typename T, int MODE>
__global__ void __launch_bounds__(QT_PIX * 6) k_warp_qtilde_levels(RegBatch b, const double *__restrict__ av, int rh, int rw,
                                                                   double eps, double *__restrict__ partial) {
    __shared__ double part[QT_PIX][6][27];
    __shared__ double pix[MODE == 1 ? QT_PIX : 1][27];
    __shared__ T tile[MODE == 0 ? (QT_TH + 2) * (QT_TW + 2) * 12 : 2];
    int li = 0;
#pragma unroll 1
    for (int k = 1; k < b.n; ++k)
        if ((int)blockIdx.x >= b.lv[k].blk0) li = k;
    const RegLevel L = b.lv[li];
    const int tb = (int)blockIdx.x - L.blk0;
    const int y0 = (tb / L.tw) * QT_TH, x0 = (tb % L.tw) * QT_TW;
    const int H = L.H, W = L.W;
    if (MODE == 0) {
        for (int i = threadIdx.x; i < (QT_TH + 2) * (QT_TW + 2) * 6; i += QT_PIX * 6) {
            const int hp = i / 6, ch = i - 6 * hp;
            const int hy = hp / (QT_TW + 2), hx = hp - hy * (QT_TW + 2);
            const int py = y0 - 1 + hy, px = x0 - 1 + hx;
            if (py >= 0 && py < H && px >= 0 && px < W)
                warp_sample((const T *)L.src, H, W, av, rh, rw, py, px, ch, &tile[i * 2], &tile[i * 2 + 1]);
        }
        __syncthreads();
    }
    const int pl = threadIdx.x / 6, sb = threadIdx.x - 6 * pl;
    {
        const int y = y0 + pl / QT_TW, x = x0 + pl % QT_TW;
        if (y < H && x < W) {
            if (MODE == 0)
                qtilde_contrib(LdTile<T>{tile, y0, x0, sb}, LdGlobal<T>{(const T *)L.ref, W, sb}, H, W, y, x, sb, eps, part[pl][sb]);
            else
                qtilde_contrib(LdGlobal<T>{(const T *)L.src, W, sb}, LdGlobal<T>{(const T *)L.ref, W, sb}, H, W, y, x, sb, eps,
                               part[pl][sb]);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < QT_PIX * 27; k += QT_PIX * 6) {
        const int p2 = k / 27, e = k - 27 * p2;
        const int y = y0 + p2 / QT_TW, x = x0 + p2 % QT_TW;
        double acc = 0.0;
        if (y < H && x < W) {
#pragma unroll
            for (int s6 = 0; s6 < 6; ++s6) acc += part[p2][s6][e];
            if (MODE == 0) L.q[((int64_t)y * W + x) * 27 + e] = acc;
        }
        if (MODE == 1) pix[p2][e] = acc;          // 0 outside the image
    }
    if (MODE == 1) {
        __syncthreads();
        if (threadIdx.x < 27) {
            double acc = 0.0;
            for (int p2 = 0; p2 < QT_PIX; ++p2) acc += pix[p2][threadIdx.x];
            partial[(int64_t)blockIdx.x * 27 + threadIdx.x] = acc;
        }
    }
}

// global estimate: Qt = sum of the block partials (fixed order), a0 = solve(Qt), and -- when the grid is small
// enough for one workgroup -- avecs[:] = a0
__global__ void __launch_bounds__(1024) k_sum_solve(const double *__restrict__ partial, int nblk, double *__restrict__ a0,
                                                    int64_t nfill, double *__restrict__ avecs) {
    __shared__ double red[32][27];
    __shared__ double a[6];
    const int j = threadIdx.x / 27, c = threadIdx.x - 27 * j;
    if (j < 32) {
        double acc = 0.0;
        for (int p = j; p < nblk; p += 32) acc += partial[(int64_t)p * 27 + c];
        red[j][c] = acc;
    }
    __syncthreads();
    if (threadIdx.x < 27) {
        double acc = 0.0;
        for (int k = 0; k < 32; ++k) acc += red[k][threadIdx.x];
        red[0][threadIdx.x] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s[6];
        solve6(red[0], s);
#pragma unroll
        for (int r = 0; r < 6; ++r) { a[r] = s[r]; a0[r] = s[r]; }
    }
    __syncthreads();
    for (int64_t i = threadIdx.x; i < nfill; i += 1024) avecs[i] = a[i % 6];
}

constexpr int BS_PIX = 8;                  // blocks of the grid per workgroup (x 27 entries = 216 threads)
__global__ void __launch_bounds__(256) k_box_solve_levels(RegBatch b, int rh, int rw, double *__restrict__ avecs) {
    __shared__ double qs[BS_PIX][27];
    const int rp = threadIdx.x / 27, c = threadIdx.x - 27 * rp;
    const int64_t r = (int64_t)blockIdx.x * BS_PIX + rp;
    if (rp < BS_PIX && r < (int64_t)rh * rw) {
        const int ry = (int)(r / rw), rx = (int)(r - (int64_t)ry * rw);
        double acc = box_rescale(b.lv[0].q, b.lv[0].H, b.lv[0].W, rh, rw, ry, rx, c);
#pragma unroll 1
        for (int k = 1; k < b.n; ++k) acc = acc + box_rescale(b.lv[k].q, b.lv[k].H, b.lv[k].W, rh, rw, ry, rx, c);
        qs[rp][c] = acc;
    }
    __syncthreads();
    const int64_t r2 = (int64_t)blockIdx.x * BS_PIX + threadIdx.x;
    if (threadIdx.x < BS_PIX && r2 < (int64_t)rh * rw) {
        double s[6];
        solve6(qs[threadIdx.x], s);
#pragma unroll
        for (int k = 0; k < 6; ++k) avecs[r2 * 6 + k] += 1.0 * s[k];
    }
}

inline unsigned blocks_for(int64_t total) { return (unsigned)((total + 255) / 256); }

// pooled scratch buffers of one estimatereg call, released (stream-ordered) on every exit path
struct Scratch {
    dtcwt_hip_ctx *ctx;
    std::vector<void *> bufs;
    explicit Scratch(dtcwt_hip_ctx *c) : ctx(c) {}
    ~Scratch() { for (void *b : bufs) dtcwt_hip_free(ctx, b); }
    template <typename T>
    T *get(int64_t count) {
        void *p = nullptr;
        if (dtcwt_hip_malloc(ctx, (size_t)count * sizeof(T), &p)) return nullptr;
        bufs.push_back(p);
        return (T *)p;
    }
};

}  // namespace

#define DT_LAUNCH_CHECK() DT_CHECK_HIP(hipGetLastError())

extern "C" {

int dtcwt_hip_qtilde(dtcwt_hip_ctx *ctx, int dtype, const void *Yh_ref, const void *Yh_target, int64_t H, int64_t W,
                     double epsilon, double *out) {
    DT_REQUIRE(ctx && Yh_ref && Yh_target && out, "NULL argument");
    DT_REQUIRE(H >= 2 && W >= 2 && H * W < ((int64_t)1 << 31), "subbands must be at least 2 x 2");
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    if (dtype == DTCWT_HIP_F32)
        k_qtilde<float><<<(unsigned)((H * W + QT_PIX - 1) / QT_PIX), QT_PIX * 6, 0, ctx->stream>>>((const float *)Yh_ref, (const float *)Yh_target, (int)H, (int)W, epsilon, out);
    else if (dtype == DTCWT_HIP_F64)
        k_qtilde<double><<<(unsigned)((H * W + QT_PIX - 1) / QT_PIX), QT_PIX * 6, 0, ctx->stream>>>((const double *)Yh_ref, (const double *)Yh_target, (int)H, (int)W, epsilon, out);
    else
        return dtcwt_set_error(-1, "bad dtype %d", dtype);
    DT_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_solve6(dtcwt_hip_ctx *ctx, const double *Qt, int64_t n, double *a) {
    DT_REQUIRE(ctx && Qt && a && n >= 0, "bad argument");
    if (n == 0) return 0;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    k_solve6<<<blocks_for(n), 256, 0, ctx->stream>>>(Qt, n, a);
    DT_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_boxfilter(dtcwt_hip_ctx *ctx, const double *in, int64_t H, int64_t W, int64_t K, int kernel_size,
                        double *out) {
    DT_REQUIRE(ctx && in && out, "NULL argument");
    DT_REQUIRE(kernel_size > 0 && kernel_size % 2 == 1, "Kernel size must be odd");
    DT_REQUIRE(H > 0 && W > 0 && K > 0 && H * W < ((int64_t)1 << 31) && K < (1 << 20), "bad extents");
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    k_boxfilter<<<blocks_for(H * W * K), 256, 0, ctx->stream>>>(in, (int)H, (int)W, (int)K, kernel_size / 2, out);
    DT_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_colsum(dtcwt_hip_ctx *ctx, const double *in, int64_t n, int64_t K, double *out) {
    DT_REQUIRE(ctx && in && out && n >= 0 && K > 0 && K < 65536, "bad argument");
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    DT_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)K * sizeof(double), ctx->stream));
    if (n == 0) return 0;
    int64_t nb = (n + 255) / 256;
    if (nb > 1024) nb = 1024;
    k_colsum<<<dim3((unsigned)nb, (unsigned)K), 256, 0, ctx->stream>>>(in, n, (int)K, out);
    DT_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_affine_velocity(dtcwt_hip_ctx *ctx, const double *avecs, int64_t h, int64_t w, double *vx, double *vy) {
    DT_REQUIRE(ctx && avecs && vx && vy && h > 0 && w > 0 && h * w < ((int64_t)1 << 31), "bad argument");
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    k_affine_velocity<<<blocks_for(h * w), 256, 0, ctx->stream>>>(avecs, (int)h, (int)w, vx, vy);
    DT_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_warp_coords(dtcwt_hip_ctx *ctx, const double *vx, const double *vy, int64_t H, int64_t W, double *xs,
                          double *ys) {
    DT_REQUIRE(ctx && vx && vy && xs && ys && H > 0 && W > 0 && H * W < ((int64_t)1 << 31), "bad argument");
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    k_warp_coords<<<blocks_for(H * W), 256, 0, ctx->stream>>>(vx, vy, (int)H, (int)W, xs, ys);
    DT_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_axpy(dtcwt_hip_ctx *ctx, int64_t n, double alpha, const double *x, double *y) {
    DT_REQUIRE(ctx && x && y && n >= 0, "bad argument");
    if (n == 0) return 0;
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    k_axpy<<<blocks_for(n), 256, 0, ctx->stream>>>(n, alpha, x, y);
    DT_LAUNCH_CHECK();
    return 0;
}

int dtcwt_hip_fill_rows(dtcwt_hip_ctx *ctx, int64_t n, int K, const double *row, double *out) {
    DT_REQUIRE(ctx && row && out && n >= 0 && K >= 1 && K <= 8, "rows of 1..8 values");
    if (n == 0) return 0;
    Row8 r{};
    for (int k = 0; k < K; ++k) r.v[k] = row[k];
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    k_fill_rows<<<blocks_for(n * K), 256, 0, ctx->stream>>>(n, K, r, out);
    DT_LAUNCH_CHECK();
    return 0;
}

#define DT_TRY(expr)              \
    do {                          \
        int rc__ = (expr);        \
        if (rc__) return rc__;    \
    } while (0)

// The whole of `estimatereg` (dtcwt/registration.py:301-372) sequenced natively: the same kernels
// the host-side sequence of dtcwt_amd/hip/registration.py launches one by one, without the
// per-launch interpreter and ctypes latency that dominates at these sizes.
//   Yh_src / Yh_ref: per level, [H_l][W_l][6] complex records (dtype = their real type);
//   shapes: [nlevels][2] = (H_l, W_l);  groups: `ngroups` level lists, group g holds
//   group_sizes[g] consecutive entries of group_levels (0-based); the first group gives the
//   global estimate, the others refine it (:339-370);  avecs: [reg_h][reg_w][6] float64 out.
// the levels of a group as one batch of tiles; q buffers only when `want_q`
static int reg_batch(Scratch &sc, const void *const *Yh_src, const void *const *Yh_ref, const int64_t *shapes,
                     const int *lv, int n, bool want_q, RegBatch *b) {
    b->n = n;
    int blk = 0;
    for (int k = 0; k < n; ++k) {
        const int l = lv[k];
        const int64_t H = shapes[2 * l], W = shapes[2 * l + 1];
        DT_REQUIRE(H >= 2 && W >= 2 && H * W * 27 < ((int64_t)1 << 31), "level too small or too large");
        RegLevel &L = b->lv[k];
        L.src = Yh_src[l]; L.ref = Yh_ref[l];
        L.H = (int)H; L.W = (int)W;
        L.tw = (int)((W + QT_TW - 1) / QT_TW);
        L.blk0 = blk;
        blk += L.tw * (int)((H + QT_TH - 1) / QT_TH);
        L.q = nullptr;
        if (want_q) {
            L.q = sc.get<double>(H * W * 27);
            DT_REQUIRE(L.q, "out of device memory");
        }
    }
    b->nblk = blk;
    return 0;
}

// one launch per step and group (see k_warp_qtilde_levels)
static int estimatereg_issue_batched(dtcwt_hip_ctx *ctx, Scratch &sc, int dtype, const void *const *Yh_src,
                                     const void *const *Yh_ref, const int64_t *shapes, int64_t reg_h, int64_t reg_w,
                                     int ngroups, const int *group_sizes, const int *group_levels, double *avecs) {
    const int64_t nreg = reg_h * reg_w;
    const bool f32 = dtype == DTCWT_HIP_F32;
    const int *lv = group_levels;
    RegBatch b;
    {
        DT_TRY(reg_batch(sc, Yh_src, Yh_ref, shapes, lv, group_sizes[0], false, &b));
        double *partial = sc.get<double>((int64_t)b.nblk * 27), *a0 = sc.get<double>(6);
        DT_REQUIRE(partial && a0, "out of device memory");
        if (f32) k_warp_qtilde_levels<float, 1><<<b.nblk, QT_PIX * 6, 0, ctx->stream>>>(b, nullptr, 0, 0, 1e-6, partial);
        else k_warp_qtilde_levels<double, 1><<<b.nblk, QT_PIX * 6, 0, ctx->stream>>>(b, nullptr, 0, 0, 1e-6, partial);
        DT_LAUNCH_CHECK();
        const bool fold = nreg * 6 <= 16384;
        k_sum_solve<<<1, 1024, 0, ctx->stream>>>(partial, b.nblk, a0, fold ? nreg * 6 : 0, avecs);
        DT_LAUNCH_CHECK();
        if (!fold) {
            k_broadcast_rows<<<blocks_for(nreg * 6), 256, 0, ctx->stream>>>(nreg, 6, a0, avecs);
            DT_LAUNCH_CHECK();
        }
        lv += group_sizes[0];
    }
    for (int g = 1; g < ngroups; lv += group_sizes[g], ++g) {
        if (group_sizes[g] < 1) continue;
        DT_TRY(reg_batch(sc, Yh_src, Yh_ref, shapes, lv, group_sizes[g], true, &b));
        if (f32) k_warp_qtilde_levels<float, 0><<<b.nblk, QT_PIX * 6, 0, ctx->stream>>>(b, avecs, (int)reg_h, (int)reg_w, 1e-6, nullptr);
        else k_warp_qtilde_levels<double, 0><<<b.nblk, QT_PIX * 6, 0, ctx->stream>>>(b, avecs, (int)reg_h, (int)reg_w, 1e-6, nullptr);
        DT_LAUNCH_CHECK();
        k_box_solve_levels<<<(unsigned)((nreg + BS_PIX - 1) / BS_PIX), 256, 0, ctx->stream>>>(b, (int)reg_h, (int)reg_w, avecs);
        DT_LAUNCH_CHECK();
    }
    return 0;
}

static int estimatereg_issue(dtcwt_hip_ctx *ctx, Scratch &sc, int dtype, int nlevels, const void *const *Yh_src,
                             const void *const *Yh_ref, const int64_t *shapes, int64_t reg_h, int64_t reg_w,
                             int ngroups, const int *group_sizes, const int *group_levels, double *avecs) {
    const size_t esz = dtype == DTCWT_HIP_F32 ? sizeof(float) : sizeof(double);
    const int64_t nreg = reg_h * reg_w;
    // DTCWT_HIP_REG_BATCH=0 keeps one launch per level and step (the A/B switch of the batched kernels)
    static const bool batch = [] { const char *e = getenv("DTCWT_HIP_REG_BATCH"); return !(e && e[0] == '0'); }();
    bool fits = group_sizes[0] >= 1 && nreg * 27 < ((int64_t)1 << 31);
    for (int g = 0; g < ngroups; ++g) fits = fits && group_sizes[g] <= REG_MAXL;
    if (batch && fits)
        return estimatereg_issue_batched(ctx, sc, dtype, Yh_src, Yh_ref, shapes, reg_h, reg_w, ngroups, group_sizes,
                                         group_levels, avecs);

    // global estimate: Q-tilde summed over every pixel of the first group's levels
    double *Qt = sc.get<double>(27), *part = sc.get<double>(27), *a0 = sc.get<double>(6);
    DT_REQUIRE(Qt && part && a0, "out of device memory");
    DT_CHECK_HIP(hipMemsetAsync(Qt, 0, 27 * sizeof(double), ctx->stream));
    const int *lv = group_levels;
    for (int k = 0; k < group_sizes[0]; ++k) {
        const int l = lv[k];
        const int64_t H = shapes[2 * l], W = shapes[2 * l + 1];
        double *q = sc.get<double>(H * W * 27);
        DT_REQUIRE(q, "out of device memory");
        DT_TRY(dtcwt_hip_qtilde(ctx, dtype, Yh_src[l], Yh_ref[l], H, W, 1e-6, q));
        DT_TRY(dtcwt_hip_colsum(ctx, q, H * W, 27, part));
        DT_TRY(dtcwt_hip_axpy(ctx, 27, 1.0, part, Qt));
    }
    DT_TRY(dtcwt_hip_solve6(ctx, Qt, 1, a0));
    k_broadcast_rows<<<blocks_for(nreg * 6), 256, 0, ctx->stream>>>(nreg, 6, a0, avecs);
    DT_LAUNCH_CHECK();
    lv += group_sizes[0];

    double *qts = sc.get<double>(nreg * 27), *da = sc.get<double>(nreg * 6);
    DT_REQUIRE(qts && da, "out of device memory");
    for (int g = 1; g < ngroups; lv += group_sizes[g], ++g) {
        if (group_sizes[g] < 1) continue;
        for (int k = 0; k < group_sizes[g]; ++k) {
            const int l = lv[k];
            const int64_t H = shapes[2 * l], W = shapes[2 * l + 1], n = H * W;
            void *wrp = sc.get<char>(n * 12 * esz);
            double *q = sc.get<double>(n * 27);
            DT_REQUIRE(wrp && q, "out of device memory");
            DT_REQUIRE(n * 27 < ((int64_t)1 << 31), "level too large");
            // warphighpass(Yh_src[l], avecs, 'bilinear')   (:397-408), one launch
            if (dtype == DTCWT_HIP_F32)
                k_warp_level<float><<<blocks_for(n * 6), 256, 0, ctx->stream>>>((const float *)Yh_src[l], (int)H, (int)W, avecs,
                                                                              (int)reg_h, (int)reg_w, (float *)wrp);
            else
                k_warp_level<double><<<blocks_for(n * 6), 256, 0, ctx->stream>>>((const double *)Yh_src[l], (int)H, (int)W, avecs,
                                                                               (int)reg_h, (int)reg_w, (double *)wrp);
            DT_LAUNCH_CHECK();
            // Q-tilde of the warped level, box filtered and resampled onto the block grid (:362-368)
            DT_TRY(dtcwt_hip_qtilde(ctx, dtype, wrp, Yh_ref[l], H, W, 1e-6, q));
            k_box_rescale_acc<<<blocks_for(nreg * 27), 256, 0, ctx->stream>>>(q, (int)H, (int)W, (int)reg_h, (int)reg_w,
                                                                               k > 0, qts);
            DT_LAUNCH_CHECK();
        }
        DT_TRY(dtcwt_hip_solve6(ctx, qts, nreg, da));
        DT_TRY(dtcwt_hip_axpy(ctx, nreg * 6, 1.0, da, avecs));
    }
    return 0;
}

// About a hundred small dependent launches: at 288 x 352 the eager sequence is bound by the host's launch
// rate (~6 us per launch, 0.61 ms), not by the device.  A call is therefore captured once as a hipGraph --
// keyed by everything baked into its nodes: the pyramid pointers, shapes, groups and the output -- and
// replayed when the same buffers come back (a registration loop over recycled pyramid buffers); up to four
// graphs are kept per context, each owning its scratch buffers.  DTCWT_HIP_REG_GRAPH=0 keeps eager launches.
namespace {
struct RegGraph {
    std::vector<int64_t> key;
    Scratch *sc;
    hipGraph_t graph;
    hipGraphExec_t exec;
};
struct RegCache {
    std::vector<RegGraph> items;       // most recently used last
    void drop(size_t i) {
        (void)hipGraphExecDestroy(items[i].exec);
        (void)hipGraphDestroy(items[i].graph);
        delete items[i].sc;
        items.erase(items.begin() + i);
    }
};
std::mutex g_reg_mu;
std::unordered_map<dtcwt_hip_ctx *, RegCache *> g_reg;
}  // namespace

int dtcwt_hip_estimatereg(dtcwt_hip_ctx *ctx, int dtype, int nlevels, const void *const *Yh_src,
                          const void *const *Yh_ref, const int64_t *shapes, int64_t reg_h, int64_t reg_w,
                          int ngroups, const int *group_sizes, const int *group_levels, double *avecs) {
    DT_REQUIRE(ctx && Yh_src && Yh_ref && shapes && group_sizes && group_levels && avecs, "NULL argument");
    DT_REQUIRE(nlevels > 0 && ngroups > 0 && reg_h > 0 && reg_w > 0, "bad extents");
    DT_CHECK_HIP(hipSetDevice(ctx->device));
    static const bool use_graph = [] { const char *e = getenv("DTCWT_HIP_REG_GRAPH"); return !(e && e[0] == '0'); }();
    if (!use_graph) {
        Scratch sc(ctx);
        return estimatereg_issue(ctx, sc, dtype, nlevels, Yh_src, Yh_ref, shapes, reg_h, reg_w, ngroups, group_sizes,
                                 group_levels, avecs);
    }
    int total = 0;
    for (int g = 0; g < ngroups; ++g) total += group_sizes[g];
    std::vector<int64_t> key = {dtype, nlevels, reg_h, reg_w, ngroups, (int64_t)(intptr_t)avecs};
    for (int l = 0; l < nlevels; ++l) {
        key.push_back((int64_t)(intptr_t)Yh_src[l]); key.push_back((int64_t)(intptr_t)Yh_ref[l]);
        key.push_back(shapes[2 * l]); key.push_back(shapes[2 * l + 1]);
    }
    for (int g = 0; g < ngroups; ++g) key.push_back(group_sizes[g]);
    for (int k = 0; k < total; ++k) key.push_back(group_levels[k]);
    std::lock_guard<std::mutex> lk(g_reg_mu);
    RegCache *&cache = g_reg[ctx];
    if (!cache) {
        cache = new RegCache();
        ctx->on_destroy.push_back([ctx] {
            std::lock_guard<std::mutex> lk2(g_reg_mu);
            auto it = g_reg.find(ctx);
            if (it == g_reg.end()) return;
            while (!it->second->items.empty()) it->second->drop(0);
            delete it->second;
            g_reg.erase(it);
        });
    }
    for (size_t i = 0; i < cache->items.size(); ++i)
        if (cache->items[i].key == key) {
            RegGraph g = cache->items[i];
            cache->items.erase(cache->items.begin() + i);
            cache->items.push_back(g);
            DT_CHECK_HIP(hipGraphLaunch(g.exec, ctx->stream));
            return 0;
        }
    // first call with these buffers: capture.  Relaxed mode: the scratch pool may have to hipMalloc meanwhile.
    Scratch *sc = new Scratch(ctx);
    hipError_t e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) {
        delete sc;
        return dtcwt_set_error(-2, "hipStreamBeginCapture failed: %s", hipGetErrorString(e));
    }
    int rc = estimatereg_issue(ctx, *sc, dtype, nlevels, Yh_src, Yh_ref, shapes, reg_h, reg_w, ngroups, group_sizes,
                               group_levels, avecs);
    hipGraph_t graph = nullptr;
    e = hipStreamEndCapture(ctx->stream, &graph);
    hipGraphExec_t exec = nullptr;
    if (!rc && e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (rc || e != hipSuccess) {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        delete sc;
        if (rc) return rc;
        return dtcwt_set_error(-2, "capturing estimatereg as a graph failed: %s", hipGetErrorString(e));
    }
    while (cache->items.size() >= 4) {
        // the evicted graph may still be running on this stream
        DT_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        cache->drop(0);
    }
    cache->items.push_back(RegGraph{key, sc, graph, exec});
    DT_CHECK_HIP(hipGraphLaunch(exec, ctx->stream));
    return 0;
}

}  // extern "C"
