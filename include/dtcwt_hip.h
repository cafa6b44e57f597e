/* dtcwt_hip.h -- C ABI of libdtcwt_hip.so, the MI355X (gfx950) DT-CWT filter bank.
 *
 * This is the drop-in boundary of the `hip` backend.  The reference (rjw57/dtcwt) is
 * pure Python and has no FFI; each entry point below states which reference function it
 * replaces (paths relative to the reference root).  Host code (dtcwt_amd/hip/, via
 * ctypes) does shape/argument checking and raises the reference's exceptions *before*
 * calling in; the library itself never throws across the boundary.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; dtcwt_hip_last_error() returns a
 *     thread-local message for the last failing call of the calling thread;
 *   - pointers named X/Y/Yl/Yh/... are DEVICE pointers unless the name says host;
 *     filter taps are HOST pointers to double (cast to the signal dtype inside, as
 *     dtcwt/numpy/lowlevel.py:33 does);
 *   - dtype: DTCWT_HIP_F32 or DTCWT_HIP_F64 for the generic filters; the fused 2-D plan
 *     is float32 (the precision the OpenCL/TF backends of the reference also use);
 *   - calls are asynchronous on the context's stream; dtcwt_hip_sync() waits;
 *   - a context (and everything created from it) is single-threaded; different contexts
 *     may be driven from different host threads;
 *   - the caller owns every device buffer it passes; plans own their workspaces.
 */
#ifndef DTCWT_HIP_H
#define DTCWT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: plan1d_*, plan3d_*, mgpu_* (round 2), mgpu_forward2d_scales, host_alloc / host_free / memcpy_*_async (round 3)
 * 3: plan2d_launches, plan2d_set_concurrency, mgpu_scatter_async / gather_async;  4: ctx_create_partition (round 4)
 * 5: plan2d_set_program, plan2d_level1_march, mgpu_create_lane, mgpu_shares, to_float kinds 9 / 10 (round 5)
 * 6: plan2d_describe; a plan reads its environment switches once, at creation (round 6) */
#define DTCWT_HIP_ABI_VERSION 6

#define DTCWT_HIP_F32 0
#define DTCWT_HIP_F64 1

#define DTCWT_HIP_MAX_TAPS 40          /* longest shipped filter is qshift_32 (32 taps) */

/* flags for the generic filters */
#define DTCWT_HIP_ACCUMULATE 1         /* Y += result instead of Y = result */

typedef struct dtcwt_hip_ctx dtcwt_hip_ctx;
typedef struct dtcwt_hip_plan2d dtcwt_hip_plan2d;
typedef struct dtcwt_hip_event dtcwt_hip_event;

/* ---------------------------------------------------------------- runtime ---------- */
int dtcwt_hip_abi_version(void);
const char *dtcwt_hip_last_error(void);
int dtcwt_hip_device_count(int *count);
/* name: buffer of >= 256 bytes; cus: compute units; mem_bytes: total HBM */
int dtcwt_hip_device_info(int device, char *name, int *cus, size_t *mem_bytes);

/* Context = device + stream.  `stream` NULL: the context creates and owns a stream;
 * otherwise it is a hipStream_t owned by the caller (e.g. torch's current stream).
 * Analogue of the `queue=` argument of the reference's OpenCL backend
 * (dtcwt/opencl/transform2d.py:108-110, dtcwt/opencl/lowlevel.py:154-167). */
int dtcwt_hip_ctx_create(int device, void *stream, dtcwt_hip_ctx **ctx);
/* (ABI 4) A context whose own stream runs on ONE of `nparts` equal shares of the device's compute units only
 * (hipExtStreamCreateWithCUMask, bits part * cus / nparts ... of the mask: on MI355X a slice of every XCD, so each share
 * keeps all eight L2s and fabric ports).  For `nparts` independent transforms in flight -- the frames of a video handed to
 * `nparts` workers (examples/register_video.py:125-156 in the reference) -- each on its own context: their kernels no
 * longer take turns on every CU, measured 0.152-0.157 against 0.165-0.170 ms per 4096 x 4096 forward + inverse with four in
 * flight (profiles/r04/ab_cu_mask*.txt).  Plans made on the context size their launches for its share (cus / nparts).
 * nparts 1..16, 0 <= part < nparts.
 * Stream semantics: hipExtStreamCreateWithCUMask takes no flags, so the stream of a partition context is a BLOCKING
 * stream -- unlike the hipStreamNonBlocking stream dtcwt_hip_ctx_create makes, it synchronises implicitly with the
 * legacy NULL stream: work a framework enqueues on the NULL stream waits for, and is waited for by, everything on every
 * partition context, which serialises the transforms this call exists to overlap.  Keep other work on explicit
 * non-blocking streams (or per-thread default streams) while partition contexts are busy. */
int dtcwt_hip_ctx_create_partition(int device, int part, int nparts, dtcwt_hip_ctx **ctx);
int dtcwt_hip_ctx_destroy(dtcwt_hip_ctx *ctx);
int dtcwt_hip_sync(dtcwt_hip_ctx *ctx);            /* the context's stream */
int dtcwt_hip_device_sync(dtcwt_hip_ctx *ctx);     /* hipDeviceSynchronize(): all streams */
void *dtcwt_hip_ctx_stream(dtcwt_hip_ctx *ctx);

/* Device buffers: replace to_device/to_array/empty of dtcwt/opencl/lowlevel.py:169-181. */
int dtcwt_hip_malloc(dtcwt_hip_ctx *ctx, size_t bytes, void **dptr);
int dtcwt_hip_free(dtcwt_hip_ctx *ctx, void *dptr);
/* malloc/free go through a per-context cache of freed buffers (stream-ordered reuse);
 * trim() returns the cached buffers to the driver (env DTCWT_HIP_POOL_MB caps the cache). */
int dtcwt_hip_trim(dtcwt_hip_ctx *ctx);
int dtcwt_hip_memcpy_h2d(dtcwt_hip_ctx *ctx, void *dst, const void *src_host, size_t bytes);
int dtcwt_hip_memcpy_d2h(dtcwt_hip_ctx *ctx, void *dst_host, const void *src, size_t bytes);
int dtcwt_hip_memcpy_d2d(dtcwt_hip_ctx *ctx, void *dst, const void *src, size_t bytes);
int dtcwt_hip_memset(dtcwt_hip_ctx *ctx, void *dst, int value, size_t bytes);
/* Page-locked host buffers and asynchronous copies (the lazy host copies of dtcwt/opencl/transform2d.py:30-84):
 * h2d_async only enqueues on the context's stream; d2h_overlapped is ordered after everything enqueued so far but
 * runs on a second stream, so kernels enqueued after it are not held up (a level's subbands go down the host link
 * while the next levels are computed); copy_sync waits for the overlapped downloads. */
int dtcwt_hip_host_alloc(size_t bytes, void **hptr);
int dtcwt_hip_host_free(void *hptr);
int dtcwt_hip_memcpy_h2d_async(dtcwt_hip_ctx *ctx, void *dst, const void *src_host, size_t bytes);
int dtcwt_hip_memcpy_d2h_overlapped(dtcwt_hip_ctx *ctx, void *dst_host, const void *src, size_t bytes);
int dtcwt_hip_copy_sync(dtcwt_hip_ctx *ctx);
/* Integer / bool samples -> float32 / float64 on the device: `asfarray` of dtcwt/utils.py:98-105 (every
 * non-float input becomes float64) done after the upload, so that an 8-bit image crosses the host link
 * as 1 byte per sample.  src_kind: 0 u8, 1 i8, 2 u16, 3 i16, 4 u32, 5 i32, 6 u64, 7 i64, 8 bool; (ABI 5) 9 float32,
 * 10 float64 -- a plain precision change on the device (complex arrays: count both components): float32 images with a
 * level of 8 samples or fewer are transformed in float64 and rounded once (their multi-bounce reflections sum the same
 * few samples many times; in float32 the error against the float64 oracle reached 9.9e-7 of a subband's maximum). */
int dtcwt_hip_to_float(dtcwt_hip_ctx *ctx, int src_kind, const void *src, int dst_dtype, void *dst,
                       int64_t count);

/* HIP events on the context's stream (timing of the benchmark harness). */
int dtcwt_hip_event_create(dtcwt_hip_ctx *ctx, dtcwt_hip_event **ev);
int dtcwt_hip_event_record(dtcwt_hip_ctx *ctx, dtcwt_hip_event *ev);
int dtcwt_hip_event_elapsed_ms(dtcwt_hip_event *start, dtcwt_hip_event *stop, float *ms);
int dtcwt_hip_event_destroy(dtcwt_hip_event *ev);

/* ---------------------------------------------------------------- generic filters --- */
/* A strided 3-D view  A[o][j][i]  (o < outer, j < n, i < inner) addresses element
 * base + o*so + j*sn + i*si  (strides in ELEMENTS).  All three filters act along j.
 * A 2-D column filter is outer=batch, n=rows, inner=cols, si=1; a row filter swaps the
 * roles (sn=1, si=row stride); 3-D volumes use the same view per axis.
 *
 * Logical extension of the input before filtering (what the transform drivers do with
 * vstack/hstack/concatenate): the filtered signal has logical length
 * n + pad_lo + pad_hi where logical sample u is input sample clamp(u-pad_lo, 0, n-1)
 * (edge replication: dtcwt/numpy/transform2d.py:86-94, :134-140; transform3d.py:322-335).
 * Output cropping (the inverse's Z[1:-1], transform2d.py:263-268): the first crop_lo
 * and last crop_hi logical output samples are not written.
 */
typedef struct dtcwt_hip_view {
    int64_t outer, n, inner;       /* input extents (n = real input samples along j) */
    int64_t xso, xsn, xsi;         /* input strides */
    int64_t yso, ysn, ysi;         /* output strides */
    int32_t pad_lo, pad_hi;        /* logical edge replication of the input */
    int32_t crop_lo, crop_hi;      /* logical output samples dropped */
} dtcwt_hip_view;

/* colfilter: replaces dtcwt/numpy/lowlevel.py:47-80.  Logical output length is
 * L (m odd) or L+1 (m even), L = n+pad_lo+pad_hi. */
int dtcwt_hip_colfilter(dtcwt_hip_ctx *ctx, int dtype, const void *X, void *Y,
                        const dtcwt_hip_view *v, const double *h_host, int m, int flags);
/* coldfilt: replaces dtcwt/numpy/lowlevel.py:82-154.  L % 4 == 0, m even; output L/2. */
int dtcwt_hip_coldfilt(dtcwt_hip_ctx *ctx, int dtype, const void *X, void *Y,
                       const dtcwt_hip_view *v, const double *ha_host, const double *hb_host,
                       int m, int flags);
/* colifilt: replaces dtcwt/numpy/lowlevel.py:156-260.  L % 2 == 0, m even; output 2L. */
int dtcwt_hip_colifilt(dtcwt_hip_ctx *ctx, int dtype, const void *X, void *Y,
                       const dtcwt_hip_view *v, const double *ha_host, const double *hb_host,
                       int m, int flags);

/* Fused pairs used by the 3-D (and 1-D) level loops, which always apply a lo AND a hi filter
 * to the same array (forward, dtcwt/numpy/transform3d.py:256-273, :353-369) or sum a lo- and
 * a hi-filtered array (inverse, :425-435, :485-495): one pass over the data instead of two.
 *   colfilter2:      Y0 = colfilter(X, h0),  Y1 = colfilter(X, h1)        (same length parity)
 *   colfilter_sum2:  Y  = colfilter(X0, h0) + colfilter(X1, h1)
 *   coldfilt2:       Y0 = coldfilt(X, ha0, hb0),  Y1 = coldfilt(X, ha1, hb1)
 *   colifilt_sum2:   Y  = colifilt(X0, ha0, hb0) + colifilt(X1, ha1, hb1)
 * X0/X1 (and Y0/Y1) share one view. */
int dtcwt_hip_colfilter2(dtcwt_hip_ctx *ctx, int dtype, const void *X, void *Y0, void *Y1,
                         const dtcwt_hip_view *v, const double *h0_host, int m0,
                         const double *h1_host, int m1);
int dtcwt_hip_colfilter_sum2(dtcwt_hip_ctx *ctx, int dtype, const void *X0, const void *X1, void *Y,
                             const dtcwt_hip_view *v, const double *h0_host, int m0,
                             const double *h1_host, int m1);
int dtcwt_hip_coldfilt2(dtcwt_hip_ctx *ctx, int dtype, const void *X, void *Y0, void *Y1,
                        const dtcwt_hip_view *v, const double *ha0_host, const double *hb0_host,
                        const double *ha1_host, const double *hb1_host, int m);
int dtcwt_hip_colifilt_sum2(dtcwt_hip_ctx *ctx, int dtype, const void *X0, const void *X1, void *Y,
                            const dtcwt_hip_view *v, const double *ha0_host, const double *hb0_host,
                            const double *ha1_host, const double *hb1_host, int m);

/* q2c: replaces dtcwt/numpy/transform2d.py:301-322 plus the slice-assign into Yh
 * (:122-127).  y: [batch][rows][cols] real plane (strides in elements), rows, cols even;
 * Yh: [batch][rows/2][cols/2][6] interleaved complex; the pair goes to subbands
 * slot0 (p-q) and slot1 (p+q). */
int dtcwt_hip_q2c(dtcwt_hip_ctx *ctx, int dtype, const void *y, int64_t batch, int64_t rows,
                  int64_t cols, int64_t y_sb, int64_t y_sr, void *Yh, int slot0, int slot1);
/* c2q: replaces dtcwt/numpy/transform2d.py:324-350.  Yh: [batch][rows][cols][6];
 * x: [batch][2 rows][2 cols]. */
int dtcwt_hip_c2q(dtcwt_hip_ctx *ctx, int dtype, const void *Yh, int64_t batch, int64_t rows,
                  int64_t cols, int slot0, int slot1, double gain0, double gain1, void *x,
                  int64_t x_sb, int64_t x_sr);
/* One whole 2-D level, any wavelet length, float32 or float64, in one launch (the path of
 * every dtype / wavelet without a fused float32 tile program; float64 is what the reference
 * computes in for non-float32 input, dtcwt/utils.py:104-134).
 *
 * level2d_forward replaces one iteration of dtcwt/numpy/transform2d.py:112-130 (kind 0,
 * level 1: Lo/Hi = colfilter(X, lo_a / hi_a) down the columns, then along the rows, then the
 * three q2c, :301-322) or of :132-160 (kind 1, levels >= 2: coldfilt with the pairs
 * (lo_a, lo_b) / (hi_a, hi_b) in the argument order of coldfilt).  X: [B][R][C]; rows / columns
 * are logically replicated by (pad_r_lo, pad_r_hi) / (pad_c_lo, pad_c_hi) (the odd-size and
 * multiple-of-4 extensions of :86-94, :134-143).  R1 x C1 = padded size (kind 0) or half of it
 * (kind 1).  Lo, Hi: scratch [B][R1][C] (touched only by the two-launch form, DTCWT_HIP_TWO_PASS=1);
 * LoLo: [B][R1][C1]; Yh: [B][R1/2][C1/2][6] complex.
 *
 * level2d_inverse replaces one iteration of :275-293 (kind 0) or :242-273 (kind 1): the three
 * c2q (:324-350, gains6[k] = gain of subband k) and the column / row synthesis filters with
 * their sums.  Zl: [B][Rl][Cl] lowpass; Yh: [B][Rl/2][Cl/2][6]; kind 1 drops crop_r / crop_c
 * output samples from both ends of the rows / columns (the size fix-up of :246-252).  With
 * (Rz, Cz) = (Rl, Cl) for kind 0, (2 Rl - 2 crop_r, 2 Cl - 2 crop_c) for kind 1:
 * Y1, Y2: scratch [B][Rz][Cl] (two-launch form only); Z: [B][Rz][Cz].
 *
 * Both return -3 (caller uses the filter-by-filter entry points above) for even-length
 * level-1 filters, pairs of unequal length, wavelets longer than the largest compile-time
 * bucket, and planes narrower than the filter bucket + 4. */
int dtcwt_hip_level2d_forward(dtcwt_hip_ctx *ctx, int dtype, int kind, const void *X, int64_t B,
                              int64_t R, int64_t C, int pad_r_lo, int pad_r_hi, int pad_c_lo,
                              int pad_c_hi, const double *lo_a, const double *lo_b,
                              const double *hi_a, const double *hi_b, int m_lo, int m_hi,
                              void *Lo, void *Hi, void *LoLo, void *Yh);
int dtcwt_hip_level2d_inverse(dtcwt_hip_ctx *ctx, int dtype, int kind, const void *Zl, const void *Yh,
                              int64_t B, int64_t Rl, int64_t Cl, const double *gains6,
                              int crop_r, int crop_c, const double *lo_a, const double *lo_b,
                              const double *hi_a, const double *hi_b, int m_lo, int m_hi,
                              void *Y1, void *Y2, void *Z);
/* One level of the 1-D transform in ONE launch, float32 / float64, X: [n][k] (k signals side by
 * side, k == 1 or k >= 32; other k return -3).
 *
 * level1d_forward replaces dtcwt/numpy/transform1d.py:79-88 (kind 0: Lo = colfilter(X, lo_a),
 * Hi = colfilter(X, hi_a)) or :93-100 (kind 1: coldfilt with the pairs (lo_a, lo_b) / (hi_a,
 * hi_b), rows replicated by (pad_lo, pad_hi) as :95-96 does) INCLUDING the packing
 * Yh = Hi[::2] + 1j*Hi[1::2] (:88, :100).  Lo: [n1][k], Yh: [n1/2][k] complex, n1 = padded n
 * (kind 0) or half of it (kind 1).
 *
 * level1d_inverse replaces :150-160 (kind 1) or :162-176 (kind 0): Z = filter(Lo, lo) +
 * filter(gain * unpack(Yh), hi) with colifilt / colfilter; kind 1 drops `crop` samples from both
 * ends (:156-157).  Lo: [n][k], Yh: [n/2][k] complex, Z: [n or 2n - 2 crop][k].
 *
 * Both return -3 where dtcwt_hip_level2d_* would. */
int dtcwt_hip_level1d_forward(dtcwt_hip_ctx *ctx, int dtype, int kind, const void *X, int64_t n,
                              int64_t k, int pad_lo, int pad_hi, const double *lo_a, const double *lo_b,
                              const double *hi_a, const double *hi_b, int m_lo, int m_hi, void *Lo,
                              void *Yh);
int dtcwt_hip_level1d_inverse(dtcwt_hip_ctx *ctx, int dtype, int kind, const void *Lo, const void *Yh,
                              int64_t n, int64_t k, double gain, int crop, const double *lo_a,
                              const double *lo_b, const double *hi_a, const double *hi_b, int m_lo,
                              int m_hi, void *Z);
/* Level 1 of the 3-D transform in its filter-by-filter form (float64, and whatever else has no
 * fused tile program), with the packings fused into the neighbouring axis pass.
 *
 * fwd3_axis0_cube2c: the last analysis pass of dtcwt/numpy/transform3d.py:256-273 on ONE of the
 * four volumes V [n0][n1][n2] left by the axis-2 and axis-1 passes: lo = colfilter(V, h0) and
 * hi = colfilter(V, h1) down axis 0, each stored either plain (octant_* = -1: the LLL lowpass,
 * into `plain`) or as subbands 4 o .. 4 o + 3 of Yh [n0/2][n1/2][n2/2][28] complex (cube2c,
 * :532-579; o = position in the concatenation order of :278-289).
 *
 * inv3_axis1_c2cube: the first synthesis pass of :425-435: out [n0][n1][n2] =
 * colfilter(lo, g0) + colfilter(hi, g1) along axis 1, where hi is octant_hi of Yh unpacked on load
 * (c2cube, :581-619) and lo is octant_lo of Yh or, for -1, the plain volume `plain`.
 *
 * Odd-length filters only; -3 otherwise (and for volumes shorter than the filter bucket). */
int dtcwt_hip_fwd3_axis0_cube2c(dtcwt_hip_ctx *ctx, int dtype, const void *V, int64_t n0, int64_t n1,
                                int64_t n2, const double *h0, int m0, const double *h1, int m1,
                                int octant_lo, int octant_hi, void *plain, void *Yh);
int dtcwt_hip_inv3_axis1_c2cube(dtcwt_hip_ctx *ctx, int dtype, const void *plain, const void *Yh,
                                int64_t n0, int64_t n1, int64_t n2, const double *g0, int m0,
                                const double *g1, int m1, int octant_lo, int octant_hi, void *out);
/* Fused float32 level 1 of the 3-D forward transform: replaces `_level1_xfm`
 * (dtcwt/numpy/transform3d.py:208-289) for odd-length biort filters -- the three axis
 * passes (h0o/h1o along axes 2, 1, 0) and the seven cube2c packings in ONE launch.
 * X, LLL: [n0][n1][n2] contiguous float32 (n* even); Yh: [n0/2][n1/2][n2/2][28] complex64,
 * octants in the reference's order (:278-289).  Filters of at most 7 taps: ONE launch (a marching pair of wavefronts for the
 * symmetric 5 / 7, 7 / 5, 5 / 3, 3 / 5 pairs where rows of axis 2 come in fours, the tile program otherwise); the 13 / 19-tap
 * filters of near_sym_b: TWO launches around two pooled axis-0 volumes (the marching pair filter along axis 0, then both
 * in-slice axes + cube2c: fused3d_long.hpp).  Returns -3 (use the generic colfilter2 + cube2c path) when no fused kernel
 * exists for the tap lengths or the volume is tiny. */
int dtcwt_hip_fwd3_level1(dtcwt_hip_ctx *ctx, const float *X, int64_t n0, int64_t n1, int64_t n2,
                          const double *h0o, int m0, const double *h1o, int m1, float *LLL,
                          float *Yh);
/* Fused float32 level >= 2 of the 3-D forward transform: replaces `_level2_xfm`
 * (dtcwt/numpy/transform3d.py:317-383) -- coldfilt(., h0b, h0a) / coldfilt(., h1b, h1a) along
 * the three axes and the seven cube2c packings -- in two launches (per-slice 2-D tile
 * program writing four planes to a pooled workspace, then axis 0 + pack).
 * X: [n0][n1][n2] float32 lowpass of the previous level; pad_a in {0, 1, 2}: planes replicated
 * per side on axis a (ext_mode 4 / 8, :322-335), n_a + 2 pad_a must be a multiple of 4.
 * LLL: [(n0+2pad0)/2][(n1+2pad1)/2][(n2+2pad2)/2]; Yh: the same extents halved, [28] complex64.
 * Returns -3 when no fused kernel exists for m-tap filters or slices are under 2m x 2m. */
int dtcwt_hip_fwd3_level2(dtcwt_hip_ctx *ctx, const float *X, int64_t n0, int64_t n1, int64_t n2,
                          int pad0, int pad1, int pad2, const double *h0b, const double *h0a,
                          const double *h1b, const double *h1a, int m, float *LLL, float *Yh);
/* Fused float32 level 1 of the 3-D inverse transform: replaces `_level1_ifm`
 * (dtcwt/numpy/transform3d.py:385-440) for odd-length biort filters -- c2cube of the seven
 * octants and the three merges colfilter(lo, g0o) + colfilter(hi, g1o) -- in two launches
 * (unpack + axis-0 merge marching along axis 0 into four pooled planes, then the 2-D
 * column/row passes per slice).  LLL, Z: [n0][n1][n2] float32; Yh: [n0/2][n1/2][n2/2][28]
 * complex64.  The 19 / 13-tap synthesis filters of near_sym_b: c2cube + both in-slice axes in one launch into two pooled
 * volumes, then the marching sum filter along axis 0.  Returns -3 when no fused kernel exists for the tap lengths or the
 * volume is small (n0 < 12 or n1/n2 under twice the tap count). */
int dtcwt_hip_inv3_level1(dtcwt_hip_ctx *ctx, const float *LLL, const float *Yh, int64_t n0, int64_t n1,
                          int64_t n2, const double *g0o, int m0, const double *g1o, int m1, float *Z);
/* Fused float32 level >= 2 of the 3-D inverse transform: replaces `_level2_ifm`
 * (dtcwt/numpy/transform3d.py:460-526): colifilt(., g0b, g0a) + colifilt(., g1b, g1a) merges
 * and the cropping of ext_mode 4 / 8 (crop_a planes per side of the 2 n_a output, :505-524).
 * LLL: [n0][n1][n2]; Yh: [n0/2][n1/2][n2/2][28]; Z: [2n0-2crop0][2n1-2crop1][2n2-2crop2]. */
int dtcwt_hip_inv3_level2(dtcwt_hip_ctx *ctx, const float *LLL, const float *Yh, int64_t n0, int64_t n1,
                          int64_t n2, int crop0, int crop1, int crop2, const double *g0b,
                          const double *g0a, const double *g1b, const double *g1a, int m, float *Z);
/* cube2c: replaces dtcwt/numpy/transform3d.py:532-579 for one octant.
 * y: real volume view [d0][d1][d2] with element strides (s0, s1, 1), d* even;
 * Yh: [d0/2][d1/2][d2/2][28] complex; writes components 4*octant .. 4*octant+3. */
int dtcwt_hip_cube2c(dtcwt_hip_ctx *ctx, int dtype, const void *y, int64_t d0, int64_t d1,
                     int64_t d2, int64_t s0, int64_t s1, void *Yh, int octant);
/* c2cube: replaces dtcwt/numpy/transform3d.py:581-619 for one octant.
 * Yh: [e0][e1][e2][28]; y: [2e0][2e1][2e2] view with strides (s0, s1, 1). */
int dtcwt_hip_c2cube(dtcwt_hip_ctx *ctx, int dtype, const void *Yh, int64_t e0, int64_t e1,
                     int64_t e2, int octant, void *y, int64_t s0, int64_t s1);
/* interleave / de-interleave of the 1-D transform: hi is [2J][k] real, Yh is [J][k]
 * complex.  pack1d: Yh[j] = Hi[2j] + i Hi[2j+1] (dtcwt/numpy/transform1d.py:88,100);
 * unpack1d: c2q1d of gain*Yh (:153,171,186-196). */
int dtcwt_hip_pack1d(dtcwt_hip_ctx *ctx, int dtype, const void *hi, int64_t J, int64_t k, void *Yh);
int dtcwt_hip_unpack1d(dtcwt_hip_ctx *ctx, int dtype, const void *Yh, int64_t J, int64_t k,
                       double gain, void *hi);
/* x[i] *= gain */
int dtcwt_hip_scale(dtcwt_hip_ctx *ctx, int dtype, void *x, int64_t count, double gain);

/* ---------------------------------------------------------------- fused 2-D plan ---- */
/* The device-resident level loop of Transform2d.forward / .inverse
 * (dtcwt/numpy/transform2d.py:40-188, :190-295) for a batch of equally sized float32
 * images: one fused kernel per level (column pass, row pass and q2c/c2q inside one LDS
 * tile), pyramid buffers in HBM in the reference's layout:
 *     X      [batch][rows][cols]              float32 (rows/cols as given, may be odd)
 *     Yl     [batch][lr][lc]                  float32
 *     Yh[l]  [batch][hr_l][hc_l][6]           complex64 (interleaved re, im)
 *     Ys[l]  [batch][sr_l][sc_l]              float32 (include_scale)
 * Shapes follow the reference exactly, including the bottom/right replication of odd
 * inputs (:86-94) and the edge padding of levels whose size is not a multiple of 4
 * (:134-140) -- done by index arithmetic, never by copying.
 *
 * biort_host:  4 vectors h0o, g0o, h1o, g1o          (odd lengths, <= DTCWT_HIP_MAX_TAPS)
 * qshift_host: 8 vectors h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b (one even length)
 * given as arrays of pointers + lengths.  The band-pass ("_bp", 6/12 vector) wavelets
 * are handled by the host through the generic filters.
 */
int dtcwt_hip_plan2d_create(dtcwt_hip_ctx *ctx, int batch, int rows, int cols, int nlevels,
                            const double *const *biort_host, const int *biort_len,
                            const double *const *qshift_host, const int *qshift_len,
                            dtcwt_hip_plan2d **plan);
int dtcwt_hip_plan2d_destroy(dtcwt_hip_plan2d *plan);
/* shapes[0..1] = extended input rows, cols; shapes[2..3] = Yl rows, cols; then for each
 * level l: Yh rows, cols, scale rows, cols  (4 ints per level). */
int dtcwt_hip_plan2d_shapes(const dtcwt_hip_plan2d *plan, int *shapes);
/* Ys may be NULL (no include_scale); otherwise nlevels device pointers. */
int dtcwt_hip_plan2d_forward(dtcwt_hip_plan2d *plan, const float *X, float *Yl,
                             void *const *Yh, float *const *Ys);
/* gain_mask_host: 6*nlevels doubles, gain_mask[d*nlevels + l] (the reference's (6,
 * nlevels) array, C order) or NULL for all ones.  Z: [batch][ext rows][ext cols]. */
int dtcwt_hip_plan2d_inverse(dtcwt_hip_plan2d *plan, const float *Yl, const void *const *Yh,
                             const double *gain_mask_host, float *Z);


/* Band-pass ("_bp") wavelet sets: the third filter of a 6-vector biort (h2o, g2o; m_biort taps, 0:
 * none) and of a 12-vector q-shift (h2a, h2b, g2a, g2b; m_qshift taps, 0: none), used for the diagonal
 * subbands (dtcwt/numpy/transform2d.py:116-129, :145-155, :250-271, :283-291).  Call once after
 * dtcwt_hip_plan2d_create; HOST pointers.  Returns -3 when no fused band-pass kernel exists for
 * these lengths (the caller then uses the generic filters). */
int dtcwt_hip_plan2d_set_bandpass(dtcwt_hip_plan2d *plan, const double *h2o, const double *g2o, int m_biort,
                                  const double *h2a, const double *h2b, const double *g2a, const double *g2b,
                                  int m_qshift);
/* Per-kernel timing for the benchmark's roofline report: when enabled, every level kernel
 * of forward/inverse is bracketed by a hipEvent pair on the plan's stream;
 * kernel_ms() synchronises and returns the durations of the LAST forward (fwd_ms[l]) and
 * inverse (inv_ms[l]) call, l = 0 .. nlevels-1.  Either pointer may be NULL. */
int dtcwt_hip_plan2d_set_profiling(dtcwt_hip_plan2d *plan, int enable);
int dtcwt_hip_plan2d_kernel_ms(dtcwt_hip_plan2d *plan, float *fwd_ms, float *inv_ms);
/* Which levels share a launch: *fwd12 = 1 when levels 1 + 2 of the forward transform (without `scales`) run as ONE
 * marching launch whose level-1 lowpass never leaves the registers (dtcwt/numpy/transform2d.py:112-160 in one pass),
 * *inv21 = 1 likewise for levels 2 + 1 of the inverse (:242-293).  kernel_ms() then reports the shared launch under
 * the level it starts with (fwd_ms[0], inv_ms[1]) and an empty event pair under the other.  Either pointer may be NULL. */
int dtcwt_hip_plan2d_launches(const dtcwt_hip_plan2d *plan, int *fwd12, int *inv21);
/* (ABI 5) *fwd1 / *inv1 = 1 when level 1 of the forward / inverse runs as a marching launch of its own (near_sym_b's 13 / 19
 * taps and antonini's 9 / 7 need a window the fused launch above has no registers for; dtcwt_amd/csrc/march2d_l1.hpp).
 * Level 2 then runs as a marching launch of its own as well where the q-shift set has one (k_fwd2m / k_inv2m: the 14- and
 * 18-tap sets, march2d_pair.hpp / march2d_ipair.hpp), levels >= 3 -- and level 2 of every other set -- on the tile programs.
 * Same choice rules and pin as dtcwt_hip_plan2d_launches. */
int dtcwt_hip_plan2d_level1_march(const dtcwt_hip_plan2d *plan, int *fwd1, int *inv1);
/* (ABI 6) Which kernel runs every level of this plan, as one line of text in `buf` (>= 64 bytes; 512 hold any plan):
 *   fwd: L1+2 k_fwd12m | L3 k_fwd2 | L4 k_fwd2 ; inv: L4 k_inv2 | L3 k_inv2 | L2+1 k_inv21p ; march=auto band=0 parts=0xff xcd_order=-1 program=auto in_flight=1 cu_shares=1
 * scales != 0: as dtcwt_hip_plan2d_forward runs it with Ys != NULL.  It consults the very predicates the forward and the inverse
 * consult -- what it says is what runs -- and ends with the environment switches AS THE PLAN READ THEM WHEN IT WAS CREATED
 * (DTCWT_HIP_MARCH, DTCWT_HIP_MARCH_BAND, DTCWT_HIP_MARCH_PARTS, DTCWT_HIP_XCD_ORDER: INTEGRATION.md section 5; a plan keeps
 * them for life, later changes of the environment reach new plans only), the pinned program, the concurrency hint and the
 * number of shares the context divides its device into. */
int dtcwt_hip_plan2d_describe(const dtcwt_hip_plan2d *plan, int scales, char *buf, size_t len);
/* How many independent transforms the caller keeps in flight on this device at a time (this plan's included; other
 * plans on other streams -- the images of a video, the members of a batch handed over one by one; default 1).  The
 * marching launches cut an image into bands of rows, each of which re-reads the rows its filters reach into above
 * and below: one image alone needs ~40-row bands to fill the GPU, four in flight are served better by ~150-row bands
 * (4096^2 fwd + inv: 0.152 against 0.167 ms per image).  A hint only: it moves the size from
 * which levels 1 + 2 run as one launch (dtcwt_hip_plan2d_launches), and the two programs agree to rounding (2e-7), not to the bit;
 * dtcwt_hip_plan2d_set_program pins one. */
int dtcwt_hip_plan2d_set_concurrency(dtcwt_hip_plan2d *plan, int transforms_in_flight);
/* (ABI 5) Which program computes levels 1 + 2 of the forward / 2 + 1 of the inverse.  AUTO (the default): the library picks
 * per call -- the one-launch marching program where the geometry and the filters allow it AND the call is large enough to
 * pay (batch x pixels, the concurrency hint, a partition context: the table in dtcwt_amd/csrc/march2d.hip), the per-level
 * tile programs otherwise.  The two programs evaluate the same sums in a different order and agree to ~2e-7 relative, NOT
 * to the bit: under AUTO the same image may therefore differ in the last place between a call on its own and a call as
 * part of a batch, and an inverse may run the other program than the forward that made the pyramid.  Callers that need
 * run-to-run / batch-to-single bit reproducibility pin one: TILES is the round-1..3 behaviour at every size, MARCH the
 * one-launch form wherever it applies (tiles where it does not: odd sizes, `scales`, other filter lengths).  A pin on
 * the plan wins over the DTCWT_HIP_MARCH environment switch, which only the AUTO mode consults. */
#define DTCWT_HIP_PROGRAM_AUTO (-1)
#define DTCWT_HIP_PROGRAM_TILES 0
#define DTCWT_HIP_PROGRAM_MARCH 1
int dtcwt_hip_plan2d_set_program(dtcwt_hip_plan2d *plan, int program);

/* A plan's level loop as a hipGraph on fixed buffers: the forward transform of X into (Yl, Yh[, Ys])
 * and, when Z is not NULL, the inverse of that pyramid into Z (gain_mask_host as for
 * dtcwt_hip_plan2d_inverse, read at capture time), captured from the plan's stream once and
 * replayed by graph_launch on the same stream.  For loops over same-shaped inputs that are copied
 * into X (dtcwt/numpy/transform2d.py:40-295 called repeatedly); profiling must be off. */
typedef struct dtcwt_hip_graph dtcwt_hip_graph;
int dtcwt_hip_plan2d_capture(dtcwt_hip_plan2d *plan, const float *X, float *Yl, void *const *Yh,
                             float *const *Ys, const double *gain_mask_host, float *Z,
                             dtcwt_hip_graph **graph);
int dtcwt_hip_graph_launch(dtcwt_hip_graph *graph);
int dtcwt_hip_graph_destroy(dtcwt_hip_graph *graph);

/* ---------------------------------------------------------------- 3-D / 1-D plans --- */
/* Whole-transform entry points, as the reference's Transform3d.forward / .inverse
 * (dtcwt/numpy/transform3d.py:37-131, :133-206) and Transform1d.forward / .inverse
 * (dtcwt/numpy/transform1d.py:26-110, :112-180) are: the plan fixes every level's geometry (ext_mode
 * 4 / 8 padding, :322-335; the replicated end samples of odd 1-D levels, transform1d.py:95-96), owns the
 * lowpass workspaces between levels and sequences the level kernels on the context's stream.
 * create returns -3 when some level has no one- or two-launch level kernel (even-length biort filters,
 * q-shift lengths outside the table, k not 1 or >= 32 for 1-D): the host then sequences the generic
 * filters itself.  forward / inverse return -3 in the same situations discovered late (tiny levels).
 *
 * plan3d: float32, X [n0][n1][n2] (multiples of 2, ext_mode 4, or 4, ext_mode 8).
 *   shapes: s[0..2] = Yl extents; then per level l six values: Yh[l] extents (x 28 complex64), scale extents.
 *   forward: Yh = nlevels device pointers (Yh[0] unused with discard_level_1 != 0, :291-315); Ys NULL or
 *            nlevels pointers (include_scale).
 *   inverse: Yh[0] may be NULL (pyramid of a discard_level_1 forward): level 1 is then the lowpass-only merge,
 *            colfilter(., g0o) along axes 1, 0, 2.  The reference's `_level1_ifm_no_highpass` (:442-458) omits
 *            a transpose there and returns cubic volumes with axes 0 and 2 exchanged; reference_quirks != 0 asks
 *            for that literal behaviour, which only the host-sequenced path provides (returns -3 here).
 * plan1d: dtype float32 / float64, X [n][k] (k signals side by side), gain_host: nlevels doubles or NULL.
 *   shapes: s[0] = lowpass length; then per level: highpass length, scale length. */
typedef struct dtcwt_hip_plan3d dtcwt_hip_plan3d;
int dtcwt_hip_plan3d_create(dtcwt_hip_ctx *ctx, int64_t n0, int64_t n1, int64_t n2, int nlevels, int ext_mode,
                            const double *const *biort_host, const int *biort_len,
                            const double *const *qshift_host, const int *qshift_len, dtcwt_hip_plan3d **plan);
int dtcwt_hip_plan3d_destroy(dtcwt_hip_plan3d *plan);
int dtcwt_hip_plan3d_shapes(const dtcwt_hip_plan3d *plan, int64_t *shapes);
int dtcwt_hip_plan3d_forward(dtcwt_hip_plan3d *plan, const float *X, float *Yl, void *const *Yh, float *const *Ys,
                             int discard_level_1);
int dtcwt_hip_plan3d_inverse(dtcwt_hip_plan3d *plan, const float *Yl, const void *const *Yh, float *Z,
                             int reference_quirks);
typedef struct dtcwt_hip_plan1d dtcwt_hip_plan1d;
int dtcwt_hip_plan1d_create(dtcwt_hip_ctx *ctx, int dtype, int64_t n, int64_t k, int nlevels,
                            const double *const *biort_host, const int *biort_len,
                            const double *const *qshift_host, const int *qshift_len, dtcwt_hip_plan1d **plan);
int dtcwt_hip_plan1d_destroy(dtcwt_hip_plan1d *plan);
int dtcwt_hip_plan1d_shapes(const dtcwt_hip_plan1d *plan, int64_t *shapes);
int dtcwt_hip_plan1d_forward(dtcwt_hip_plan1d *plan, const void *X, void *Yl, void *const *Yh, void *const *Ys);
int dtcwt_hip_plan1d_inverse(dtcwt_hip_plan1d *plan, const void *Yl, const void *const *Yh, const double *gain_host,
                             void *Z);

/* ---------------------------------------------------------------- multi-GPU --------- */
/* A batch of independent images sharded over the GPUs of one node from ONE process: contiguous
 * split (shard d owns images [start_d, start_d + count_d), sizes differing by at most one), one
 * context + stream + fused plan + host worker thread per shard, no data-path collective.  This is
 * the scatter / transform / gather of the reference's only parallel code, the MPI frame groups of
 * examples/register_video.py:125-156, with device-resident shards.  `devices` may name a device
 * more than once (several shards on one GPU).
 * flags: DTCWT_HIP_MGPU_BCAST_TAPS = the packed tap table travels from shard 0's device to every
 * other device by ONE RCCL broadcast (single-process communicator over the shard devices, xGMI) and
 * each shard builds its plan from the copy that arrived on its device; without it the host table
 * is used directly.  Per-shard pointer arrays are indexed [shard] (X, Yl, Z) and
 * [shard * nlevels + level] (Yh).  Calls return once every shard's launches are ENQUEUED;
 * dtcwt_hip_mgpu_sync() waits for the devices. */
#define DTCWT_HIP_MGPU_BCAST_TAPS 1
typedef struct dtcwt_hip_mgpu dtcwt_hip_mgpu;
int dtcwt_hip_mgpu_create(int ndev, const int *devices, int batch, int rows, int cols, int nlevels,
                          const double *const *biort_host, const int *biort_len,
                          const double *const *qshift_host, const int *qshift_len, int flags,
                          dtcwt_hip_mgpu **mgpu);
/* (ABI 5) The same object as one of `nlanes` the caller keeps in flight on the same devices, each with batches of its own
 * (the frames of a video handed to the node group by group; bench.py --mgpu rotates its steps over four): lane `lane`'s
 * shard contexts are share `lane` of `nlanes` of their device's compute units (dtcwt_hip_ctx_create_partition) where that
 * measured faster -- FOUR lanes from images of 1024 x 1024, TWO lanes from images of 2048 x 2048 (mgpu.hip: lane_on_a_share;
 * profiles/r04/ab_partition.txt, profiles/r05/batch_streams.txt) -- and plain contexts whose plans carry the concurrency hint
 * `nlanes` otherwise.  Partition contexts own BLOCKING streams (see dtcwt_hip_ctx_create_partition): a caller whose framework
 * enqueues on the NULL stream of the same device serialises with them at every size the rule above partitions; DTCWT_HIP_MGPU_PARTITION / _NO_PARTITION in `flags`
 * force one or the other.  This gives the one-process path the engine of the one-process-per-GPU path: one 4096 x 4096
 * image per device per call runs at the four-in-flight rate (0.15-0.16 ms) instead of the one-at-a-time rate (0.19 ms).
 * dtcwt_hip_mgpu_create is lane 0 of 1.  dtcwt_hip_mgpu_shares: the number of shares a lane's contexts divide their
 * device into (1: whole devices). */
#define DTCWT_HIP_MGPU_PARTITION 2
#define DTCWT_HIP_MGPU_NO_PARTITION 4
int dtcwt_hip_mgpu_create_lane(int ndev, const int *devices, int batch, int rows, int cols, int nlevels,
                               const double *const *biort_host, const int *biort_len,
                               const double *const *qshift_host, const int *qshift_len, int flags,
                               int lane, int nlanes, dtcwt_hip_mgpu **mgpu);
int dtcwt_hip_mgpu_shares(const dtcwt_hip_mgpu *mgpu);
int dtcwt_hip_mgpu_destroy(dtcwt_hip_mgpu *mgpu);
int dtcwt_hip_mgpu_ndev(const dtcwt_hip_mgpu *mgpu);
int dtcwt_hip_mgpu_taps_broadcast(const dtcwt_hip_mgpu *mgpu);          /* 1: plans built from RCCL-delivered taps */
int dtcwt_hip_mgpu_shard(const dtcwt_hip_mgpu *mgpu, int shard, int *device, int *start, int *count);
dtcwt_hip_ctx *dtcwt_hip_mgpu_ctx(dtcwt_hip_mgpu *mgpu, int shard);     /* for allocations / copies on that shard */
int dtcwt_hip_mgpu_shapes(const dtcwt_hip_mgpu *mgpu, int *shapes);     /* per image, as dtcwt_hip_plan2d_shapes */
int dtcwt_hip_mgpu_forward2d(dtcwt_hip_mgpu *mgpu, const float *const *X, float *const *Yl, void *const *Yh);
/* the same with include_scale (transform2d.py:96-99, :160-163): Ys[shard * nlevels + level] receives the lowpass
 * image of every level, [count][lo_r][lo_c] as dtcwt_hip_plan2d_shapes reports them; Ys == NULL: no scales */
int dtcwt_hip_mgpu_forward2d_scales(dtcwt_hip_mgpu *mgpu, const float *const *X, float *const *Yl, void *const *Yh,
                                    float *const *Ys);
int dtcwt_hip_mgpu_inverse2d(dtcwt_hip_mgpu *mgpu, const float *const *Yl, const void *const *Yh,
                             const double *gain_mask_host, float *const *Z);
int dtcwt_hip_mgpu_sync(dtcwt_hip_mgpu *mgpu);
/* host batch [batch][bytes_per_image] <-> the shards' device buffers (each shard copies its own slice from its thread) */
int dtcwt_hip_mgpu_scatter(dtcwt_hip_mgpu *mgpu, const void *host, size_t bytes_per_image, void *const *dev);
int dtcwt_hip_mgpu_gather(dtcwt_hip_mgpu *mgpu, const void *const *dev, size_t bytes_per_image, void *host);
/* The same without waiting: `host` must be page-locked (dtcwt_hip_host_alloc) and stay untouched until
 * dtcwt_hip_mgpu_sync().  Every shard's upload is enqueued on its own stream (ordered before the transforms issued
 * after it), every shard's download on its context's copy stream behind what its stream holds so far -- a host-fed
 * batch (C5: 8.6 GB in, 43 GB out over eight host links) then overlaps its copies with the other shards' kernels
 * instead of eight blocking, staged, pageable copies (the reference's examples/register_video.py:125-156 scatters
 * frames with blocking MPI sends). */
int dtcwt_hip_mgpu_scatter_async(dtcwt_hip_mgpu *mgpu, const void *host, size_t bytes_per_image, void *const *dev);
int dtcwt_hip_mgpu_gather_async(dtcwt_hip_mgpu *mgpu, const void *const *dev, size_t bytes_per_image, void *host);

/* ---------------------------------------------------------------- re-sampling ------ */
/* Replaces dtcwt/sampling.py.  An image is [H][W][ncomp] of the real
 * dtype (channels; complex data counts two components per channel).  Coordinates are DEVICE
 * arrays of double; (x, y) is the centre of im[y][x]; outside the image the half-sample
 * symmetric extension applies (sampling.py:36-40).  method: */
#define DTCWT_HIP_SAMPLE_NEAREST 0     /* sampling.py:42-43 */
#define DTCWT_HIP_SAMPLE_BILINEAR 1    /* sampling.py:45-66 */
#define DTCWT_HIP_SAMPLE_LANCZOS 2     /* sampling.py:68-103, window radius 3 */
/* sample: dtcwt/sampling.py:105-129.  out: [npts][ncomp]. */
int dtcwt_hip_sample(dtcwt_hip_ctx *ctx, int dtype, const void *im, int64_t H, int64_t W, int64_t ncomp,
                     const double *xs, const double *ys, int64_t npts, int method, void *out);
/* rescale: dtcwt/sampling.py:131-165 (the sample grid is computed in the kernel).
 * out: [out_h][out_w][ncomp]. */
int dtcwt_hip_rescale(dtcwt_hip_ctx *ctx, int dtype, const void *im, int64_t H, int64_t W, int64_t ncomp,
                      int64_t out_h, int64_t out_w, int method, void *out);
/* upsample by two along both axes: dtcwt/sampling.py:280-367.  offsets / w_even / w_odd are
 * HOST arrays of the per-axis taps (outputs 2i and 2i+1 sit at i - 1/4 and i + 1/4).
 * out: [2H][2W][ncomp]. */
int dtcwt_hip_upsample2(dtcwt_hip_ctx *ctx, int dtype, const void *im, int64_t H, int64_t W, int64_t ncomp,
                        int ntaps, const int *offsets, const double *w_even, const double *w_odd, void *out);
/* Phase rolling of complex subbands: dtcwt/sampling.py:167-190 (`_phase_image`) fused with
 * the subband selection `im[:, :, sbs]` (:213, :270).
 *   out[p][k] = in[p][src[k]] * exp(sign * j * (dtheta_dx[k] * x_p + dtheta_dy[k] * y_p))
 * in: [..][nin] complex, out: [..][nch] complex (nch <= 6; src, dtheta_* are HOST arrays).
 * _grid: p runs over an H x W array whose pixel (py, px) sits at
 * (xscale (px + 1/2) - 1/2, yscale (py + 1/2) - 1/2); _points: x_p, y_p from device arrays. */
int dtcwt_hip_phase_roll_grid(dtcwt_hip_ctx *ctx, int dtype, const void *in, int64_t H, int64_t W, int64_t nin,
                              int nch, const int *src, const double *dtheta_dx, const double *dtheta_dy,
                              double xscale, double yscale, double sign, void *out);
int dtcwt_hip_phase_roll_points(dtcwt_hip_ctx *ctx, int dtype, const void *in, int64_t npts, int64_t nin, int nch,
                                const int *src, const double *dtheta_dx, const double *dtheta_dy,
                                const double *xs, const double *ys, double sign, void *out);

/* ---------------------------------------------------------------- registration ----- */
/* Replaces the per-pixel loops of dtcwt/registration.py.  All results
 * are float64; Yh_* are the [H][W][6] complex subband records of one pyramid level (dtype =
 * their real type). */
/* One level of `qtildematrices` (registration.py:140-214) with `confidence` (:83-137) and
 * `phasegradient` (:31-75) folded in.  out: [H][W][27]. */
int dtcwt_hip_qtilde(dtcwt_hip_ctx *ctx, int dtype, const void *Yh_ref, const void *Yh_target, int64_t H, int64_t W,
                     double epsilon, double *out);
/* `solvetransform` (registration.py:216-250): a = -Q^{-1} q per row of Qt [n][27] -> a [n][6]
 * (the reference fills only the upper triangle of Q, :231-232: a back substitution). */
int dtcwt_hip_solve6(dtcwt_hip_ctx *ctx, const double *Qt, int64_t n, double *a);
/* `_boxfilter` (registration.py:417-446): odd kernel_size, first two axes of in [H][W][K]. */
int dtcwt_hip_boxfilter(dtcwt_hip_ctx *ctx, const double *in, int64_t H, int64_t W, int64_t K, int kernel_size,
                        double *out);
/* out[c] = sum_p in[p][c]: the `np.sum(np.sum(x, axis=0), axis=0)` of registration.py:341-344. */
int dtcwt_hip_colsum(dtcwt_hip_ctx *ctx, const double *in, int64_t n, int64_t K, double *out);
/* `velocityfield` before its rescale (registration.py:385-390): avecs [h][w][6] -> vx, vy [h][w]. */
int dtcwt_hip_affine_velocity(dtcwt_hip_ctx *ctx, const double *avecs, int64_t h, int64_t w, double *vx, double *vy);
/* sample positions of `warp` / `warphighpass` (registration.py:401-415) in pixels:
 * xs = ((x / W)_float32 + vx) W, ys likewise. */
int dtcwt_hip_warp_coords(dtcwt_hip_ctx *ctx, const double *vx, const double *vy, int64_t H, int64_t W, double *xs,
                          double *ys);
/* y += alpha x (the `qts +=` / `avecs +=` of registration.py:368-370); out[p][k] = row[k] (HOST
 * row of K <= 8 values: the broadcast of the global estimate, :347-348). */
int dtcwt_hip_axpy(dtcwt_hip_ctx *ctx, int64_t n, double alpha, const double *x, double *y);
int dtcwt_hip_fill_rows(dtcwt_hip_ctx *ctx, int64_t n, int K, const double *row, double *out);
/* The whole of `estimatereg` (registration.py:301-372) in one call: the sequence of the kernels
 * above and of the re-sampling ones, without a host round trip per launch.
 * Yh_src / Yh_ref: HOST arrays of `nlevels` device pointers to the [H_l][W_l][6] complex records
 * of the two pyramids (entries of unused levels may be NULL); shapes: HOST [nlevels][2] = (H_l, W_l);
 * the level schedule is `ngroups` lists, group g = group_sizes[g] consecutive entries of
 * group_levels (0-based level indices): group 0 gives the global estimate, the others refine it.
 * avecs: DEVICE [reg_h][reg_w][6] float64 out. */
int dtcwt_hip_estimatereg(dtcwt_hip_ctx *ctx, int dtype, int nlevels, const void *const *Yh_src,
                          const void *const *Yh_ref, const int64_t *shapes, int64_t reg_h, int64_t reg_w,
                          int ngroups, const int *group_sizes, const int *group_levels, double *avecs);

#ifdef __cplusplus
}
#endif
#endif /* DTCWT_HIP_H */
