// Whole-transform entry points for the 3-D and the 1-D DT-CWT: dtcwt_hip_plan3d_* and
// dtcwt_hip_plan1d_* of include/dtcwt_hip.h.
//
// Transform3d.forward / .inverse (dtcwt/numpy/transform3d.py:37-131, :133-206) and
// Transform1d.forward / .inverse (dtcwt/numpy/transform1d.py:26-110, :112-180) are single calls
// in the reference; here too: a plan fixes the geometry of every level (the edge padding of
// ext_mode 4 / 8, the odd-sized 1-D levels), owns the lowpass workspaces between levels, and
// sequences the per-level kernels on the context's stream from native code -- one library call
// per transform instead of one per level and pass from Python.  A plan only exists where EVERY
// level has a one- or two-launch level kernel (float32 3-D with odd-length biort filters and
// table q-shift lengths; 1-D float32 / float64 with one signal or >= 32 signals side by side):
// creation returns -3 otherwise and the host keeps sequencing the generic filters.
#include <cstring>
#include <vector>

#include "common.hpp"

namespace {

struct Lv3 {
    int64_t in[3];      // input extents of the level
    int pad[3];         // planes replicated per side (levels >= 2)
    int64_t lo[3];      // lowpass output extents
    int64_t hi[3];      // highpass extents (x 28 complex)
};

std::vector<double> vec(const double *p, int n) { return std::vector<double>(p, p + n); }

}  // namespace

struct dtcwt_hip_plan3d {
    dtcwt_hip_ctx *ctx;
    int nlevels, ext_mode;
    std::vector<Lv3> lv;
    std::vector<double> biort[4];     // h0o g0o h1o g1o
    std::vector<double> qshift[8];    // h0a h0b g0a g0b h1a h1b g1a g1b
    std::vector<float *> work;        // lowpass of level l (l < nlevels - 1), also the inverse's intermediate
    float *tmp[2] = {nullptr, nullptr};   // highpass-free level 1 (discard_level_1 / Yh[0] == NULL): axis passes
    bool fwd_ok = false, inv_ok = false;  // every level of that direction has a fused level kernel
};

struct dtcwt_hip_plan1d {
    dtcwt_hip_ctx *ctx;
    int dtype, nlevels;
    int64_t n, k;
    std::vector<int64_t> lo_n;        // lowpass length after level l
    std::vector<int> pad;             // 1: level l (>= 1) replicates one sample per end (transform1d.py:95-96)
    std::vector<double> biort[4], qshift[8];
    std::vector<void *> work;
};

extern "C" {

// ------------------------------------------------------------------------------- 3-D
int dtcwt_hip_plan3d_create(dtcwt_hip_ctx *ctx, int64_t n0, int64_t n1, int64_t n2, int nlevels, int ext_mode,
                            const double *const *biort_host, const int *biort_len,
                            const double *const *qshift_host, const int *qshift_len, dtcwt_hip_plan3d **out) {
    DT_REQUIRE(ctx && out && biort_host && biort_len && qshift_host && qshift_len, "NULL argument");
    DT_REQUIRE(nlevels >= 1 && n0 >= 1 && n1 >= 1 && n2 >= 1, "bad plan extents");
    DT_REQUIRE(ext_mode == 4 || ext_mode == 8, "ext_mode must be one of 4 or 8");
    const int mult1 = ext_mode == 4 ? 2 : 4;
    DT_REQUIRE(n0 % mult1 == 0 && n1 % mult1 == 0 && n2 % mult1 == 0,
               "Input shape should be a multiple of %d in each direction when ext_mode == %d", mult1, ext_mode);
    for (int i = 0; i < 4; ++i) DT_REQUIRE(biort_len[i] >= 1 && biort_len[i] <= DTCWT_HIP_MAX_TAPS, "biort length out of range");
    for (int i = 0; i < 8; ++i)
        DT_REQUIRE(qshift_len[i] == qshift_len[0] && qshift_len[i] <= DTCWT_HIP_MAX_TAPS, "qshift lengths differ");
    if (biort_len[0] % 2 == 0 || biort_len[1] % 2 == 0 || biort_len[2] % 2 == 0 || biort_len[3] % 2 == 0)
        return dtcwt_set_error(-3, "even-length biort filters: the 3-D level-1 kernels need odd lengths");
    dtcwt_hip_plan3d *p = new dtcwt_hip_plan3d();
    p->ctx = ctx; p->nlevels = nlevels; p->ext_mode = ext_mode;
    for (int i = 0; i < 4; ++i) p->biort[i] = vec(biort_host[i], biort_len[i]);
    for (int i = 0; i < 8; ++i) p->qshift[i] = vec(qshift_host[i], qshift_len[i]);
    Lv3 l0{{n0, n1, n2}, {0, 0, 0}, {n0, n1, n2}, {n0 / 2, n1 / 2, n2 / 2}};
    p->lv.push_back(l0);
    const int mult = ext_mode == 4 ? 4 : 8, npad = ext_mode == 4 ? 1 : 2;
    for (int l = 1; l < nlevels; ++l) {                       // transform3d.py:317-335
        const Lv3 &pr = p->lv.back();
        Lv3 L{};
        for (int a = 0; a < 3; ++a) {
            L.in[a] = pr.lo[a];
            L.pad[a] = (L.in[a] % mult) ? npad : 0;
            const int64_t ext = L.in[a] + 2 * L.pad[a];
            if (ext % 4) {
                delete p;
                return dtcwt_set_error(-1, "level %d extent %lld is not a multiple of 4 after padding", l + 1, (long long)ext);
            }
            L.lo[a] = ext / 2; L.hi[a] = ext / 4;
        }
        p->lv.push_back(L);
    }
    // which directions run natively at every level (the conditions of the level entry points, fused3d.hip)
    const int mq = qshift_len[0];
    p->fwd_ok = dtcwt_fwd3_level1_ok(n0, n1, n2, biort_len[0], biort_len[2], biort_host[0], biort_host[2]);
    p->inv_ok = dtcwt_inv3_level1_ok(n0, n1, n2, biort_len[1], biort_len[3], biort_host[1], biort_host[3]);
    for (int l = 1; l < nlevels; ++l) {
        const Lv3 &L = p->lv[l];
        p->fwd_ok = p->fwd_ok && dtcwt_fwd3_level2_ok(L.in[0], L.in[1], L.in[2], L.pad[0], L.pad[1], L.pad[2], mq);
        p->inv_ok = p->inv_ok && dtcwt_inv3_level2_ok(L.lo[0], L.lo[1], L.lo[2], L.pad[0], mq);
    }
    if (!p->fwd_ok && !p->inv_ok) {
        delete p;
        return dtcwt_set_error(-3, "some level of this volume has no fused 3-D level kernel in either direction");
    }
    p->work.assign(nlevels, nullptr);
    for (int l = 0; l < nlevels; ++l) {
        void *d = nullptr;
        size_t bytes = (size_t)(p->lv[l].lo[0] * p->lv[l].lo[1] * p->lv[l].lo[2]) * sizeof(float);
        int rc = dtcwt_hip_malloc(ctx, bytes, &d);
        if (rc) { dtcwt_hip_plan3d_destroy(p); return rc; }
        p->work[l] = (float *)d;
    }
    *out = p;
    return 0;
}

int dtcwt_hip_plan3d_destroy(dtcwt_hip_plan3d *p) {
    if (!p) return 0;
    for (float *w : p->work) if (w) dtcwt_hip_free(p->ctx, w);
    for (float *w : p->tmp) if (w) dtcwt_hip_free(p->ctx, w);
    delete p;
    return 0;
}

// shapes[0..2]: Yl; then per level 6 values: Yh extents (x 28), scale extents
int dtcwt_hip_plan3d_shapes(const dtcwt_hip_plan3d *p, int64_t *s) {
    DT_REQUIRE(p && s, "NULL argument");
    for (int a = 0; a < 3; ++a) s[a] = p->lv.back().lo[a];
    for (int l = 0; l < p->nlevels; ++l)
        for (int a = 0; a < 3; ++a) { s[3 + 6 * l + a] = p->lv[l].hi[a]; s[6 + 6 * l + a] = p->lv[l].lo[a]; }
    return 0;
}

// colfilter of a whole [n0][n1][n2] float32 volume along `axis` (odd-length h: same size out)
static int axis_colfilter3(dtcwt_hip_ctx *ctx, const float *X, float *Y, const int64_t n[3], int axis,
                           const std::vector<double> &h) {
    dtcwt_hip_view v{};
    v.outer = 1; for (int a = 0; a < axis; ++a) v.outer *= n[a];
    v.n = n[axis];
    v.inner = 1; for (int a = axis + 1; a < 3; ++a) v.inner *= n[a];
    v.xso = v.n * v.inner; v.xsn = v.inner; v.xsi = 1;
    v.yso = v.n * v.inner; v.ysn = v.inner; v.ysi = 1;
    return dtcwt_hip_colfilter(ctx, DTCWT_HIP_F32, X, Y, &v, h.data(), (int)h.size(), 0);
}

static int lowpass_only_level1(dtcwt_hip_plan3d *p, const float *X, float *Y, const std::vector<double> &h,
                               const int order[3]) {
    const Lv3 &L = p->lv[0];
    const size_t bytes = (size_t)(L.in[0] * L.in[1] * L.in[2]) * sizeof(float);
    for (float *&t : p->tmp)
        if (!t) {
            void *d = nullptr;
            if (int rc = dtcwt_hip_malloc(p->ctx, bytes, &d)) return rc;
            t = (float *)d;
        }
    if (int rc = axis_colfilter3(p->ctx, X, p->tmp[0], L.in, order[0], h)) return rc;
    if (int rc = axis_colfilter3(p->ctx, p->tmp[0], p->tmp[1], L.in, order[1], h)) return rc;
    return axis_colfilter3(p->ctx, p->tmp[1], Y, L.in, order[2], h);
}

// Yh[l] NULL at level 0 with discard_level_1 (transform3d.py:291-315); Ys NULL or nlevels pointers
int dtcwt_hip_plan3d_forward(dtcwt_hip_plan3d *p, const float *X, float *Yl, void *const *Yh, float *const *Ys,
                             int discard_level_1) {
    DT_REQUIRE(p && X && Yl && Yh, "NULL argument");
    if (!p->fwd_ok) return dtcwt_set_error(-3, "a level of the forward transform has no fused kernel for this volume");
    const int nl = p->nlevels;
    const float *in = X;
    for (int l = 0; l < nl; ++l) {
        const Lv3 &L = p->lv[l];
        float *lo = (l == nl - 1 && !Ys) ? Yl : (Ys ? Ys[l] : p->work[l]);
        DT_REQUIRE(lo, "NULL output buffer at level %d", l);
        int rc;
        if (l == 0 && discard_level_1) {
            const int order[3] = {2, 1, 0};
            rc = lowpass_only_level1(p, in, lo, p->biort[0], order);
        } else if (l == 0) {
            DT_REQUIRE(Yh[0], "NULL Yh at level 0");
            rc = dtcwt_hip_fwd3_level1(p->ctx, in, L.in[0], L.in[1], L.in[2], p->biort[0].data(), (int)p->biort[0].size(),
                                       p->biort[2].data(), (int)p->biort[2].size(), lo, (float *)Yh[0]);
        } else {
            DT_REQUIRE(Yh[l], "NULL Yh at level %d", l);
            // coldfilt(., h0b, h0a) / coldfilt(., h1b, h1a)   (transform3d.py:353-369)
            rc = dtcwt_hip_fwd3_level2(p->ctx, in, L.in[0], L.in[1], L.in[2], L.pad[0], L.pad[1], L.pad[2],
                                       p->qshift[1].data(), p->qshift[0].data(), p->qshift[5].data(), p->qshift[4].data(),
                                       (int)p->qshift[0].size(), lo, (float *)Yh[l]);
        }
        if (rc) return rc;
        in = lo;
    }
    if (Ys) {
        const Lv3 &L = p->lv[nl - 1];
        DT_CHECK_HIP(hipMemcpyAsync(Yl, Ys[nl - 1], (size_t)(L.lo[0] * L.lo[1] * L.lo[2]) * sizeof(float),
                                    hipMemcpyDeviceToDevice, p->ctx->stream));
    }
    return 0;
}

// Yh[0] may be NULL (the pyramid of a discard_level_1 forward): level 1 is then the lowpass-only merge
// colfilter(., g0o) along axes 1, 0, 2.  quirks != 0 reproduces the reference's `_level1_ifm_no_highpass`
// (transform3d.py:442-458) literally: it filters along 1, 0, 2 but omits the final transpose, so a cubic
// volume comes back with axes 0 and 2 exchanged (non-cubic volumes raise in the reference; here: error).
int dtcwt_hip_plan3d_inverse(dtcwt_hip_plan3d *p, const float *Yl, const void *const *Yh, float *Z, int quirks) {
    DT_REQUIRE(p && Yl && Yh && Z, "NULL argument");
    if (!p->inv_ok) return dtcwt_set_error(-3, "a level of the inverse transform has no fused kernel for this volume");
    const int nl = p->nlevels;
    const float *in = Yl;
    for (int l = nl - 1; l >= 0; --l) {
        const Lv3 &L = p->lv[l];
        int rc;
        if (l == 0) {
            if (!Yh[0]) {
                if (quirks) return dtcwt_set_error(-3, "reference-quirk mode of the highpass-free level 1 is host-sequenced");
                const int order[3] = {1, 0, 2};
                rc = lowpass_only_level1(p, in, Z, p->biort[1], order);
            } else {
                rc = dtcwt_hip_inv3_level1(p->ctx, in, (const float *)Yh[0], L.in[0], L.in[1], L.in[2], p->biort[1].data(),
                                           (int)p->biort[1].size(), p->biort[3].data(), (int)p->biort[3].size(), Z);
            }
        } else {
            DT_REQUIRE(Yh[l], "NULL Yh at level %d", l);
            float *outb = p->work[l - 1];
            // colifilt(., g0b, g0a) + colifilt(., g1b, g1a), crop where the forward level padded (:460-526)
            rc = dtcwt_hip_inv3_level2(p->ctx, in, (const float *)Yh[l], L.lo[0], L.lo[1], L.lo[2], L.pad[0], L.pad[1],
                                       L.pad[2], p->qshift[3].data(), p->qshift[2].data(), p->qshift[7].data(),
                                       p->qshift[6].data(), (int)p->qshift[0].size(), outb);
            in = outb;
        }
        if (rc) return rc;
    }
    return 0;
}

// ------------------------------------------------------------------------------- 1-D
int dtcwt_hip_plan1d_create(dtcwt_hip_ctx *ctx, int dtype, int64_t n, int64_t k, int nlevels,
                            const double *const *biort_host, const int *biort_len,
                            const double *const *qshift_host, const int *qshift_len, dtcwt_hip_plan1d **out) {
    DT_REQUIRE(ctx && out && biort_host && biort_len && qshift_host && qshift_len, "NULL argument");
    DT_REQUIRE(dtype == DTCWT_HIP_F32 || dtype == DTCWT_HIP_F64, "bad dtype %d", dtype);
    DT_REQUIRE(nlevels >= 1 && n >= 2 && k >= 1, "bad plan extents");
    DT_REQUIRE(n % 2 == 0, "Size of input X must be a multiple of 2");
    if (!(k == 1 || k >= 32)) return dtcwt_set_error(-3, "one-launch 1-D levels take one signal or >= 32 side by side");
    for (int i = 0; i < 4; ++i) DT_REQUIRE(biort_len[i] >= 1 && biort_len[i] <= DTCWT_HIP_MAX_TAPS, "biort length out of range");
    for (int i = 0; i < 8; ++i)
        DT_REQUIRE(qshift_len[i] == qshift_len[0] && qshift_len[i] <= DTCWT_HIP_MAX_TAPS, "qshift lengths differ");
    if (biort_len[0] % 2 == 0 || biort_len[1] % 2 == 0 || biort_len[2] % 2 == 0 || biort_len[3] % 2 == 0)
        return dtcwt_set_error(-3, "even-length biort filters go through the generic filters");
    dtcwt_hip_plan1d *p = new dtcwt_hip_plan1d();
    p->ctx = ctx; p->dtype = dtype; p->nlevels = nlevels; p->n = n; p->k = k;
    for (int i = 0; i < 4; ++i) p->biort[i] = vec(biort_host[i], biort_len[i]);
    for (int i = 0; i < 8; ++i) p->qshift[i] = vec(qshift_host[i], qshift_len[i]);
    p->lo_n.push_back(n); p->pad.push_back(0);
    for (int l = 1; l < nlevels; ++l) {                       // transform1d.py:93-100
        const int64_t prev = p->lo_n.back();
        const int pd = (prev % 4) ? 1 : 0;
        p->pad.push_back(pd);
        p->lo_n.push_back((prev + 2 * pd) / 2);
    }
    const size_t es = dtype == DTCWT_HIP_F32 ? 4 : 8;
    p->work.assign(nlevels, nullptr);
    for (int l = 0; l < nlevels; ++l) {
        int rc = dtcwt_hip_malloc(ctx, (size_t)(p->lo_n[l] * k) * es, &p->work[l]);
        if (rc) { dtcwt_hip_plan1d_destroy(p); return rc; }
    }
    *out = p;
    return 0;
}

int dtcwt_hip_plan1d_destroy(dtcwt_hip_plan1d *p) {
    if (!p) return 0;
    for (void *w : p->work) if (w) dtcwt_hip_free(p->ctx, w);
    delete p;
    return 0;
}

// shapes[0] = lowpass length; then per level: highpass length, scale length
int dtcwt_hip_plan1d_shapes(const dtcwt_hip_plan1d *p, int64_t *s) {
    DT_REQUIRE(p && s, "NULL argument");
    s[0] = p->lo_n.back();
    for (int l = 0; l < p->nlevels; ++l) { s[1 + 2 * l] = p->lo_n[l] / 2; s[2 + 2 * l] = p->lo_n[l]; }
    return 0;
}

int dtcwt_hip_plan1d_forward(dtcwt_hip_plan1d *p, const void *X, void *Yl, void *const *Yh, void *const *Ys) {
    DT_REQUIRE(p && X && Yl && Yh, "NULL argument");
    const int nl = p->nlevels;
    const size_t es = p->dtype == DTCWT_HIP_F32 ? 4 : 8;
    const void *in = X;
    int64_t n_in = p->n;
    for (int l = 0; l < nl; ++l) {
        void *lo = (l == nl - 1 && !Ys) ? Yl : (Ys ? Ys[l] : p->work[l]);
        DT_REQUIRE(lo && Yh[l], "NULL output buffer at level %d", l);
        int rc;
        if (l == 0)     // Lo = colfilter(X, h0o), Hi = colfilter(X, h1o), Yh = Hi[::2] + 1j Hi[1::2]   (:79-88)
            rc = dtcwt_hip_level1d_forward(p->ctx, p->dtype, 0, in, n_in, p->k, 0, 0, p->biort[0].data(), nullptr,
                                           p->biort[2].data(), nullptr, (int)p->biort[0].size(), (int)p->biort[2].size(),
                                           lo, Yh[0]);
        else            // coldfilt(., h0b, h0a) / coldfilt(., h1b, h1a)   (:93-100)
            rc = dtcwt_hip_level1d_forward(p->ctx, p->dtype, 1, in, n_in, p->k, p->pad[l], p->pad[l], p->qshift[1].data(),
                                           p->qshift[0].data(), p->qshift[5].data(), p->qshift[4].data(),
                                           (int)p->qshift[0].size(), (int)p->qshift[0].size(), lo, Yh[l]);
        if (rc) return rc;
        in = lo; n_in = p->lo_n[l];
    }
    if (Ys)
        DT_CHECK_HIP(hipMemcpyAsync(Yl, Ys[nl - 1], (size_t)(p->lo_n[nl - 1] * p->k) * es, hipMemcpyDeviceToDevice,
                                    p->ctx->stream));
    return 0;
}

// gain_host: nlevels doubles (the reference's gain_mask, one per level) or NULL
int dtcwt_hip_plan1d_inverse(dtcwt_hip_plan1d *p, const void *Yl, const void *const *Yh, const double *gain_host,
                             void *Z) {
    DT_REQUIRE(p && Yl && Yh && Z, "NULL argument");
    const int nl = p->nlevels;
    const void *in = Yl;
    for (int l = nl - 1; l >= 0; --l) {
        DT_REQUIRE(Yh[l], "NULL Yh at level %d", l);
        const double g = gain_host ? gain_host[l] : 1.0;
        int rc;
        if (l == 0)     // Z = colfilter(Lo, g0o) + colfilter(unpack(Yh[0]), g1o)   (:162-176)
            rc = dtcwt_hip_level1d_inverse(p->ctx, p->dtype, 0, in, Yh[0], p->n, p->k, g, 0, p->biort[1].data(), nullptr,
                                           p->biort[3].data(), nullptr, (int)p->biort[1].size(), (int)p->biort[3].size(), Z);
        else {          // colifilt(Lo, g0b, g0a) + colifilt(unpack(Yh), g1b, g1a), cropped where the forward padded (:150-160)
            void *outb = p->work[l - 1];
            rc = dtcwt_hip_level1d_inverse(p->ctx, p->dtype, 1, in, Yh[l], p->lo_n[l], p->k, g, p->pad[l], p->qshift[3].data(),
                                           p->qshift[2].data(), p->qshift[7].data(), p->qshift[6].data(),
                                           (int)p->qshift[0].size(), (int)p->qshift[0].size(), outb);
            in = outb;
        }
        if (rc) return rc;
    }
    return 0;
}

}  // extern "C"
