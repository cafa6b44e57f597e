// Batch-sharded 2-D transforms over the GPUs of one node, driven from ONE process:
// dtcwt_hip_mgpu_* of include/dtcwt_hip.h.
//
// The path shards by image: a batch of B independent images is split
// contiguously over the devices, device d transforms images [start_d, start_d + count_d) with
// its own context, stream and fused plan -- no data-path collective, exactly the shape of the
// reference's only parallel code, the MPI frame scatter / gather of
// examples/register_video.py:125-156.  Every shard has a host worker thread, so the level
// launches of all devices are issued concurrently (one thread driving eight devices would
// serialise ~8 x 8 launches of ~4 us each per step, more than a step takes).
//
// The one collective is the broadcast of the packed filter-tap table from shard 0 at set-up
// (flag DTCWT_HIP_MGPU_BCAST_TAPS): a single-process RCCL communicator over the shard
// devices (ncclCommInitAll), ncclBroadcast over xGMI, and every shard builds its plan from
// the copy that arrived on ITS device.  RCCL is loaded with dlopen so that single-GPU users
// of the library do not need it.
#include <dlfcn.h>

#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>

#include "common.hpp"

namespace {

struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = true, quit = false;
    int rc = 0;
    std::string err;

    void start() {
        th = std::thread([this] {
            for (;;) {
                std::function<int()> j;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [this] { return has_job || quit; });
                    if (quit && !has_job) return;
                    j = std::move(job);
                    has_job = false;
                }
                int r = j();
                std::string e = r ? dtcwt_hip_last_error() : "";
                {
                    std::lock_guard<std::mutex> lk(mu);
                    rc = r; err = std::move(e); done = true;
                }
                cv.notify_all();
            }
        });
    }
    void post(std::function<int()> j) {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = std::move(j); has_job = true; done = false;
        }
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return done; });
        return rc;
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

struct Shard {
    int device = 0, start = 0, count = 0;
    dtcwt_hip_ctx *ctx = nullptr;
    dtcwt_hip_plan2d *plan = nullptr;      // NULL when count == 0
    Worker *w = nullptr;
};

// ---- RCCL through dlopen: just the four entry points the tap broadcast needs ------------
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, int /*datatype*/, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load() {
        if (h) return true;
        for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        Broadcast = (decltype(Broadcast))dlsym(h, "ncclBroadcast");
        GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
        return CommInitAll && CommDestroy && Broadcast && GroupStart && GroupEnd;
    }
};
constexpr int kNcclFloat64 = 8;        // ncclDouble (rccl.h: ncclFloat64 = 8)

}  // namespace

struct dtcwt_hip_mgpu {
    int batch, rows, cols, nlevels;
    std::vector<Shard> sh;
    std::vector<Worker *> workers;
    int taps_broadcast = 0;            // 1: the plans were built from taps that travelled through RCCL
    int on_shares = 1;                 // > 1: every shard's context runs on 1 / on_shares of its device (a lane)
};

namespace {

// run fn(d) on every shard's worker thread, wait for all, report the first failure
int on_all(dtcwt_hip_mgpu *m, const std::function<int(int)> &fn) {
    const int n = (int)m->sh.size();
    for (int d = 0; d < n; ++d) m->sh[d].w->post([&fn, d] { return fn(d); });
    int rc = 0;
    std::string err;
    for (int d = 0; d < n; ++d) {
        int r = m->sh[d].w->wait();
        if (r && !rc) { rc = r; err = m->sh[d].w->err; }
    }
    if (rc) return dtcwt_set_error(rc, "%s", err.c_str());
    return 0;
}

// Broadcast `flat` (n doubles, valid on shard 0) to every distinct device and read each copy back.
int broadcast_taps(dtcwt_hip_mgpu *m, const std::vector<double> &flat, std::vector<std::vector<double>> &per_shard) {
    static Rccl R;
    if (!R.load()) return dtcwt_set_error(-2, "RCCL (librccl.so) could not be loaded for the tap broadcast");
    // one communicator rank per DISTINCT device (a device may carry several shards in tests)
    std::vector<int> devs;
    std::vector<int> rank_of(m->sh.size());
    for (size_t d = 0; d < m->sh.size(); ++d) {
        size_t k = 0;
        while (k < devs.size() && devs[k] != m->sh[d].device) ++k;
        if (k == devs.size()) devs.push_back(m->sh[d].device);
        rank_of[d] = (int)k;
    }
    const int nr = (int)devs.size();
    std::vector<ncclComm_t> comms(nr);
    int caller_device = 0;                  // this function walks the devices: put the caller's back at the end
    (void)hipGetDevice(&caller_device);
    struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{caller_device};
    ncclResult_t r = R.CommInitAll(comms.data(), nr, devs.data());
    if (r != 0) return dtcwt_set_error(-2, "ncclCommInitAll failed: %s", R.GetErrorString ? R.GetErrorString(r) : "?");
    std::vector<double *> buf(nr, nullptr);
    std::vector<hipStream_t> st(nr);
    int rc = 0;
    const size_t n = flat.size();
    for (int k = 0; k < nr && !rc; ++k) {
        if (hipSetDevice(devs[k]) != hipSuccess || hipMalloc((void **)&buf[k], n * sizeof(double)) != hipSuccess ||
            hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking) != hipSuccess)
            rc = dtcwt_set_error(-2, "device set-up for the tap broadcast failed on device %d", devs[k]);
        else if (hipMemset(buf[k], 0, n * sizeof(double)) != hipSuccess)
            rc = dtcwt_set_error(-2, "hipMemset failed");
    }
    if (!rc) {
        (void)hipSetDevice(devs[0]);
        if (hipMemcpy(buf[0], flat.data(), n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
            rc = dtcwt_set_error(-2, "upload of the tap table failed");
    }
    if (!rc) {
        R.GroupStart();
        for (int k = 0; k < nr; ++k) {
            (void)hipSetDevice(devs[k]);
            ncclResult_t q = R.Broadcast(buf[k], buf[k], n, kNcclFloat64, 0, comms[k], st[k]);
            if (q != 0 && !rc) rc = dtcwt_set_error(-2, "ncclBroadcast failed: %s", R.GetErrorString ? R.GetErrorString(q) : "?");
        }
        ncclResult_t q = R.GroupEnd();
        if (q != 0 && !rc) rc = dtcwt_set_error(-2, "ncclGroupEnd failed: %s", R.GetErrorString ? R.GetErrorString(q) : "?");
    }
    std::vector<std::vector<double>> got(nr, std::vector<double>(n, 0.0));
    for (int k = 0; k < nr && !rc; ++k) {
        (void)hipSetDevice(devs[k]);
        if (hipStreamSynchronize(st[k]) != hipSuccess ||
            hipMemcpy(got[k].data(), buf[k], n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
            rc = dtcwt_set_error(-2, "read-back of the broadcast tap table failed on device %d", devs[k]);
    }
    for (int k = 0; k < nr; ++k) {
        (void)hipSetDevice(devs[k]);
        if (buf[k]) (void)hipFree(buf[k]);
        if (st[k]) (void)hipStreamDestroy(st[k]);
        R.CommDestroy(comms[k]);
    }
    if (rc) return rc;
    per_shard.resize(m->sh.size());
    for (size_t d = 0; d < m->sh.size(); ++d) per_shard[d] = got[rank_of[d]];
    return 0;
}

}  // namespace

extern "C" {

// Does a lane's shard run on a share of its device?  Measured under the bench protocol with four in flight
// (profiles/r04/ab_partition.txt): images of 2048^2 and more gain from contexts on quarters of the compute units
// (4096^2 -5 .. -7 %, 64 x 2048^2 -2 %); 64 x 1024^2 with five levels lost 4 % in round 4 and gains 2.5-3 % on QUARTERS with the
// round-5 kernels (0.604 against 0.623 ms per step, profiles/r05/batch_streams.txt), while on halves it still loses 5 %; two and
// four shares divide the XCDs evenly, three do not.
static bool lane_on_a_share(int flags, int nlanes, int rows, int cols) {
    if (nlanes < 2 || (flags & DTCWT_HIP_MGPU_NO_PARTITION)) return false;
    if (flags & DTCWT_HIP_MGPU_PARTITION) return nlanes <= 16;
    const int64_t px = (int64_t)rows * cols;
    return (nlanes == 4 && px >= (int64_t)1024 * 1024) || (nlanes == 2 && px >= (int64_t)2048 * 2048);
}

int dtcwt_hip_mgpu_create(int ndev, const int *devices, int batch, int rows, int cols, int nlevels,
                          const double *const *biort_host, const int *biort_len,
                          const double *const *qshift_host, const int *qshift_len, int flags,
                          dtcwt_hip_mgpu **out) {
    return dtcwt_hip_mgpu_create_lane(ndev, devices, batch, rows, cols, nlevels, biort_host, biort_len, qshift_host, qshift_len,
                                      flags, 0, 1, out);
}

int dtcwt_hip_mgpu_create_lane(int ndev, const int *devices, int batch, int rows, int cols, int nlevels,
                               const double *const *biort_host, const int *biort_len,
                               const double *const *qshift_host, const int *qshift_len, int flags,
                               int lane, int nlanes, dtcwt_hip_mgpu **out) {
    DT_REQUIRE(out && devices && biort_host && biort_len && qshift_host && qshift_len, "NULL argument");
    DT_REQUIRE(nlanes >= 1 && nlanes <= 16 && lane >= 0 && lane < nlanes, "lane %d of %d: need 1 <= nlanes <= 16, 0 <= lane < nlanes", lane, nlanes);
    DT_REQUIRE(ndev >= 1 && ndev <= 64, "bad device count %d", ndev);
    DT_REQUIRE(batch >= 1 && rows >= 1 && cols >= 1 && nlevels >= 1, "bad extents");
    // the tables are copied below, before any plan looks at them: lengths first
    for (int i = 0; i < 4; ++i)
        DT_REQUIRE(biort_host[i] && biort_len[i] >= 1 && biort_len[i] <= DTCWT_HIP_MAX_TAPS, "biort vector %d: NULL or length %d out of range", i, biort_len[i]);
    for (int i = 0; i < 8; ++i)
        DT_REQUIRE(qshift_host[i] && qshift_len[i] >= 1 && qshift_len[i] <= DTCWT_HIP_MAX_TAPS, "q-shift vector %d: NULL or length %d out of range", i, qshift_len[i]);
    int nvis = 0;
    DT_CHECK_HIP(hipGetDeviceCount(&nvis));
    for (int d = 0; d < ndev; ++d)
        DT_REQUIRE(devices[d] >= 0 && devices[d] < nvis, "device %d out of range (%d visible)", devices[d], nvis);
    dtcwt_hip_mgpu *m = new dtcwt_hip_mgpu();
    m->batch = batch; m->rows = rows; m->cols = cols; m->nlevels = nlevels;
    m->sh.resize(ndev);
    // contiguous split, sizes differ by at most one (dtcwt_amd/hip/sharding.py shard_range)
    const int base = batch / ndev, rem = batch % ndev;
    for (int d = 0; d < ndev; ++d) {
        Shard &s = m->sh[d];
        s.device = devices[d];
        s.start = d * base + (d < rem ? d : rem);
        s.count = base + (d < rem ? 1 : 0);
        s.w = new Worker();
        s.w->start();
        m->workers.push_back(s.w);
    }
    // tap tables per shard: either the caller's, or the copy that RCCL delivered to the shard's device
    std::vector<int> lens;
    std::vector<double> flat;
    for (int i = 0; i < 4; ++i) { lens.push_back(biort_len[i]); flat.insert(flat.end(), biort_host[i], biort_host[i] + biort_len[i]); }
    for (int i = 0; i < 8; ++i) { lens.push_back(qshift_len[i]); flat.insert(flat.end(), qshift_host[i], qshift_host[i] + qshift_len[i]); }
    std::vector<std::vector<double>> taps(ndev, flat);
    int rc = 0;
    if (flags & DTCWT_HIP_MGPU_BCAST_TAPS) {
        rc = broadcast_taps(m, flat, taps);
        if (!rc) m->taps_broadcast = 1;
    }
    if (!rc)
        rc = on_all(m, [&](int d) -> int {
            Shard &s = m->sh[d];
            // one of `nlanes` objects in flight on the same devices: the engine of the one-process-per-GPU path (bench.py)
            // -- each lane's context on its own share of the compute units where that measured faster, otherwise plain
            // contexts whose plans are told how many transforms share the device
            const bool share = lane_on_a_share(flags, nlanes, rows, cols);
            int r = share ? dtcwt_hip_ctx_create_partition(s.device, lane, nlanes, &s.ctx) : dtcwt_hip_ctx_create(s.device, nullptr, &s.ctx);
            if (r || s.count == 0) return r;
            const double *bp[4], *qp[8];
            const double *p = taps[d].data();
            for (int i = 0; i < 4; ++i) { bp[i] = p; p += lens[i]; }
            for (int i = 0; i < 8; ++i) { qp[i] = p; p += lens[4 + i]; }
            r = dtcwt_hip_plan2d_create(s.ctx, s.count, rows, cols, nlevels, bp, biort_len, qp, qshift_len, &s.plan);
            if (!r && nlanes > 1 && !share) r = dtcwt_hip_plan2d_set_concurrency(s.plan, nlanes);
            return r;
        });
    m->on_shares = lane_on_a_share(flags, nlanes, rows, cols) ? nlanes : 1;
    if (rc) {
        std::string keep = dtcwt_hip_last_error();
        dtcwt_hip_mgpu_destroy(m);
        return dtcwt_set_error(rc, "%s", keep.c_str());
    }
    *out = m;
    return 0;
}

int dtcwt_hip_mgpu_destroy(dtcwt_hip_mgpu *m) {
    if (!m) return 0;
    for (Shard &s : m->sh) {
        if (s.plan) dtcwt_hip_plan2d_destroy(s.plan);
        if (s.ctx) dtcwt_hip_ctx_destroy(s.ctx);
    }
    for (Worker *w : m->workers) { w->stop(); delete w; }
    delete m;
    return 0;
}

int dtcwt_hip_mgpu_ndev(const dtcwt_hip_mgpu *m) { return m ? (int)m->sh.size() : 0; }

int dtcwt_hip_mgpu_taps_broadcast(const dtcwt_hip_mgpu *m) { return m ? m->taps_broadcast : 0; }

int dtcwt_hip_mgpu_shares(const dtcwt_hip_mgpu *m) { return m ? m->on_shares : 0; }

int dtcwt_hip_mgpu_shard(const dtcwt_hip_mgpu *m, int d, int *device, int *start, int *count) {
    DT_REQUIRE(m && d >= 0 && d < (int)m->sh.size(), "bad shard index");
    if (device) *device = m->sh[d].device;
    if (start) *start = m->sh[d].start;
    if (count) *count = m->sh[d].count;
    return 0;
}

dtcwt_hip_ctx *dtcwt_hip_mgpu_ctx(dtcwt_hip_mgpu *m, int d) {
    return (m && d >= 0 && d < (int)m->sh.size()) ? m->sh[d].ctx : nullptr;
}

int dtcwt_hip_mgpu_shapes(const dtcwt_hip_mgpu *m, int *shapes) {
    DT_REQUIRE(m && shapes, "NULL argument");
    for (const Shard &s : m->sh)
        if (s.plan) return dtcwt_hip_plan2d_shapes(s.plan, shapes);
    return dtcwt_set_error(-1, "no shard holds an image");
}

int dtcwt_hip_mgpu_forward2d_scales(dtcwt_hip_mgpu *m, const float *const *X, float *const *Yl, void *const *Yh,
                                    float *const *Ys) {
    DT_REQUIRE(m && X && Yl && Yh, "NULL argument");
    const int nl = m->nlevels;
    return on_all(m, [&](int d) -> int {
        Shard &s = m->sh[d];
        if (!s.plan) return 0;
        return dtcwt_hip_plan2d_forward(s.plan, X[d], Yl[d], Yh + (size_t)d * nl, Ys ? Ys + (size_t)d * nl : nullptr);
    });
}

int dtcwt_hip_mgpu_forward2d(dtcwt_hip_mgpu *m, const float *const *X, float *const *Yl, void *const *Yh) {
    return dtcwt_hip_mgpu_forward2d_scales(m, X, Yl, Yh, nullptr);
}

int dtcwt_hip_mgpu_inverse2d(dtcwt_hip_mgpu *m, const float *const *Yl, const void *const *Yh,
                             const double *gain_mask_host, float *const *Z) {
    DT_REQUIRE(m && Yl && Yh && Z, "NULL argument");
    const int nl = m->nlevels;
    return on_all(m, [&](int d) -> int {
        Shard &s = m->sh[d];
        if (!s.plan) return 0;
        return dtcwt_hip_plan2d_inverse(s.plan, Yl[d], Yh + (size_t)d * nl, gain_mask_host, Z[d]);
    });
}

int dtcwt_hip_mgpu_sync(dtcwt_hip_mgpu *m) {
    DT_REQUIRE(m, "NULL argument");
    return on_all(m, [&](int d) -> int { return dtcwt_hip_sync(m->sh[d].ctx); });
}

// host [batch][elems_per_image * elem_bytes] <-> per-shard device buffers, every shard's copy issued from
// its own thread (the uploads / downloads of different devices overlap on the host links)
int dtcwt_hip_mgpu_scatter(dtcwt_hip_mgpu *m, const void *host, size_t bytes_per_image, void *const *dev) {
    DT_REQUIRE(m && host && dev, "NULL argument");
    return on_all(m, [&](int d) -> int {
        Shard &s = m->sh[d];
        if (!s.count) return 0;
        return dtcwt_hip_memcpy_h2d(s.ctx, dev[d], (const char *)host + (size_t)s.start * bytes_per_image,
                                    (size_t)s.count * bytes_per_image);
    });
}

int dtcwt_hip_mgpu_gather(dtcwt_hip_mgpu *m, const void *const *dev, size_t bytes_per_image, void *host) {
    DT_REQUIRE(m && host && dev, "NULL argument");
    return on_all(m, [&](int d) -> int {
        Shard &s = m->sh[d];
        if (!s.count) return 0;
        return dtcwt_hip_memcpy_d2h(s.ctx, (char *)host + (size_t)s.start * bytes_per_image, dev[d],
                                    (size_t)s.count * bytes_per_image);
    });
}

// page-locked host memory, enqueue only: completion at dtcwt_hip_mgpu_sync() (dtcwt_hip_sync waits for a context's
// stream and for its copy stream)
int dtcwt_hip_mgpu_scatter_async(dtcwt_hip_mgpu *m, const void *host, size_t bytes_per_image, void *const *dev) {
    DT_REQUIRE(m && host && dev, "NULL argument");
    return on_all(m, [&](int d) -> int {
        Shard &s = m->sh[d];
        if (!s.count) return 0;
        return dtcwt_hip_memcpy_h2d_async(s.ctx, dev[d], (const char *)host + (size_t)s.start * bytes_per_image,
                                          (size_t)s.count * bytes_per_image);
    });
}

int dtcwt_hip_mgpu_gather_async(dtcwt_hip_mgpu *m, const void *const *dev, size_t bytes_per_image, void *host) {
    DT_REQUIRE(m && host && dev, "NULL argument");
    return on_all(m, [&](int d) -> int {
        Shard &s = m->sh[d];
        if (!s.count) return 0;
        return dtcwt_hip_memcpy_d2h_overlapped(s.ctx, (char *)host + (size_t)s.start * bytes_per_image, dev[d],
                                               (size_t)s.count * bytes_per_image);
    });
}

}  // extern "C"
