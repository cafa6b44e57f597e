// Runtime part of the C ABI: contexts, device buffers, events (include/dtcwt_hip.h).
#include <cstdlib>
#include <cstring>

#include "common.hpp"

static thread_local char g_err[512] = "";

int dtcwt_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// integer / bool samples -> floating point on the device: the reference converts every non-float
// input to float64 on the host first (dtcwt/utils.py:98-105 asfarray); an 8-bit image is an eighth of
// the bytes over the host link when it is widened here instead.
template <typename S, typename D>
__global__ void __launch_bounds__(256) k_to_float(const S *__restrict__ src, D *__restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (D)src[i];
}

template <typename D>
static int to_float_dispatch(dtcwt_hip_ctx *c, int src_kind, const void *src, D *dst, int64_t n) {
    const unsigned blocks = (unsigned)((n + 255) / 256);
#define DT_TF(S_) k_to_float<S_, D><<<blocks, 256, 0, c->stream>>>((const S_ *)src, dst, n)
    switch (src_kind) {
        case 0: DT_TF(uint8_t); break;
        case 1: DT_TF(int8_t); break;
        case 2: DT_TF(uint16_t); break;
        case 3: DT_TF(int16_t); break;
        case 4: DT_TF(uint32_t); break;
        case 5: DT_TF(int32_t); break;
        case 6: DT_TF(uint64_t); break;
        case 7: DT_TF(int64_t); break;
        case 8: DT_TF(bool); break;
        case 9: DT_TF(float); break;        // float32 <-> float64: degenerate float32 shapes are computed in float64
        case 10: DT_TF(double); break;
        default: return dtcwt_set_error(-1, "bad integer kind %d", src_kind);
    }
#undef DT_TF
    DT_CHECK_HIP(hipGetLastError());
    return 0;
}

// pool_mu held by the caller
static int trim_locked(dtcwt_hip_ctx *c) {
    DT_CHECK_HIP(hipSetDevice(c->device));
    // A capturing stream must not be synchronised (the call fails and invalidates the capture: estimatereg
    // allocates its scratch while capturing).  Nothing captured has run yet, and hipFree waits for the work
    // that was submitted before the capture began, so the pooled buffers are idle when they are released.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c->stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (cap == hipStreamCaptureStatusNone) DT_CHECK_HIP(hipStreamSynchronize(c->stream));
    for (auto &kv : c->pool)
        for (void *b : kv.second) (void)hipFree(b);
    c->pool.clear();
    c->pooled_bytes = 0;
    return 0;
}

extern "C" {

int dtcwt_hip_abi_version(void) { return DTCWT_HIP_ABI_VERSION; }

const char *dtcwt_hip_last_error(void) { return g_err; }

int dtcwt_hip_device_count(int *count) {
    DT_REQUIRE(count, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return dtcwt_set_error(-2, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    *count = n;
    return 0;
}

int dtcwt_hip_device_info(int device, char *name, int *cus, size_t *mem_bytes) {
    hipDeviceProp_t p;
    DT_CHECK_HIP(hipGetDeviceProperties(&p, device));
    if (name) {
        snprintf(name, 256, "%s (%s)", p.name, p.gcnArchName);
    }
    if (cus) *cus = p.multiProcessorCount;
    if (mem_bytes) *mem_bytes = p.totalGlobalMem;
    return 0;
}

static int ctx_create(int device, void *stream, int part, int nparts, dtcwt_hip_ctx **out) {
    DT_REQUIRE(out, "ctx out pointer is NULL");
    int n = 0;
    DT_CHECK_HIP(hipGetDeviceCount(&n));
    DT_REQUIRE(device >= 0 && device < n, "device %d out of range (%d devices)", device, n);
    DT_REQUIRE(nparts >= 1 && nparts <= 16 && part >= 0 && part < nparts, "partition %d of %d: need 1 <= nparts <= 16, 0 <= part < nparts", part, nparts);
    DT_CHECK_HIP(hipSetDevice(device));
    hipDeviceProp_t p;
    const bool have_props = hipGetDeviceProperties(&p, device) == hipSuccess;
    const int cus = have_props ? p.multiProcessorCount : 256;
    dtcwt_hip_ctx *c = new dtcwt_hip_ctx();
    c->device = device;
    c->owns_stream = (stream == nullptr);
    c->cus = cus;
    if (stream) {
        c->stream = reinterpret_cast<hipStream_t>(stream);
    } else if (nparts > 1) {
        // share `part`: the mask bits part * per .. (part + 1) * per - 1 (contiguous ranges measured best, and far better
        // than whole XCDs: tools/ab_cu_mask.py)
        const int per = cus / nparts;
        std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
        for (int b = part * per; b < (part + 1) * per; ++b) mask[(size_t)b / 32] |= 1u << (b % 32);
        hipError_t e = per >= 1 ? hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)mask.size(), mask.data()) : hipErrorInvalidValue;
        if (e != hipSuccess) {
            delete c;
            return dtcwt_set_error(-2, "hipExtStreamCreateWithCUMask (share %d of %d, %d CUs) failed: %s", part, nparts, per, hipGetErrorString(e));
        }
        c->cus = per;
        c->nparts = nparts;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            return dtcwt_set_error(-2, "hipStreamCreate failed: %s", hipGetErrorString(e));
        }
    }
    // cache of freed buffers: up to a quarter of the HBM (288 GB on MI355X) -- pyramids of large
    // volumes are several GB per buffer, and every hipFree / hipMalloc is a device-wide sync
    if (have_props && p.totalGlobalMem / 4 > c->pool_limit) c->pool_limit = p.totalGlobalMem / 4;
    if (const char *e = getenv("DTCWT_HIP_POOL_MB")) c->pool_limit = (size_t)atoll(e) << 20;
    *out = c;
    return 0;
}

int dtcwt_hip_ctx_create(int device, void *stream, dtcwt_hip_ctx **out) { return ctx_create(device, stream, 0, 1, out); }

int dtcwt_hip_ctx_create_partition(int device, int part, int nparts, dtcwt_hip_ctx **out) {
    return ctx_create(device, nullptr, part, nparts, out);
}

int dtcwt_hip_ctx_destroy(dtcwt_hip_ctx *c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto &fn : c->on_destroy) fn();
    c->on_destroy.clear();
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->copy_event) (void)hipEventDestroy(c->copy_event);
    for (auto &kv : c->pool)
        for (void *b : kv.second) (void)hipFree(b);
    c->pool.clear();
    if (c->owns_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int dtcwt_hip_sync(dtcwt_hip_ctx *c) {
    DT_REQUIRE(c, "ctx is NULL");
    DT_CHECK_HIP(hipStreamSynchronize(c->stream));
    // "the context is idle" includes the downloads that overlap its kernels (dtcwt_hip_memcpy_d2h_overlapped)
    if (c->copy_stream) DT_CHECK_HIP(hipStreamSynchronize(c->copy_stream));
    return 0;
}

int dtcwt_hip_device_sync(dtcwt_hip_ctx *c) {
    DT_REQUIRE(c, "ctx is NULL");
    DT_CHECK_HIP(hipSetDevice(c->device));
    DT_CHECK_HIP(hipDeviceSynchronize());
    return 0;
}

void *dtcwt_hip_ctx_stream(dtcwt_hip_ctx *c) { return c ? (void *)c->stream : nullptr; }

int dtcwt_hip_malloc(dtcwt_hip_ctx *c, size_t bytes, void **dptr) {
    DT_REQUIRE(c && dptr, "NULL argument");
    *dptr = nullptr;
    if (bytes == 0) bytes = 16;
    bytes = (bytes + 255) & ~(size_t)255;
    std::lock_guard<std::mutex> lk(c->pool_mu);
    auto it = c->pool.find(bytes);
    if (it != c->pool.end() && !it->second.empty()) {
        *dptr = it->second.back();
        it->second.pop_back();
        c->pooled_bytes -= bytes;
        c->live[*dptr] = bytes;
        return 0;
    }
    DT_CHECK_HIP(hipSetDevice(c->device));
    hipError_t e = hipMalloc(dptr, bytes);
    if (e != hipSuccess && c->pooled_bytes) {        // out of memory: drop the cache and retry
        (void)hipGetLastError();
        trim_locked(c);
        e = hipMalloc(dptr, bytes);
    }
    if (e != hipSuccess)
        return dtcwt_set_error(-2, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    c->live[*dptr] = bytes;
    return 0;
}

int dtcwt_hip_free(dtcwt_hip_ctx *c, void *dptr) {
    DT_REQUIRE(c, "ctx is NULL");
    if (!dptr) return 0;
    std::lock_guard<std::mutex> lk(c->pool_mu);
    auto it = c->live.find(dptr);
    if (it == c->live.end()) return dtcwt_set_error(-1, "free of a pointer this context did not allocate");
    size_t bytes = it->second;
    c->live.erase(it);
    if (c->pooled_bytes + bytes <= c->pool_limit) {
        c->pool[bytes].push_back(dptr);
        c->pooled_bytes += bytes;
        return 0;
    }
    DT_CHECK_HIP(hipSetDevice(c->device));
    DT_CHECK_HIP(hipStreamSynchronize(c->stream));
    DT_CHECK_HIP(hipFree(dptr));
    return 0;
}

int dtcwt_hip_trim(dtcwt_hip_ctx *c) {
    DT_REQUIRE(c, "ctx is NULL");
    std::lock_guard<std::mutex> lk(c->pool_mu);
    return trim_locked(c);
}

int dtcwt_hip_memcpy_h2d(dtcwt_hip_ctx *c, void *dst, const void *src, size_t bytes) {
    DT_REQUIRE(c, "ctx is NULL");
    if (!bytes) return 0;
    DT_CHECK_HIP(hipSetDevice(c->device));
    DT_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    // pageable host memory: make the call safe to follow by a host-side free/reuse
    DT_CHECK_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int dtcwt_hip_memcpy_d2h(dtcwt_hip_ctx *c, void *dst, const void *src, size_t bytes) {
    DT_REQUIRE(c, "ctx is NULL");
    if (!bytes) return 0;
    DT_CHECK_HIP(hipSetDevice(c->device));
    // a blocking download is ordered after the overlapped ones as well: its destination may be a pooled host buffer
    // that one of them was still writing when its owner let go of it
    if (c->copy_stream) DT_CHECK_HIP(hipStreamSynchronize(c->copy_stream));
    DT_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    DT_CHECK_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

// ---- page-locked host memory and asynchronous copies -------------------------------------------
// The reference's OpenCL backend keeps device buffers and copies lazily (dtcwt/opencl/transform2d.py:30-84); the
// literal drop-in call Transform2d.forward(numpy array) -> numpy arrays moves 64 MiB up and 320 MiB down for a
// 4096^2 image, so the host link, not the kernels, is what it costs.  Page-locked buffers let those copies run
// as DMA at the link rate without a staging pass, and asynchronously.
int dtcwt_hip_host_alloc(size_t bytes, void **hptr) {
    DT_REQUIRE(hptr, "NULL argument");
    *hptr = nullptr;
    if (!bytes) bytes = 16;
    hipError_t e = hipHostMalloc(hptr, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return dtcwt_set_error(-2, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    return 0;
}

int dtcwt_hip_host_free(void *hptr) {
    if (!hptr) return 0;
    DT_CHECK_HIP(hipHostFree(hptr));
    return 0;
}

// enqueue only: the host buffer must stay untouched until dtcwt_hip_sync() (page-locked memory: a DMA transfer;
// pageable memory: the runtime stages it and the call may block)
int dtcwt_hip_memcpy_h2d_async(dtcwt_hip_ctx *c, void *dst, const void *src, size_t bytes) {
    DT_REQUIRE(c, "ctx is NULL");
    if (!bytes) return 0;
    DT_CHECK_HIP(hipSetDevice(c->device));
    DT_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    return 0;
}

// Download that does not hold up the kernels enqueued after it: ordered after everything issued on the
// context's stream so far, executed on the context's copy stream.  dtcwt_hip_copy_sync() waits for all of them.
int dtcwt_hip_memcpy_d2h_overlapped(dtcwt_hip_ctx *c, void *dst, const void *src, size_t bytes) {
    DT_REQUIRE(c, "ctx is NULL");
    if (!bytes) return 0;
    DT_CHECK_HIP(hipSetDevice(c->device));
    {
        std::lock_guard<std::mutex> lk(c->pool_mu);         // two threads of one context: one stream, not two
        if (!c->copy_stream) {
            hipStream_t st = nullptr;
            DT_CHECK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            DT_CHECK_HIP(hipEventCreateWithFlags(&c->copy_event, hipEventDisableTiming));
            c->copy_stream = st;
        }
    }
    DT_CHECK_HIP(hipEventRecord(c->copy_event, c->stream));
    DT_CHECK_HIP(hipStreamWaitEvent(c->copy_stream, c->copy_event, 0));
    DT_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->copy_stream));
    return 0;
}

int dtcwt_hip_copy_sync(dtcwt_hip_ctx *c) {
    DT_REQUIRE(c, "ctx is NULL");
    if (c->copy_stream) DT_CHECK_HIP(hipStreamSynchronize(c->copy_stream));
    return 0;
}

int dtcwt_hip_memcpy_d2d(dtcwt_hip_ctx *c, void *dst, const void *src, size_t bytes) {
    DT_REQUIRE(c, "ctx is NULL");
    if (!bytes) return 0;
    DT_CHECK_HIP(hipSetDevice(c->device));
    DT_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int dtcwt_hip_memset(dtcwt_hip_ctx *c, void *dst, int value, size_t bytes) {
    DT_REQUIRE(c, "ctx is NULL");
    if (!bytes) return 0;
    DT_CHECK_HIP(hipSetDevice(c->device));
    DT_CHECK_HIP(hipMemsetAsync(dst, value, bytes, c->stream));
    return 0;
}

int dtcwt_hip_event_create(dtcwt_hip_ctx *c, dtcwt_hip_event **ev) {
    DT_REQUIRE(c && ev, "NULL argument");
    DT_CHECK_HIP(hipSetDevice(c->device));
    dtcwt_hip_event *e = new dtcwt_hip_event();
    hipError_t r = hipEventCreate(&e->ev);
    if (r != hipSuccess) {
        delete e;
        return dtcwt_set_error(-2, "hipEventCreate failed: %s", hipGetErrorString(r));
    }
    *ev = e;
    return 0;
}

int dtcwt_hip_event_record(dtcwt_hip_ctx *c, dtcwt_hip_event *ev) {
    DT_REQUIRE(c && ev, "NULL argument");
    DT_CHECK_HIP(hipEventRecord(ev->ev, c->stream));
    return 0;
}

int dtcwt_hip_event_elapsed_ms(dtcwt_hip_event *a, dtcwt_hip_event *b, float *ms) {
    DT_REQUIRE(a && b && ms, "NULL argument");
    DT_CHECK_HIP(hipEventSynchronize(b->ev));
    DT_CHECK_HIP(hipEventElapsedTime(ms, a->ev, b->ev));
    return 0;
}

int dtcwt_hip_event_destroy(dtcwt_hip_event *ev) {
    if (!ev) return 0;
    (void)hipEventDestroy(ev->ev);
    delete ev;
    return 0;
}

int dtcwt_hip_to_float(dtcwt_hip_ctx *c, int src_kind, const void *src, int dst_dtype, void *dst, int64_t count) {
    DT_REQUIRE(c && src && dst, "NULL argument");
    DT_REQUIRE(count >= 0 && count < ((int64_t)1 << 39), "bad count");
    if (!count) return 0;
    DT_CHECK_HIP(hipSetDevice(c->device));
    if (dst_dtype == DTCWT_HIP_F32) return to_float_dispatch<float>(c, src_kind, src, (float *)dst, count);
    if (dst_dtype == DTCWT_HIP_F64) return to_float_dispatch<double>(c, src_kind, src, (double *)dst, count);
    return dtcwt_set_error(-1, "bad dtype %d", dst_dtype);
}

}  // extern "C"
