// Fused float32 3-D DT-CWT level kernels: __global__ wrappers of the marching tile
// programs in fused3d_tiles.hpp behind dtcwt_hip_fwd3_level1 (include/dtcwt_hip.h).
//
// Replaces one `_level1_xfm` of dtcwt/numpy/transform3d.py:208-289 (three axis passes
// over the whole volume in Python slice loops plus seven cube2c packings) by a single
// launch that reads the volume once and writes the lowpass volume and the 28-subband
// highpass records once.
#include "common.hpp"
#include "fused3d_tiles.hpp"

using namespace dt3d;

namespace {

template <class C>
__global__ void __launch_bounds__(DT_NT) k_fwd3_l1(Fwd3L1Params p) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_FLOATS];
    float *S0 = smem, *S1 = smem + C::S0F;
    const int bid = blockIdx.x;
    const int tk = bid % p.tilesK, tj = (bid / p.tilesK) % p.tilesJ, ch = bid / (p.tilesK * p.tilesJ);
    const int j0 = tj * C::TJ, k0 = tk * C::TK, i0 = ch * p.chunk;
    const int iend = min(i0 + p.chunk, p.n0);
    const int tid = threadIdx.x;
    Fwd3L1State<C> st;
    f3l1_init<C>(p, st, tid, j0, k0);
    f3l1_prologue<C>(p, st, i0);
    for (int i = i0; i < iend; ++i) {
        f3l1_axis0<C>(p, st, S0, i, i + 1 < iend);
        __syncthreads();
        f3l1_axis2<C>(p, S0, S1, tid);
        __syncthreads();
        f3l1_axis1_pack<C>(p, st, S1, tid, i, j0, k0);
    }
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

template <class C>
int launch_fwd3_l1(Fwd3L1Params &p, int cus, hipStream_t s) {
    p.tilesJ = cdiv(p.n1, C::TJ); p.tilesK = cdiv(p.n2, C::TK);
    // slices per workgroup: long marches amortise the 2H-slice warm-up, but the launch
    // still has to put several workgroups on every CU
    int chunk = 64;
    while (chunk > 8 && (int64_t)p.tilesJ * p.tilesK * cdiv(p.n0, chunk) < 4 * (int64_t)cus) chunk /= 2;
    if (const char *e = getenv("DTCWT_HIP_CHUNK3D")) {
        int v = atoi(e);
        if (v >= 2 && v % 2 == 0) chunk = v;
    }
    p.chunk = chunk;
    p.chunks = cdiv(p.n0, chunk);
    k_fwd3_l1<C><<<(unsigned)(p.tilesJ * p.tilesK * p.chunks), DT_NT, 0, s>>>(p);
    return 0;
}

void put_taps(float *dst, const double *src, int m) {
    for (int k = 0; k < DT_MAXT; ++k) dst[k] = k < m ? (float)src[k] : 0.f;
}

}  // namespace

#define DT_FWD3_L1_TABLE(X) X(5, 7) X(9, 7) X(5, 3) X(7, 5) X(7, 9) X(3, 5)

extern "C" int dtcwt_hip_fwd3_level1(dtcwt_hip_ctx *ctx, const float *X, int64_t n0, int64_t n1, int64_t n2,
                                     const double *h0o, int m0, const double *h1o, int m1, float *LLL,
                                     float *Yh) {
    DT_REQUIRE(ctx && X && h0o && h1o && LLL && Yh, "NULL argument");
    DT_REQUIRE(n0 > 0 && n1 > 0 && n2 > 0 && n0 % 2 == 0 && n1 % 2 == 0 && n2 % 2 == 0,
               "level-1 volume extents must be even (transform3d.py:214-217)");
    DT_REQUIRE(m0 > 0 && m1 > 0 && m0 <= DT_MAXT && m1 <= DT_MAXT, "bad tap counts");
    if (n0 < 8 || n1 < 8 || n2 < 8 || n0 >= (1 << 30) || n1 * n2 >= ((int64_t)1 << 31))
        return dtcwt_set_error(-3, "fused 3-D level 1 needs extents >= 8 and slices < 2^31 samples");
    Fwd3L1Params p{};
    p.X = X; p.LLL = LLL; p.Yh = Yh;
    p.n0 = (int)n0; p.n1 = (int)n1; p.n2 = (int)n2;
    put_taps(p.h0, h0o, m0); put_taps(p.h1, h1o, m1);
    DT_CHECK_HIP(hipSetDevice(ctx->device));
#define X_(A, B)                                                                           \
    if (m0 == A && m1 == B) {                                                              \
        launch_fwd3_l1<Fwd3L1Cfg<A, B>>(p, ctx->cus, ctx->stream);                         \
        DT_CHECK_HIP(hipGetLastError());                                                   \
        return 0;                                                                          \
    }
    DT_FWD3_L1_TABLE(X_)
#undef X_
    return dtcwt_set_error(-3, "no fused 3-D level-1 kernel for %d/%d-tap biort filters", m0, m1);
}
