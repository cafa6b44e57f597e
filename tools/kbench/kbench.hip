// Kernel micro-benchmark harness (development tool, not part of the product):
// runs variants of the level-1 tile programs on a 4096x4096 image, checks them against the
// first variant and prints time / effective bandwidth.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dtcwt_amd/csrc -I include tools/kbench/kbench.hip -o /tmp/kbench && /tmp/kbench
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "fused2d_tiles.hpp"
#include "fused2d_tiles_v2.hpp"

using namespace dt2d;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- reference copy kernels: the practical ceiling for this read/write mix -------------
__global__ void __launch_bounds__(256) k_copy_1r4w(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        f4 v = in[i];
        out[4 * i] = v; out[4 * i + 1] = v; out[4 * i + 2] = v; out[4 * i + 3] = v;
    }
}
__global__ void __launch_bounds__(256) k_copy_4r1w(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        f4 a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
        out[i] = f4{a.x + b.x, a.y + c.y, a.z + d.z, a.w + b.w + c.w + d.w};
    }
}

// ---- kernel wrappers ------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(DT_NT) k_fwd1_v0(Fwd1Params p) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_FLOATS];
    const int ntile = p.tilesR * p.tilesC * p.B;
    int t = xcd_tile(blockIdx.x, ntile);
    if (t >= ntile) return;
    int tc = t % p.tilesC, tr = (t / p.tilesC) % p.tilesR, b = t / (p.tilesC * p.tilesR);
    float *sx = smem, *sLo = smem + C::SX, *sHi = sLo + C::SL;
    int r0 = tr * C::TR, c0 = tc * C::TC;
    fwd1_load<C>(p, sx, threadIdx.x, b, r0, c0);
    __syncthreads();
    fwd1_cols<C>(p, sx, sLo, sHi, threadIdx.x);
    __syncthreads();
    fwd1_rows<C>(p, sLo, sHi, threadIdx.x, b, r0, c0);
}

template <class C, bool XCD, int MINW>
__global__ void __launch_bounds__(DT_NT, MINW) k_fwd1_d(Fwd1Params p) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_FLOATS];
    const int ntile = p.tilesR * p.tilesC * p.B;
    int t = XCD ? xcd_tile(blockIdx.x, ntile) : (int)blockIdx.x;
    if (t >= ntile) return;
    int tc = t % p.tilesC, tr = (t / p.tilesC) % p.tilesR, b = t / (p.tilesC * p.tilesR);
    float *sLo = smem, *sHi = sLo + C::SL;
    int r0 = tr * C::TR, c0 = tc * C::TC;
    fwd1d_cols<C>(p, sLo, sHi, threadIdx.x, b, r0, c0);
    __syncthreads();
    fwd1d_rows<C, false>(p, sLo, sHi, nullptr, threadIdx.x, b, r0, c0);
}

struct Variant {
    std::string name;
    std::function<void(Fwd1Params &)> launch;
};

template <class C>
void launch_v0(Fwd1Params &p) {
    p.tilesR = cdiv(p.LR, C::TR); p.tilesC = cdiv(p.LC, C::TC);
    int nt = p.tilesR * p.tilesC * p.B;
    k_fwd1_v0<C><<<cdiv(nt, 8) * 8, DT_NT>>>(p);
}
template <class C, bool XCD, int MINW>
void launch_d(Fwd1Params &p) {
    p.tilesR = cdiv(p.LR, C::TR); p.tilesC = cdiv(p.LC, C::TC);
    int nt = p.tilesR * p.tilesC * p.B;
    k_fwd1_d<C, XCD, MINW><<<cdiv(nt, 8) * 8, DT_NT>>>(p);
}

int main(int argc, char **argv) {
    const int R = 4096, Cc = 4096, B = 1;
    const size_t npx = (size_t)R * Cc;
    std::vector<float> hX(npx);
    srand(1);
    for (auto &v : hX) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *dX, *dLo, *dYh;
    CK(hipMalloc(&dX, npx * 4));
    CK(hipMalloc(&dLo, npx * 4));
    CK(hipMalloc(&dYh, npx * 12));
    CK(hipMemcpy(dX, hX.data(), npx * 4, hipMemcpyHostToDevice));

    Fwd1Params p{};
    p.X = dX; p.LoLo = dLo; p.Yh = dYh; p.B = B; p.inR = R; p.inC = Cc; p.LR = R; p.LC = Cc;
    const double h0[5] = {-0.05, 0.25, 0.6, 0.25, -0.05};
    const double h1[7] = {-0.0107142857, 0.0535714286, 0.2607142857, -0.6071428571, 0.2607142857, 0.0535714286, -0.0107142857};
    for (int k = 0; k < 5; ++k) p.h0[k] = (float)h0[k];
    for (int k = 0; k < 7; ++k) p.h1[k] = (float)h1[k];

    std::vector<Variant> vs;
    vs.push_back({"v0 staged 32x64", launch_v0<Fwd1Cfg<32, 64, 5, 7>>});
    vs.push_back({"v0 staged 64x64", launch_v0<Fwd1Cfg<64, 64, 5, 7>>});
    vs.push_back({"d 32x64 rs8 xcd", launch_d<Fwd1DCfg<32, 64, 8, 5, 7>, true, 1>});
    vs.push_back({"d 32x56 rs8 xcd", launch_d<Fwd1DCfg<32, 56, 8, 5, 7>, true, 1>});
    vs.push_back({"d 32x120 rs8 xcd", launch_d<Fwd1DCfg<32, 120, 8, 5, 7>, true, 1>});
    vs.push_back({"d 32x120 rs8 lin", launch_d<Fwd1DCfg<32, 120, 8, 5, 7>, false, 1>});
    vs.push_back({"d 32x120 rs16 xcd", launch_d<Fwd1DCfg<32, 120, 16, 5, 7>, true, 1>});
    vs.push_back({"d 16x120 rs8 xcd", launch_d<Fwd1DCfg<16, 120, 8, 5, 7>, true, 1>});
    vs.push_back({"d 16x120 rs4 xcd", launch_d<Fwd1DCfg<16, 120, 4, 5, 7>, true, 1>});
    vs.push_back({"d 16x248 rs8 xcd", launch_d<Fwd1DCfg<16, 248, 8, 5, 7>, true, 1>});
    vs.push_back({"d 64x56 rs8 xcd", launch_d<Fwd1DCfg<64, 56, 8, 5, 7>, true, 1>});
    vs.push_back({"d 32x56 rs8 xcd w2", launch_d<Fwd1DCfg<32, 56, 8, 5, 7>, true, 2>});

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ref_lo, ref_yh, lo(npx), yh(npx * 3);
    const int reps = 20;
    // copy ceilings
    {
        f4 *a = (f4 *)dX, *o = (f4 *)dYh;   // 64 MiB in, 256 MiB out needs npx*16 bytes: use Yh(192MiB)+Lo
        size_t n4 = npx / 4 * 3 / 4;        // keep inside dYh: out elements = 4*n4 f4 = n4*64 B <= npx*12
        for (int w = 0; w < 3; ++w) k_copy_1r4w<<<2048, 256>>>(a, o, n4);
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) k_copy_1r4w<<<2048, 256>>>(a, o, n4);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-22s %8.1f us  %7.1f GB/s (1 read : 4 write copy, %zu MB)\n", "copy_1r4w", ms * 1e3, n4 * 80.0 / ms / 1e6, n4 * 80 >> 20);
        for (int w = 0; w < 3; ++w) k_copy_4r1w<<<2048, 256>>>(o, a, n4);
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) k_copy_4r1w<<<2048, 256>>>(o, a, n4);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-22s %8.1f us  %7.1f GB/s (4 read : 1 write copy)\n", "copy_4r1w", ms * 1e3, n4 * 80.0 / ms / 1e6);
        CK(hipMemcpy(dX, hX.data(), npx * 4, hipMemcpyHostToDevice));
    }
    for (size_t i = 0; i < vs.size(); ++i) {
        CK(hipMemset(dLo, 0xff, npx * 4)); CK(hipMemset(dYh, 0xff, npx * 12));
        for (int w = 0; w < 3; ++w) vs[i].launch(p);
        CK(hipGetLastError());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) vs[i].launch(p);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        CK(hipMemcpy(lo.data(), dLo, npx * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(yh.data(), dYh, npx * 12, hipMemcpyDeviceToHost));
        double dmax = 0;
        if (i == 0) { ref_lo = lo; ref_yh = yh; }
        else {
            for (size_t k = 0; k < npx; ++k) { double d = fabs((double)lo[k] - ref_lo[k]); if (!(d <= dmax)) dmax = d; }
            for (size_t k = 0; k < npx * 3; ++k) { double d = fabs((double)yh[k] - ref_yh[k]); if (!(d <= dmax)) dmax = d; }
        }
        printf("%-22s %8.1f us  %7.1f GB/s algorithmic  maxdiff %.2e\n", vs[i].name.c_str(), ms * 1e3,
               npx * 20.0 / ms / 1e6, dmax);
    }
    return 0;
}
