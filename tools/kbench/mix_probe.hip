// What does HBM deliver for the read : write mixes of the level-1 kernels?  Trivially coalesced float4 streams:
//   r4w1: four input streams of 67 MB -> one output stream (k_inv1: 272 MB in, 69 MB out)
//   r1w4: one input stream -> four output streams        (k_fwd1: 73 MB in, 262 MB out)
//   r1w1: copy
// The launches rotate over NSET buffer sets (2.1 GB in all) so that nothing is served by the 256 MB Infinity Cache
// from the previous launch, as in bench.py.
// usage: mix_probe            (prints us and TB/s per mix; nt = non-temporal stores)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NR, int NW, bool NT>
__global__ void __launch_bounds__(256) k_mix(const v4 *__restrict__ in, v4 *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    v4 a = in[i];
#pragma unroll
    for (int r = 1; r < NR; ++r) a += in[i + r * n];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        v4 b = a * (float)(w + 1);
        if (NT) __builtin_nontemporal_store(b, out + i + w * n); else out[i + w * n] = b;
    }
}

constexpr int NSET = 4;
template <int NR, int NW, bool NT>
int run(const char *name, v4 *in0, v4 *out0, size_t n) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((n + 255) / 256);
    for (int i = 0; i < 5; ++i) k_mix<NR, NW, NT><<<grid, 256>>>(in0, out0, n);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) {
            const size_t o = (size_t)(i % NSET) * 4 * n;
            k_mix<NR, NW, NT><<<grid, 256>>>(in0 + o, out0 + o, n);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / 20 < best) best = ms / 20;
    }
    const double bytes = (double)(NR + NW) * n * 16;
    printf("%-10s %s  %7.1f us  %6.1f MB  %.2f TB/s\n", name, NT ? "nt" : "  ", best * 1e3, bytes / 1e6, bytes / (best * 1e-3) / 1e12);
    return 0;
}

int main() {
    const size_t n = (size_t)4096 * 4096 / 4;       // float4 elements of one 67 MB plane
    v4 *in, *out;
    CK(hipMalloc(&in, NSET * 4 * n * 16)); CK(hipMalloc(&out, NSET * 4 * n * 16));
    CK(hipMemset(in, 0x3c, NSET * 4 * n * 16));
    run<1, 1, false>("r1w1", in, out, n); run<1, 1, true>("r1w1", in, out, n);
    run<4, 1, false>("r4w1", in, out, n); run<4, 1, true>("r4w1", in, out, n);
    run<1, 4, false>("r1w4", in, out, n); run<1, 4, true>("r1w4", in, out, n);
    run<2, 2, true>("r2w2", in, out, n);
    run<0 + 1, 3, true>("r1w3", in, out, n);
    return 0;
}
