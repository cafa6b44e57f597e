// does a raw buffer store with voffset >= num_records get dropped on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
__global__ void k(float *base, unsigned nrec, unsigned voff_bad, unsigned soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)nrec, 0x00020000);
    unsigned vo = threadIdx.x < 32 ? threadIdx.x * 16u : voff_bad + threadIdx.x * 16u;
    __builtin_amdgcn_raw_buffer_store_b128(v4{1.f, 2.f, 3.f, 4.f}, r, vo, soff, 0);
}
int main(int argc, char **argv) {
    int only = argc > 1 ? atoi(argv[1]) : -1; int ci = -1;
    float *d; size_t n = 1 << 20; hipMalloc(&d, n * 4);
    std::vector<float> h(n);
    struct { unsigned nrec, bad, soff; } cases[] = {{0x80000000u, 0x80000000u, 0}, {0x7fffffffu, 0x80000000u, 0}, {0x40000000u, 0x40000000u, 0}, {0x40000000u, 0x80000000u, 0}, {0x40000000u, 0xfffff000u, 0}, {0x80000000u, 0x80000000u, 4096}, {65536u, 65536u, 0}, {65536u, 65536u, 4096}, {65536u, 0x80000000u, 4096}, {0xffffffffu, 0x80000000u, 0}};
    for (auto c : cases) {
        if (++ci != only) continue;
        hipMemset(d, 0, n * 4);
        k<<<1, 64>>>(d, c.nrec, c.bad, c.soff);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
        size_t nz = 0, first = 0, last = 0;
        for (size_t i = 0; i < n; ++i) if (h[i] != 0.f) { if (!nz) first = i; last = i; ++nz; }
        printf("num_records %08x bad voffset %08x soffset %u: %zu nonzero floats, first %zu last %zu (expect 128 = lanes 0-31 only)\n", c.nrec, c.bad, c.soff, nz, first, last);
    }
    return 0;
}
