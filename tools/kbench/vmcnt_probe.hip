// Do vector-memory operations that the buffer range check drops entirely (every lane out of range) still retire IN
// ORDER with respect to older loads?  s_waitcnt vmcnt(N) is a counter, not a queue: the march kernels (march2d.hpp)
// issue clipped stores for rows outside their band so that every path through a step issues the same number of
// memory operations; if a clipped store retired at once, vmcnt(N) would let an older, still outstanding load through.
//
// Each wavefront: sentinel -> v; one cold global load into v (a fresh 4 KiB page per wavefront of a 1 GiB buffer); K
// clipped buffer stores; s_waitcnt vmcnt(K); copy v.  A copy that still holds the sentinel = out-of-order retirement.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>   // 0: clipped stores (num_records = 0), 1: real stores to a scratch line, 2: no wait at all (the test must FAIL here), 3: clipped LOADS
__global__ void probe(const unsigned *big, unsigned *scratch, unsigned *out, size_t stride_words) {
    const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const unsigned *src = big + w * stride_words + (threadIdx.x & 63);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(scratch, 0, MODE == 1 ? 4096 : 0, 0x00020000);
    unsigned v = 0xdeadbeefu, copy, zero = 0, voff = (threadIdx.x & 63) * 4;
    if (MODE == 3) {
        unsigned d0, d1, d2, d3;
        asm volatile("global_load_dword %0, %6, off\n\t"
                     "buffer_load_dword %2, %7, %8, 0 offen\n\tbuffer_load_dword %3, %7, %8, 0 offen offset:256\n\t"
                     "buffer_load_dword %4, %7, %8, 0 offen offset:512\n\tbuffer_load_dword %5, %7, %8, 0 offen offset:768\n\t"
                     "buffer_load_dword %2, %7, %8, 0 offen offset:1024\n\tbuffer_load_dword %3, %7, %8, 0 offen offset:1280\n\t"
                     "buffer_load_dword %4, %7, %8, 0 offen offset:1536\n\tbuffer_load_dword %5, %7, %8, 0 offen offset:1792\n\t"
                     "s_waitcnt vmcnt(8)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                     : "+v"(v), "=&v"(copy), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(src), "v"(voff), "s"(r) : "memory");
        if (d0 | d1 | d2 | d3) copy = 0;        // clipped loads return 0
    } else if (MODE == 2) {
        asm volatile("global_load_dword %0, %2, off\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                     : "+v"(v), "=&v"(copy) : "v"(src) : "memory");
    } else {
        asm volatile("global_load_dword %0, %2, off\n\t"
                     "buffer_store_dword %3, %4, %5, 0 offen\n\tbuffer_store_dword %3, %4, %5, 0 offen offset:256\n\t"
                     "buffer_store_dword %3, %4, %5, 0 offen offset:512\n\tbuffer_store_dword %3, %4, %5, 0 offen offset:768\n\t"
                     "buffer_store_dword %3, %4, %5, 0 offen offset:1024\n\tbuffer_store_dword %3, %4, %5, 0 offen offset:1280\n\t"
                     "buffer_store_dword %3, %4, %5, 0 offen offset:1536\n\tbuffer_store_dword %3, %4, %5, 0 offen offset:1792\n\t"
                     "s_waitcnt vmcnt(8)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                     : "+v"(v), "=&v"(copy) : "v"(src), "v"(zero), "v"(voff), "s"(r) : "memory");
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = copy;
}

int main() {
    const size_t big_words = (size_t)1 << 28;        // 1 GiB
    unsigned *big, *scratch, *out;
    CK(hipMalloc(&big, big_words * 4)); CK(hipMalloc(&scratch, 1 << 16)); 
    const int blocks = 8192, threads = 256;
    const size_t nthreads = (size_t)blocks * threads, nw = nthreads / 64, stride = big_words / nw;
    CK(hipMalloc(&out, nthreads * 4));
    CK(hipMemset(big, 0x5a, big_words * 4));
    std::vector<unsigned> h(nthreads);
    const char *names[4] = {"8 clipped stores + vmcnt(8)", "8 real stores + vmcnt(8)", "no wait (must fail)", "8 clipped loads + vmcnt(8)"};
    for (int mode = 0; mode < 4; ++mode) {
        size_t bad = 0;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemset(out, 0, nthreads * 4));
            // evict: touch another big region
            CK(hipMemset(big, 0x5a, big_words * 4));
            if (mode == 0) probe<0><<<blocks, threads>>>(big, scratch, out, stride);
            if (mode == 1) probe<1><<<blocks, threads>>>(big, scratch, out, stride);
            if (mode == 2) probe<2><<<blocks, threads>>>(big, scratch, out, stride);
            if (mode == 3) probe<3><<<blocks, threads>>>(big, scratch, out, stride);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), out, nthreads * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < nthreads; ++i) bad += h[i] != 0x5a5a5a5au;
        }
        printf("%-32s stale copies: %zu of %zu\n", names[mode], bad, 5 * nthreads);
    }
    return 0;
}
