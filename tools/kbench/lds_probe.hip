// LDS access-pattern probe (measurement tool): time per wave-instruction of ds_read_b128 / _b64 / _b32 and
// ds_write_b32 / _b64 / _b128 for the lane -> address mappings the tile programs use, relative to the lane-linear
// (conflict-free) mapping of the same instruction.  The instructions are issued from inline asm (the compiler
// would hoist loop-invariant LDS reads), 16 per s_waitcnt, by 4 workgroups x 4 waves per CU on every CU.
//
//   make -C tools/kbench lds_probe && tools/kbench/lds_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

enum { R128, R64, R32, W32, W64, W128 };

template <int OP>
__global__ void __launch_bounds__(256) k_probe(const int *__restrict__ addr_bytes, float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += 256) lds[i] = (float)i;
    __syncthreads();
    const unsigned a = (unsigned)addr_bytes[tid];
    v4f r4 = {0.f, 0.f, 0.f, 0.f};
    v2f r2 = {0.f, 0.f};
    float r1 = 0.f;
    const v4f w4 = {1.f, 2.f, 3.f, (float)tid};
    const v2f w2 = {1.f, (float)tid};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (OP == R128) asm volatile("ds_read_b128 %0, %1" : "=v"(r4) : "v"(a));
            if (OP == R64) asm volatile("ds_read_b64 %0, %1" : "=v"(r2) : "v"(a));
            if (OP == R32) asm volatile("ds_read_b32 %0, %1" : "=v"(r1) : "v"(a));
            if (OP == W32) asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(w2.y));
            if (OP == W64) asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(w2));
            if (OP == W128) asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(w4));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * 256 + tid] = r4.x + r4.w + r2.x + r2.y + r1;
}

static int *d_addr;
static float *d_out;
static hipStream_t st;

template <int OP>
static double run(const std::vector<int> &addr, int iters = 2000) {
    CK(hipMemcpy(d_addr, addr.data(), 256 * sizeof(int), hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = 256 * 4;
    k_probe<OP><<<grid, 256, 65536 / 2, st>>>(d_addr, d_out, 50);          // 32 KB: 4-5 workgroups per CU
    CK(hipEventRecord(a, st));
    k_probe<OP><<<grid, 256, 65536 / 2, st>>>(d_addr, d_out, iters);
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    // wave-instructions per CU: 4 workgroups x 4 waves x iters x 16
    return ms * 1e6 / (4.0 * 4 * iters * 16);      // ns per wave-instruction per CU
}

static int lds128_perm(int tid) {
    const int l = tid & 31;
    const int p = l < 4 ? l : l < 12 ? l + 12 : l < 16 ? l - 8 : l < 20 ? l + 8 : l < 28 ? l - 12 : l;
    return (tid & ~31) | p;
}

int main() {
    CK(hipSetDevice(0)); CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipMalloc(&d_addr, 256 * sizeof(int))); CK(hipMalloc(&d_out, 256 * 4 * 256 * sizeof(float)));
    auto mk = [](std::function<int(int)> f) { std::vector<int> v(256); for (int t = 0; t < 256; ++t) v[t] = 4 * (f(t) % 8000); return v; };
    struct Case { std::string name; int op; std::vector<int> addr; };
    std::vector<Case> cases;
    // ---- linear references
    cases.push_back({"r128 linear (16 B/lane)", R128, mk([](int t) { return 4 * t; })});
    cases.push_back({"r64  linear (8 B/lane)", R64, mk([](int t) { return 2 * t; })});
    cases.push_back({"r32  linear", R32, mk([](int t) { return t; })});
    cases.push_back({"w32  linear", W32, mk([](int t) { return t; })});
    cases.push_back({"w64  linear", W64, mk([](int t) { return 2 * t; })});
    cases.push_back({"w128 linear", W128, mk([](int t) { return 4 * t; })});
    // ---- b128 reads, 16 tasks to a row, row stride S floats, identity vs permuted lanes
    for (int S : {64, 72, 80, 88, 96, 128}) {
        cases.push_back({"r128 16/row stride " + std::to_string(S), R128, mk([S](int t) { return (t / 16) * S + 4 * (t % 16); })});
        cases.push_back({"r128 16/row stride " + std::to_string(S) + " perm", R128, mk([S](int t) { int l = lds128_perm(t); return (l / 16) * S + 4 * (l % 16); })});
    }
    // every second row (row pairs: the core row pass reads rows 2u + er): stride 2 S
    for (int S : {72, 88}) {
        cases.push_back({"r128 16/row stride 2x" + std::to_string(S), R128, mk([S](int t) { return (t / 16) * 2 * S + 4 * (t % 16); })});
        cases.push_back({"r128 16/row stride 2x" + std::to_string(S) + " perm", R128, mk([S](int t) { int l = lds128_perm(t); return (l / 16) * 2 * S + 4 * (l % 16); })});
    }
    // level-2 row pass of k_fwd2: 28 tasks to a row pair, NCI = 128, reads at 4 jl
    cases.push_back({"r128 28/row stride 2x128 (k_fwd2 rows)", R128, mk([](int t) { return (t / 28) * 256 + 4 * (t % 28); })});
    // ---- b64 reads: 32 tasks to a row (k_fwd1 row pass), row stride 2 W
    for (int S : {72, 128}) cases.push_back({"r64  32/row stride 2x" + std::to_string(S), R64, mk([S](int t) { return (t / 32) * 2 * S + 2 * (t % 32); })});
    // inverse row pass: 30 tasks to a row, windows at 4 q (b64 then b128)
    cases.push_back({"r64  30/row stride 128 at 4q (k_inv1 rows, old)", R64, mk([](int t) { return (t / 30) * 128 + 4 * (t % 30); })});
    cases.push_back({"r128 32/row stride 128 at 4q (k_inv1 rows, new)", R128, mk([](int t) { return (t / 32) * 128 + 4 * (t % 32); })});
    // record reads of the inverse gather: lane i of a parity reads record (row, i): 48-byte lane stride
    cases.push_back({"r128 record stride 12 floats (inverse gather)", R128, mk([](int t) { return (t % 64) * 12; })});
    // ---- b32 column reads (level-2 column pass from LDS): lanes along cc, two strips
    cases.push_back({"r32  80 cols, strip stride 16 x 88", R32, mk([](int t) { return (t / 80) * 16 * 88 + (t % 80); })});
    // ---- writes
    cases.push_back({"w32  70/row stride 72 (3-D axis0 -> S0)", W32, mk([](int t) { return (t / 70) * 72 + t % 70; })});
    cases.push_back({"w32  72/row stride 72 (k_fwd1 column pass)", W32, mk([](int t) { return (t / 72) * 8 * 72 + t % 72; })});
    cases.push_back({"w64  42/row stride 8 x 88 (k_fwd12 column pass)", W64, mk([](int t) { return (t / 42) * 8 * 88 + 2 + 2 * (t % 42); })});
    cases.push_back({"w128 record slab stride 12 floats", W128, mk([](int t) { return (t % 64) * 12; })});
    cases.push_back({"w128 record slab stride 24 floats", W128, mk([](int t) { return (t % 32) * 24; })});
    cases.push_back({"w128 record slab stride 15 f4 (3-D, 60 floats)", W128, mk([](int t) { return (t % 32) * 60; })});
    cases.push_back({"w128 16/row stride 64 (3-D axis2 -> S1)", W128, mk([](int t) { return (t / 16) * 64 + 4 * (t % 16); })});
    cases.push_back({"w32  column: lanes along 64-float rows? stride 64", W32, mk([](int t) { return (t % 64) * 64 + t / 64; })});
    for (auto &c : cases) {
        double ns = 0;
        switch (c.op) {
            case R128: ns = run<R128>(c.addr); break;
            case R64: ns = run<R64>(c.addr); break;
            case R32: ns = run<R32>(c.addr); break;
            case W32: ns = run<W32>(c.addr); break;
            case W64: ns = run<W64>(c.addr); break;
            case W128: ns = run<W128>(c.addr); break;
        }
        printf("%-58s %7.2f ns per wave-instruction per CU\n", c.name.c_str(), ns);
        fflush(stdout);
    }
    return 0;
}
