// Where do the wavefronts of two-wavefront workgroups land?  128-thread workgroups, two wavefronts per SIMD (the shape of
// the marching pairs, march2d_pair.hpp): every wavefront records HW_REG_HW_ID (WAVE_ID [3:0], SIMD_ID [5:4], CU_ID [11:8],
// SE_ID [15:13]) and HW_REG_XCC_ID, then spins so that the chip fills.  Prints, per (xcc, se, cu), which (workgroup, wavefront)
// sits on which SIMD and slot.
//   hipcc --offload-arch=gfx950 -O2 tools/kbench/hwid_probe.hip -o /tmp/hwid_probe && /tmp/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <tuple>
#include <algorithm>

__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) k_probe(unsigned *out, int spin) {
    __shared__ float pad[4096];            // 16 KB: a few workgroups per CU at most by LDS as well
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    float a = (float)lane;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    pad[threadIdx.x] = a;
    __syncthreads();
    if (lane == 0) { out[4 * (2 * blockIdx.x + wv)] = hw; out[4 * (2 * blockIdx.x + wv) + 1] = xcc; out[4 * (2 * blockIdx.x + wv) + 2] = (unsigned)pad[(threadIdx.x + 1) & 127]; }
}

int main() {
    const int nwg = 2048;
    unsigned *d; hipMalloc(&d, nwg * 2 * 4 * sizeof(unsigned));
    hipMemset(d, 0, nwg * 2 * 4 * sizeof(unsigned));
    k_probe<<<nwg, 128>>>(d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nwg * 2 * 4);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    // per CU: list of (wg, wv, simd, slot)
    std::map<std::tuple<int, int, int>, std::vector<std::tuple<int, int, int, int>>> cu;
    int same_slot = 0, adj = 0, cnt[2][4] = {{0}};
    for (int w = 0; w < nwg; ++w) {
        unsigned h0 = h[4 * (2 * w)], h1 = h[4 * (2 * w + 1)];
        for (int v = 0; v < 2; ++v) {
            const unsigned hw = h[4 * (2 * w + v)], xcc = h[4 * (2 * w + v) + 1] & 0xf;
            cu[{(int)xcc, (int)((hw >> 13) & 7), (int)((hw >> 8) & 15)}].push_back({w, v, (int)((hw >> 4) & 3), (int)(hw & 15)});
            cnt[v][(hw >> 4) & 3]++;
        }
        same_slot += (h0 & 15) == (h1 & 15);
        adj += (((h0 >> 4) & 3) ^ 1) == ((h1 >> 4) & 3);
    }
    printf("workgroups %d: both wavefronts in the same slot index %d, on SIMDs s and s^1 %d\n", nwg, same_slot, adj);
    printf("wavefront 0 on SIMD 0..3: %d %d %d %d   wavefront 1: %d %d %d %d\n", cnt[0][0], cnt[0][1], cnt[0][2], cnt[0][3], cnt[1][0], cnt[1][1], cnt[1][2], cnt[1][3]);
    int shown = 0;
    for (auto &kv : cu) {
        if (shown++ >= 6) break;
        auto v = kv.second; std::sort(v.begin(), v.end());
        printf("xcc %d se %d cu %d:", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first));
        for (auto &t : v) printf("  wg%d.%d->simd%d/slot%d", std::get<0>(t), std::get<1>(t), std::get<2>(t), std::get<3>(t));
        printf("\n");
    }
    printf("CUs seen %zu\n", cu.size());
    return 0;
}
