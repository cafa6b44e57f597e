// The one-launch level-1+2 forward kernel (fused2d_l12.hpp) and its launcher: measurement only.
// It was part of libdtcwt_hip.so in round 2 (opt-in, DTCWT_HIP_FUSE12=1) and measured slower than the two launches
// it replaces in every protocol (profiles/r02/fwd12_*.txt, profiles/r03/fuse12_two_streams.txt): the arithmetic of
// the two levels (55 us at 4096^2) does not hide behind its traffic.  Kept here with its micro-benchmark
// (fwd12_bench.hip) and its host-emulator test (tests/test_emu_tiles.py).
// Include AFTER dtcwt_amd/csrc/fused2d.hip (Fwd1Params / Fwd2Params, tile_of, grid_for, cdiv, dt_pack_lh).
#pragma once
#include "fused2d_l12.hpp"

/* X(level-2 tile rows, cols, level-1 column-pass strip, level-2 (A,B) pairs per strip, len h0o, len h1o, q-shift length) */
#define DT_FWD12_TABLE(X) \
    X(16, 32, 8, 4, 5, 7, 10)     /* near_sym_a + qshift_a / qshift_06 */ \
    X(16, 28, 8, 4, 9, 7, 10)     /* antonini (9-tap lowpass: a wider halo, so a narrower core keeps one column-pass task per thread) */ \
    X(16, 32, 8, 4, 5, 3, 10)     /* legall */

namespace {

#define DT_WAIT_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70)      /* s_waitcnt vmcnt(0), gfx9 encoding (expcnt 7, lgkmcnt 15 = no wait) */
#define DT_OPAQUE(v_) asm volatile("" : "+v"(v_))             /* the value may have changed: nothing derived from it is loop-invariant */

// Levels 1 and 2 forward in one launch (fused2d_l12.hpp): LoLo1 stays in LDS.
// SKIP: phase knock-out bits for tools/kbench/fwd12_bench (timing experiments only; always 0 in the library)
template <class C, int SKIP = 0>
__global__ void __launch_bounds__(C::NT, C::MIN_WAVES) k_fwd12(Fwd1Params p1, Fwd2Params p2) {
    extern __shared__ __attribute__((aligned(16))) float smem[];       // C::LDS_FLOATS (may exceed 64 KiB)
    const int ntile = p2.tilesR * p2.tilesC * p2.B;
    int t = tile_of(blockIdx.x, ntile, p2.xcd_order);
    if (t >= ntile) return;
    int tc = t % p2.tilesC, tr = (t / p2.tilesC) % p2.tilesR, b = t / (p2.tilesC * p2.tilesR);
    float *sLo = smem, *sHi = sLo + C::SLO, *stage = sHi + C::SB;
    float *sLo2 = sHi, *sHi2 = sHi + C::S2;            // level-2 planes over the (dead) Hi plane
    const int tid = threadIdx.x;
    const int tidp = lds128_perm(tid);              // task index of this lane in the phases that read LDS 16 bytes at a time
    if (SKIP & 64) { tr = 2 + (tr & 3); tc = 2 + (tc & 3); }     // timing experiment: every workgroup on the same few (cache-resident) tiles
    const int r2 = tr * C::T2R, c2 = tc * C::T2C, r1 = 2 * r2, c1 = 2 * c2;
    if (!(SKIP & 1)) fwd12_cols<C>(p1, sLo, sHi, tid, b, r1, c1);
    // Every load of this workgroup has been consumed by now.  Saying so keeps the compiler from guarding later
    // re-uses of the window registers with s_waitcnt vmcnt(0) -- which, further down, would also wait for the
    // record STORES issued in between (vmcnt counts loads and stores in order) and put the HBM write latency
    // on the critical path of the tile.
    DT_WAIT_VMEM();
    __syncthreads();
    Fwd12State<C> st;
    if (!(SKIP & 2))
#pragma unroll
    for (int round = 0; round < C::NCR; ++round) {
        alignas(16) float rec[2][12];
        fwd12_core_compute<C>(p1, sLo, sHi, tidp, round, st, rec);
        if (!(SKIP & 32))
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            fwd12_core_deposit<C>(stage, tidp, round, half, rec);
            DT_WAVE_LDS_SYNC();
            fwd12_core_flush<C>(p1, stage, tid, round, half, b, r1, c1);
            DT_WAVE_LDS_SYNC();
        }
    }
    if (!(SKIP & 4)) fwd12_halo_compute<C>(p1, sLo, tidp, st);
    __syncthreads();                                // every read of the Lo plane is done: LoLo1 goes over it
    if (!(SKIP & 6)) fwd12_writeback<C>(p1, sLo, tidp, b, r1, c1, st);
    __syncthreads();
    if (fwd12_needs_fix<C>(p1, r1, c1)) {           // uniform per workgroup
        fwd12_fix<C>(p1, sLo, tid, r1, c1);
        __syncthreads();
    }
    if (!(SKIP & 8)) fwd12_cols2<C>(p2, sLo, sLo2, sHi2, tid);
    __syncthreads();
    if (!(SKIP & 16))
    for (int base = 0; base < C::TI * C::TJ; base += C::NT) {
        fwd2s_rows_compute<typename C::L2View>(p2, sLo2, sHi2, stage, tidp, base, b, r2, c2);
        DT_WAVE_LDS_SYNC();
        fwd2s_rows_flush<typename C::L2View>(p2, stage, tid, base, b, r2, c2);
        DT_WAVE_LDS_SYNC();
    }
}

// extra_lds: bytes of LDS requested on top of what the tile needs (occupancy experiments of tools/kbench only)
template <class C, int SKIP = 0>
int launch_fwd12(Fwd1Params &p1, Fwd2Params &p2, hipStream_t s, size_t extra_lds = 0) {
    p2.tilesR = cdiv(p2.LR / 2, C::T2R); p2.tilesC = cdiv(p2.LC / 2, C::T2C);
    dt_pack_lh(p2);
    const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float) + extra_lds;
    // the attribute is per device: set it on every launch (cheap) rather than once per process
    if (lds > (48u << 10) &&
        hipFuncSetAttribute((const void *)k_fwd12<C, SKIP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return -2;
    k_fwd12<C, SKIP><<<grid_for(p2.tilesR * p2.tilesC * p2.B, p2.xcd_order), C::NT, lds, s>>>(p1, p2);
    return 0;
}


}  // namespace
