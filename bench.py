#!/usr/bin/env python
"""Benchmark of the hot path: 2-D DT-CWT forward + inverse, 4096x4096 float32, nlevels=4,
near_sym_a / qshift_a (BASELINE.json metric, configs[1]) on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one forward + one inverse of one image per GPU, input and pyramid resident in
HBM.  N > 1 is weak scaling over independent images (the path shards by image, no
data-path collective; the only collective is one RCCL broadcast of the filter taps at
set-up).  Rank 0 prints ONE JSON line.  The roofline object is for the dominant kernel
(the slower of the two level-1 kernels), its duration measured with hipEvent pairs on the
library's stream; cpu_baseline times the NumPy oracle (a port of the reference's
algorithm) on one image on the host.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ROWS = COLS = 4096
NLEVELS = 4
BIORT, QSHIFT = 'near_sym_a', 'qshift_a'
HBM_PEAK = 8.0e12             # B/s, MI355X spec (MI355X_MICROARCH.md)
FWD_BYTES_PER_PX = 20.0       # read X, write Yl + all Yh
STEP_BYTES_PER_PX = 40.0      # forward + inverse
L1_BYTES_PER_PX = 20.0        # level-1 kernel: X 4 + LoLo 4 + Yh[0] 12 (inverse: mirrored)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=500)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--graph', action='store_true', help='replay the eight level kernels of a step as one hipGraph '
                    '(measured: 0.234 vs 0.229 ms/step for plain stream launches, so not the default)')
    ap.add_argument('--settle-ms', type=float, default=300.0,
                    help='untimed steps for this long before the W warmup steps: after an idle period the '
                         'device needs ~20 ms of load to reach its sustained clock (0 disables)')
    ap.add_argument('--rows', type=int, default=ROWS)
    ap.add_argument('--cols', type=int, default=COLS)
    ap.add_argument('--batch', type=int, default=1, help='images per GPU per step')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    torch = None
    # torch is plumbing only (RCCL barrier / broadcast, device sync): imported for N > 1 or
    # on request; at N = 1 the bracket is hipDeviceSynchronize() through the library, which is
    # what torch.cuda.synchronize() calls (and a cold `import torch` costs minutes on a fresh box).
    use_dist = world > 1 or os.environ.get('DTCWT_BENCH_FORCE_DIST', '0') == '1'   # latter: exercise RCCL at N=1
    if use_dist or os.environ.get('DTCWT_BENCH_TORCH', '0') == '1':
        import torch
    saved_stdout = None
    if use_dist:
        # RCCL prints its own banner lines on stdout at initialisation: keep stdout for the one
        # JSON line of the contract, send everything else to stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    import dtcwt_amd
    from dtcwt_amd.coeffs import biort, qshift
    from dtcwt_amd.hip import Context, DeviceArray, _lib

    if not _lib.have_hip():
        raise _lib.NoHIPPresentError('bench.py needs a GPU and dtcwt_amd/libdtcwt_hip.so')
    ctx = Context(local_rank % max(_lib.device_count(), 1))

    # filter taps: rank 0 owns the table, one RCCL broadcast over xGMI hands it to the others
    bt, qt = biort(BIORT), qshift(QSHIFT)
    if use_dist:
        from dtcwt_amd.hip.sharding import broadcast_taps
        bt, qt = broadcast_taps(bt, qt, dist, device=torch.device('cuda', local_rank), src=0)

    B, R, C = args.batch, args.rows, args.cols
    t2 = dtcwt_amd.hip.Transform2d(tuple(bt), tuple(qt), ctx=ctx)
    plan = t2.plan(B, R, C, NLEVELS)
    rs = np.random.RandomState(1000 * rank)             # random, not zero: DVFS
    X = ctx.to_device(rs.standard_normal((B, R, C)).astype(np.float32))
    Yl = DeviceArray(ctx, (B,) + plan.low, np.float32)
    Yh = [DeviceArray(ctx, (B,) + plan.high[l] + (6,), np.complex64) for l in range(NLEVELS)]
    Z = DeviceArray(ctx, (B,) + plan.ext, np.float32)

    # one step = the forward and the inverse level loops on fixed buffers, 8 launches on one stream; --graph
    # replays them as one captured hipGraph instead (same kernels, same order)
    graph = plan.capture(X, Yl, Yh, Z) if args.graph else None

    def step_direct():
        plan.forward_into(X, Yl, Yh)
        plan.inverse_into(Yl, Yh, None, Z)

    def step():
        if graph is not None:
            graph.launch()
        else:
            step_direct()

    def fence():
        ctx.sync()
        if use_dist:
            dist.barrier()
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()
        ctx.device_sync()

    # Device settle: measured on MI355X, the first ~20 ms of work after an idle period run ~20 % slower
    # (20 timed steps: 0.275 ms/step after 10 warmup steps, 0.225 after 200), so a short run would report
    # the clock ramp, not the kernels.  Same work, untimed, then the W warmup steps of the contract.
    if args.settle_ms > 0:
        t_settle = time.perf_counter()
        for _ in range(20):
            step()
        ctx.sync()
        per_step = max((time.perf_counter() - t_settle) / 20, 1e-5)
        for _ in range(int(args.settle_ms * 1e-3 / per_step) + 1):     # no sync: the load stays continuous
            step()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # sanity of the timed work: reconstruction equals the input
    err = float(np.abs(Z.get()[0, :64, :64] - X.get()[0, :64, :64]).max())

    # ---- roofline of the dominant kernel: hipEvent pair around every level kernel -------
    plan.set_profiling(True)
    kf = np.zeros(NLEVELS); ki = np.zeros(NLEVELS)
    nprof = max(5, min(args.steps, 50))
    for _ in range(nprof):
        step_direct()          # per-kernel hipEvent pairs need the plain launches
        f, i = plan.kernel_ms()
        kf += f; ki += i
    plan.set_profiling(False)
    kf /= nprof; ki /= nprof
    px = float(B) * R * C
    cand = [('k_fwd1 (level-1 forward)', kf[0]), ('k_inv1 (level-1 inverse)', ki[0])]
    name, ms = max(cand, key=lambda c: c[1])
    achieved = L1_BYTES_PER_PX * px / (ms * 1e-3) / 1e9       # GB/s
    roofline = {'bound': 'hbm', 'kernel': name, 'achieved': round(achieved, 1), 'peak': HBM_PEAK / 1e9,
                'unit': 'GB/s', 'frac': round(achieved * 1e9 / HBM_PEAK, 4), 'traffic': None,
                'kernel_ms': round(float(ms), 5), 'algorithmic_bytes_per_launch': L1_BYTES_PER_PX * px,
                'fwd_kernel_ms': [round(float(x), 5) for x in kf],
                'inv_kernel_ms': [round(float(x), 5) for x in ki],
                'step_frac': round(STEP_BYTES_PER_PX * px / (dt / args.steps) / HBM_PEAK, 4)}
    tr = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.exists(tr):
        try:
            roofline['traffic'] = json.load(open(tr)).get(name.split(' ')[0])
        except Exception:
            pass

    value = world * px * args.steps / dt / 1e6
    out = {
        'metric': 'Mpixels/s 2D DT-CWT fwd+inv, 4096^2 f32 nlevels=4',
        'value': round(value, 1), 'unit': 'Mpixels/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'settle_ms': args.settle_ms, 'launch': 'hipGraph' if graph is not None else 'stream', 'ms_per_step': round(dt / args.steps * 1e3, 5),
        'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '2D forward+inverse %dx%d f32, nlevels=%d, %s/%s, %d image(s) per GPU per step'
                               % (R, C, NLEVELS, BIORT, QSHIFT, B),
                   'sharding': 'independent images per GPU, no data-path collective'},
        'roofline': roofline, 'recon_max_abs_err': err,
    }

    # ---- CPU baseline: the oracle (a NumPy port of the reference's algorithm), rank 0 ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import dtcwt_oracle as o
        Xh = X.get()[0]
        to = o.Transform2d(biort(BIORT), qshift(QSHIFT))
        c0 = time.perf_counter()
        p = to.forward(Xh, nlevels=NLEVELS)
        zc = to.inverse(p)
        cdt = time.perf_counter() - c0
        out['cpu_baseline'] = {'value': round(R * C / cdt / 1e6, 3), 'unit': 'Mpixels/s', 'cores': 1,
                               'kind': 'port', 'host_cpus': os.cpu_count(), 'numpy': np.__version__,
                               'sample': '1 image %dx%d f32 fwd+inv nlevels=%d (%.1f s)' % (R, C, NLEVELS, cdt)}
        # the timed GPU output against the CPU port on the same input
        out['gpu_vs_cpu_recon_max_abs_diff'] = float(np.abs(Z.get()[0] - zc).max())
    elif rank == 0:
        out['cpu_baseline'] = None
    if rank == 0:
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        if saved_stdout is not None:
            os.dup2(2, 1)           # teardown chatter of the collective library, if any
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
