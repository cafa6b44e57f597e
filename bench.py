#!/usr/bin/env python
"""Benchmark of the hot path: 2-D DT-CWT forward + inverse, 4096x4096 float32, nlevels=4,
near_sym_a / qshift_a (BASELINE.json metric, configs[1]) on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c5|c5full|c4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one forward + one inverse of one batch per GPU, input and pyramid resident in
HBM.  N > 1 is weak scaling over independent images (the path shards by image, no data-path
collective; the only collective is one RCCL broadcast of the filter taps at set-up), one
process per GPU: started without a launcher (`python bench.py --gpus N`, no WORLD_SIZE in the
environment) the script re-executes itself under `torch.distributed.run` with N ranks;
`--mgpu` instead drives the N devices from ONE process through dtcwt_hip_mgpu_* (one host
thread per device).  Rank 0 prints ONE JSON line.

Steps rotate over `--sets` (default 8) distinct sets of input / pyramid / output buffers, about
0.4 GB each for the headline config, so that no step finds its input or the previous step's
pyramid in the 256 MiB Infinity Cache; `resident_ms_per_step` in the JSON line is the same
step on ONE buffer set (what round 1 reported), for comparison.

`ms_per_step` / `value` are throughput over `--streams` (default 4; two until round 4: with the
marching kernels, which run one or two wavefronts to a SIMD, four images in flight measured 0.178
against 0.187 ms per step, profiles/r04/ab_streams.txt) HIP streams of independent
images (the coarse-level kernels of one image overlap the level-1 kernels of the next); it is not
the latency of one forward + inverse: `one_stream_ms_per_step` in the same line is the same
rotating-buffer protocol on ONE stream.  For c2, c3 and c5 each of the streams belongs to a context on its own share of the
compute units (`--cu-partition`, dtcwt_hip_ctx_create_partition: -5 to -7 % per step, profiles/r04/ab_partition.txt; c3 since the
round-5 kernels: -2.5 to -3 %, profiles/r05/batch_streams.txt); the
one-at-a-time phases and the roofline object's kernel times use a context on the whole device.

The roofline object is for the dominant kernel (the launch with the longest median duration: the
forward's levels 1 + 2, which run as ONE marching launch k_fwd12m when the geometry allows, or the
level-1 kernels k_fwd1 / k_inv1, or the inverse's one-launch levels 2 + 1, k_inv21m): `kernel_ms` is the MEDIAN over the profiled steps of the raw
hipEvent-pair time around that kernel on the library's stream (an empty pair costs
`event_pair_overhead_ms`, reported, not subtracted); `rocprof_kernel_ms` next to it is the median of
the same kernel under `rocprofv3 --kernel-trace` of this command (tools/profile_round.sh ->
profiles/traffic.json), as is `traffic`.  cpu_baseline times the NumPy oracle (a port of the
reference's algorithm) on the host, rank 0 only, after the process group is gone.

At N = 1 the default (c2) run also times short versions of the other single-GPU BASELINE configs -- c3, one GPU's
share of c5, c4 -- as sub-runs of this script after its own measurement and before the CPU baseline, and reports
them under `other_configs` (`--no-other-configs` skips them).  It then runs tools/kbench/step_probe (a trivial float4
streaming program that moves the algorithmic bytes of a step in the same six launches, in the same three protocols:
one stream / four plain streams / four streams on quarters of the compute units) and reports its ms per step and the
transform's figures over it as `streaming_probe` (`--no-probe` skips it): not a roofline -- `roofline` quotes the 8 TB/s
the contract asks for -- but what this box gives a program WITHOUT arithmetic, halo or warm-up rows in this very run.  `device_under_load` is rocm-smi's sclk and
socket power read while the configuration keeps running (`--no-clocks` skips it): boxes of one pool differed by +-10 % on one tree.

`--config c4` (BASELINE configs[3], one volume: it does not shard, N = 1 only) times the 3-D transform the same way:
a step = Transform3d forward + inverse of one 256^3 float32 volume, nlevels=3, rotating over `--sets` volumes on
`--streams` streams (independent volumes in flight; `ms_per_step_one_stream` = one at a time); the roofline object is for k_fwd3_l1 (level 1 of the forward, 36 B/voxel), timed by a raw event pair around
its own C entry.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BIORT, QSHIFT = 'near_sym_a', 'qshift_a'
HBM_PEAK = 8.0e12             # B/s, MI355X spec (MI355X_MICROARCH.md)
HBM_COPY = 6.29e12            # B/s, what a float4 copy kernel reaches on this part (MI355X_MICROARCH.md): SURVEY 8(d) asks for both
FWD_BYTES_PER_PX = 20.0       # read X, write Yl + all Yh
STEP_BYTES_PER_PX = 40.0      # forward + inverse

# BASELINE.json configs that run on one GPU (c5: one GPU's share of the 512-image batch)
CONFIGS = {
    'c2': dict(rows=4096, cols=4096, batch=1, nlevels=4, seed=lambda rank: 1000 * rank, cu_partition=True,
               name='2D forward+inverse 4096x4096 f32, nlevels=4'),
    'c3': dict(rows=1024, cols=1024, batch=64, nlevels=5, seed=lambda rank: 2 + 1000 * rank, cu_partition=True,
               name='batched 2D 64x1024x1024 f32, nlevels=5'),
    'c5': dict(rows=2048, cols=2048, batch=64, nlevels=4, seed=lambda rank: 3 + 1000 * rank, cu_partition=True,
               name='batched 2D 512x2048x2048 f32 nlevels=4 sharded over 8 GPUs: 64 images per GPU'),
    # the WHOLE C5 batch on one GPU: 2.1 G pixels, Yh[0] 6.4 G floats (> 2^31 elements), 58 GB per buffer set
    'c5full': dict(rows=2048, cols=2048, batch=512, nlevels=4, seed=lambda rank: 3 + 1000 * rank, sets=1, streams=1,
                   name='batched 2D 512x2048x2048 f32 nlevels=4, the whole batch on ONE GPU'),
}


def shard_plan(cfg, world):
    """Weak scaling over independent images: what each of `world` ranks transforms per step, as
    [(seed of its synthetic images, images per step)], and the global batch (the reference's shape:
    examples/register_video.py:125-156 hands a contiguous run of frames to each worker)."""
    return [(cfg['seed'](r), cfg['batch']) for r in range(world)], world * cfg['batch']


def aggregate_value(world, px_per_rank, steps, dt_max):
    """Whole-job Mpixels/s: the pixels ALL ranks processed in `steps` steps over the slowest rank's time."""
    return world * px_per_rank * steps / dt_max / 1e6


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=500)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', choices=sorted(CONFIGS) + ['c4'], default='c2')
    ap.add_argument('--sets', type=int, default=8, help='distinct buffer sets the steps rotate over')
    ap.add_argument('--streams', type=int, default=4, help='HIP streams the steps alternate over (independent images: step k '
                    'runs on stream k %% S, each with its own plan and buffer sets): the small coarse-level kernels of one '
                    'image overlap the large level-1 kernels of the next')
    ap.add_argument('--cu-partition', choices=['auto', 'on', 'off'], default='auto', help='with S > 1 streams: one context per '
                    'share of the compute units (dtcwt_hip_ctx_create_partition) instead of plain streams whose kernels '
                    'share every CU.  auto: where it measured faster (profiles/r04/ab_partition.txt): c2 -5 %%, c5 -2 %%; '
                    'not c3 (+4 %%) and c4 (+4 %%)')
    ap.add_argument('--per-share', type=int, default=1, help='with a CU partition: this many streams on EVERY share of the compute units '
                    '(--streams 8 --per-share 2 = eight images in flight, two on each quarter: while one stream sits in the gap between two '
                    'of its launches, the other one\'s kernel has the share)')
    ap.add_argument('--mgpu', action='store_true', help='N > 1 from ONE process: dtcwt_hip_mgpu_* with a host '
                    'thread per device instead of one process per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-n', type=int, default=0, help='--config c4: the CPU sample is the leading n^3 corner of the first volume '
                    '(default: the whole volume; the c4_qbgn sub-run of the default line passes 192: the long filters take ~40 s at 256^3)')
    ap.add_argument('--cpu-baseline-s', type=float, default=0.0, help='the CPU sample repeats its image until this many seconds have '
                    'passed (the sub-runs of the batch configurations)')
    ap.add_argument('--ab-protocols', type=int, default=0, help='c3 / c5 at N = 1: after the timed region, time the same steps on four '
                    'PLAIN streams and on four streams on QUARTERS of the compute units, alternating, this many times each '
                    '(`ab_protocols` in the line; the sub-runs of the default line pass 3)')
    ap.add_argument('--no-clocks', action='store_true', help='c2 at N = 1 only: do not read rocm-smi (sclk, socket power) under the load')
    ap.add_argument('--no-probe', action='store_true', help='c2 at N = 1 only: do not run tools/kbench/step_probe (a trivial float4 '
                    'streaming program moving the algorithmic bytes of a step in the same launch structure) beside the transform')
    ap.add_argument('--no-other-configs', action='store_true', help='c2 at N = 1 only: do not append short runs of '
                    'c3, c5 (the share of one GPU) and c4 as "other_configs"')
    ap.add_argument('--graph', action='store_true', help='replay the level kernels of a step as one hipGraph '
                    '(measured: not faster than plain stream launches, so not the default)')
    ap.add_argument('--settle-ms', type=float, default=300.0,
                    help='untimed steps for this long before the W warmup steps: after an idle period the '
                         'device needs ~20 ms of load to reach its sustained clock (0 disables)')
    ap.add_argument('--nlevels', type=int, default=None, help='experiments only: another level count than the config names')
    ap.add_argument('--rows', type=int, default=None)
    ap.add_argument('--cols', type=int, default=None)
    ap.add_argument('--batch', type=int, default=None, help='images per GPU per step')
    ap.add_argument('--biort', default=None, help='another level-1 wavelet than the headline near_sym_a (e.g. near_sym_b): the line '
                    'names it in config.workload and is then NOT the BASELINE metric')
    ap.add_argument('--qshift', default=None, help='another level >= 2 wavelet than qshift_a (e.g. qshift_b)')
    a = ap.parse_args(argv)
    global BIORT, QSHIFT
    if a.biort:
        BIORT = a.biort
    if a.qshift:
        QSHIFT = a.qshift
    return a


def respawn_under_launcher(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: one process per GPU via torch.distributed.run."""
    from dtcwt_amd.hip import _lib
    ndev = _lib.device_count()
    if ndev < args.gpus:
        raise SystemExit('bench.py --gpus %d: only %d HIP device(s) visible' % (args.gpus, ndev))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    raise SystemExit(subprocess.call(cmd, env=env))


def upload_random(ctx, rs, B, R, C, chunk=64):
    """[B][R][C] float32 standard-normal samples in HBM, generated and uploaded `chunk` images at a time."""
    from dtcwt_amd.hip import DeviceArray
    if B <= chunk:
        return ctx.to_device(rs.standard_normal((B, R, C)).astype(np.float32))
    X = DeviceArray(ctx, (B, R, C), np.float32)
    per = R * C * 4
    for b0 in range(0, B, chunk):
        n = min(chunk, B - b0)
        DeviceArray(ctx, (n, R, C), np.float32, ptr=X.ptr + b0 * per, owner=X).set(
            rs.standard_normal((n, R, C)).astype(np.float32))
    return X


def cpu_baseline(cfg, Xh, min_s=0.0):
    """The NumPy oracle (port of the reference's algorithm, one core) on a bounded sample: one image, repeated until
    `min_s` seconds have passed (the sub-runs of the batch configurations: their images take 0.3 - 2 s each)."""
    from dtcwt_amd.coeffs import biort, qshift
    from oracle import dtcwt_oracle as o
    R, C = Xh.shape
    to = o.Transform2d(biort(BIORT), qshift(QSHIFT))
    c0 = time.perf_counter()
    n = 0
    while True:
        p = to.forward(Xh, nlevels=cfg['nlevels'])
        zc = to.inverse(p)
        n += 1
        cdt = time.perf_counter() - c0
        if cdt >= min_s:
            break
    return {'value': round(n * R * C / cdt / 1e6, 3), 'unit': 'Mpixels/s', 'cores': 1, 'kind': 'port',
            'host_cpus': os.cpu_count(), 'numpy': np.__version__,
            'sample': '%d image%s %dx%d f32 fwd+inv nlevels=%d (%.1f s)' % (n, '' if n == 1 else 's', R, C, cfg['nlevels'], cdt)}, zc


def main():
    args = parse_args()
    if args.config == 'c4':
        return main_c4(args)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and not args.mgpu:
        respawn_under_launcher(args)
    if args.mgpu:
        return main_mgpu(args)

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and rank == 0:
        print('bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus=%d'
              % (args.gpus, world, world), file=sys.stderr)
    cfg = dict(CONFIGS[args.config])
    for k in ('rows', 'cols', 'batch', 'nlevels'):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)
    dist = None
    torch = None
    # torch is plumbing only (RCCL barrier / broadcast, device sync): imported for N > 1 or
    # on request; at N = 1 the bracket is hipDeviceSynchronize() through the library, which is
    # what torch.cuda.synchronize() calls (and a cold `import torch` costs minutes on a fresh box).
    use_dist = world > 1 or os.environ.get('DTCWT_BENCH_FORCE_DIST', '0') == '1'   # latter: exercise RCCL at N=1
    if use_dist or os.environ.get('DTCWT_BENCH_TORCH', '0') == '1':
        import torch
    saved_stdout = None
    if use_dist and 'RANK' not in os.environ:          # DTCWT_BENCH_FORCE_DIST=1 without a launcher: a one-rank group
        os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
                          MASTER_PORT=str(_free_port()))
    if use_dist:
        # RCCL prints its own banner lines on stdout at initialisation: keep stdout for the one
        # JSON line of the contract, send everything else to stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        # Two groups: RCCL (backend "nccl") carries the path's one collective, the broadcast of the filter taps over
        # xGMI, and is torn down right after it; the barriers and the reduction of the ranks' times around the timed
        # region run on a gloo group (host side) next to explicit device synchronisation.  With the RCCL communicator
        # left alive the step itself ran slower -- its watchdog / proxy threads share the HIP runtime with the
        # launching thread: 0.214-0.233 ms per step at one rank against 0.197 ms (profiles/r03/rccl_alive.txt).
        # DTCWT_BENCH_BARRIER=nccl keeps the round-2 arrangement (everything on one RCCL group).
        barrier_backend = os.environ.get('DTCWT_BENCH_BARRIER', 'gloo')
        if barrier_backend != 'nccl' and 'GLOO_SOCKET_IFNAME' not in os.environ:
            # gloo picks its interface by resolving the host name, which containers often cannot: all ranks are on
            # this node, the loopback interface will do
            try:
                socket.gethostbyname(socket.gethostname())
            except OSError:
                os.environ['GLOO_SOCKET_IFNAME'] = 'lo'
        if barrier_backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
            rccl_group = None
        else:
            try:
                dist.init_process_group('gloo')
                rccl_group = dist.new_group(backend='nccl', device_id=torch.device('cuda', local_rank))
            except Exception as exc:       # no usable gloo transport here: everything on RCCL, as in round 2
                print('bench.py: gloo group unavailable (%s); barriers on RCCL' % exc, file=sys.stderr)
                if dist.is_initialized():
                    dist.destroy_process_group()
                barrier_backend = 'nccl'
                dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
                rccl_group = None

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
        bt, qt = broadcast_taps(bt, qt, dist, device=torch.device('cuda', local_rank), src=0, group=rccl_group)
        rccl_ranks = dist.get_world_size(group=rccl_group)
        if rccl_group is not None:
            torch.cuda.synchronize()
            dist.destroy_process_group(rccl_group)
        red_dev = 'cuda' if barrier_backend == 'nccl' else 'cpu'      # where the timing reductions live

    B, R, C, NL = cfg['batch'], cfg['rows'], cfg['cols'], cfg['nlevels']
    nstreams = max(1, cfg.get('streams', args.streams))
    nsets = max(1, cfg.get('sets', args.sets))
    if nsets % nstreams:
        nsets = (nsets // nstreams + 1) * nstreams      # every buffer set belongs to exactly one stream
    # S images in flight: one context per image stream, each on its own share of the compute units (a slice of every
    # XCD): 0.152-0.157 against 0.165-0.170 ms per step on plain streams (profiles/r04/ab_cu_mask_2.txt).  `ctx` keeps the
    # whole device for the uploads and the one-at-a-time phases below.
    # (two and four shares measured faster than plain streams, three slower: 85 CUs do not divide the XCDs evenly)
    per_share = max(1, args.per_share)
    nshares = nstreams // per_share if nstreams % per_share == 0 else nstreams
    partitioned = nstreams > 1 and (args.cu_partition == 'on' or (args.cu_partition == 'auto' and cfg.get('cu_partition', False) and nshares in (2, 4)))
    if partitioned:
        try:
            ctxs = [Context(ctx.device, partition=(s % nshares, nshares)) for s in range(nstreams)]
        except _lib.HipError as exc:        # a runtime that refuses CU masks: plain streams, and the line says so
            print('bench.py: no CU partition (%s); plain streams' % exc, file=sys.stderr)
            partitioned = False
    if not partitioned:
        ctxs = [ctx] + [Context(ctx.device) for _ in range(nstreams - 1)]
    t2s = [dtcwt_amd.hip.Transform2d(tuple(bt), tuple(qt), ctx=c) for c in ctxs]
    plans = [t.plan(B, R, C, NL) for t in t2s]
    t2 = dtcwt_amd.hip.Transform2d(tuple(bt), tuple(qt), ctx=ctx) if partitioned else t2s[0]
    plan = t2.plan(B, R, C, NL) if partitioned else plans[0]          # the whole device, one transform at a time
    if not partitioned:
        for pl in plans:
            pl.set_concurrency(nstreams)    # the images in flight on the device: the marching launches size their bands by it
    rs = np.random.RandomState(shard_plan(cfg, world)[0][rank][0])   # random, not zero: DVFS; per-shard seed
    sets = []
    for k in range(nsets):
        c = ctxs[k % nstreams]
        X = upload_random(c, rs, B, R, C)
        Yl = DeviceArray(c, (B,) + plan.low, np.float32)
        Yh = [DeviceArray(c, (B,) + plan.high[l] + (6,), np.complex64) for l in range(NL)]
        Z = DeviceArray(c, (B,) + plan.ext, np.float32)
        sets.append((X, Yl, Yh, Z))
    set_bytes = sum(a.nbytes for a in (sets[0][0], sets[0][1], sets[0][3])) + sum(a.nbytes for a in sets[0][2])

    # one step = the forward and the inverse level loops of one buffer set on one stream; --graph replays
    # them as one captured hipGraph per set instead (same kernels, same order)
    graphs = [plans[k % nstreams].capture(*s[:3], s[3]) for k, s in enumerate(sets)] if args.graph else None
    counter = [0]

    def step_on(k, pl=None):
        X, Yl, Yh, Z = sets[k]
        pl = pl or plans[k % nstreams]
        pl.forward_into(X, Yl, Yh)
        pl.inverse_into(Yl, Yh, None, Z)

    def step():
        k = counter[0] % nsets
        counter[0] += 1
        if graphs is not None:
            graphs[k].launch()
        else:
            step_on(k)

    def drain():
        """This rank's GPU work is done: every stream of the library, then the whole device."""
        for c in ctxs:
            c.sync()
        ctx.sync()
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()
        ctx.device_sync()

    def fence():
        """Barrier over the ranks + device synchronisation (the bracket of the timed region)."""
        drain()
        if use_dist:
            dist.barrier()
            drain()             # the barrier is itself a collective kernel on this device

    # Device settle: measured on MI355X, the first ~20 ms of work after an idle period run ~20 % slower
    # (20 timed steps: 0.275 ms/step after 10 warmup steps, 0.225 after 200), so a short run would report
    # the clock ramp, not the kernels.  Same work, untimed, then the W warmup steps of the contract.
    if args.settle_ms > 0:
        t_settle = time.perf_counter()
        for _ in range(20):
            step()
        for c in ctxs:
            c.sync()
        per_step = max((time.perf_counter() - t_settle) / 20, 1e-5)
        for _ in range(int(args.settle_ms * 1e-3 / per_step) + 1):     # no sync: the load stays continuous
            step()
    for _ in range(args.warmup):
        step()
    # The timed region: barrier + synchronise, K steps, synchronise -- the clock stops when THIS rank's work is
    # done -- barrier + synchronise again; the job's time is the MAX over the ranks' times.  (With the clock stopped
    # after the closing barrier instead, a RCCL barrier -- a collective launch plus its completion, ~1 ms -- would
    # be charged to the K steps: +30 % on a 20-step run of 0.2 ms steps, measured at one rank.)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    dt = time.perf_counter() - t0
    fence()
    rank_ms = [dt / args.steps * 1e3]
    # the same region over >= 300 steps (bounded to ~0.5 s for the big batches): the 20-step figure of the driver's command
    # carries the start / drain of the streams (profiles/r04/stream_timeline.txt), this one does not
    n_sus = max(args.steps, min(300, int(0.5 / max(dt / args.steps, 1e-6)) + 1))
    sustained_ms = None
    if n_sus > args.steps and not use_dist:
        t1 = time.perf_counter()
        for _ in range(n_sus):
            step()
        drain()
        sustained_ms = (time.perf_counter() - t1) / n_sus * 1e3
    # the device's clock and socket power UNDER this load (boxes of one pool differed by +-10 % on the same tree: the line says what
    # the box was doing): rocm-smi read while the same steps run for ~1.5 s more; None where there is no rocm-smi
    under_load = None
    if not use_dist and rank == 0 and not args.no_clocks:
        under_load = clocks_under_load(step, drain, dt / args.steps)
    # ---- the two launch protocols in ONE run: the same steps on four plain streams and on four streams on
    # quarters of the compute units, alternating, the same buffer sets; min and median per protocol.  Box-to-box variance (+-8 %)
    # is larger than the difference between them, so only an A/B inside one process on one box says which the box prefers.
    ab = None
    if args.ab_protocols > 0 and not use_dist and nstreams == 4 and graphs is None:
        try:
            fam = {'quarters' if partitioned else 'plain_streams': plans}
            if partitioned:
                octx = [Context(ctx.device) for _ in range(nstreams)]
            else:
                octx = [Context(ctx.device, partition=(q, nstreams)) for q in range(nstreams)]
            ot2 = [dtcwt_amd.hip.Transform2d(tuple(bt), tuple(qt), ctx=c) for c in octx]
            opl = [t.plan(B, R, C, NL) for t in ot2]
            if partitioned:
                for pl in opl:
                    pl.set_concurrency(nstreams)
            fam['plain_streams' if partitioned else 'quarters'] = opl
            n_ab = max(args.steps, 8)
            times = {k: [] for k in fam}

            def drain_all():
                for c in octx:
                    c.sync()
                drain()
            for rep in range(args.ab_protocols + 1):        # the first round of each is warm-up
                for key in ('plain_streams', 'quarters'):
                    pls = fam[key]
                    for k in range(nstreams):
                        step_on(k, pls[k % nstreams])
                    drain_all()
                    ta = time.perf_counter()
                    for i in range(n_ab):
                        k = i % nsets
                        step_on(k, pls[k % nstreams])
                    drain_all()
                    if rep:
                        times[key].append((time.perf_counter() - ta) / n_ab * 1e3)
            ab = {'steps': n_ab, 'reps': args.ab_protocols,
                  'plain_streams_ms_per_step': round(float(np.median(times['plain_streams'])), 5),
                  'quarters_ms_per_step': round(float(np.median(times['quarters'])), 5),
                  'plain_streams_min': round(min(times['plain_streams']), 5), 'quarters_min': round(min(times['quarters']), 5),
                  'timed_region_ran_on': 'quarters' if partitioned else 'plain_streams',
                  'is': 'the timed steps again on four plain streams / four streams on quarters of the compute units, alternating in this '
                        'process on the same buffer sets; median (and min) ms per step of `reps` regions of `steps` steps each'}
            del opl, ot2, octx
        except Exception as exc:
            ab = {'error': repr(exc)[:300]}
    nccl_ranks = 1
    if use_dist:
        nccl_ranks = rccl_ranks
        mine = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        every = [torch.zeros_like(mine) for _ in range(nccl_ranks)]
        dist.all_gather(every, mine)                    # every rank's own time: min / max in the JSON line
        rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]
        tt = mine.clone()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # sanity of the timed work: reconstruction equals the input, on every buffer set
    def first_image(a, shape):      # image 0 of a batch without downloading the whole batch
        return DeviceArray(a.ctx, (1,) + tuple(shape), np.float32, ptr=a.ptr, owner=a).get()[0]
    done = sets[:min(nsets, counter[0])]         # a very short run does not reach every buffer set
    # the WHOLE first image of every buffer set the timed steps wrote (a wrong band anywhere in it shows)
    err = max(float(np.abs(first_image(s[3], plan.ext)[:R, :C] - first_image(s[0], (R, C))).max()) for s in done)

    # the level kernels as they run in the timed region -- S images in flight, each stream on its share of the compute
    # units: event pairs around the launches of the LAST step of bursts of three steps per stream (the other streams are in
    # their own last steps meanwhile; the host reads the events between bursts, so the figure is approximate)
    in_flight = None
    if nstreams > 1 and graphs is None:
        fence()
        for pl in plans:
            pl.set_profiling(True)
        F, I = [], []
        for rep in range(8):
            for _ in range(3 * nstreams):
                step()
            for pl in plans:
                f_, i_ = pl.kernel_ms()
                F.append(f_); I.append(i_)
        for pl in plans:
            pl.set_profiling(False)
        in_flight = (np.median(np.array(F), axis=0), np.median(np.array(I), axis=0))

    # the same step on one buffer set and one stream only (input and pyramid may stay in the Infinity Cache): ONE
    # transform in flight, and the plan is told so
    fence()
    plan.set_concurrency(1)
    px_step = B * R * C
    nres = 100 if px_step <= 2 ** 25 else max(5, min(args.steps, 100))      # short timed regions carry ~0.5 ms of start / drain cost
    for _ in range(5):
        step_on(0, plan)
    ctx.sync()
    r0 = time.perf_counter()
    for _ in range(nres):
        step_on(0, plan)
    ctx.sync()
    resident_ms = (time.perf_counter() - r0) / nres * 1e3

    # ... and rotating over the buffer sets of stream 0 on that ONE stream: what a single forward + inverse
    # costs end to end, without the overlap of independent images
    own = [k for k in range(nsets) if k % nstreams == 0]
    for i in range(5):
        step_on(own[i % len(own)], plan)
    ctx.sync()
    r0 = time.perf_counter()
    for i in range(nres):
        step_on(own[i % len(own)], plan)
    ctx.sync()
    one_stream_ms = (time.perf_counter() - r0) / nres * 1e3

    # ---- roofline of the dominant kernel: hipEvent pair around every level kernel -------
    ev0, ev1 = ctx.event(), ctx.event()
    null_ms = []
    for _ in range(20):
        ev0.record(); ev1.record()
        null_ms.append(ev0.elapsed_ms(ev1))
    null_ms = float(np.median(null_ms))                 # an event pair with nothing in between: reported, NOT subtracted
    plan.set_profiling(True)
    nprof = max(5, min(args.steps, 50))
    kf = np.zeros((nprof, NL)); ki = np.zeros((nprof, NL))
    for i in range(nprof):
        step_on(own[i % len(own)], plan)    # per-kernel hipEvent pairs need the plain launches, on the whole-device stream
        kf[i], ki[i] = plan.kernel_ms()
    plan.set_profiling(False)
    kf = np.median(kf, axis=0); ki = np.median(ki, axis=0)
    px = float(B) * R * C
    # algorithmic bytes per launch (SURVEY 8(d) per-unit figures x pixels), inverse mirrored:
    #   level-1 kernels              X 4 + LoLo1 4 + Yh[0] 12            = 20 B/px
    #   levels 1 + 2 in one launch   X 4 + Yh[0] 12 + Yh[1] 3 + LoLo2 1  = 20 B/px (LoLo1 stays in registers)
    fwd12, inv21 = plan.launches() if NL >= 2 else (False, False)
    # one transform at a time up to 4096^2 (18 M useful pixels) the library runs levels 2 + 1 of the inverse as a marching PAIR of
    # wavefronts (k_inv21p, march2d.hip: dtcwt_march_inv21) -- that is the launch the one-at-a-time event pairs below time
    nstrip58 = -(-C // 232)
    alone_pair = BIORT in ('near_sym_a', 'legall') and QSHIFT in ('qshift_a', 'qshift_06') and px * C / (nstrip58 * 232.0) <= 1.8e7
    inv21_name = ('k_inv21p (levels 2+1 inverse, one launch: the marching pair one transform at a time; k_inv21m with several in flight)'
                  if alone_pair else 'k_inv21m (levels 2+1 inverse, one launch)')
    cand = [('k_fwd12m (levels 1+2 forward, one launch)', kf[0], 20.0) if fwd12 else ('k_fwd1 (level-1 forward)', kf[0], 20.0),
            (inv21_name, ki[1], 20.0) if inv21 else ('k_inv1 (level-1 inverse)', ki[0], 20.0)]
    # The dominant kernel is the one that tops the COMMITTED rocprof trace of this command (profiles/traffic.json `dominant`: the
    # levels-2+1 inverse launch, 39-43 % of the kernel time with one and with four streams); the two candidates are within
    # 5 % of each other, and picking by this run's event pairs named the second kernel of the trace on some boxes.  The
    # other one is in fwd_kernel_ms / inv_kernel_ms.  No side file: the slower of the two by this run's event pairs.
    dom = None
    try:
        tj0 = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
        tj0 = tj0 if args.config == 'c2' else (tj0.get(args.config) or {})
        dom = tj0.get('dominant')
    except Exception:
        pass
    pick = [c for c in cand if dom and c[0].startswith(dom)]
    name, ms, bpp = pick[0] if pick else max(cand, key=lambda c: c[1])
    achieved = bpp * px / (ms * 1e-3) / 1e9       # GB/s
    roofline = {'bound': 'hbm', 'kernel': name, 'achieved': round(achieved, 1), 'peak': HBM_PEAK / 1e9,
                'unit': 'GB/s', 'frac': round(achieved * 1e9 / HBM_PEAK, 4),
                'frac_of_copy_ceiling': round(achieved * 1e9 / HBM_COPY, 4), 'copy_ceiling': HBM_COPY / 1e9, 'traffic': None,
                'kernel_ms': round(float(ms), 5), 'kernel_ms_is': 'median raw hipEvent pair, %d steps' % nprof,
                'rocprof_kernel_ms': None, 'algorithmic_bytes_per_launch': bpp * px,
                'fwd_kernel_ms': [round(float(x), 5) for x in kf],
                'inv_kernel_ms': [round(float(x), 5) for x in ki],
                'kernel_chosen_by': ('profiles/traffic.json: top kernel of the committed trace' if pick else 'slowest launch of this run'),
                'launches': {'fwd_levels_1_2_one_launch': bool(fwd12), 'inv_levels_2_1_one_launch': bool(inv21),
                             'note': 'a shared launch is timed under the level it starts with; the other level shows an empty event pair'},
                'sum_kernel_ms': round(float(kf.sum() + ki.sum()), 5), 'event_pair_overhead_ms': round(null_ms, 5),
                'step_frac': round(STEP_BYTES_PER_PX * px / (dt / args.steps) / HBM_PEAK, 4),
                'step_frac_of_copy_ceiling': round(STEP_BYTES_PER_PX * px / (dt / args.steps) / HBM_COPY, 4)}
    if in_flight is not None:
        share = 1.0 / nshares if partitioned else None
        kms4 = float(in_flight[0][0] if name.startswith('k_fwd') else in_flight[1][1 if inv21 else 0])
        roofline['in_flight'] = {
            'streams': nstreams, 'cu_share': share, 'kernel_ms': round(kms4, 5),
            'achieved': round(bpp * px / (kms4 * 1e-3) / 1e9, 1),
            'frac_of_share': None if share is None else round(bpp * px / (kms4 * 1e-3) / (HBM_PEAK * share), 4),
            'fwd_kernel_ms': [round(float(x), 5) for x in in_flight[0]], 'inv_kernel_ms': [round(float(x), 5) for x in in_flight[1]],
            'is': 'the same kernels as the timed region runs them: %d images in flight%s; `achieved` = algorithmic bytes / kernel_ms, '
                  '`frac_of_share` = achieved / (peak x cu_share) -- bandwidth is not partitioned, so this is an occupancy-normalised '
                  'figure, not a roofline fraction' % (nstreams, ', each stream on 1/%d of the compute units' % nstreams if partitioned else '')}
    tr = os.path.join(ROOT, 'profiles', 'traffic.json')
    default_shape = all(getattr(args, k) is None for k in ('rows', 'cols', 'batch', 'nlevels', 'biort', 'qshift'))
    if os.path.exists(tr) and default_shape and args.config in ('c2', 'c3', 'c5'):
        try:
            tj = json.load(open(tr))
            if args.config != 'c2':
                tj = tj.get(args.config) or {}
            short = name.split(' ')[0]
            roofline['traffic'] = tj.get(short)
            for key, src in (('rocprof_kernel_ms', 'rocprof_median_us_one_stream'), ('rocprof_kernel_ms_two_streams', 'rocprof_median_us')):
                v = (tj.get(src) or {}).get(short, None)
                roofline[key] = None if v is None else round(v / 1e3, 5)
            roofline['traffic_source'] = tj.get('source')
        except Exception:
            pass

    value = aggregate_value(world, px, args.steps, dt)
    out = {
        'metric': 'Mpixels/s 2D DT-CWT fwd+inv, 4096^2 f32 nlevels=4' if args.config == 'c2' else
                  'Mpixels/s 2D DT-CWT fwd+inv, %s' % cfg['name'],
        'value': round(value, 1), 'unit': 'Mpixels/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'settle_ms': args.settle_ms, 'launch': 'hipGraph' if graphs is not None else 'stream',
        'ms_per_step': round(dt / args.steps * 1e3, 5),
        'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s, %s/%s, %d image(s) per GPU per step' % (cfg['name'], BIORT, QSHIFT, B),
                   'sharding': 'independent images per GPU, no data-path collective',
                   'buffer_sets': nsets, 'bytes_per_set': set_bytes, 'streams': nstreams,
                   'cu_partition': ('%d contexts, each on 1/%d of the compute units' % (nstreams, nshares)) if partitioned else None,
                   'ms_per_step_is': 'throughput over %d overlapped stream(s) of independent images' % nstreams},
        'sustained_ms_per_step': None if sustained_ms is None else round(sustained_ms, 5), 'sustained_steps': n_sus if sustained_ms is not None else None,
        'one_stream_ms_per_step': round(one_stream_ms, 5), 'resident_ms_per_step': round(resident_ms, 5),
        'roofline': roofline, 'recon_max_abs_err': err, 'device_under_load': under_load,
    }
    if ab is not None:
        out['ab_protocols'] = ab

    out['nccl_ranks'] = nccl_ranks
    out['rank_ms_per_step'] = {'min': round(min(rank_ms), 5), 'max': round(max(rank_ms), 5)}
    Xh0 = Zh0 = None
    if rank == 0 and not args.no_cpu_baseline:
        Xh0, Zh0 = first_image(sets[0][0], (R, C)), first_image(sets[0][3], plan.ext)
    # the collective library goes first: the other ranks must not sit in a RCCL barrier while rank 0 spends
    # ~13 s in the single-core CPU baseline
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()

    # ---- the other single-GPU BASELINE configs, short, as sub-runs of this script (c2 at N = 1 only) ----
    if rank == 0 and world == 1 and args.config == 'c2' and not args.no_other_configs and not args.mgpu:
        sets.clear()            # hand this run's 1.6 GB of buffers back before the sub-runs allocate theirs
        out['other_configs'] = other_configs()

    # ---- what the device gives a trivial streaming program with the step's bytes and launches (c2 at N = 1, default shape) ----
    if rank == 0 and world == 1 and args.config == 'c2' and default_shape and not args.no_probe and not args.mgpu:
        sets.clear()
        out['streaming_probe'] = streaming_probe(out['ms_per_step'], out.get('sustained_ms_per_step'), out['one_stream_ms_per_step'], partitioned, nstreams)

    # ---- CPU baseline: the oracle (a NumPy port of the reference's algorithm), rank 0 ----
    if rank == 0 and not args.no_cpu_baseline:
        out['cpu_baseline'], zc = cpu_baseline(cfg, Xh0, args.cpu_baseline_s)
        # the timed GPU output against the CPU port on the same input
        out['gpu_vs_cpu_recon_max_abs_diff'] = float(np.abs(Zh0 - zc).max())
    elif rank == 0:
        out['cpu_baseline'] = None
    if rank == 0:
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        if saved_stdout is not None:
            os.dup2(2, 1)           # what the collective library still has in its stdio buffer (its banner) goes to stderr


def clocks_under_load(step, drain, s_per_step):
    """rocm-smi --showclocks --showpower while the bench's own steps keep the device busy (it takes a few hundred ms to answer)."""
    import re
    import shutil
    exe = shutil.which('rocm-smi') or ('/opt/rocm/bin/rocm-smi' if os.path.exists('/opt/rocm/bin/rocm-smi') else None)
    if exe is None:
        return None
    try:
        pr = subprocess.Popen([exe, '--showclocks', '--showpower'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        t_end = time.perf_counter() + 4.0
        burst = max(8, int(0.02 / max(s_per_step, 1e-6)))
        while pr.poll() is None and time.perf_counter() < t_end:
            for _ in range(burst):
                step()
            drain()
        if pr.poll() is None:
            pr.kill()
            return None
        txt = pr.stdout.read().decode(errors='replace')
        sclk = re.search(r'sclk clock level:[^(]*\((\d+)Mhz\)', txt)
        pw = re.search(r'Power \(W\):\s*([0-9.]+)', txt)
        return {'sclk_mhz': int(sclk.group(1)) if sclk else None, 'socket_power_w': float(pw.group(1)) if pw else None,
                'is': 'rocm-smi --showclocks --showpower of GPU 0 while this configuration keeps running'}
    except Exception as exc:
        return {'error': repr(exc)[:200]}


def streaming_probe(ms20, ms_sus, ms_one, partitioned, nstreams):
    """tools/kbench/step_probe (built by __graft_entry__.build()): per step and stream six float4 streaming kernels that move the
    algorithmic bytes of a 4096^2 nlevels=4 forward + inverse (20 B/px per direction + the LoLo2 / LoLo3 round trips of the
    per-level tail) on eight rotating buffer sets, in the three launch protocols.  Not a roofline: the rate a program with NO
    arithmetic, no halo and no band warm-up reaches on this box in this run -- what is left between the transform and it is
    what the kernels themselves cost."""
    exe = os.path.join(ROOT, 'tools', 'kbench', 'step_probe')
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, '--json'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=60)
        pj = json.loads(r.stdout.decode().strip().splitlines()[-1])
    except Exception as exc:
        return {'error': repr(exc)[:200]}
    key = 'four_streams_on_quarters' if (partitioned and nstreams == 4) else ('four_plain_streams' if nstreams == 4 else 'one_stream')
    pj['is'] = ('tools/kbench/step_probe.hip in this run: ms per step of a trivial float4 streaming program with the bytes and the six '
                'launches of a step, 20 / 200 steps between device syncs, eight 403 MB buffer sets; transform_over_probe = this '
                'line\'s figure / the probe\'s in the same protocol')
    pj['transform_over_probe'] = {
        'ms_per_step_%d_streams' % nstreams: round(ms20 / pj[key]['ms_per_step_20'], 4),
        'sustained': None if ms_sus is None else round(ms_sus / pj[key]['ms_per_step_200'], 4),
        'one_stream': round(ms_one / pj['one_stream']['ms_per_step_200'], 4)}
    pj['probe_frac_of_8TBs'] = {k: round(STEP_BYTES_PER_PX * 4096 * 4096 / (pj[k]['ms_per_step_200'] * 1e-3) / HBM_PEAK, 4)
                                for k in ('one_stream', 'four_plain_streams', 'four_streams_on_quarters')}
    return pj


def other_configs():
    """Short runs of the other single-GPU BASELINE configs under the same protocol (rotating buffer sets, settle phase,
    device-synchronised timed region), each as `python bench.py --config X --no-cpu-baseline`: c3 (64 x 1024^2 nl=5), c5
    (one GPU's share of the 512-image batch: 64 x 2048^2 nl=4) and c4 (3-D 256^3 nl=3).  The reference's own benchmark
    script times every case it names in one run (scripts/benchmark_opencl.py:57-100)."""
    res = {}
    # c4_qbgn: BASELINE configs[3] reads "qbgn-style" -- the reference's own 3-D vectors (tests/test_againstmatlab.py:115-124) use
    # near_sym_b / qshift_b: 13 / 19-tap level-1 filters, two launches per direction around four plane volumes (fused3d_long.hpp)
    # SURVEY 8(d) asks for the CPU path beside C3 / C4 / C5 as well: a bounded sample each (the oracle on one image repeated for ~3 s;
    # one 256^3 volume, ~15 s; c4_qbgn: the leading 192^3 corner of its volume, ~3-20 s by host -- the long filters take ~40 s at 256^3)
    for name, extra in (('c3', ['--steps', '40', '--ab-protocols', '3', '--cpu-baseline-s', '3']),
                        ('c5_share', ['--config', 'c5', '--steps', '20', '--ab-protocols', '3', '--cpu-baseline-s', '3']),
                        ('c4', ['--steps', '40']),
                        ('c4_qbgn', ['--steps', '40', '--biort', 'near_sym_b', '--qshift', 'qshift_b', '--cpu-sample-n', '192'])):
        cfgname = 'c5' if name == 'c5_share' else ('c4' if name == 'c4_qbgn' else name)
        cmd = [sys.executable, os.path.abspath(__file__), '--config', cfgname, '--no-other-configs',
               '--warmup', '5', '--settle-ms', '150'] + [e for e in extra if e not in ('--config', 'c5')]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=180)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
            d = json.loads(line)
            keep = {'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'], 'steps': d['steps'],
                    'workload': d['config']['workload'], 'step_frac': d['roofline'].get('step_frac'),
                    'wall_s': round(time.perf_counter() - t0, 1)}
            if cfgname == 'c4':
                keep.update({k: d.get(k) for k in ('fwd_ms_per_step', 'inv_ms_per_step', 'ms_per_step_one_stream')})
                keep['streams'] = d['config'].get('streams')
                keep['k_fwd3_l1_frac'] = d['roofline'].get('frac')
                keep['step_frac'] = d.get('step_frac')
            else:
                keep['one_stream_ms_per_step'] = d.get('one_stream_ms_per_step')
                keep['sustained_ms_per_step'] = d.get('sustained_ms_per_step')
                keep['launches'] = d['roofline'].get('launches')
            for k in ('cpu_baseline', 'ab_protocols', 'device_under_load'):
                if d.get(k) is not None:
                    keep[k] = d[k]
            keep['roofline'] = {k: d['roofline'].get(k) for k in ('kernel', 'kernel_ms', 'rocprof_kernel_ms', 'frac', 'frac_of_copy_ceiling',
                                                                  'algorithmic_bytes_per_launch', 'traffic', 'traffic_source')}
            res[name] = keep
        except Exception as exc:            # a failed sub-run must not cost the headline line
            res[name] = {'error': '%s: %s' % (type(exc).__name__, exc)}
    return res


def main_c4(args):
    """BASELINE configs[3]: 3-D forward + inverse of ONE 256^3 float32 volume, nlevels=3 (near_sym_a / qshift_a).  A volume
    is not sharded (replicas only), so this mode runs at N = 1."""
    if args.gpus != 1 or int(os.environ.get('WORLD_SIZE', '1')) != 1:
        raise SystemExit('bench.py --config c4: one volume does not shard; run it with --gpus 1')
    import ctypes
    from dtcwt_amd.coeffs import biort, qshift
    from dtcwt_amd.hip import Context, DeviceArray, Transform3d, _lib
    from dtcwt_amd.utils import flat_taps
    n = args.rows or 256
    nl = 3
    ctx = Context(0)
    # independent volumes in flight on separate streams, as the default configuration does with images: the coarse levels
    # (a few hundred workgroups) of one volume run beside the level-1 launches of another
    nstreams = max(1, args.streams)
    nsets = max(nstreams, args.sets)
    if nsets % nstreams:
        nsets = (nsets // nstreams + 1) * nstreams
    # each stream on its own share of the compute units: slower for the short filters (forward 0.200 -> 0.211 ms), 3.5 % faster per forward +
    # inverse for near_sym_b / qshift_b (0.626 -> 0.604, all of it the inverse; the forward alone level): profiles/r05/ab_c4_partition.txt
    partitioned = nstreams in (2, 4) and (args.cu_partition == 'on' or (args.cu_partition == 'auto' and BIORT == 'near_sym_b'))
    ctxs = [Context(0, partition=(s, nstreams)) for s in range(nstreams)] if partitioned else [ctx] + [Context(0) for _ in range(nstreams - 1)]
    t3s = [Transform3d(BIORT, QSHIFT, ctx=c) for c in ctxs]
    rs = np.random.RandomState(4)
    vols = [ctxs[k % nstreams].to_device(rs.standard_normal((n, n, n)).astype(np.float32)) for k in range(nsets)]
    t3 = Transform3d(BIORT, QSHIFT, ctx=ctx) if partitioned else t3s[0]      # the whole device, one volume at a time
    state = {}

    def step(k):
        s = k % nsets
        t = t3s[s % nstreams]
        p = t.forward(vols[s], nlevels=nl)
        state['p', s % nstreams] = p
        state['z', s % nstreams] = t.inverse(p, device_output=True)

    t_end = time.perf_counter() + args.settle_ms / 1e3
    k = 0
    while time.perf_counter() < t_end:
        step(k); k += 1
        if k % 8 == 0:
            ctx.device_sync()
    for w in range(args.warmup):
        step(w)
    ctx.device_sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    ctx.device_sync()
    dt = time.perf_counter() - t0
    ls = (args.steps - 1) % nsets
    err = float(np.abs(state['z', ls % nstreams].get() - vols[ls].get()).max())
    # forward-only and inverse-only times of the same protocol
    ctx.device_sync(); t1 = time.perf_counter()
    for k in range(args.steps):
        s = k % nsets
        state['p', s % nstreams] = t3s[s % nstreams].forward(vols[s], nlevels=nl)
    ctx.device_sync(); dt_f = time.perf_counter() - t1
    for q in range(nstreams):
        if ('p', q) not in state:
            state['p', q] = t3s[q].forward(vols[q], nlevels=nl)
    ctx.device_sync(); t1 = time.perf_counter()
    for k in range(args.steps):
        q = k % nstreams
        state['z', q] = t3s[q].inverse(state['p', q], device_output=True)
    ctx.device_sync(); dt_i = time.perf_counter() - t1
    # one volume at a time on one stream: the latency of a single forward + inverse
    ctx.device_sync(); t1 = time.perf_counter()
    own = [s for s in range(nsets) if s % nstreams == 0]
    for k in range(args.steps):
        p1 = t3.forward(vols[own[k % len(own)]], nlevels=nl)
        z1 = t3.inverse(p1, device_output=True)
    ctx.device_sync(); dt_1 = time.perf_counter() - t1
    del p1, z1

    # the dominant kernel, k_fwd3_l1, alone: raw event pair around dtcwt_hip_fwd3_level1 (one launch), median
    h0, h1 = flat_taps(biort(BIORT)[0]), flat_taps(biort(BIORT)[2])
    pd = ctypes.POINTER(ctypes.c_double)
    LLL = DeviceArray(ctx, (n, n, n), np.float32)
    Yh0 = DeviceArray(ctx, (n // 2, n // 2, n // 2, 28), np.complex64)
    e0, e1 = ctx.event(), ctx.event()
    ks, empty = [], []
    fused_l1 = True
    for r in range(24):
        X = vols[r % len(vols)]
        e0.record()
        if fused_l1:
            rc = _lib.lib().dtcwt_hip_fwd3_level1(ctx.handle, X.ptr, n, n, n, h0.ctypes.data_as(pd), h0.shape[0],
                                                  h1.ctypes.data_as(pd), h1.shape[0], LLL.ptr, Yh0.ptr)
            if rc == -3 and r == 0:
                fused_l1 = False        # a wavelet set without a one-launch level 1: time the level as Transform3d runs it
            else:
                _lib.check(rc)
        if not fused_l1:
            pl1 = t3.forward(X, nlevels=1)
        e1.record()
        ctx.device_sync()
        ks.append(e0.elapsed_ms(e1))
        e0.record(); e1.record(); ctx.device_sync()
        empty.append(e0.elapsed_ms(e1))
    kms = float(np.median(ks[4:]))
    vox = float(n) ** 3
    ms = dt / args.steps * 1e3
    out = {
        'metric': 'Mvoxels/s 3D DT-CWT fwd+inv, %d^3 f32, nlevels=%d, %s/%s, resident in HBM' % (n, nl, BIORT, QSHIFT),
        'value': round(vox * args.steps / dt / 1e6, 1), 'unit': 'Mvoxels/s', 'n_gpus': 1, 'steps': args.steps,
        'warmup': args.warmup, 'settle_ms': args.settle_ms, 'ms_per_step': round(ms, 5), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '3D forward+inverse %dx%dx%d f32, nlevels=%d, %s/%s, one volume per step' % (n, n, n, nl, BIORT, QSHIFT),
                   'sharding': 'replicas only: a volume is not sharded', 'buffer_sets': len(vols), 'streams': nstreams,
                   'cu_partition': ('%d contexts, each on 1/%d of the compute units' % (nstreams, nstreams)) if partitioned else None,
                   'ms_per_step_is': 'throughput over %d overlapped stream(s) of independent volumes' % nstreams},
        'ms_per_step_one_stream': round(dt_1 / args.steps * 1e3, 5),
        'fwd_ms_per_step': round(dt_f / args.steps * 1e3, 5), 'inv_ms_per_step': round(dt_i / args.steps * 1e3, 5),
        'step_frac': round(72.0 * vox / (ms * 1e-3) / HBM_PEAK, 4),
        'roofline': {'bound': 'hbm', 'kernel': ('k_fwd3m_l1 (level-1 forward, one launch: marching pairs of wavefronts; the tile program k_fwd3_l1 where it does not apply)' if BIORT != 'near_sym_b'
                                                 else 'colfilter2 along axis 0 (k_g2_fwd_p1) + k_fwd3l_slices (level-1 forward for the 13 / 19-tap filters: two launches around two axis-0 volumes, 52 B/voxel moved; until round 5 k_fwd1m<PLANES> + k_fwd3l_axis0 around four plane volumes, 68 B/voxel)') if fused_l1 else 'level-1 forward as Transform3d runs it for this wavelet set (several launches)',
                     'achieved': round(36.0 * vox / (kms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                     'frac': round(36.0 * vox / (kms * 1e-3) / HBM_PEAK, 4), 'traffic': None,
                     'kernel_ms': round(kms, 5), 'kernel_ms_is': 'median raw hipEvent pair around dtcwt_hip_fwd3_level1, 20 launches',
                     'event_pair_overhead_ms': round(float(np.median(empty)), 5),
                     'frac_of_copy_ceiling': round(36.0 * vox / (kms * 1e-3) / HBM_COPY, 4), 'copy_ceiling': HBM_COPY / 1e9,
                     'rocprof_kernel_ms': None, 'algorithmic_bytes_per_launch': 36.0 * vox},
        'recon_max_abs_err': err,
    }
    tr = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.exists(tr) and n == 256 and fused_l1 and (BIORT, QSHIFT) == ('near_sym_b', 'qshift_b'):
        try:            # the two launches of the level together
            tj = json.load(open(tr)).get('c4_qbgn') or {}
            out['roofline']['traffic'] = tj.get('level1_forward')
            v = (tj.get('rocprof_median_us_one_stream') or {}).get('level1_forward')
            out['roofline']['rocprof_kernel_ms'] = None if v is None else round(v / 1e3, 5)
            out['roofline']['traffic_source'] = tj.get('source')
        except Exception:
            pass
    if os.path.exists(tr) and n == 256 and fused_l1 and (BIORT, QSHIFT) == ('near_sym_a', 'qshift_a'):
        try:
            tj = json.load(open(tr)).get('c4') or {}
            out['roofline']['traffic'] = tj.get('k_fwd3_l1')
            v = (tj.get('rocprof_median_us_one_stream') or {}).get('k_fwd3_l1')
            out['roofline']['rocprof_kernel_ms'] = None if v is None else round(v / 1e3, 5)
            out['roofline']['traffic_source'] = tj.get('source')
        except Exception:
            pass
    if not args.no_cpu_baseline:
        # the NumPy oracle (one core) on the first volume: ~15 s at 256^3
        from oracle import dtcwt_oracle as o
        m = min(n, args.cpu_sample_n) if args.cpu_sample_n > 0 else n
        Xh = np.ascontiguousarray(vols[0].get()[:m, :m, :m].astype(np.float32))
        to = o.Transform3d(biort(BIORT), qshift(QSHIFT))
        c0 = time.perf_counter()
        po = to.forward(Xh, nlevels=nl)
        zo = to.inverse(po)
        cdt = time.perf_counter() - c0
        out['cpu_baseline'] = {'value': round(m ** 3 / cdt / 1e6, 3), 'unit': 'Mvoxels/s', 'cores': 1, 'kind': 'port',
                               'host_cpus': os.cpu_count(), 'numpy': np.__version__,
                               'sample': 'one %d^3 f32 volume%s fwd+inv nlevels=%d (%.1f s)' % (m, '' if m == n else ' (the leading corner of the %d^3 one)' % n, nl, cdt)}
        zg = np.asarray(t3.inverse(t3.forward(Xh, nlevels=nl)))
        out['gpu_vs_cpu_recon_max_abs_diff'] = float(np.abs(zg - zo).max())
    print(json.dumps(out))


def main_mgpu(args):
    """N devices from one process: dtcwt_hip_mgpu_* (a host thread per device, taps broadcast with RCCL)."""
    from dtcwt_amd.coeffs import biort, qshift
    from dtcwt_amd.hip import _lib
    from dtcwt_amd.hip.multigpu import MultiGPUTransform2d
    cfg = dict(CONFIGS[args.config])
    for k in ('rows', 'cols', 'batch'):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)
    ndev = _lib.device_count()
    if ndev < args.gpus:
        raise SystemExit('bench.py --mgpu --gpus %d: only %d HIP device(s) visible' % (args.gpus, ndev))
    N = args.gpus
    B, R, C, NL = cfg['batch'], cfg['rows'], cfg['cols'], cfg['nlevels']
    # RCCL (loaded by the library for the tap broadcast) prints its banner on stdout, some of it at exit: stdout is
    # kept for the one JSON line of the contract, everything else goes to stderr
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    # the engine of the one-process-per-GPU path: `--streams` batches in flight per device, each lane's contexts on a share of
    # the compute units where that measured faster (the library's rule; --cu-partition on / off forces it)
    lanes = max(1, cfg.get('streams', args.streams))
    m = MultiGPUTransform2d(biort(BIORT), qshift(QSHIFT), devices=list(range(N)), batch=N * B, rows=R, cols=C,
                            nlevels=NL, broadcast_taps=True, lanes=lanes,
                            partition=None if args.cu_partition == 'auto' else args.cu_partition == 'on')
    nsets = max(1, cfg.get('sets', args.sets))
    if nsets % lanes:
        nsets = (nsets // lanes + 1) * lanes
    sets = []
    for k in range(nsets):
        rs = np.random.RandomState(cfg['seed'](0) + 17 * k)
        bufs = m.alloc(lane=k % lanes)
        m.scatter(rs.standard_normal((N * B, R, C)).astype(np.float32), bufs.X)
        sets.append(bufs)
    counter = [0]

    def step():
        s = sets[counter[0] % nsets]
        counter[0] += 1
        m.forward_into(s)
        m.inverse_into(s)

    if args.settle_ms > 0:
        t_settle = time.perf_counter()
        for _ in range(20):
            step()
        m.sync()
        per_step = max((time.perf_counter() - t_settle) / 20, 1e-5)
        for _ in range(int(args.settle_ms * 1e-3 / per_step) + 1):
            step()
    for _ in range(args.warmup):
        step()
    m.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    m.sync()
    dt = time.perf_counter() - t0
    px = float(N * B) * R * C
    Zh, Xh = m.gather(sets[0].Z, (R + (R & 1), C + (C & 1)), np.float32), m.gather(sets[0].X, (R, C), np.float32)
    err = float(np.abs(Zh[:, :R, :C] - Xh).max())          # every image of the first buffer set, whole
    out = {
        'metric': 'Mpixels/s 2D DT-CWT fwd+inv, 4096^2 f32 nlevels=4' if args.config == 'c2' else
                  'Mpixels/s 2D DT-CWT fwd+inv, %s' % cfg['name'],
        'value': round(px * args.steps / dt / 1e6, 1), 'unit': 'Mpixels/s', 'n_gpus': N, 'steps': args.steps,
        'warmup': args.warmup, 'settle_ms': args.settle_ms, 'launch': 'mgpu (one process, a host thread per device)',
        'ms_per_step': round(dt / args.steps * 1e3, 5), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s, %s/%s, %d image(s) per GPU per step' % (cfg['name'], BIORT, QSHIFT, B),
                   'sharding': 'contiguous batch split over %d device(s), no data-path collective' % N,
                   'buffer_sets': nsets, 'taps_broadcast_with_rccl': bool(m.taps_broadcast), 'streams': lanes,
                   'cu_partition': ('%d lanes per device, each on 1/%d of the compute units' % (lanes, m.shares)) if m.shares > 1 else None,
                   'ms_per_step_is': 'throughput over %d overlapped lane(s) of independent batches per device' % lanes},
        'roofline': {'bound': 'hbm', 'kernel': 'whole step', 'achieved': round(STEP_BYTES_PER_PX * px / dt * args.steps / N / 1e9, 1),
                     'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                     'frac': round(STEP_BYTES_PER_PX * px / N / (dt / args.steps) / HBM_PEAK, 4), 'traffic': None},
        'recon_max_abs_err': err,
    }
    if not args.no_cpu_baseline:
        out['cpu_baseline'], _ = cpu_baseline(cfg, Xh[0])
    else:
        out['cpu_baseline'] = None
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)


if __name__ == '__main__':
    main()
