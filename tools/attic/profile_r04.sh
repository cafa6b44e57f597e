#!/bin/bash
# round-4 profile collection (GPU box): the driver's command under rocprofv3 (kernel trace, HBM-traffic counters), the
# marching kernels' knock-out bench and counters, the other configs, A/B runs
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04prof
mkdir -p $O
bash $R/tools/profile_round.sh gpurun_out/r04prof r04 > $O/profile_round.log 2>&1
cd $R/tools/kbench && timeout 300 ./march_bench_noslp 4096 60 > $O/march_bench.txt 2>&1
timeout 60 ./vmcnt_probe > $O/vmcnt_probe.txt 2>&1
cd $R
for c in c3 c5 c4; do python bench.py --config $c --no-cpu-baseline --steps 40 > $O/bench_$c.json 2>/dev/null; done
python bench.py --config c5full --no-cpu-baseline --steps 5 > $O/bench_c5full.json 2>/dev/null
DTCWT_HIP_MARCH=0 python bench.py --cu-partition off --no-cpu-baseline --no-other-configs --steps 100 > $O/bench_march_off.json 2>/dev/null
DTCWT_HIP_MARCH=0 python bench.py --cu-partition off --no-cpu-baseline --no-other-configs --steps 100 --streams 2 --sets 4 > $O/bench_march_off_streams2.json 2>/dev/null
python bench.py --no-cpu-baseline --no-other-configs --steps 500 > $O/bench_500steps.json 2>/dev/null
timeout 200 python tools/soak_march.py 100 7 > $O/soak_march.txt 2>&1
cat $O/status.txt
