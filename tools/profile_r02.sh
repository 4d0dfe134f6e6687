#!/bin/bash
# rocprofv3 evidence for the round: kernel stats of the driver's bench command, then separate --pmc passes
# (FETCH_SIZE / WRITE_SIZE cannot share a pass) over the same 20-step build (12 deNoise points). Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r02_x}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/kt.log || echo "kernel-trace run failed"
f=$(find $out/kt -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv
for P in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv --pmc $P -d $out/pm_$P -o p -- python3 bench.py --steps 20 --warmup 0 --no-cpu-baseline > $out/pm_$P.log 2>&1 || echo "pass $P failed"
done
python3 tools/pmc_summary.py $out/pm_FETCH_SIZE $out/pm_WRITE_SIZE > $out/pmc_traffic.txt
rm -rf $out/kt $out/pm_FETCH_SIZE $out/pm_WRITE_SIZE
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err
