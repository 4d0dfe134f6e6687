#!/bin/bash
# rocprofv3 evidence for a round: kernel stats of the driver's bench command, then separate --pmc passes
# (FETCH_SIZE / WRITE_SIZE cannot share a pass) over the same 20-step build with its deNoise points. Run on the GPU box from
# the repo root: bash tools/profile_round.sh r03_a   (results under gpurun_out/<tag>/; copy what is to be judged into profiles/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r03_a}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $out/bench_under_rocprof.json 2> $out/kt.log || echo "kernel-trace run failed"
f=$(find $out/kt -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv
for P in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv --pmc $P -d $out/pm_$P -o p -- python3 bench.py --steps 20 --warmup 0 --no-cpu-baseline --no-secondary > $out/pm_$P.log 2>&1 || echo "pass $P failed"
done
python3 tools/pmc_summary.py $out/pm_FETCH_SIZE $out/pm_WRITE_SIZE > $out/pmc_traffic.txt
# (a bench run holds TWO 20-step builds of the same batches -- the overlapped one it times and the serial one it takes its
# per-kernel times from -- so the counters cover 40 steps)
python3 tools/pmc_to_json.py $out/pmc_traffic.txt 40 "profiles/${tag}_pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --steps 20 --warmup 0 = two 20-step builds, tools/profile_round.sh)" > $out/pmc_traffic.json
rm -rf $out/kt $out/pm_FETCH_SIZE $out/pm_WRITE_SIZE
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --serial --no-secondary --no-cpu-baseline > $out/bench_serial.json 2> $out/bench_serial.err
