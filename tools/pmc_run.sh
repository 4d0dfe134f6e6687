#!/bin/bash
# usage: pmc_run.sh OUTFILE "COUNTERS pass 1" "COUNTERS pass 2" ...   (one rocprofv3 --pmc pass per argument; bench.py 1 step)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$1; shift
i=0; dirs=""
for P in "$@"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $P -d gpurun_out/pm_$i -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pm_$i.log 2>&1 || echo "pass $i failed"
  dirs="$dirs gpurun_out/pm_$i"
done
python3 tools/pmc_summary.py $dirs > $out
rm -rf $dirs
