#!/bin/bash
# Round-end measurements on the GPU box: bench line (with CPU baseline), rocprofv3 kernel stats (two-stream default
# and single-stream), PMC traffic passes.  Usage: bash tools/profile_round.sh <tag>   (outputs under gpurun_out/<tag>/)
set -e
TAG=${1:-r02_a}
EXTRA=${2:-}          # extra bench.py flags for every run, e.g. "--dtype bf16"
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 10 --warmup 3 $EXTRA > $O/bench.json 2> $O/bench.err
echo "bench done"; tail -c 600 $O/bench.json
rocprofv3 --kernel-trace --stats -d $O/stats2 -o b --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-inference-leg $EXTRA > $O/stats2.log 2>&1
echo "stats (2 streams) done"
rocprofv3 --kernel-trace --stats -d $O/stats1 -o b --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-inference-leg --single-stream $EXTRA > $O/stats1.log 2>&1
echo "stats (1 stream) done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o b --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-inference-leg --single-stream $EXTRA > $O/pmc_fetch.log 2>&1
echo "pmc fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o b --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-inference-leg --single-stream $EXTRA > $O/pmc_write.log 2>&1
echo "pmc write done"
python3 $R/tools/pmc_family_traffic.py $O/pmc_fetch/b_counter_collection.csv $O/pmc_write/b_counter_collection.csv $O/pmc_traffic_by_family.json
# keep the merged output small: drop the raw traces, keep the stats tables
rm -f $O/stats*/b_kernel_trace.csv $O/pmc_*/b_kernel_trace.csv
ls -la $O $O/stats2 | head -30
