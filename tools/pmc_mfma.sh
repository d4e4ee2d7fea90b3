#!/bin/bash
# MFMA-busy counters of the step's kernels, calibrated against a pure-MFMA loop (tools/bin/mfma_peak).
# Usage (on the GPU box): bash tools/pmc_mfma.sh <tag>   -> gpurun_out/<tag>/pmc_mfma_summary.json
set -e
TAG=${1:-r01_i}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_all.txt 2>&1 || true
WANT="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32"
HAVE=""
for c in $WANT; do if grep -q -w "$c" $O/counters_all.txt; then HAVE="$HAVE $c"; fi; done
echo "counters:$HAVE"
grep -i -E "mfma" $O/counters_all.txt | cut -c1-160 | sort -u | head -40 > $O/counters_mfma.txt || true
rocprofv3 --kernel-trace --pmc $HAVE -d $O/cal -o b --output-format csv -- $R/tools/bin/mfma_peak > $O/cal.log 2>&1
echo "calibration pass done"
rocprofv3 --kernel-trace --pmc $HAVE -d $O/step -o b --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --single-stream > $O/step.log 2>&1
echo "step pass done"
python3 $R/tools/pmc_mfma.py $O/cal $O/step $O/pmc_mfma_summary.json
rm -f $O/counters_all.txt
ls -la $O $O/step | head -20
