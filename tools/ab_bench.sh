#!/bin/bash
# A/B two builds of the library on the same box: tools/ab_bench.sh <libA.so> <libB.so> [rounds]   (paths relative to the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${3:-3}
for i in $(seq 1 $N); do
  for L in "$1" "$2"; do
    CIDNET_LIB_PATH=$R/$L python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference-leg 2>/dev/null \
      | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done
