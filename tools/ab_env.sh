#!/bin/bash
# A/B environment settings on the same box: tools/ab_env.sh "<ENV A>" "<ENV B>" ... (each argument is a list of VAR=VALUE words)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2; do
  for E in "$@"; do
    env $E python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference-leg 2>/dev/null \
      | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$E]', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done
