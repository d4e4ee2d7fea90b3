#!/bin/bash
# alternate two bench.py configurations on ONE box (box-to-box spread is +-3 %): tools/ab_bench.sh OUTDIR "flags A" "flags B" [rounds]
out=$1; fa=$2; fb=$3; n=${4:-2}
mkdir -p $out
for r in $(seq 1 $n); do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference-leg --no-variants $fa 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('A', d['value'], d['ms_per_step'])" | tee -a $out/ab.txt || exit 1
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference-leg --no-variants $fb 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B', d['value'], d['ms_per_step'])" | tee -a $out/ab.txt || exit 1
done
