"""Sweep cidnet_pw_conv launch knobs over the shapes of one CIDNet step (dev tool): what does the dispatcher leave on the table?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from micro_pw import run

# (M, K, HW, calls per step) of the 400x600 bs=8 step (from bench.py --op-table)
SHAPES = [(72, 382, 15000, 3), (36, 190, 60000, 4), (144, 766, 3750, 4), (72, 72, 60000, 4), (190, 36, 60000, 4), (36, 95, 60000, 4),
          (766, 144, 3750, 4), (36, 36, 60000, 22), (288, 288, 3750, 4), (72, 36, 60000, 4), (382, 72, 15000, 3), (72, 191, 15000, 3),
          (144, 383, 3750, 4), (95, 36, 60000, 4), (144, 144, 15000, 3), (383, 144, 3750, 4), (144, 144, 3750, 16), (36, 72, 60000, 4),
          (36, 36, 240000, 2), (72, 144, 15000, 3), (144, 288, 3750, 4), (288, 144, 3750, 4), (72, 72, 15000, 14), (191, 72, 15000, 3),
          (144, 72, 15000, 3), (72, 72, 3750, 4), (36, 36, 15000, 4)]
tot_def = tot_best = 0.0
for M, K, HW, n in SHAPES:
    sh = (8, M, K, HW)
    d = run(*sh, flags=0)[0]
    res = {"default": d}
    for tb in (256, 512, 1024):
        res[f"tb{tb}"] = run(*sh, flags=(tb << 8))[0]
        res[f"lds,tb{tb}"] = run(*sh, flags=4 | (tb << 8))[0]
        for mt in (1, 2, 3, 4):
            res[f"mt{mt},tb{tb}"] = run(*sh, flags=(tb << 8) | (mt << 28))[0]
    res["nosplitk"] = run(*sh, flags=16)[0]
    best = min(res, key=res.get)
    tot_def += d * n; tot_best += res[best] * n
    print(f"M={M:4d} K={K:4d} HW={HW:6d} x{n:2d}: default {d:6.1f} us  best {res[best]:6.1f} ({best})  " +
          " ".join(f"{k}:{v:.0f}" for k, v in sorted(res.items(), key=lambda kv: kv[1])[:5]), flush=True)
print(f"per step: default {tot_def / 1e3:.2f} ms, best-of-sweep {tot_best / 1e3:.2f} ms")
