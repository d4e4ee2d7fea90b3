#!/usr/bin/env python3
"""List every GPU kernel of one training step that is NOT ours (not `cidnet::`), with the ATen op, its input shapes and
the Python frames that launched it.  VERDICT r3 item 5: the step should launch only cidnet:: code (+ RCCL at N > 1).

    python tools/foreign_kernels_in_step.py [--single-stream]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch


def foreign_kernels(trainer, x, gt, steps=1):
    """{kernel name: [(aten op, shapes, stack), ...]} over `steps` steps of an already warmed-up trainer"""
    from torch.profiler import ProfilerActivity, profile
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        for _ in range(steps):
            trainer.step(x, gt)
        torch.cuda.synchronize()
    out = {}
    for ev in prof.events():
        for k in getattr(ev, "kernels", []) or []:
            name = k.name
            if "cidnet::" in name:
                continue
            out.setdefault(name, []).append((ev.name, str(getattr(ev, "input_shapes", "")), [s for s in (ev.stack or [])][:8]))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--single-stream", action="store_true")
    a = ap.parse_args()
    import hvi_cidnet_amd as P
    from hvi_cidnet_amd.dp import DataParallelTrainer
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = P.CIDNet().to(dev)
    model.two_streams = not a.single_stream
    tr = DataParallelTrainer(model, wgrad_stream=not a.single_stream)
    x = torch.rand((8, 3, 400, 600), device=dev)
    gt = torch.rand((8, 3, 400, 600), device=dev)
    for _ in range(3):
        tr.step(x, gt)
    fk = foreign_kernels(tr, x, gt)
    n = sum(len(v) for v in fk.values())
    print(f"{n} launches of {len(fk)} foreign kernels in one step")
    for name, uses in sorted(fk.items(), key=lambda kv: -len(kv[1])):
        print(f"\n== {len(uses)} x {name[:150]}")
        seen = {}
        for op, shapes, stack in uses:
            key = (op, shapes, tuple(s for s in stack if "hvi-cidnet_amd" in s or "hvi_cidnet_amd" in s or "bench" in s)[:4])
            seen[key] = seen.get(key, 0) + 1
        for (op, shapes, stack), c in sorted(seen.items(), key=lambda kv: -kv[1]):
            print(f"   {c:3d} x {op} {shapes[:120]}")
            for s in stack:
                print(f"          {s}")
