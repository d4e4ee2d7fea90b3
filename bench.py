#!/usr/bin/env python3
"""Headline benchmark: CIDNet training step (forward + L1 loss + backward + gradient all-reduce +
fused Adam) on synthetic 8x3x400x600 fp32 batches per GPU -- BASELINE.json config[1].

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` = images/s over all ranks (weak scaling: 8 images per rank).
`roofline` is measured live with HIP events (torch.cuda.Event on the launch stream) around every
launch of the dominant kernel family in extra, separately-run steps; `cpu_baseline` times the CPU
oracle (oracle/cidnet_oracle.py, a port of the reference's algorithm) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense fp32
PEAK_HBM_GBS = 8000.0
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA (the 2:1-sparsity figure is twice that)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--height", type=int, default=400)
    ap.add_argument("--width", type=int, default=600)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="storage type of the IEL chain's hidden tensors (arithmetic is fp32 either way); f32 is the parity mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inference-leg", action="store_true",
                    help="skip the legs after the timed region: the bf16-mode step (configs[2]), 32x3x1024x1024 inference (configs[3]), the variants (configs[4])")
    ap.add_argument("--no-prepared-weights", action="store_true", help="A/B: prepare the bf16x3 weight operands per launch (as a plain model(x) call does)")
    ap.add_argument("--no-variants", action="store_true", help="skip the MSSA / TNSM bs=16 measurement (configs[4])")
    ap.add_argument("--op-table", action="store_true", help="print per-entry-point time shares to stderr")
    ap.add_argument("--op-rows", type=int, default=70, help="rows of the per-shape part of --op-table")
    ap.add_argument("--no-wgrad-stream", action="store_true", help="keep the weight-gradient GEMMs on the branch streams")
    ap.add_argument("--graph", action="store_true", help="replay forward+backward as one hipGraph (experiment)")
    ap.add_argument("--full-loss", action="store_true",
                    help="train.py's objective without the VGG term (L1 + SSIM + Edge on RGB and HVI) instead of the metric's L1")
    ap.add_argument("--p-weight", type=float, default=0.0,
                    help="with --full-loss: weight of the VGG19 perceptual term (train.py's P_weight, 1e-2 there; random VGG weights here)")
    ap.add_argument("--single-stream", action="store_true",
                    help="run the I and HV branches on one stream (default: two streams, kernels of the two branches overlap)")
    ap.add_argument("--dual-norms", action="store_true", help="A/B: the two LayerNorms of an LCA input as one pass (CIDNet.dual_norms)")
    return ap.parse_args()


class OpTimer:
    """Times every C-ABI call with an event pair (diagnostic pass only, never the timed region)."""

    def __init__(self):
        self.rec = []

    def install(self):
        from hvi_cidnet_amd import _lib
        L = _lib.lib()
        self._orig = L.call
        timer = self

        def timed(name, *args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            timer._orig(name, *args)
            e1.record()
            timer.rec.append((name, args, e0, e1))
        L.call = timed
        return self

    def remove(self):
        from hvi_cidnet_amd import _lib
        _lib.lib().call = self._orig

    def table(self, by_shape=False):
        torch.cuda.synchronize()
        agg = {}
        for name, args, e0, e1 in self.rec:
            key = name
            if by_shape:
                ints = [int(v) for v in args if isinstance(v, int) and not isinstance(v, bool)]
                key = name + " " + ",".join(str(v) for v in ints[-6:])
            a = agg.setdefault(key, [0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1)
        return agg


def iel_stream_bytes(name, args):
    """algorithmic HBM bytes of one launch of the IEL chain's streaming kernels (every operand once), from the call's
    own arguments: element size follows the storage-type code `dt`"""
    ints = [v for v in args if isinstance(v, int) and not isinstance(v, bool)]
    if name == "cidnet_iel_dw_gate_fwd_t":          # (pin, wdw, w1, w2, u, g, dt, B, h, H, W): read 2h, write 2h (u) + h
        dt, B, h, H, W = ints[-5:]
        return (2 + (2 if args[4] is not None and getattr(args[4], "value", 1) else 0) + 1) * h * B * H * W * (2 if dt else 4)
    if name == "cidnet_iel_gate_dw_bwd_t":          # (u, w1, w2, dg, du, dt, gw1, gw2, ws, ws_floats, B, h, H, W): read 2h + h, write 2h
        dt = ints[0]
        B, h, H, W = ints[-4:]
        return 5 * h * B * H * W * (2 if dt else 4)
    if name == "cidnet_dw3x3_bwd_t":                # (in, gout, w1, w2, csplit, addend, gin, dt, ..., B, C, H, W): read 2C, write C
        dt = ints[1]
        B, C, H, W = ints[-4:]
        return 3 * C * B * H * W * (2 if dt else 4)
    return 0


def conv3x3_flops(args):
    # cidnet_conv3x3(X, x_bs, Wt, w_ms, w_ks, flip, replicate, Y, y_bs, B, M, K, H, W, stream)
    B, M, K, H, W = args[9], args[10], args[11], args[12], args[13]
    return 2.0 * 9 * M * K * H * W * B


def conv3x_shape(name, args):
    """(R, B, M, K, H, W) of a dense bf16x3 3x3 launch, either entry point:
    cidnet_conv3x3_bf16x3(X, x_bs, Wt, w_ms, w_ks, flip, R, r_bs, Y, y_bs, ws, ws_floats, B, M, K, H, W, stream)
    cidnet_conv3x3_bf16x3_pre(X, x_bs, Wprep, R, r_bs, Y, y_bs, B, M, K, H, W, stream)   (weights prepared once per step)"""
    if name in ("cidnet_conv3x3_bf16x3_pre", "cidnet_conv3x3_bf16x3_pre_lv"):
        return args[3], args[7], args[8], args[9], args[10], args[11]
    return args[6], args[12], args[13], args[14], args[15], args[16]


def conv3x_flops(sh):
    R, B, M, K, H, W = sh
    return 2.0 * 9 * M * K * H * W * B


def conv3x_bytes(sh):
    # X (K channels) once + Y (M channels) once (+ the addend R): the compulsory HBM bytes of one launch
    R, B, M, K, H, W = sh
    return (K + M + (M if R is not None else 0)) * 4.0 * H * W * B


def pw_work(name, args):
    """(algorithmic FLOPs, algorithmic HBM bytes: every operand once) of one launch of the 1x1-conv family, from the call's
    own arguments (include/cidnet_hip.h); None for other entry points"""
    es = lambda dt: 2 if dt else 4
    if name == "cidnet_pw_conv_t":        # (X, x_dt, x_bs, Wt, w_bs, w_ms, w_ks, Y, y_dt, y_bs, R, r_bs, B, M, K, HW, stream)
        x_dt, y_dt, R, B, M, K, HW = args[1], args[8], args[10], args[12], args[13], args[14], args[15]
        return 2.0 * M * K * HW * B, (K * es(x_dt) + M * es(y_dt) + (M * 4 if R is not None else 0)) * HW * B
    if name == "cidnet_pw_conv_bf16x3":   # (X, x_bs, Wt, w_bs, w_ms, w_ks, Y, y_bs, R, r_bs, ws, ws_floats, B, M, K, HW, stream)
        R, B, M, K, HW = args[8], args[12], args[13], args[14], args[15]
        return 2.0 * M * K * HW * B, (K + M + (M if R is not None else 0)) * 4 * HW * B
    if name in ("cidnet_pw_conv_bf16x3_pre", "cidnet_pw_conv_bf16x3_pre_lv"):   # (X, x_bs, Wprep, per_sample, Y, y_bs, R, r_bs, B, M, K, HW, [w_levels, x_levels,] stream)
        R, B, M, K, HW = args[6], args[8], args[9], args[10], args[11]
        return 2.0 * M * K * HW * B, (K + M + (M if R is not None else 0)) * 4 * HW * B
    if name == "cidnet_pw_conv_bf16x3_pre_t":   # (X, x_dt, x_bs, Wprep, per_sample, Y, y_dt, y_bs, R, r_bs, B, M, K, HW, w_levels, x_levels, stream)
        x_dt, y_dt, R, B, M, K, HW = args[1], args[6], args[8], args[10], args[11], args[12], args[13]
        return 2.0 * M * K * HW * B, (K * es(x_dt) + M * es(y_dt) + (M * 4 if R is not None else 0)) * HW * B
    if name == "cidnet_pw_conv_up_prelu":  # (skip, x_bs, Wt, w_ms, w_ks, Z, slope, Y, Ypre, B, M, K, zh, zw, stream)
        Ypre, B, M, K, zh, zw = args[8], args[9], args[10], args[11], args[12], args[13]
        hw = 4 * zh * zw
        return 2.0 * M * K * hw * B, ((K + M + (M if Ypre is not None else 0)) * hw + M * zh * zw) * 4 * B
    if name == "cidnet_pw_bwd_fused":     # (gY, gy_bs, X, x_bs, Wt, gX, gx_bs, dW, ws, ws_floats, B, M, N, HW, stream): both products
        B, M, N, HW = args[10], args[11], args[12], args[13]
        return 4.0 * M * N * HW * B, (M + 2 * N) * 4 * HW * B
    if name == "cidnet_pw_wgrad_t":       # (dY, dy_dt, dy_bs, X, x_dt, x_bs, dW, dw_ld, per_sample, acc, ws, ws_floats, B, M, N, HW, stream)
        dy_dt, x_dt, B, M, N, HW = args[1], args[4], args[12], args[13], args[14], args[15]
        return 2.0 * M * N * HW * B, (M * es(dy_dt) + N * es(x_dt)) * HW * B
    return None


def relaunch_ranks(a):
    """`python bench.py --gpus N` without a launcher: this parent process (which has not touched the GPU) starts
    N ranks through torch.distributed.run as a CHILD process, relays its output and exits with its code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.run(cmd, env=env).returncode)


def roofline_plan(B, H, W, channels=(36, 36, 72, 144), heads=(1, 2, 4, 8)):
    """Algorithmic work of one training step (forward + backward = 3x the forward) under the fused-kernel
    plan of SURVEY 2a: each fused kernel reads its inputs once and writes its outputs once.  Returns
    [(name, flops, bytes)] for the whole batch and step.  FLOPs are the reference's (the dead I_LCA5 included)."""
    c1, c2, c3, c4 = channels
    hw = [H * W, H * W // 4, H * W // 16, H * W // 64]
    k = []

    def add(name, flops, nbytes):
        k.append((name, 3.0 * B * flops, 3.0 * B * nbytes))
    add("hvit", 0.0, 24.0 * hw[0])
    add("stems (1->36, 3->36, replicate)", 2.0 * 9 * (1 + 3) * c1 * hw[0], 4.0 * (1 + 3 + 2 * c1) * hw[0])
    for lvl, (ci, co) in enumerate(((c1, c2), (c2, c3), (c3, c4))):
        for br in ("I", "HV"):
            add(f"down L{lvl} {br} (conv3x3+bilinear+prelu)", 2.0 * 9 * ci * co * hw[lvl], 4.0 * (ci * hw[lvl] + co * hw[lvl + 1]))
    lcas = [(c2, 1, heads[1])] * 2 + [(c3, 2, heads[2])] * 2 + [(c4, 3, heads[3])] * 4 + [(c3, 2, heads[2])] * 2 + [(c2, 1, heads[1])] * 2
    for i, (c, lvl, nh) in enumerate(lcas):
        px = hw[lvl]
        h = int(c * 2.66)
        cab = 2.0 * c * 3 * c * px + 2.0 * 9 * 3 * c * px + 2.0 * c * (c // nh) * px * 2 + 2.0 * c * c * px
        add(f"lca{i} CAB (LN+1x1+dw+gram | attn.v+proj+res)", cab, 4.0 * 5 * c * px)
        iel = 2.0 * c * 2 * h * px + 2.0 * 9 * 2 * h * px + 2.0 * 9 * 2 * h * px + 2.0 * h * c * px
        add(f"lca{i} IEL (LN+1x1+dw+gate+1x1)", iel, 4.0 * 2 * c * px)
    for lvl, (ci, co) in ((3, (c4, c3)), (2, (c3, c2)), (1, (c2, c1))):
        for br in ("I", "HV"):
            add(f"up L{lvl} {br} (conv3x3+bilinear+cat+1x1+prelu)", 2.0 * 9 * ci * co * hw[lvl] + 2.0 * 2 * co * co * hw[lvl - 1],
                4.0 * (ci * hw[lvl] + 2 * co * hw[lvl - 1]))
    add("heads (36->2, 36->1, replicate)", 2.0 * 9 * c1 * 3 * hw[0], 4.0 * (2 * c1 + 3) * hw[0])
    add("phvit + residual", 0.0, 36.0 * hw[0])
    return k


def whole_step_roofline(a, ms_per_step):
    plan = roofline_plan(a.batch, a.height, a.width)
    floor_s = sum(max(b / (PEAK_HBM_GBS * 1e9), f / (PEAK_F32_MFMA_TFLOPS * 1e12)) for _, f, b in plan)
    flops, nbytes = sum(f for _, f, _ in plan), sum(b for _, _, b in plan)
    return {"definition": "sum over the fused-kernel plan (SURVEY 2a/8d) of max(bytes/8 TB/s, flops/157.3 TFLOP/s) / measured step time",
            "floor_ms": round(1e3 * floor_s, 3), "measured_ms": round(ms_per_step, 3), "frac": round(1e3 * floor_s / ms_per_step, 4),
            "alg_tflop_per_step": round(flops / 1e12, 4), "alg_gb_per_step": round(nbytes / 1e9, 3),
            "achieved_tflops": round(flops / 1e12 / (ms_per_step * 1e-3), 2)}


def inference_1024(dev, world=1, local=0):
    """configs[3]: CIDNet inference on 32x3x1024x1024 (img/s) and the HBM rate of the HVIT / PHVIT kernels on that batch,
    measured after the timed region.  With N ranks every rank runs its own batch (weak scaling, no communication); the
    line reports the aggregate over the slowest rank's time next to rank 0's own numbers."""
    import hvi_cidnet_amd as P
    from hvi_cidnet_amd import ops
    B, H, W = 32, 1024, 1024
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    x = torch.rand((B, 3, H, W), device=dev, generator=g)
    k = torch.full([1], 0.2, device=dev)

    def timeit(fn, iters):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    px = B * H * W
    m = P.CIDNet().to(dev).eval()
    with torch.no_grad():
        hvi = ops.HVITFn.apply(x, k)
        ms_h = timeit(lambda: ops.HVITFn.apply(x, k), 10)
        ms_p = timeit(lambda: ops.PHVITFn.apply(hvi, None, None, 0.2, False, 1.3, False, 1.0), 10)
        if world > 1:
            dist.barrier(device_ids=[local])
        # frozen weights: the prepared (split) weight operands are kept across the forward calls (ops.prepared_weights)
        with ops.prepared_weights(True):
            # set-up, not warm-up: a 32-image 1024x1024 forward cycles ~40 GB through the caching allocator, which starts
            # empty here; run synchronised forwards until one adds no device segment (a forward that still has to hipMalloc
            # gigabytes was once timed at 4x its steady state)
            segs = lambda: torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
            quiet = 0
            for it in range(8):
                before = segs()
                t0 = time.perf_counter()
                m(x)
                torch.cuda.synchronize()
                if local == 0:
                    print(f"[bench] inference leg set-up forward {it}: {1e3 * (time.perf_counter() - t0):.1f} ms, {segs() - before} new device segments, "
                          f"{torch.cuda.memory_reserved(dev) / 2**30:.1f} GiB reserved", file=sys.stderr, flush=True)
                quiet = quiet + 1 if segs() == before else 0
                if quiet >= 2:
                    break
            # timed: every forward followed by a synchronisation, as a caller that consumes each batch's output runs it.  (Several
            # un-synchronised 40 GB forwards in flight make the allocator hipMalloc a second and third footprint in the middle of
            # the timed loop: 302 instead of 117 ms per batch were measured that way.)
            n_inf = 4
            t0 = time.perf_counter()
            for _ in range(n_inf):
                m(x)
                torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / n_inf
        ops.clear_prepared_weights()
    del m, x, hvi
    torch.cuda.empty_cache()
    ms_all = ms
    if world > 1:
        t = torch.tensor([ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms_all = t.item()
    prec = "fp32" if ops.MATH["levels"] == 3 else "bf16 mode"
    return {"workload": f"CIDNet inference 32x3x1024x1024 {prec} per GPU (BASELINE.json configs[3]), {world} rank(s), synchronised per batch",
            "images_per_s": round(world * B / (ms_all * 1e-3), 1), "n_gpus": world, "ms_per_batch": round(ms_all, 2),
            "rank0_images_per_s": round(B / (ms * 1e-3), 1), "hvit_GBs": round(24.0 * px / (ms_h * 1e-3) / 1e9, 1),
            "phvit_GBs": round(24.0 * px / (ms_p * 1e-3) / 1e9, 1), "hbm_peak_GBs": PEAK_HBM_GBS, "alg_bytes_per_px": 24}


def settle(trainer, x, gt, dev, world=1, max_steps=12):
    """Set-up, not warm-up: the trainer's first steps build its buckets / gradient arena and grow the caching allocator to
    the footprint of three queued steps (the host runs ahead of the GPU, see dp.DataParallelTrainer); a step that still
    has to hipMalloc costs 60-80 ms on the host -- and hundreds of ms once the queue is deep.  Run un-synchronised steps until
    three in a row add no device segment (N > 1: a fixed count, every step holds a collective)."""
    segs = lambda: torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    quiet = 0
    for _ in range(max_steps if world == 1 else 8):
        before = segs()
        trainer.step(x, gt)
        quiet = quiet + 1 if segs() == before else 0
        if quiet >= 3 and world == 1:
            break
    torch.cuda.synchronize()


def bf16_mode_leg(a, dev, world, local, rank, f32_ms):
    """configs[2]'s numeric mode on the SAME box right after the fp32 timed region: P.set_precision("bf16") -- convolution
    operands rounded to bf16 on the matrix cores (one product per term, fp32 accumulation) and the LCA-internal tensors
    stored as bf16 -- same step (fwd + L1 + bwd + gradient all-reduce + fused Adam), same batch per GPU, all ranks."""
    import hvi_cidnet_amd as P
    from hvi_cidnet_amd.dp import DataParallelTrainer
    P.set_precision("bf16")
    try:
        torch.manual_seed(0)
        model = P.CIDNet().to(dev)
        model.two_streams = not a.single_stream
        tr = DataParallelTrainer(model, lr=1e-4, n_buckets=4, wgrad_stream=not (a.no_wgrad_stream or a.single_stream))
        g = torch.Generator(device=dev)
        g.manual_seed(1000 + rank)
        x = torch.rand((a.batch, 3, a.height, a.width), device=dev, generator=g)
        gt = torch.rand((a.batch, 3, a.height, a.width), device=dev, generator=g)
        settle(tr, x, gt, dev, world)
        for _ in range(3):
            tr.step(x, gt)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(device_ids=[local])
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            loss = tr.step(x, gt)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(device_ids=[local])
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        lossv = float(loss.item())
        del tr, model
    finally:
        P.set_precision("f32")
        from hvi_cidnet_amd import ops
        ops.set_grad_arena(None, None)
        ops.clear_prepared_weights()
        torch.cuda.empty_cache()
    ms = 1e3 * dt / n
    return {"workload": f"the same step in the bf16 mode (BASELINE.json configs[2]'s numeric type), bs={a.batch}/GPU, {world} rank(s)",
            "images_per_s": round(world * a.batch * n / dt, 1), "ms_per_step": round(ms, 3), "steps": n,
            "vs_f32_line": round(f32_ms / ms, 3), "loss": round(lossv, 6),
            "dtype": "bf16 matrix-core operands, fp32 accumulation; bf16 storage of LayerNorm outputs, q/k/v, IEL hidden tensors and their gradients"}


def variants_bs16(dev):
    """configs[4]: the MSSA and TNSM variants, fwd + L1 (+ 0.1 mean(noise map) for TNSM, so that its noise branch trains) +
    bwd + fused Adam at bs=16 3x400x600 on one GPU, measured after the timed region (rank 0)."""
    import hvi_cidnet_amd as P
    from hvi_cidnet_amd.dp import DataParallelTrainer
    from hvi_cidnet_amd import ops
    B = 16
    prec = "fp32" if ops.MATH["levels"] == 3 else "bf16 mode"
    out = {"workload": f"CIDNet_MSSA / CIDNet_TNSM fwd+bwd bs=16 3x400x600 {prec} (BASELINE.json configs[4])", "step": "fwd + L1 + bwd + fused Adam"}
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    x = torch.rand((B, 3, 400, 600), device=dev, generator=g)
    gt = torch.rand((B, 3, 400, 600), device=dev, generator=g)
    for name, ctor, tnsm in (("CIDNet_MSSA", P.CIDNet_MSSA, False), ("CIDNet_TNSM", P.CIDNet_TNSM, True)):
        torch.manual_seed(0)
        m = ctor().to(dev)
        lf = (lambda y, t: ops.L1LossFn.apply(y[0], t) + 0.1 * y[1].mean()) if tnsm else None
        tr = DataParallelTrainer(m, lr=1e-4, loss_fn=lf)
        settle(tr, x, gt, dev)
        for _ in range(2):
            tr.step(x, gt)
        torch.cuda.synchronize()
        n = 6
        t0 = time.perf_counter()
        for _ in range(n):
            tr.step(x, gt)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        out[name] = {"images_per_s": round(B / dt, 1), "ms_per_step": round(1e3 * dt, 2)}
        del tr, m
        ops.set_grad_arena(None, None)
        ops.clear_prepared_weights()
        torch.cuda.empty_cache()
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_ranks(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python bench.py --gpus N does it itself)"
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP library is the only compute path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or os.environ.get("CIDNET_DP_FORCE_ALLREDUCE") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # no device_id: binding the group to a device at init (eager communicator) slows EVERY later step by ~5 ms on
        # this stack (39.5 vs 34.6 ms/step with no collective issued at all, tools/comm_probe.py); lazy init does not
        dist.init_process_group("nccl")

    import hvi_cidnet_amd as P
    from hvi_cidnet_amd.dp import DataParallelTrainer, pin_rank_to_cpus
    cpus = pin_rank_to_cpus(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))      # per-rank CPU slice (N > 1 only)
    if cpus is not None:
        print(f"[bench] rank {rank}: pinned to {len(cpus)} CPUs ({cpus[0]}..{cpus[-1]})", file=sys.stderr)
    P.set_precision(a.dtype)
    torch.manual_seed(0)
    model = P.CIDNet().to(dev)
    model.two_streams = not a.single_stream
    model.dual_norms = a.dual_norms
    trainer = DataParallelTrainer(model, lr=1e-4, n_buckets=4, wgrad_stream=not (a.no_wgrad_stream or a.single_stream),
                                  use_graph=a.graph, prepared_weights=not a.no_prepared_weights, loss_fn=P.CIDNetLoss(model, P_weight=a.p_weight).to(dev) if a.full_loss else None)
    g = torch.Generator(device=dev)
    g.manual_seed(1000 + rank)
    shape = (a.batch, 3, a.height, a.width)
    x = torch.rand(shape, device=dev, generator=g)
    gt = torch.rand(shape, device=dev, generator=g)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(device_ids=[local])
            torch.cuda.synchronize()

    settle(trainer, x, gt, dev, world)                  # set-up steps (allocator footprint), see settle()
    sync()
    for _ in range(max(a.warmup, 1)):
        loss = trainer.step(x, gt)
    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]   # per-step diagnostics only (stderr)
    host = []
    t0 = time.perf_counter()
    marks[0].record()
    watchdog = bool(os.environ.get("CIDNET_BENCH_WATCHDOG"))   # diagnostics: dump all Python stacks if one step's enqueue takes > 1 s
    if watchdog:
        import faulthandler
    for i in range(a.steps):
        if watchdog:
            faulthandler.dump_traceback_later(1.0, file=sys.stderr)
        loss = trainer.step(x, gt)
        if watchdog:
            faulthandler.cancel_dump_traceback_later()
        marks[i + 1].record()
        host.append(time.perf_counter())
    sync()
    dt = time.perf_counter() - t0
    if rank == 0:
        raw_enq = [1e3 * (b - c) for b, c in zip(host, [t0] + host[:-1])]
        gpu_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
        enq_ms = sorted(raw_enq)
        slow = [(i, round(v, 1)) for i, v in enumerate(raw_enq) if v > 5 * enq_ms[len(enq_ms) // 2]]
        print(f"[bench] slowest host enqueue at timed step {raw_enq.index(enq_ms[-1])}; steps slower than 5x the median: {slow}", file=sys.stderr)
        print(f"[bench] per-step ms on the main stream: min {gpu_ms[0]:.2f} median {gpu_ms[len(gpu_ms) // 2]:.2f} max {gpu_ms[-1]:.2f}; "
              f"host enqueue per step: min {enq_ms[0]:.2f} median {enq_ms[len(enq_ms) // 2]:.2f} max {enq_ms[-1]:.2f}", file=sys.stderr, flush=True)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    lossv = float(loss.item())
    assert lossv == lossv, "loss is NaN"

    # ---- roofline of the dominant kernel family (dense 3x3 conv on the fp32 MFMA), live events ----
    # every rank runs these extra steps (they contain the gradient all-reduce); only rank 0 instruments
    # They run single-stream so that an event pair brackets ONE kernel family at a time: with the two branch
    # streams of the timed region, launches of different families overlap and an event pair would charge a
    # kernel for the time it shares the GPU with the other branch.
    from hvi_cidnet_amd import ops as _ops
    model.two_streams = False
    _ops.enable_wgrad_stream(False)
    graph_mode, trainer.use_graph = trainer.use_graph, False      # the instrumented steps launch eagerly
    timer = OpTimer().install() if rank == 0 else None
    for _ in range(2):
        trainer.step(x, gt)
    sync()
    model.two_streams = not a.single_stream
    trainer.use_graph = graph_mode
    _ops.enable_wgrad_stream(trainer.wgrad_stream)
    if rank == 0:
        agg = timer.table()
        timer.remove()
        tot = sum(v[1] for v in agg.values())
        # the MFMA kernel only: the <= 4-channel stem / head launches of cidnet_conv3x3 run on streaming VALU kernels
        # cidnet_conv3x3_add(X, x_bs, Wt, w_ms, w_ks, flip, replicate, R, r_bs, Y, y_bs, B, M, K, H, W): same kernel with an addend
        c3 = [(args, e0.elapsed_time(e1)) for name, args, e0, e1 in timer.rec
              if name == "cidnet_conv3x3" and min(args[10], args[11]) > 4]
        c3 += [(args[:7] + args[9:], e0.elapsed_time(e1)) for name, args, e0, e1 in timer.rec
               if name == "cidnet_conv3x3_add" and min(args[12], args[13]) > 4]
        c3_flops = sum(conv3x3_flops(ar) for ar, _ in c3)
        c3_ms = sum(ms for _, ms in c3)
        # the same convs on the bf16 matrix cores (csrc/conv3x.hip, the default): six bf16 products per fp32 product
        cx = [(conv3x_shape(name, args), e0.elapsed_time(e1)) for name, args, e0, e1 in timer.rec
              if name in ("cidnet_conv3x3_bf16x3", "cidnet_conv3x3_bf16x3_pre", "cidnet_conv3x3_bf16x3_pre_lv")]
        cx_flops = sum(conv3x_flops(ar) for ar, _ in cx)
        cx_ms = sum(ms for _, ms in cx)
        if a.op_table:
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                print(f"  {k:40s} calls/step {v[0] // 2:5d}  ms/step {v[1] / 2:9.3f}  {100 * v[1] / tot:5.1f}%", file=sys.stderr)
            print(f"  sum of kernel families: {tot / 2:.3f} ms/step; wall {1e3 * dt / a.steps:.3f} ms/step", file=sys.stderr)
            for k, v in sorted(timer.table(by_shape=True).items(), key=lambda kv: -kv[1][1])[:a.op_rows]:
                print(f"    {k:70s} x{v[0] // 2:3d}  {v[1] / 2:8.3f} ms", file=sys.stderr)
        if cx_ms > c3_ms:
            # fp32-equivalent (algorithmic) rate, and the rate of the bf16 MFMA work it issues for it against the dense bf16 peak
            eq = cx_flops / (cx_ms * 1e-3) / 1e12
            nprod = 6 if _ops.MATH["levels"] == 3 else 1
            roof = {"bound": "mfma", "kernel": "conv3x_kernel (cidnet_conv3x3_bf16x3: dense 3x3 fwd + dgrad as six exact bf16 "
                                               "products per fp32 product on v_mfma_f32_16x16x32_bf16)",
                    "achieved": round(nprod * eq, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(nprod * eq / PEAK_BF16_MFMA_TFLOPS, 4), "bf16_products_per_term": nprod,
                    "fp32_equivalent_tflops": round(eq, 2), "vs_fp32_mfma_peak": round(eq / PEAK_F32_MFMA_TFLOPS, 4),
                    "traffic": pmc_traffic("conv3x"), "traffic_source": PMC_TRAFFIC_FILE,
                    "alg_bytes_per_launch": int(sum(conv3x_bytes(ar) for ar, _ in cx) / max(len(cx), 1)),
                    "launches_per_step": len(cx) // 2, "avg_launch_ms": round(cx_ms / max(len(cx), 1), 4),
                    "measured_in": "2 extra single-stream steps after the timed region (HIP events per launch)",
                    "share_of_step_kernel_time": round(cx_ms / tot, 3) if tot else None,
                    "note": "frac counts algorithmic products only: the kernel pads 36 -> 48 output channels and 324 -> 352 k "
                            "(0.69 of its issued MFMAs are algorithmic) and the chip sustains ~1.6 GHz under bf16-MFMA load"}
        else:
            achieved = c3_flops / (c3_ms * 1e-3) / 1e12 if c3_ms > 0 else 0.0
            roof = {"bound": "mfma", "kernel": "conv3_kernel (cidnet_conv3x3 / _add: dense 3x3 fwd + dgrad, fp32 MFMA launches)",
                    "achieved": round(achieved, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": pmc_traffic("conv3"), "traffic_source": PMC_TRAFFIC_FILE,
                    "launches_per_step": len(c3) // 2, "avg_launch_ms": round(c3_ms / max(len(c3), 1), 4),
                    "measured_in": "2 extra single-stream steps after the timed region (HIP events per launch)",
                    "share_of_step_kernel_time": round(c3_ms / tot, 3) if tot else None}

        # the family that bounds the step: the 1x1 convs and their weight gradients
        pw = [(pw_work(name, args), e0.elapsed_time(e1)) for name, args, e0, e1 in timer.rec if pw_work(name, args) is not None]
        pw_fl, pw_by, pw_ms = sum(w[0] for w, _ in pw), sum(w[1] for w, _ in pw), sum(t for _, t in pw)
        roof_pw = {"bound": "hbm", "kernel": "1x1-conv family (cidnet_pw_conv_t, cidnet_pw_conv_bf16x3[_pre, _pre_lv, _pre_t], cidnet_pw_conv_up_prelu, cidnet_pw_wgrad_t, cidnet_pw_bwd_fused)",
                   "achieved": round(pw_by / (pw_ms * 1e-3) / 1e9, 1) if pw_ms > 0 else 0.0, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                   "frac": round(pw_by / (pw_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if pw_ms > 0 else 0.0,
                   "tflops": round(pw_fl / (pw_ms * 1e-3) / 1e12, 2) if pw_ms > 0 else 0.0,
                   "vs_fp32_mfma_peak": round(pw_fl / (pw_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4) if pw_ms > 0 else 0.0,
                   "traffic": pmc_traffic("pw"), "traffic_source": PMC_TRAFFIC_FILE, "launches_per_step": len(pw) // 2,
                   "alg_gb_per_step": round(pw_by / 2 / 1e9, 3), "alg_gflop_per_step": round(pw_fl / 2 / 1e9, 1),
                   "ms_per_step": round(pw_ms / 2, 3), "share_of_step_kernel_time": round(pw_ms / tot, 3) if tot else None}

        # the HBM-bound side: the streaming kernels of the IEL chain (the tensors the bf16 storage mode halves)
        hb = [(iel_stream_bytes(name, args), e0.elapsed_time(e1)) for name, args, e0, e1 in timer.rec
              if name in ("cidnet_iel_dw_gate_fwd_t", "cidnet_iel_gate_dw_bwd_t", "cidnet_dw3x3_bwd_t")]
        hb_bytes, hb_ms = sum(b for b, _ in hb), sum(t for _, t in hb)
        roof_hbm = {"bound": "hbm", "kernel": "IEL-chain streaming kernels (iel_dw_gate_kernel, iel_gate_dw_bwd_kernel, dw3x3_bwd_kernel)",
                    "achieved": round(hb_bytes / (hb_ms * 1e-3) / 1e9, 1) if hb_ms > 0 else 0.0, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(hb_bytes / (hb_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if hb_ms > 0 else 0.0,
                    "traffic": None, "launches_per_step": len(hb) // 2, "alg_gb_per_step": round(hb_bytes / 2 / 1e9, 3),
                    "ms_per_step": round(hb_ms / 2, 3), "storage": a.dtype}
        roof["whole_step"] = whole_step_roofline(a, 1e3 * dt / a.steps)
    # the 1024x1024 inference leg runs on EVERY rank (north_star: both sizes at 1/2/4/8 GPUs); the variants on rank 0 at N = 1
    infer = variants = bf16 = None
    default_cfg = (a.height, a.width, a.batch) == (400, 600, 8)
    if not a.no_inference_leg and default_cfg:
        del trainer
        _ops.set_grad_arena(None, None)
        _ops.clear_prepared_weights()
        torch.cuda.empty_cache()
        if a.dtype == "f32" and not a.full_loss:
            bf16 = bf16_mode_leg(a, dev, world, local, rank, 1e3 * dt / a.steps)
        infer = inference_1024(dev, world, local)
        if world == 1 and not a.no_variants:
            variants = variants_bs16(dev)
    if rank == 0:
        cpu = None
        if world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline(a)

        out = {
            "metric": "images/sec fwd+bwd, CIDNet 400x600 bs=8 (step = fwd + " + (("L1+SSIM+Edge" + ("+VGG19-perceptual" if a.p_weight > 0 else "") + " loss on RGB and HVI") if a.full_loss else "L1 loss")
                      + " + bwd + grad all-reduce + Adam)",
            "value": round(world * a.batch * a.steps / dt, 3), "unit": "images/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if a.dtype == "f32" else "bf16 (matrix-core operands rounded to bf16, fp32 accumulation; bf16 storage of the LCA-internal tensors)", "data": "synthetic",
            "config": {"workload": f"CIDNet fwd+bwd bs={a.batch}/GPU 3x{a.height}x{a.width} fp32 (BASELINE.json configs[1])",
                       "global_batch": world * a.batch, "parallelism": f"dp{world}", "rccl_ranks": (dist.get_world_size() if dist.is_initialized() else 1), "streams": 1 if a.single_stream else (2 if a.no_wgrad_stream else 3), "loss": round(lossv, 6)},
            "roofline": roof if a.dtype == "f32" else dict(roof_hbm, mfma_conv3=roof), "roofline_pw": roof_pw, "roofline_hbm": roof_hbm,
            "cpu_baseline": cpu,
            "bf16_mode": bf16,
            "inference_1024": infer,
            "variants_bs16": variants,
        }
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


PMC_TRAFFIC_FILE = "profiles/r04_z_pmc_traffic_by_family.json"


def pmc_traffic(family):
    """HBM bytes per launch of a kernel family ("conv3x", "conv3", "pw") from the committed rocprofv3 PMC passes
    (written by tools/pmc_family_traffic.py from separate FETCH_SIZE / WRITE_SIZE runs of `bench.py --steps 1 --single-stream`,
    gfx950 x2 read correction); counters cannot be read from inside this process, so this is null if the summary is absent."""
    try:
        d = json.load(open(os.path.join(ROOT, PMC_TRAFFIC_FILE)))
        return int(d["families"][family]["avg_hbm_bytes_per_launch"])
    except Exception:
        return None


def usable_cores():
    """CPU threads this process may really use: affinity mask and cgroup quota, not the host's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 32))


def cpu_baseline(a):
    """The oracle (port of the reference algorithm, plain PyTorch CPU eager) forward+backward on a
    bounded sample: 2 images of the same 3xHxW shape (1/4 of one GPU batch), all host cores."""
    from oracle import cidnet_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    p = O.params_to(O.make_params(0, jitter=False), requires_grad=True)
    n = 2
    x = O.synthetic_batch(1, (n, 3, a.height, a.width))
    gt = O.synthetic_batch(2, (n, 3, a.height, a.width))
    y = O.cidnet_forward(p, x[:1, :, :64, :96])          # page-in / thread-pool warm-up
    y.abs().mean().backward()
    t0 = time.perf_counter()
    y = O.cidnet_forward(p, x)
    (y - gt).abs().mean().backward()
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"1 fwd+bwd of {n}x3x{a.height}x{a.width} fp32, torch CPU eager, {cores} threads ({dt:.1f} s)"}


if __name__ == "__main__":
    main()
