"""CPU checks of bench.py's bookkeeping (no GPU): the fused-kernel plan behind `roofline.whole_step` reproduces SURVEY.md 8(d)'s
per-image totals, and the per-launch work formulas agree with the entry points' argument lists in include/cidnet_hip.h."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("_bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_roofline_plan_totals_match_the_survey():
    b = _bench()
    plan = b.roofline_plan(1, 400, 600)
    flops = sum(f for _, f, _ in plan) / 3.0            # the plan is forward + backward = 3 x forward
    nbytes = sum(n for _, _, n in plan) / 3.0
    # 59.9 GFLOP and ~0.93 GB per image forward under the fused plan (dead I_LCA5 included)
    assert abs(flops / 1e9 - 59.9) < 0.3, flops / 1e9
    assert abs(nbytes / 1e9 - 0.93) < 0.02, nbytes / 1e9
    # the floor bench.py divides by: sum of max(bytes / 8 TB/s, flops / 157.3 TFLOP/s) over the 40 fused kernels, x8 images
    plan8 = b.roofline_plan(8, 400, 600)
    assert len(plan8) == 40
    floor_ms = 1e3 * sum(max(n / (b.PEAK_HBM_GBS * 1e9), f / (b.PEAK_F32_MFMA_TFLOPS * 1e12)) for _, f, n in plan8)
    assert abs(floor_ms - 9.452) < 0.01, floor_ms


def test_work_formulas_follow_the_header_argument_order():
    b = _bench()
    from hvi_cidnet_amd import _lib
    protos = _lib.parse_header()

    def names(fn):
        return [n for _, n in protos[fn][1]]
    # conv3x_shape reads (R, B, M, K, H, W) by position from both entry points
    a = names("cidnet_conv3x3_bf16x3_pre_lv")
    assert [a[i] for i in (3, 7, 8, 9, 10, 11)] == ["R", "B", "M", "K", "H", "W"]
    a = names("cidnet_conv3x3_bf16x3")
    assert [a[i] for i in (6, 12, 13, 14, 15, 16)] == ["R", "B", "M", "K", "H", "W"]
    # pw_work
    a = names("cidnet_pw_conv_bf16x3_pre_lv")
    assert [a[i] for i in (6, 8, 9, 10, 11)] == ["R", "B", "M", "K", "HW"]
    a = names("cidnet_pw_wgrad_t")
    assert [a[i] for i in (1, 4, 12, 13, 14, 15)] == ["dy_dt", "x_dt", "B", "M", "N", "HW"]
    a = names("cidnet_pw_conv_t")
    assert [a[i] for i in (1, 8, 10, 12, 13, 14, 15)] == ["x_dt", "y_dt", "R", "B", "M", "K", "HW"]
    a = names("cidnet_pw_conv_bf16x3_pre_t")
    assert [a[i] for i in (1, 6, 8, 10, 11, 12, 13)] == ["x_dt", "y_dt", "R", "B", "M", "K", "HW"]
    # the typed 1x1 entry point the step uses (ops.pw_conv_bf16x3): bench.pw_work must know it -- for a while it did not, and the
    # 1x1 family's roofline silently lost its largest member
    fl, by = b.pw_work("cidnet_pw_conv_bf16x3_pre_t", (0, 0, 0, 0, 0, 0, 0, 0, None, 0, 8, 36, 95, 60000, 3, 3, 0))
    assert fl == 2.0 * 36 * 95 * 60000 * 8 and by == (95 * 4 + 36 * 4) * 60000 * 8
    # every 1x1 entry point ops.py calls is priced
    import re
    src = open(os.path.join(ROOT, "hvi-cidnet_amd", "ops.py")).read()
    called = set(re.findall(r'lib\(\)\.call\("(cidnet_pw_(?:conv|wgrad|bwd)[a-z0-9_]*)"', src))
    called -= {n for n in called if n.endswith("_prep") or n.endswith("_prep_batch")}
    for n in called:
        nargs = len(protos[n][1])
        assert b.pw_work(n, tuple([0] * nargs)) is not None, f"bench.pw_work does not price {n}"
