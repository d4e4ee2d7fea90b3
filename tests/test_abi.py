"""CPU checks of the C-ABI boundary: the library builds for gfx950, loads, and exports every symbol
that include/cidnet_hip.h declares (no kernel is launched here)."""
import ctypes
import os

import pytest


def test_header_parses_and_library_exports_every_symbol():
    from hvi_cidnet_amd import _lib
    protos = _lib.parse_header()
    assert "cidnet_hvit_fwd" in protos and "cidnet_phvit_bwd" in protos
    assert os.path.exists(_lib.LIB_PATH), "run `python hvi-cidnet_amd/build.py`"
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), f"libcidnet_hip.so lacks {name}"
    assert _lib.lib().raw("cidnet_abi_version")() >= 1


def test_shipped_library_has_no_debug_state():
    """include/cidnet_hip.h promises stateless, re-entrant entry points: the timing-study switches
    (cidnet_debug_*) exist only in -DCIDNET_DEBUG builds, never in the in-tree library"""
    from hvi_cidnet_amd import _lib
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in ("cidnet_debug_pw_flags", "cidnet_debug_dw_rows", "cidnet_debug_c3_flags", "cidnet_debug_c3_phases", "cidnet_debug_c3xw_flags"):
        assert not hasattr(dll, name), f"{name} exported by the production library"
    assert not any(n.startswith("cidnet_debug") for n in _lib.parse_header())


def test_product_path_refuses_cpu_tensors():
    import torch
    from hvi_cidnet_amd.hvi_transform import RGB_HVI
    m = RGB_HVI()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.HVIT(torch.rand(1, 3, 4, 4))


def test_product_package_does_not_import_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "hvi-cidnet_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_library_has_no_packed_fp32_or_sdwa_instructions(tmp_path):
    """The shipped code objects contain no v_pk_{fma,mul,add}_f32 and no SDWA instruction (hvi-cidnet_amd/build.py): beside
    LDS-fed bf16 MFMAs (csrc/conv3x.hip runs on both branch streams) packed-fp32 ops with op_sel in a neighbouring
    kernel's waves were seen to drop products (tools/mfma_pk_probe.hip)."""
    import glob
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    from hvi_cidnet_amd import _lib
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not present")
    lib = shutil.copy(_lib.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)
    cos = glob.glob(str(tmp_path / "lib.so.*gfx950"))
    assert len(cos) >= 15, cos
    mfma = 0
    for co in cos:
        dis = subprocess.run([objdump, "-d", co], check=True, capture_output=True, text=True).stdout
        bad = [l for l in dis.splitlines() if "sdwa" in l or any(f"v_pk_{op}_f32" in l for op in ("fma", "mul", "add"))]
        assert not bad, (co, bad[:3])
        mfma += dis.count("v_mfma_f32_16x16x32_bf16")
    assert mfma > 0          # the split-product kernels are in the library


def test_inline_asm_loads_are_not_read_before_their_wait(tmp_path):
    """csrc/conv3xw.hip issues its staging loads as inline assembly so that they stay in flight behind the MFMA loop (and
    across the loop edge of its staging waves); the compiler does not know their results are pending, so on every path
    from such a load to the next `s_waitcnt vmcnt(0)` no instruction may touch a destination register -- checked on the
    shipped code object by a data-flow pass over its basic blocks (tools/asm_load_hazard.py)."""
    import glob
    import shutil
    import subprocess
    import sys
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    from hvi_cidnet_amd import _lib
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    try:
        import asm_load_hazard
    finally:
        sys.path.pop(0)
    lib = shutil.copy(_lib.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)
    found = False
    for co in glob.glob(str(tmp_path / "lib.so.*gfx950")):
        dis = subprocess.run([objdump, "-d", "--symbolize-operands", co], check=True, capture_output=True, text=True).stdout
        if "conv3xw_kernel" not in dis:
            continue
        found = True
        nloads, bad = asm_load_hazard.check(dis, "conv3xw_kernel")
        assert nloads >= 8 and not bad, bad[:4]
        # the checker sees a planted hazard: a read of a load destination right after the barrier that follows the loads
        lines = dis.splitlines()
        last = max(i for i, l in enumerate(lines) if "global_load_dwordx4" in l and "conv3xw" not in l)
        import re
        reg = re.search(r"global_load_dwordx4 v\[(\d+):", lines[last]).group(1)
        bar = next(i for i in range(last, len(lines)) if "s_barrier" in lines[i])
        lines.insert(bar + 1, f"\tv_mov_b32_e32 v250, v{reg}")
        assert asm_load_hazard.check("\n".join(lines), "conv3xw_kernel")[1]
    assert found


def test_asm_load_hazard_checker_on_synthetic_assembly():
    """tools/asm_load_hazard.py models vmcnt as an in-order queue over the kernel's basic blocks: a read of a pending
    destination is found across a loop edge, a counted wait retires only the older loads, and clean code passes"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    try:
        import asm_load_hazard as h
    finally:
        sys.path.pop(0)
    clean = """
k_kernel:
\tglobal_load_dwordx4 v[0:3], v[10:11], off
.LBB0_1:
\ts_waitcnt vmcnt(0)
\tv_add_f32_e32 v20, v0, v1
\tglobal_load_dwordx4 v[0:3], v[10:11], off
\ts_barrier
\ts_cbranch_scc1 .LBB0_1
\ts_waitcnt vmcnt(0)
\ts_endpgm
"""
    assert h.check(clean, "k_kernel") == (2, [])
    # a copy of the destination right after the barrier, i.e. before the wait at the loop top
    loop_edge = clean.replace("\ts_barrier\n", "\ts_barrier\n\tv_mov_b32_e32 v30, v2\n")
    assert h.check(loop_edge, "k_kernel")[1] == ["v_mov_b32_e32 v30, v2"]
    # vmcnt(1) retires the older of two loads only
    counted = """
k_kernel:
\tglobal_load_dwordx4 v[0:3], v[10:11], off
\tglobal_load_dwordx4 v[4:7], v[12:13], off
\ts_waitcnt vmcnt(1)
\tv_add_f32_e32 v20, v0, v1
\tv_add_f32_e32 v21, v4, v5
\ts_waitcnt vmcnt(0)
\ts_endpgm
"""
    assert h.check(counted, "k_kernel")[1] == ["v_add_f32_e32 v21, v4, v5"]
    # an address register that is itself a pending destination
    addr = counted.replace("\ts_waitcnt vmcnt(1)\n", "\tglobal_load_dwordx4 v[8:11], v[4:5], off\n\ts_waitcnt vmcnt(0)\n")
    assert h.check(addr, "k_kernel")[1] == ["global_load_dwordx4 v[8:11], v[4:5], off"]
