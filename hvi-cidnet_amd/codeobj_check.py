"""Static checks on the gfx950 code objects of libcidnet_hip.so, run by build.py after linking (a violation fails the
build) and by tests/test_abi.py on the shipped library.

1. No packed-fp32 (v_pk_{fma,mul,add}_f32) and no SDWA instruction anywhere: beside LDS-fed bf16 MFMAs a packed-fp32 op
   with op_sel in ANOTHER kernel's waves was seen to lose a product (tools/mfma_pk_probe.hip), and
   the branch / weight-gradient streams make every kernel of the library a possible neighbour of conv3x / pwx / conv3xw.
2. csrc/conv3xw.hip issues global loads as inline assembly, so the compiler does not know their results are pending:
   no instruction may touch a load's destination registers before the wait that retires it (tools/asm_load_hazard.py,
   an in-order vmcnt model over the kernel's basic blocks).
"""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("CIDNET_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
_TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")


def _hazard_checker():
    sys.path.insert(0, _TOOLS)
    try:
        import asm_load_hazard
    finally:
        sys.path.pop(0)
    return asm_load_hazard


def extract_code_objects(lib_path, workdir):
    """-> paths of the gfx950 code objects bundled in the shared library"""
    lib = shutil.copy(lib_path, os.path.join(workdir, "lib.so"))
    subprocess.run([OBJDUMP, "--offloading", lib], check=True, capture_output=True, cwd=workdir)
    return sorted(glob.glob(os.path.join(workdir, "lib.so.*gfx950")))


def disassemble(co, symbolize=False):
    cmd = [OBJDUMP, "-d"] + (["--symbolize-operands"] if symbolize else []) + [co]
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def kernel_symbols(dis, substring):
    """names of the function symbols of a disassembly that contain `substring` (one per template instantiation)"""
    import re
    names = []
    for m in re.finditer(r"^(?:[0-9a-f]+ )?<([\w.$]+)>:$", dis, flags=re.M):
        n = m.group(1)
        if substring in n and not n.startswith(("L", ".L")) and n not in names:
            names.append(n)
    return names


def check_library(lib_path, allow_packed=False):
    """-> (violations, facts): violations is a list of strings (empty = clean), facts counts what was looked at"""
    if not os.path.exists(OBJDUMP):
        raise RuntimeError(f"{OBJDUMP} not found: the code-object checks need llvm-objdump (set CIDNET_OBJDUMP)")
    bad = []
    facts = {"code_objects": 0, "bf16_mfma": 0, "asm_loads_checked": 0}
    with tempfile.TemporaryDirectory() as tmp:
        cos = extract_code_objects(lib_path, tmp)
        facts["code_objects"] = len(cos)
        for co in cos:
            dis = disassemble(co)
            facts["bf16_mfma"] += dis.count("v_mfma_f32_16x16x32_bf16")
            if not allow_packed:
                hits = [l.strip() for l in dis.splitlines()
                        if "sdwa" in l or any(f"v_pk_{op}_f32" in l for op in ("fma", "mul", "add"))]
                if hits:
                    bad.append(f"{os.path.basename(co)}: {len(hits)} packed-fp32 / SDWA instructions, e.g. {hits[0]}")
            if "conv3xw_kernel" in dis:
                sym = disassemble(co, symbolize=True)
                for name in kernel_symbols(sym, "conv3xw_kernel"):         # every instantiation (operand levels 3 and 1)
                    nloads, hz = _hazard_checker().check(sym, name)
                    facts["asm_loads_checked"] += nloads
                    if nloads < 8:
                        bad.append(f"{name}: only {nloads} inline-assembly loads found (expected >= 8)")
                    bad += [f"{name}: load destination touched before its wait: {h}" for h in hz[:4]]
    if facts["code_objects"] < 15:
        bad.append(f"only {facts['code_objects']} gfx950 code objects found in {lib_path}")
    if facts["asm_loads_checked"] == 0:
        bad.append("conv3xw_kernel not found in any code object")
    return bad, facts


if __name__ == "__main__":
    v, f = check_library(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcidnet_hip.so"))
    print(f)
    for line in v:
        print("VIOLATION:", line)
    sys.exit(1 if v else 0)
