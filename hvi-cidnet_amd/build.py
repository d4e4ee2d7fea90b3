"""Builds libcidnet_hip.so (all csrc/*.hip, gfx950 only) in-tree with hipcc.

    python hvi-cidnet_amd/build.py [--force]

hipcc cross-compiles for gfx950 without a GPU, so this runs in the development container; the
resulting .so travels to the GPU box with the repository snapshot.  Objects are cached by mtime.
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.environ.get("CIDNET_OBJ_DIR") or os.path.join(CSRC, "build")
LIB = os.environ.get("CIDNET_LIB_OUT") or os.path.join(HERE, "libcidnet_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# No packed-fp32 (v_pk_fma/mul/add_f32) and no SDWA instructions anywhere in the library: beside waves that feed bf16 MFMAs
# from LDS (csrc/conv3x.hip, on by default since round 3; csrc/pws.hip) a packed-fp32 op with op_sel in ANOTHER kernel's
# waves was seen to lose one of its two products (tools/mfma_pk_probe.hip), and the two branch
# streams / the weight-gradient stream make any kernel a possible neighbour.  The feature switch removes the instructions
# at the source (-fno-slp-vectorize alone leaves the ones that come from float4 arithmetic); SDWA shares the operand-select
# path.  The host pass of hipcc prints "not a recognized feature" for the switch and ignores it (filtered below).
NO_PACKED = [] if os.environ.get("CIDNET_ALLOW_PACKED") else ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
                                                               "-mllvm", "-amdgpu-sdwa-peephole=0"]      # (unrestricted build: A/B only)
COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", *NO_PACKED,
          "-I", os.path.join(os.path.dirname(HERE), "include"), "-I", CSRC, *os.environ.get("CIDNET_EXTRA_FLAGS", "").split()]
# hvi.hip mirrors the reference's fp32 operation order (bit-exact masks/sextants): no implicit FMA
# dw.hip: the SLP vectoriser packs the stencil FMAs into v_pk_fma_f32 and pays for it with register-pair shuffles
# (more instructions in total, and past 256 VGPRs in the gate backward): the kernels there are VALU-issue-bound
# iel.hip: same for the stencil stages of the tile-resident IEL kernel
PER_FILE = {"hvi.hip": ["-ffp-contract=off"], "dw.hip": ["-fno-slp-vectorize"] if not os.environ.get("CIDNET_DW_SLP") else [],
            "iel.hip": ["-fno-slp-vectorize"],
            "conv3x.hip": ["-fno-slp-vectorize", *os.environ.get("CIDNET_C3X_FLAGS", "").split()],
            "pwx.hip": ["-fno-slp-vectorize"],
            "conv3xw.hip": ["-fno-slp-vectorize"],
            "pwb.hip": ["-fno-slp-vectorize"],
            "conv3_thin.hip": os.environ.get("CIDNET_THIN_FLAGS", "").split()}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    ts = [os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    inc = os.path.join(os.path.dirname(HERE), "include")
    ts += [os.path.getmtime(os.path.join(inc, f)) for f in os.listdir(inc)] if os.path.isdir(inc) else []
    return max(ts) if ts else 0.0


def _compile(src, force):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    spath = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(spath), _deps_mtime()):
        return obj, False
    cmd = [HIPCC, *COMMON, *PER_FILE.get(src, []), "-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    err = "\n".join(l for l in r.stderr.splitlines() if "is not a recognized feature for this target" not in l)
    if err.strip():
        sys.stderr.write(err + "\n")
    return obj, True


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in res]
    if force or any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[build] linked {LIB} from {len(objs)} objects")
        _check_code_objects(verbose)
    elif verbose:
        print(f"[build] {LIB} up to date")
    return LIB


def _check_code_objects(verbose):
    """A freshly linked library must pass the static code-object checks (codeobj_check.py: no packed-fp32 / SDWA
    instructions, no inline-assembly load read before its wait) -- a compiler or flag change (CIDNET_EXTRA_FLAGS) that
    reintroduces either fails the BUILD, not only a test.  CIDNET_ALLOW_PACKED builds (A/B measurements) skip check 1."""
    if os.environ.get("CIDNET_SKIP_CODEOBJ_CHECK") == "1":      # explicit opt-out (a box without llvm-objdump); tests still check
        if verbose:
            print("[build] code-object checks skipped (CIDNET_SKIP_CODEOBJ_CHECK=1)")
        return
    sys.path.insert(0, HERE)
    try:
        import codeobj_check
    finally:
        sys.path.pop(0)
    bad, facts = codeobj_check.check_library(LIB, allow_packed=bool(os.environ.get("CIDNET_ALLOW_PACKED")))
    if bad:
        os.replace(LIB, LIB + ".rejected")
        raise RuntimeError("code-object check failed (library moved to %s.rejected):\n  " % LIB + "\n  ".join(bad))
    if verbose:
        print(f"[build] code objects clean: {facts}")


if __name__ == "__main__":
    build(force="--force" in sys.argv)
