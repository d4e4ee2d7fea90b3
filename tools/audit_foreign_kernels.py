#!/usr/bin/env python3
"""Which instructions do the kernels that run beside ours -- but that we do not compile -- contain?

DESIGN.md section 4 (i) / 4.1 (b): beside LDS-fed bf16 MFMAs, packed-fp32 instructions WITH op_sel in another kernel's
waves were seen to lose a product (tools/mfma_pk_probe.hip).  libcidnet_hip.so is built without them (build.py, checked
by tests/test_abi.py); this tool applies the same disassembly check to the gfx950 code objects of libraries whose kernels
share the GPU with a training step: RCCL's reduction kernels (the bucket all-reduce overlaps the backward) and ATen's
elementwise kernels.

    python tools/audit_foreign_kernels.py <lib.so> <symbol-regex> [--out report.json]

For every offload bundle in the library's .hip_fatbin section (compressed "CCOB" bundles included, through
clang-offload-bundler) the gfx950 code object is extracted, the kernels whose (demangled) name matches the regex are
disassembled, and per kernel the tool counts v_pk_{fma,mul,add}_f32 with and without op_sel / op_sel_hi modifiers,
SDWA forms and MFMAs.
"""
import argparse
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
PK = re.compile(r"\bv_pk_(fma|mul|add)_f32\b")


def bundles(fatbin):
    """(offset, size) of every offload bundle in a .hip_fatbin image"""
    d = fatbin
    pos, out = 0, []
    while pos < len(d):
        if d[pos:pos + 4] == b"CCOB":
            ver = struct.unpack_from("<H", d, pos + 4)[0]
            size = struct.unpack_from("<I", d, pos + 8)[0] if ver == 2 else struct.unpack_from("<Q", d, pos + 8)[0]
            out.append((pos, size))
            pos += size
        elif d[pos:pos + 24] == b"__CLANG_OFFLOAD_BUNDLE__":
            n = struct.unpack_from("<Q", d, pos + 24)[0]
            end, p = pos, pos + 32
            for _ in range(n):
                off, sz, tl = struct.unpack_from("<QQQ", d, p)
                p += 24 + tl
                end = max(end, pos + off + sz)
            out.append((pos, end - pos))
            pos = end
        else:
            nxt = min([x for x in (d.find(b"CCOB", pos + 1), d.find(b"__CLANG_OFFLOAD_BUNDLE__", pos + 1)) if x >= 0], default=-1)
            if nxt < 0:
                break
            pos = nxt
    return out


def op_sel_nondefault(line):
    """packed-fp32 with operand routing: any op_sel bit set, or an op_sel_hi bit cleared (default op_sel_hi is all ones)"""
    m = re.search(r"op_sel:\[([01,]+)\]", line)
    if m and "1" in m.group(1):
        return True
    m = re.search(r"op_sel_hi:\[([01,]+)\]", line)
    return bool(m and "0" in m.group(1))


def audit(lib, sym_re, verbose=True):
    rx = re.compile(sym_re)
    report = {"library": lib, "symbol_regex": sym_re, "target": TARGET, "kernels": {}}
    with tempfile.TemporaryDirectory() as td:
        fb = os.path.join(td, "fatbin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fb], check=True)
        data = open(fb, "rb").read()
        bl = bundles(data)
        report["bundles"] = len(bl)
        for bi, (off, size) in enumerate(bl):
            bpath, co = os.path.join(td, "b.bin"), os.path.join(td, "b.co")
            open(bpath, "wb").write(data[off:off + size])
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={bpath}", f"--targets={TARGET}",
                                f"--output={co}"], capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            syms = subprocess.run([f"{LLVM}/llvm-readelf", "-sW", co], capture_output=True, text=True).stdout
            funcs = [l.split()[7] for l in syms.splitlines() if len(l.split()) >= 8 and l.split()[3] == "FUNC" and l.split()[6] != "UND"]
            dem = subprocess.run(["c++filt"], input="\n".join(funcs), capture_output=True, text=True).stdout.splitlines()
            want = [(f, d) for f, d in zip(funcs, dem) if rx.search(d)]
            os.remove(co) if not want else None
            if not want:
                continue
            for i in range(0, len(want), 200):
                chunk = want[i:i + 200]
                dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--disassemble-symbols=" + ",".join(s for s, _ in chunk), co],
                                     capture_output=True, text=True).stdout
                cur = None
                names = dict(chunk)
                for line in dis.splitlines():
                    m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                    if m:
                        cur = report["kernels"].setdefault(names.get(m.group(1), m.group(1)),
                                                           {"bundle": bi, "instructions": 0, "pk_f32": 0, "pk_f32_op_sel": 0, "sdwa": 0, "mfma": 0,
                                                            "examples": []})
                        continue
                    if cur is None or not line.startswith("\t"):
                        continue
                    cur["instructions"] += 1
                    if PK.search(line):
                        cur["pk_f32"] += 1
                        if op_sel_nondefault(line):
                            cur["pk_f32_op_sel"] += 1
                            if len(cur["examples"]) < 2:
                                cur["examples"].append(line.split("//")[0].strip())
                    if "sdwa" in line:
                        cur["sdwa"] += 1
                    if "v_mfma" in line:
                        cur["mfma"] += 1
            os.remove(co)
            if verbose:
                print(f"[audit] bundle {bi}: {len(want)} matching kernels", file=sys.stderr)
    ks = report["kernels"]
    report["summary"] = {"kernels": len(ks), "with_pk_f32": sum(1 for k in ks.values() if k["pk_f32"]),
                         "with_pk_f32_op_sel": sum(1 for k in ks.values() if k["pk_f32_op_sel"]),
                         "with_sdwa": sum(1 for k in ks.values() if k["sdwa"]),
                         "pk_f32_total": sum(k["pk_f32"] for k in ks.values()),
                         "pk_f32_op_sel_total": sum(k["pk_f32_op_sel"] for k in ks.values())}
    return report


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("regex")
    ap.add_argument("--out")
    ap.add_argument("--max-list", type=int, default=40, help="kernels listed individually in the report (those with packed ops first)")
    a = ap.parse_args()
    rep = audit(a.lib, a.regex)
    ks = sorted(rep["kernels"].items(), key=lambda kv: (-kv[1]["pk_f32_op_sel"], -kv[1]["pk_f32"], kv[0]))
    rep["kernels"] = dict(ks[:a.max_list])
    txt = json.dumps(rep, indent=1)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    print(json.dumps(rep["summary"]))
