"""Static check of a compiled kernel (assembly text or llvm-objdump -d output on stdin / file argument): between an
inline-assembly global_load_dwordx4 and the next `s_waitcnt vmcnt(0)` no instruction may touch a destination register of
the load (the compiler does not know the result is pending).  Prints the offending instructions (dev tool; the shipped
library is checked by tests/test_abi.py)."""
import re, sys
text = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
pending, bad, loads = set(), [], 0
for n, line in enumerate(text.splitlines(), 1):
    ins = line.split("//")[0].strip()
    if not ins or ins.startswith(";") or ins.startswith("."):
        continue
    m = re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\]", ins)
    if m:
        pending.update(range(int(m.group(1)), int(m.group(2)) + 1)); loads += 1
        continue
    if "vmcnt(0)" in ins:
        pending.clear()
        continue
    regs = set()
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", ins):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(r) for r in re.findall(r"\bv(\d+)\b", ins))
    if regs & pending:
        bad.append((n, ins))
print(f"{loads} loads, {len(bad)} instructions touch a register with a pending load")
for n, ins in bad[:8]:
    print(f"  line {n}: {ins}")
sys.exit(1 if bad else 0)
