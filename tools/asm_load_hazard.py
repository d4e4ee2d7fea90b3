"""Static check of a compiled kernel: the staging loads of csrc/conv3xw.hip are inline assembly (so that they stay in flight
behind the MFMA loop), and the compiler does not know their results are pending.  On every path from such a
global_load_dwordx4 to the next `s_waitcnt vmcnt(0)` no instruction may touch a destination register of the load.

Forward data-flow over the kernel's basic blocks (labels, s_branch / s_cbranch_*, s_endpgm); works on the compiler's
assembly text (`hipcc -save-temps`, labels `.LBB0_12:`) and on `llvm-objdump -d --symbolize-operands` output (labels `<L12>:`).

  python tools/asm_load_hazard.py file.s [kernel-name-substring]       exit code 1 when a hazard is found
(the shipped library is checked by tests/test_abi.py through check().)"""
import re
import sys


def _instructions(text, kernel):
    """-> list of (label or None, mnemonic + operands) for the named kernel"""
    out, on = [], False
    for raw in text.splitlines():
        line = raw.split("//")[0].split(";")[0].rstrip()
        s = line.strip()
        if not s:
            continue
        m = re.match(r"^(?:[0-9a-f]+ )?<?([\w.$]+)>?:$", s)
        if m:
            name = m.group(1)
            if not on and kernel in name and not name.startswith((".L", "L")):
                on = True
                continue
            if on and not name.startswith((".L", "L")):
                break                                      # next symbol
            if on:
                out.append((name, None))
            continue
        if not on or s.startswith("."):
            continue
        out.append((None, s))
    return out


def _regs(ins):
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", ins):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(r) for r in re.findall(r"\bv(\d+)\b", ins))
    return regs


def check(text, kernel="conv3xw_kernel"):
    """-> (number of inline-assembly loads, list of offending instructions)"""
    seq = _instructions(text, kernel)
    # basic blocks
    blocks, cur, labels = [], [], {}
    for lab, ins in seq:
        if lab is not None:
            if cur:
                blocks.append(cur)
            cur = []
            labels[lab] = len(blocks)
            continue
        cur.append(ins)
        if ins.startswith(("s_branch", "s_cbranch", "s_endpgm")):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    # a label may point at a block that starts after an emptied `cur`: labels were recorded with the index the NEXT block gets
    succ = []
    for i, b in enumerate(blocks):
        last = b[-1] if b else ""
        s = []
        m = re.match(r"s_c?branch\w*\s+<?([\w.$]+)>?", last)
        if m and m.group(1) in labels:
            s.append(labels[m.group(1)])
        if not last.startswith(("s_branch", "s_endpgm")) and i + 1 < len(blocks):
            s.append(i + 1)
        succ.append([t for t in s if t < len(blocks)])
    pend_in = [set() for _ in blocks]
    bad, loads = {}, 0
    work = list(range(len(blocks)))
    first = True
    while work:
        i = work.pop(0)
        pending = set(pend_in[i])
        for ins in blocks[i]:
            m = re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\]", ins)
            if m:
                if pending and (_regs(ins.split(",", 1)[1]) & pending):      # its address registers
                    bad[(i, ins)] = True
                pending |= set(range(int(m.group(1)), int(m.group(2)) + 1))
                continue
            if re.search(r"s_waitcnt.*vmcnt\(0\)", ins):
                pending = set()
                continue
            if pending and (_regs(ins) & pending):
                bad[(i, ins)] = True
        for t in succ[i]:
            if not pending <= pend_in[t]:
                pend_in[t] |= pending
                if t not in work:
                    work.append(t)
    for b in blocks:
        loads += sum(1 for ins in b if ins.startswith("global_load_dwordx4"))
    return loads, [ins for (_, ins) in bad]


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    n, bad = check(text, sys.argv[2] if len(sys.argv) > 2 else "conv3xw_kernel")
    print(f"{n} inline-assembly loads, {len(bad)} instructions touch a register with a pending load")
    for ins in bad[:8]:
        print("   ", ins)
    sys.exit(1 if bad or n == 0 else 0)
