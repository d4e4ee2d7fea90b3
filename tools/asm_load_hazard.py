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
    """-> (number of global_load_dwordx4, list of instructions that touch a register with a pending load)"""
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
    # vmcnt model: vector-memory operations complete in order; `s_waitcnt vmcnt(N)` returns when at most N are outstanding.
    # State = registers pending at block entry (a set: everything inherited is older than the block's own operations) plus
    # the block's own operations in order (destination sets; empty for stores).  vmcnt(N) with at least N own operations
    # retires everything inherited and all but the last N own ones; with fewer it keeps the inherited set (conservative).
    # Compiler-generated loads are modelled the same way as the inline-assembly ones: their waits are correct by
    # construction, and they decide how long the assembly loads between them stay pending.
    pend_in = [set() for _ in blocks]
    bad = {}
    work = list(range(len(blocks)))
    mem = re.compile(r"^(global|buffer|scratch|flat)_(load|store|atomic)")
    while work:
        i = work.pop(0)
        inherited, own = set(pend_in[i]), []
        for ins in blocks[i]:
            pending = inherited.union(*own) if own else inherited
            if mem.match(ins):
                ops = ins.split(None, 1)[1] if " " in ins else ""
                first = ops.split(",", 1)[0]
                rest = ops.split(",", 1)[1] if "," in ops else ""
                is_load = "_load" in ins.split()[0] or ("_atomic" in ins.split()[0] and " glc" in ins)
                touched = _regs(rest) | (_regs(first) if not is_load else set())     # addresses / store data are read now
                if pending and (touched & pending):
                    bad[(i, ins)] = True
                own.append(_regs(first) if is_load else set())
                continue
            m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", ins)
            if m:
                nleft = int(m.group(1))
                if len(own) >= nleft:
                    inherited = set()
                    own = own[len(own) - nleft:] if nleft else []
                continue
            if pending and (_regs(ins) & pending):
                bad[(i, ins)] = True
        out = inherited.union(*own) if own else inherited
        for t in succ[i]:
            if not out <= pend_in[t]:
                pend_in[t] |= out
                if t not in work:
                    work.append(t)
    loads = sum(1 for b in blocks for ins in b if ins.startswith("global_load_dwordx4"))
    return loads, [ins for (_, ins) in bad]


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    n, bad = check(text, sys.argv[2] if len(sys.argv) > 2 else "conv3xw_kernel")
    print(f"{n} dwordx4 loads, {len(bad)} instructions touch a register with a pending load")
    for ins in bad[:8]:
        print("   ", ins)
    sys.exit(1 if bad or n == 0 else 0)
