#!/usr/bin/env python3
"""Lists the s_waitcnt vmcnt(...) of a kernel's ISA that exist only because the next instruction OVERWRITES a register a load
is still to deliver (the register allocator handed a dead load result - a record word nobody reads - to the next temporary):
such a wait sits right behind the loads and serialises what could run under them.
Usage: python tools/find_waw_waits.py file.s <kernel symbol substring>"""
import re
import sys

text = open(sys.argv[1]).read()
sym = sys.argv[2]
start = text.index(sym + ":") if (sym + ":") in text else text.index(sym)
body = text[start:text.index(".Lfunc_end", start)].split("\n")

def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out

pending = []      # list of (dst regs, line) in issue order
lines = [(i, l.strip()) for i, l in enumerate(body) if l.strip() and not l.strip().startswith((";", ".", "//"))]
for k, (i, l) in enumerate(lines):
    if l.endswith(":"):
        continue
    op = l.split()[0]
    ops = l[len(op):].split(",")
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        pending.append((regs(ops[0]), i, l))
        continue
    if op == "s_waitcnt" and "vmcnt" in l:
        n = int(re.search(r"vmcnt\((\d+)\)", l).group(1))
        waited = pending[:max(0, len(pending) - n)]
        # what the following instruction (the one the wait was placed for) does
        nxt = lines[k + 1][1] if k + 1 < len(lines) else ""
        nop = nxt.split()[0] if nxt else ""
        nops = nxt[len(nop):].split(",")
        dst, src = regs(nops[0]), set().union(*[regs(x) for x in nops[1:]]) if len(nops) > 1 else set()
        if nop.startswith(("ds_write", "global_store", "ds_min", "ds_max", "ds_add", "ds_or")):
            src |= dst; dst = set()
        waited_regs = set().union(*[w[0] for w in waited]) if waited else set()
        if waited and not (src & waited_regs) and (dst & waited_regs):
            culprit = [w for w in waited if w[0] & dst]
            print("line %d: %s  before  '%s'  overwrites %s of '%s' (line %d), %d loads behind it" % (
                i, l, nxt, sorted(dst & waited_regs), culprit[-1][2], culprit[-1][1], len(pending) - 1 - pending.index(culprit[-1])))
        pending = pending[len(waited):]
