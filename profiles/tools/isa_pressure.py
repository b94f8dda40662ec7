#!/usr/bin/env python3
"""Rough VGPR liveness profile of a straight-line stretch of a hipcc -S listing (lines [a, b) of the file, e.g. one
unrolled rollout step): treats the stretch as straight-line code, computes backward liveness of v-registers and
prints the live count every N instructions plus the registers live across the whole stretch (loop-carried).
    python profiles/tools/isa_pressure.py kernel.s 603 2111 [every]
"""
import re
import sys

src, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
every = int(sys.argv[4]) if len(sys.argv) > 4 else 40
lines = open(src).read().split("\n")[a - 1:b - 1]
RX = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(tok):
    out = []
    for m in RX.finditer(tok):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


ins = []
for no, l in enumerate(lines):
    t = l.strip()
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    t = t.split(";")[0].strip()
    op, _, rest = t.partition(" ")
    ops = [x.strip() for x in rest.split(",")] if rest else []
    if op.startswith(("s_", "ds_write", "global_store", "scratch_store", "buffer_store")) and not op.startswith("s_"):
        d, u = [], sum((regs(x) for x in ops), [])
    elif op.startswith(("global_store", "scratch_store", "ds_write")):
        d, u = [], sum((regs(x) for x in ops), [])
    elif op.startswith("s_") or op.startswith(";;"):
        d, u = [], sum((regs(x) for x in ops), [])
    elif op.startswith("v_cmp") or op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        d, u = [], sum((regs(x) for x in ops), [])
    elif op.startswith("v_writelane"):
        d, u = [], sum((regs(x) for x in ops), [])      # partial write: keeps the register live
    else:
        d = regs(ops[0]) if ops else []
        u = sum((regs(x) for x in ops[1:]), [])
        if op.startswith(("v_fmac", "v_mac", "v_dot")) or "dpp" in t:
            u += d
    ins.append((a + no, op, d, u))

live = set()
for _pass in range(3):                 # a loop body: live-out at the bottom = live-in at the top (fixpoint)
    prof = []
    for (no, op, d, u) in reversed(ins):
        live -= set(d)
        live |= set(u)
        prof.append((no, op, len(live)))
prof.reverse()
carried = sorted(live)
print(f"{len(ins)} instructions; live-in at the top: {len(carried)} registers")
peak = max(prof, key=lambda x: x[2])
print("peak", peak)
for i in range(0, len(prof), every):
    no, op, n = prof[i]
    print(f"  line {no:6d} {op:28s} live {n}")

# registers live at the peak, with the line and opcode of their latest definition above it (or "carried")
pk = peak[0]
live2 = set(carried)
lastdef = {}
for (no, op, d, u) in ins:
    if no > pk:
        break
    for r in d:
        lastdef[r] = (no, op)
live_at = set()
l = set(carried)
for _p in range(2):
    for (no, op, d, u) in reversed(ins):
        l -= set(d)
        l |= set(u)
        if no == pk:
            live_at = set(l)
by = {}
for r in sorted(live_at):
    by.setdefault(lastdef.get(r, (0, "carried")), []).append(r)
for k in sorted(by):
    print(f"   def line {k[0]:6d} {k[1]:24s} -> v{by[k]}")
