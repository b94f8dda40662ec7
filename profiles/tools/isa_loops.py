#!/usr/bin/env python3
"""Loops of one kernel in a hipcc -S listing, with their instruction mix: every backward branch closes a loop
[label .. branch]; loops are printed outermost-last with per-class counts of the instructions between label and
branch (inner loops included).  Helper for DESIGN.md's per-step instruction tables.
    python profiles/tools/isa_loops.py kernel.s <mangled-name fragment> [min_instructions]
"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
frag = sys.argv[2]
min_ins = int(sys.argv[3]) if len(sys.argv) > 3 else 100
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3nig") and frag in l and ":" in l and not l.startswith("\t"))
end = start
while not lines[end].startswith(".Lfunc_end"):
    end += 1
body = lines[start + 1:end]
ins, labels = [], {}
for l in body:
    t = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", t):
        labels[t.split(":")[0]] = len(ins)
    elif l.startswith("\t") and t and not t.startswith((".", ";")):
        ins.append(t)


def classes(seq):
    c = Counter(x.split()[0] for x in seq)
    g = lambda *p: sum(v for k, v in c.items() if k.startswith(p))
    f64 = sum(v for k, v in c.items() if "f64" in k)
    return dict(total=len(seq), valu=g("v_"), salu=g("s_"), f64=f64, vmem=g("global_", "buffer_", "scratch_"), lds=g("ds_"),
                readlane=g("v_readlane", "v_readfirstlane", "v_writelane"), nop=c.get("s_nop", 0), mov=g("v_mov", "v_accvgpr"),
                mul_hi=g("v_mul_hi"), mul_lo=g("v_mul_lo"), mad64=g("v_mad_u64"), cndmask=g("v_cndmask"), cmp=g("v_cmp"),
                cvt=g("v_cvt"), waitcnt=c.get("s_waitcnt", 0), scratch=g("scratch_"))


print(lines[start].split(":")[0][:100], "instructions", len(ins))
loops = []
for k, t in enumerate(ins):
    op = t.split()[0]
    if op == "s_branch" or op.startswith("s_cbranch"):
        tgt = labels.get(t.split()[-1])
        if tgt is not None and tgt <= k and k - tgt + 1 >= min_ins:
            loops.append((tgt, k))
for a, b in sorted(loops, key=lambda x: (x[1] - x[0])):
    print("loop [%d, %d]" % (a, b), classes(ins[a:b + 1]))
if len(sys.argv) > 5:      # histogram of one loop: isa_loops.py file frag min a b
    a, b = int(sys.argv[4]), int(sys.argv[5])
    for op, n in Counter(x.split()[0] for x in ins[a:b + 1]).most_common(60):
        print("   %-28s %d" % (op, n))
