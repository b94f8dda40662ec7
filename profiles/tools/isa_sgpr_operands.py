#!/usr/bin/env python3
"""Vector instructions of one kernel that would issue at the fast rate (~2.7 cycles per wave-instruction with >= 2 waves
per SIMD) but take a scalar-register source, which makes them issue at the slow rate (~4.4): profiles/r04/valu_rate_operands.txt.
Lists them per mnemonic for an instruction range (default: the whole kernel).
    python profiles/tools/isa_sgpr_operands.py kernel.s <mangled-name fragment> [first last]"""
import re
import sys
from collections import Counter

FAST = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_bitop3_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32")
lines = open(sys.argv[1]).read().split("\n")
frag = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3nig") and frag in l and ":" in l and not l.startswith("\t"))
ins = []
for l in lines[start + 1:]:
    if l.startswith(".Lfunc_end"):
        break
    t = l.strip()
    if l.startswith("\t") and t and not t.startswith((".", ";")):
        ins.append(t.split(";")[0].strip())
a, b = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, len(ins) - 1)
hit, tot = Counter(), Counter()
for t in ins[a:b + 1]:
    op = t.split()[0]
    base = op.replace("_e32", "").replace("_e64", "")
    if base in FAST:
        tot[base] += 1
        srcs = t.split(",")[1:]
        if any(re.search(r"\bs\d+\b|\bs\[\d+:\d+\]|\bvcc\b|\bexec", x) for x in srcs):
            hit[base] += 1
print("range [%d, %d]: %d instructions, %d of the fast class, %d of those with a scalar source" % (a, b, b - a + 1, sum(tot.values()), sum(hit.values())))
for k, v in hit.most_common():
    print("   %-16s %4d of %4d" % (k, v, tot[k]))
