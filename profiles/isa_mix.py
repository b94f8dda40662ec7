#!/usr/bin/env python3
"""Static instruction mix of each kernel in a hipcc -S listing (helper for DESIGN.md tables)."""
import sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2] if len(sys.argv) > 2 else 'step_kernel'
for i, l in enumerate(lines):
    if l.startswith('_ZN3nig') and pat in l and l.rstrip().endswith(tuple('E0123456789')) is not None and ':' in l and not l.startswith('\t'):
        name = l.split(':')[0]
        k = i
        while k < len(lines) and not lines[k].startswith('.Lfunc_end'):
            k += 1
        ins = [x.strip().split()[0] for x in lines[i + 1:k] if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
        c = Counter(ins)
        cnt = lambda p: sum(v for kk, v in c.items() if kk.startswith(p))
        print(name[7:70], "total", len(ins), "valu", cnt('v_'), "salu", cnt('s_'), "vmem", cnt('global_') + cnt('buffer_'),
              "lds", cnt('ds_'), "f64", sum(v for kk, v in c.items() if 'f64' in kk), "mulhi", cnt('v_mul_hi'),
              "mul_lo", cnt('v_mul_lo'), "mad64", cnt('v_mad_u64'), "waitcnt", c.get('s_waitcnt', 0))
