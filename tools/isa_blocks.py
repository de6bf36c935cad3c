#!/usr/bin/env python3
"""Basic-block structure of one kernel in a hipcc -S dump:  tools/isa_blocks.py file.s <name-substring> [min_instr]
Per block: instructions, FP64 VALU, LDS, global/flat memory operations, and the branch that ends it."""
import re
import sys

s = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
minc = int(sys.argv[3]) if len(sys.argv) > 3 else 0
start = next(i for i, l in enumerate(s) if re.match(r'^_Z\S*:', l) and pat in l)
cnt = f64 = lds = gl = 0
nblocks = 0
for l in s[start + 1:]:
    if l.startswith('.Lfunc_end'):
        break
    t = l.strip()
    if not t or t.startswith(';') or (t.startswith('.') and not t.endswith(':')):
        continue
    if t.endswith(':'):
        if cnt >= minc:
            print(f"   [{cnt} instr, {f64} f64, {lds} lds, {gl} mem] (falls into {t})")
        nblocks += 1
        cnt = f64 = lds = gl = 0
        continue
    op = t.split()[0]
    cnt += 1
    f64 += '_f64' in op
    lds += op.startswith('ds_')
    gl += op.startswith(('global_', 'flat_', 'buffer_'))
    if op.startswith('s_cbranch') or op == 's_branch':
        if cnt >= minc:
            print(f"   [{cnt} instr, {f64} f64, {lds} lds, {gl} mem] -> {t}")
        nblocks += 1
        cnt = f64 = lds = gl = 0
print("blocks:", nblocks)
