#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -S dump:  tools/isa_hist.py file.s <name-substring>"""
import collections
import re
import sys

s = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = None
for i, l in enumerate(s):
    if re.match(r'^_Z\S*:', l) and pat in l:
        start = i
        break
assert start is not None, "kernel not found"
ops = collections.Counter()
for l in s[start + 1:]:
    if l.startswith('.Lfunc_end'):
        break
    l = l.strip()
    if not l or l.startswith(('.', ';', '//')) or l.endswith(':'):
        continue
    ops[l.split()[0]] += 1
tot = sum(ops.values())
f64 = sum(v for k, v in ops.items() if k.endswith('_f64'))
print(f"total {tot}  f64-valu {f64}")
for k, v in ops.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 45):
    print(f"  {k:32s} {v}")
