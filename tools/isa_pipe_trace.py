#!/usr/bin/env python3
"""Per-instruction issue times of ONE basic block under the model of tools/isa_pipe_model.py:
   tools/isa_pipe_trace.py file.s <kernel-substring> <block-length> [nth]   (the nth block of that many instructions)"""
import re
import sys

s = open(sys.argv[1]).read().split('\n')
pat, want = sys.argv[2], int(sys.argv[3])
nth = int(sys.argv[4]) if len(sys.argv) > 4 else 0
start = next(i for i, l in enumerate(s) if re.match(r'^_Z\S*:', l) and pat in l)
blocks, cur = [], []
for l in s[start + 1:]:
    if l.startswith('.Lfunc_end'):
        break
    t = l.split(';')[0].strip()
    if not t or (t.startswith('.') and not t.endswith(':')):
        continue
    if t.endswith(':'):
        if cur:
            blocks.append(cur)
        cur = []
        continue
    cur.append(t)
    if t.startswith('s_cbranch') or t.startswith('s_branch'):
        blocks.append(cur)
        cur = []


def regs(tok):
    m = re.search(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.search(r'\bv(\d+)\b', tok)
    return [int(m.group(1))] if m else []


b = [b for b in blocks if len(b) == want][nth]
ready, t = {}, 0
for ins in b:
    parts = ins.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
    is_valu = op.startswith('v_')
    f64 = is_valu and '_f64' in op
    cost = 8 if f64 else (4 if is_valu else 2)
    lat = 40 if f64 else (8 if is_valu else 0)
    srcs, dst = [], []
    if is_valu and ops:
        dst = regs(ops[0])
        for o in ops[1:]:
            srcs += regs(o)
        if op.startswith(('v_fmac', 'v_mac')):
            srcs += dst
    t0 = t
    for r in srcs:
        t0 = max(t0, ready.get(r, 0))
    print(f"{t0:5d} {('+' + str(t0 - t)) if t0 > t else '':>5s}  {ins[:90]}")
    t = t0 + cost
    for r in dst:
        ready[r] = t0 + lat
