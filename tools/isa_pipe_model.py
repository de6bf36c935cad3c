#!/usr/bin/env python3
"""In-order issue model of the basic blocks of one kernel (hipcc -S dump): where does a single wave per SIMD stall on
dependent FP64 results?   tools/isa_pipe_model.py file.s <kernel-substring> [min_instr]
Model (MI355X, tools/ubench/fp64_lat: dependent FP64 FMA 40 cycles, independent ~7.5): an instruction issues when the
previous one has issued and its VGPR sources are ready; FP64 VALU: issue 8, result 40; other VALU: issue 4, result 8;
SALU 2; LDS/VMEM results are ignored (they are prefetched a trip ahead).  Prints per block: instructions, issue cycles,
modelled cycles, stall share."""
import re
import sys

s = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
minc = int(sys.argv[3]) if len(sys.argv) > 3 else 40
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
if cur:
    blocks.append(cur)


def regs(tok):
    """VGPR indices named by an operand token like v5, v[6:7], -v[8:9], |v[2:3]|."""
    m = re.search(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.search(r'\bv(\d+)\b', tok)
    return [int(m.group(1))] if m else []


tot_issue = tot_model = 0
for b in blocks:
    if len(b) < minc:
        continue
    ready = {}
    t = 0
    issue_sum = 0
    for ins in b:
        parts = ins.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
        is_valu = op.startswith('v_')
        f64 = is_valu and ('_f64' in op)
        cost = 8 if f64 else (4 if is_valu else (2 if op.startswith('s_') else 4))
        lat = 40 if f64 else (8 if is_valu else 0)
        srcs = []
        dst = []
        if is_valu and ops:
            dst = regs(ops[0])
            for o in ops[1:]:
                srcs += regs(o)
            if op.startswith(('v_fmac', 'v_mac')):
                srcs += dst
        elif op.startswith('ds_write') or op.startswith('global_store'):
            for o in ops:
                srcs += regs(o)
        t0 = t
        for r in srcs:
            t0 = max(t0, ready.get(r, 0))
        t = t0 + cost
        issue_sum += cost
        for r in dst:
            ready[r] = t0 + lat
    tot_issue += issue_sum
    tot_model += t
    print(f"block of {len(b):4d} instr: issue {issue_sum:6d} cycles, modelled {t:6d} cycles, stall {100.0 * (t - issue_sum) / t:5.1f} %   ends: {b[-1][:40]}")
print(f"sum over listed blocks: issue {tot_issue}, modelled {tot_model}")
