#!/usr/bin/env python3
"""Instruction mix of every loop body in a gfx950 assembly listing (hipcc -S): for each backward branch, the instructions between
its target label and itself, split into v_mad_u64_u32 / other VALU / SALU (s_nop separately) / memory.  Usage: loops.py rings.s [min]"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
minn = int(sys.argv[2]) if len(sys.argv) > 2 else 300
labels = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
def mix(a, b):
    c = collections.Counter()
    for x in lines[a:b + 1]:
        if not x.startswith('\t'): continue
        t = x.split()
        if not t or t[0].startswith(('.', ';')): continue
        op = t[0]
        if op.startswith('v_mad_u64'): c['mac'] += 1
        elif op.startswith('v_mov'): c['vmov'] += 1
        elif op.startswith('v_'): c['valu'] += 1
        elif op == 's_nop': c['nop'] += 1
        elif op.startswith('s_'): c['salu'] += 1
        elif op.startswith(('global_', 'flat_', 'buffer_')): c['gmem'] += 1
        elif op.startswith('scratch_'): c['scratch'] += 1
        elif op.startswith('ds_'): c['lds'] += 1
    return c
for i, l in enumerate(lines):
    m = re.match(r'\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)', l)
    if m and m.group(2) in labels and labels[m.group(2)] < i:
        a = labels[m.group(2)]; c = mix(a, i)
        v = c['mac'] + c['valu'] + c['vmov']
        if v >= minn:
            print(f"{m.group(2):>10} lines {a:6d}-{i:6d}  VALU {v:5d} = mac {c['mac']:4d} + mov {c['vmov']:3d} + other {c['valu']:4d} | nop {c['nop']:3d} salu {c['salu']:3d} gmem {c['gmem']:2d} scratch {c['scratch']:2d} lds {c['lds']:2d}")
for l in lines:
    if 'private_seg_size' in l or '.num_vgpr' in l or 'codeLenInByte' in l: print(l.strip())
