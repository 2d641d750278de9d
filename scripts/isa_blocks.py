#!/usr/bin/env python3
"""Basic-block statistics of one kernel in a hipcc -S listing: instruction counts, fp64 ops, LDS / scratch / readlane traffic.
usage: scripts/isa_blocks.py file.s <substring of the kernel's mangled name> [min_instr]"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
minins = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split('\n')
start = None
for i, l in enumerate(lines):
    if re.match(r'^[A-Za-z_][\w$.]*:', l) and key in l.split(':')[0]:
        start = i
        break
assert start is not None, "kernel not found"
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('.end_amdhsa_kernel') or lines[i].strip().startswith('s_endpgm') and False) if False else None
# kernel body ends at the first '.Lfunc_end'
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
blocks, cur, name = [], [], 'entry'
for l in body:
    t = l.strip()
    m = re.match(r'^([\w$.]+):', t)
    if m:
        if cur: blocks.append((name, cur))
        name, cur = m.group(1), []
        continue
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    cur.append(t.split(';')[0].strip())
if cur: blocks.append((name, cur))
def cnt(b, pat): return sum(1 for x in b if re.match(pat, x))
tot = sum(len(b) for _, b in blocks)
print(f"kernel {lines[start]} blocks {len(blocks)} instructions {tot}")
print("block instr f64 ldexp exp/rcp ds_rd ds_wr scratch rdlane wrlane waitcnt nop branch_to")
for nme, b in blocks:
    if len(b) < minins: continue
    f64 = cnt(b, r'v_(fma|mul|add|fmac|max|min|ldexp|rndne|trunc|cvt|div|rcp|rsq|sqrt|cmp\w*|cndmask)\w*_f64')
    ex = cnt(b, r'v_(exp|rcp|rsq|sqrt|log)\w*')
    br = [x.split()[-1] for x in b if x.startswith('s_cbranch') or x.startswith('s_branch')]
    print(nme, len(b), f64, cnt(b, r'v_ldexp_f64'), ex, cnt(b, r'ds_read|ds_load'), cnt(b, r'ds_write|ds_store'), cnt(b, r'scratch_'), cnt(b, r'v_readlane'),
          cnt(b, r'v_writelane'), cnt(b, r's_waitcnt'), cnt(b, r's_nop'), ','.join(br[-2:]))
