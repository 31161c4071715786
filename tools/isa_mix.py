#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in the audit listing (make -C tc-gnn_atc23_amd/csrc audit).
usage: tools/isa_mix.py <mangled-name-substring> [min_instructions] [--dump LABEL]"""
import re, sys
S = '/tmp/tcgnn_audit/tcgnn_device-hip-amdgcn-amd-amdhsa-gfx950.s'
s = open(S).read()
pat = sys.argv[1]
minn = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 30
m = re.search(r'\n(_Z\w*%s\w*):' % re.escape(pat), s)
name = m.group(1)
i = m.start(); j = s.index('.Lfunc_end', i)
body = s[i:j]
print(name, re.findall(r'; NumVgprs: \d+|; ScratchSize: \d+|; Occupancy: \d+', s[j:j + 6000])[:3])
if '--dump' in sys.argv:
    lab = sys.argv[sys.argv.index('--dump') + 1]
    k = body.index(lab + ':'); e = re.search(r'\n\.LBB\d+_\d+:', body[k + 5:])
    print(body[k:k + 5 + (e.start() if e else 20000)]); sys.exit(0)
cur = None; blocks = []
for l in body.splitlines():
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm: cur = [mm.group(1), []]; blocks.append(cur); continue
    if cur is not None and l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;'): cur[1].append(l.strip())
for n, ins in blocks:
    if len(ins) < minn: continue
    c = lambda p: sum(1 for x in ins if x.startswith(p))
    print(f"{n:14s} n={len(ins):4d} valu={c('v_') - c('v_mfma'):4d} mfma={c('v_mfma'):2d} salu={c('s_'):3d} ds={c('ds_'):3d} mem={c('buffer_') + c('global_'):2d}")
