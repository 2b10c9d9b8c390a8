#!/usr/bin/env python3
"""Instruction mix per kernel from a --save-temps .s file (dev tool): isa_mix.py file.s substr [substr...]"""
import sys, re, collections
txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
parts = re.split(r"\n(_Z[^\n:]*):[^\n]*\n", txt)
for i in range(1, len(parts) - 1, 2):
    name, body = parts[i], parts[i + 1]
    if pats and not any(p in name for p in pats):
        continue
    body = body.split('.Lfunc_end')[0]
    ins = [l.strip().split()[0] for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = collections.Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    print(name[:70], 'total', len(ins), 'VALU', valu, 'pk_fma', c['v_pk_fma_f32'], 'mfma', sum(v for k, v in c.items() if 'mfma' in k),
          'ds_read', sum(v for k, v in c.items() if k.startswith('ds_read')), 'ds_write', sum(v for k, v in c.items() if k.startswith('ds_write')),
          'readlane', c['v_readlane_b32'], 'writelane', c['v_writelane_b32'],
          's_load', sum(v for k, v in c.items() if k.startswith('s_load')), 'waitcnt', c['s_waitcnt'], 'v_mov', c['v_mov_b32'], 'cndmask', c['v_cndmask_b32'],
          'scratch', sum(v for k, v in c.items() if k.startswith('scratch')))
