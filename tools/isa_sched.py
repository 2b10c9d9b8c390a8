#!/usr/bin/env python3
"""Compact schedule of one kernel's ISA (dev tool): isa_sched.py file.s name-substring [start-label]"""
import re, sys
txt = open(sys.argv[1]).read()
parts = re.split(r"\n(_Z[^\n:]*):[^\n]*\n", txt)
for i in range(1, len(parts) - 1, 2):
    if sys.argv[2] not in parts[i]:
        continue
    body = parts[i + 1].split('.Lfunc_end')[0].split('\n')
    out = []; run = None; cnt = 0
    for l in body:
        s = l.strip()
        if not s or (s.startswith((';', '.')) and not s.startswith('.LBB')):
            continue
        if s.startswith('.LBB'):
            tok = '\n' + s.split(':')[0] + ':'
        else:
            op = s.split()[0]
            if op.startswith('v_mfma'): tok = 'MFMA'
            elif op.startswith('s_waitcnt'): tok = s.replace('s_waitcnt ', 'W:').replace(' ', ',')
            elif op.startswith('ds_read'): tok = 'dsr'
            elif op.startswith('ds_write'): tok = 'dsw'
            elif op.startswith('global_load_dwordx4'): tok = 'gA'
            elif op.startswith('global_load'): tok = 'gB'
            elif op.startswith('global_store'): tok = 'gst'
            elif op.startswith('s_barrier'): tok = 'BARRIER'
            elif op.startswith(('s_cbranch', 's_branch')): tok = s.replace(' ', '>')
            elif op.startswith('v_'): tok = 'v'
            elif op.startswith('s_'): tok = 's'
            else: tok = op
        if tok == run: cnt += 1
        else:
            if run: out.append(f"{run}x{cnt}" if cnt > 1 else run)
            run = tok; cnt = 1
    out.append(f"{run}x{cnt}")
    text = ' '.join(out)
    if len(sys.argv) > 3:
        text = text[text.find(sys.argv[3]):]
    print(text[:int(sys.argv[4]) if len(sys.argv) > 4 else 100000])
    break
