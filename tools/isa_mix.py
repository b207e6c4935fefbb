#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -save-temps .s file (dev tool): python tools/isa_mix.py file.s substring [first last]"""
import sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
starts = [i for i, l in enumerate(lines) if l.startswith('_Z') and key in l.split(':')[0] and l.rstrip().split(';')[0].strip().endswith(':')]
for st in starts:
    en = next(i for i in range(st, len(lines)) if lines[i].strip() == 's_endpgm')
    body = [l.strip() for l in lines[st + 1:en + 1]]
    ops = [l.split()[0] for l in body if l and not l.startswith(('.', ';', '//')) and not l.split(';')[0].strip().endswith(':')]
    c = Counter(ops)
    valu = sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k)
    salu = sum(v for k, v in c.items() if k.startswith('s_') and k not in ('s_waitcnt', 's_barrier', 's_nop', 's_endpgm'))
    mf = sum(v for k, v in c.items() if 'mfma' in k)
    w = [l for l in body if l.startswith('s_waitcnt')]
    print(lines[st].split(':')[0][-60:])
    print(f"  instrs {len(ops)}  mfma {mf}  valu {valu}  salu {salu}  ds_read {sum(v for k, v in c.items() if k.startswith('ds_read'))}  ds_write "
          f"{sum(v for k, v in c.items() if k.startswith('ds_write'))}  vmem_load {sum(v for k, v in c.items() if 'load' in k and not k.startswith(('s_', 'ds_', 'scratch')))}  "
          f"vmem_store {sum(v for k, v in c.items() if 'store' in k and not k.startswith(('ds_', 'scratch')))}  waitcnt {len(w)} (vmcnt(0): {sum('vmcnt(0)' in l for l in w)})  "
          f"saveexec {c['s_and_saveexec_b64']}  cbranch {sum(v for k, v in c.items() if k.startswith('s_cbranch'))}  barrier {c['s_barrier']}  nop {c['s_nop']}  scratch "
          f"{sum(v for k, v in c.items() if k.startswith('scratch'))}")
    print("  top valu:", [(k, v) for k, v in c.most_common(40) if k.startswith('v_') and 'mfma' not in k][:12])
