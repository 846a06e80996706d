#!/usr/bin/env python3
"""Static instruction mix per kernel from a hipcc -save-temps .s file: isa_mix.py file.s [name-substring ...]"""
import re, sys
s = open(sys.argv[1]).read()
pats = sys.argv[2:]
cur, bodies = None, {}
for line in s.split('\n'):
    m = re.match(r'^(_Z\S+):\s*(;.*)?$', line)
    if m:
        cur = m.group(1); bodies[cur] = []
        continue
    if line.startswith('\t.end_amdhsa_kernel') or line.startswith('.Lfunc_end'):
        cur = None
    if cur is not None:
        t = line.strip()
        if t and not t.startswith(('.', ';', '//')) and not t.endswith(':'):
            bodies[cur].append(t.split()[0])
for name, ops in bodies.items():
    if pats and not any(p in name for p in pats):
        continue
    cls = {}
    for op in ops:
        if 'mfma' in op: k = 'MFMA'
        elif op.startswith('ds_read') or op.startswith('ds_load'): k = 'ds_read'
        elif op.startswith('ds_'): k = 'ds_write/other'
        elif op.startswith(('global_load', 'buffer_load', 'flat_load')): k = 'vmem_load'
        elif op.startswith(('global_store', 'buffer_store', 'flat_store')): k = 'vmem_store'
        elif op.startswith('s_waitcnt'): k = 's_waitcnt'
        elif op.startswith('s_barrier'): k = 's_barrier'
        elif op.startswith('s_'): k = 'SALU'
        elif op.startswith('v_'): k = 'VALU'
        else: k = 'other'
        cls[k] = cls.get(k, 0) + 1
    print(f"{name}: {len(ops)} instrs  " + "  ".join(f"{k}={v}" for k, v in sorted(cls.items(), key=lambda kv: -kv[1])))
