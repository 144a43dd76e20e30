#!/usr/bin/env python3
"""Order of vector-memory instructions, waits and barriers in a kernel's gfx950 assembly: L = global load, S = global store, wN = s_waitcnt vmcnt(N),
|B| = barrier, c = branch; one line per basic block ('*' = inside a loop). A 'w0' right behind a group of L's means the prefetch is not one: the wave
stalls on the loads it has just issued.      python tools/memseq.py /tmp/kregs_k_sgs.s 'k_strain_tileIjLi0ELi14ELi0ELi1ELi1E'"""
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):', s, re.M):
    if sys.argv[2] not in m.group(1):
        continue
    body = s[m.start():s.index('.end_amdhsa_kernel', m.start())]
    print(m.group(1)); out = []; cur = ''
    for line in body.split('\n'):
        t = line.strip(); op = t.split(' ')[0]
        if re.match(r'\.LBB\d+_\d+:', t):
            if cur.strip('c'): out.append(cur)
            cur = ('*' if 'Loop' in t else ' ') + t.split(':')[0][4:] + ': '
        elif op.startswith('global_load') or op.startswith('buffer_load'): cur += 'L'
        elif op.startswith('global_store') or op.startswith('buffer_store'): cur += 'S'
        elif op == 's_barrier': cur += '|B|'
        elif op == 's_waitcnt' and 'vmcnt' in t: cur += ' w' + re.search(r'vmcnt\((\d+)\)', t).group(1) + ' '
        elif op.startswith('ds_read') or op.startswith('ds_load'): cur += 'r'
        elif op.startswith('ds_write') or op.startswith('ds_store'): cur += 'd'
    out.append(cur)
    print('\n'.join(out))
