#!/usr/bin/env python3
"""Summarise the MFMA-bearing basic blocks of a gfx950 assembly dump (hipcc -S --cuda-device-only):
instruction mix and s_waitcnt vmcnt values per block.  usage: isa_loops.py file.s <substring of kernel name>..."""
import re
import sys

s = open(sys.argv[1]).read()
for key in sys.argv[2:]:
    m = re.search(r'^(_Z\S*' + re.escape(key) + r'\S*):.*?\n(.*?)\n\.Lfunc_end', s, re.S | re.M)
    if not m:
        print(key, 'not found')
        continue
    body = m.group(2).split('\n')
    blocks, cur, name = [], [], 'entry'
    for l in body:
        if re.match(r'^\.LBB\S+:', l):
            blocks.append((name, cur)); cur = []; name = l.split(':')[0]
        else:
            cur.append(l)
    blocks.append((name, cur))
    print(key)
    for n, b in blocks:
        ins = [l for l in b if l.strip() and not l.strip().startswith(('.', ';'))]
        mf = sum('v_mfma' in l for l in ins)
        if mf < 8:
            continue
        valu = sum(1 for l in ins if re.match(r'\s+v_', l) and 'mfma' not in l)
        salu = sum(1 for l in ins if re.match(r'\s+s_', l))
        waits = [re.search(r'vmcnt\((\d+)\)', l).group(1) for l in ins if 'vmcnt' in l]
        print(f'  {n}: instr {len(ins)} mfma {mf} valu {valu} salu {salu} gload {sum("global_load" in l for l in ins)}'
              f' dsr {sum("ds_read" in l for l in ins)} dsw {sum("ds_write" in l for l in ins)}'
              f' br {sum("s_cbranch" in l or "s_branch" in l for l in ins)} vmcnt {waits}')
