#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS per kernel from hipcc's -save-temps assembly (scripts/kernel_resources.py file.s [filter])."""
import re, subprocess, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2:] or [""]
rows = []
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, flags=re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r'\.amdhsa_' + k + r' (\d+)', body) or [None, '-'])[1]
    rows.append((name, g('next_free_vgpr'), g('accum_offset'), g('next_free_sgpr'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
dem = subprocess.run(['c++filt'], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
print("vgpr accum_off sgpr scratch lds  kernel")
for r, d in zip(rows, dem):
    d = d.replace('void clc::', '')
    if any(f in d for f in flt):
        print(f"{r[1]:>4} {r[2]:>9} {r[3]:>4} {r[4]:>7} {r[5]:>5}  {d[:150]}")
