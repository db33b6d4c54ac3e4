#!/usr/bin/env python3
"""kernel -> (vgprs, agprs, sgprs, spills, lds) table from a -save-temps .s file; `regs.py a.s [b.s]` prints it or the diff"""
import re, subprocess, sys


def table(path):
    out, name = {}, None
    cur = {}
    for line in open(path):
        m = re.match(r'\s*\.amdhsa_kernel\s+(\S+)', line)
        if m:
            name, cur = m.group(1), {}
            continue
        if name:
            m = re.match(r'\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|accum_offset|group_segment_fixed_size|private_segment_fixed_size)\s+(\d+)', line)
            if m:
                cur[m.group(1)] = int(m.group(2))
            if '.end_amdhsa_kernel' in line:
                out[name] = cur
                name = None
    return out


def demangle(names):
    p = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True)
    return dict(zip(names, p.stdout.split('\n')))


if __name__ == '__main__':
    a = table(sys.argv[1])
    b = table(sys.argv[2]) if len(sys.argv) > 2 else None
    dm = demangle(sorted(set(a) | set(b or {})))
    for k in sorted(a):
        short = re.sub(r'\(.*', '', dm[k].replace('void ', '').replace('(anonymous namespace)::', ''))[:90]
        va = a[k]
        row = (va.get('next_free_vgpr'), va.get('accum_offset'), va.get('next_free_sgpr'), va.get('private_segment_fixed_size'), va.get('group_segment_fixed_size'))
        if b is None:
            print('%-90s vgpr %4s acc_off %4s sgpr %4s scratch %5s lds %6s' % ((short,) + row))
        elif k in b:
            vb = b[k]
            rowb = (vb.get('next_free_vgpr'), vb.get('accum_offset'), vb.get('next_free_sgpr'), vb.get('private_segment_fixed_size'), vb.get('group_segment_fixed_size'))
            if row != rowb:
                print('%-90s %s -> %s' % (short, row, rowb))
    if b is not None:
        for k in sorted(set(b) - set(a)):
            print('NEW %s' % re.sub(r'\(.*', '', dm[k])[:100])
