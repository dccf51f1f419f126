#!/usr/bin/env python3
"""Opcode histogram of one kernel from the assembly tools/kstats.sh leaves in /tmp/ptgpu_k.s (no GPU needed):
tools/opmix.py '<demangled name prefix>' [top]. Classes: f64 arithmetic, f64 transcendental, selects / moves,
compares, int32, conversions, scalar ALU, scalar memory, LDS, global, scratch, lane moves (SGPR spills), control."""
import collections
import re
import subprocess
import sys

flt = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
txt = open('/tmp/ptgpu_k.s').read()


def cls(op):
    if op.startswith('scratch_'): return 'scratch'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_'): return 'global'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('s_load') or op.startswith('s_buffer'): return 'smem'
    if op in ('v_readlane_b32', 'v_writelane_b32', 'v_readfirstlane_b32'): return 'lane moves'
    if op.startswith('s_waitcnt') or op.startswith('s_nop') or op.startswith('s_cbranch') or op.startswith('s_branch') or op in ('s_endpgm', 's_barrier'): return 'control'
    if op.startswith('s_'): return 'salu'
    if re.match(r'v_(rcp|rsq|sqrt|sin|cos|log|exp|frexp|ldexp|trig|div_scale|div_fmas|div_fixup)', op): return 'f64/f32 transcendental+div'
    if re.match(r'v_cmp|v_cmpx', op): return 'compare'
    if re.match(r'v_cndmask|v_mov|v_accvgpr|v_swap', op): return 'select/move'
    if re.match(r'v_cvt', op): return 'convert'
    if re.search(r'_f64', op): return 'f64 arithmetic'
    if re.search(r'_f32|_f16', op): return 'f32 arithmetic'
    if op.startswith('v_'): return 'int32 valu'
    return 'other'


for m in re.finditer(r'\n(_Z\w+):\s*; @\1\n(.*?)\n\s*s_endpgm', txt, re.S):
    name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0].replace('void ', '')
    if not name.startswith(flt):
        continue
    ops = collections.Counter()
    for line in m.group(2).splitlines():
        mm = re.match(r'\s+([a-z_0-9]+)\s', line)
        if mm:
            ops[mm.group(1)] += 1
    tot = sum(ops.values())
    by = collections.Counter()
    for op, n in ops.items():
        by[cls(op)] += n
    print(f'{name}: {tot} instructions (static)')
    for c, n in by.most_common():
        print(f'  {c:32s} {n:6d} {100.0 * n / tot:5.1f} %')
    print('  top opcodes: ' + ', '.join(f'{op} {n}' for op, n in ops.most_common(top)))
