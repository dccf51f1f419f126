#!/bin/bash
# Prints registers / scratch / instruction mix of the device kernels (no GPU needed): tools/kstats.sh [name filter]
set -e
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -mllvm -two-entry-phi-node-folding-threshold=8 ${KFLAGS} -Iinclude -Ipath-tracer_amd/csrc --cuda-device-only -S"
/opt/rocm/bin/hipcc $FLAGS -o /tmp/ptgpu_k1.s path-tracer_amd/csrc/ptgpu.hip 2>/dev/null
# the list path pool is a translation unit of its own (scheduler flags of its own: POOL_LIST_FLAGS of path-tracer_amd/Makefile)
/opt/rocm/bin/hipcc $FLAGS ${POOL_LIST_FLAGS} -o /tmp/ptgpu_k2.s path-tracer_amd/csrc/pool_list.hip 2>/dev/null
cat /tmp/ptgpu_k1.s /tmp/ptgpu_k2.s > /tmp/ptgpu_k.s
python3 - "$1" <<'PY'
import re, sys, subprocess, collections
flt = sys.argv[1] if len(sys.argv) > 1 else ''
txt = open('/tmp/ptgpu_k.s').read()
bodies = {}
for m in re.finditer(r'\n(_Z\w+):\s*; @\1\n(.*?)\n\s*s_endpgm', txt, re.S):
    bodies[m.group(1)] = m.group(2)
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, re.S):
    name, body = m.group(1), m.group(2)
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r'\(anonymous namespace\)::', '', dn).split('(')[0].replace('void ', '')
    if flt and flt not in dn: continue
    g = lambda k: re.search(k + r'\s+(\S+)', body).group(1)
    ops = collections.Counter()
    for line in bodies.get(name, '').splitlines():
        mm = re.match(r'\s+([a-z_0-9]+)\s', line)
        if mm: ops[mm.group(1)] += 1
    tot = sum(ops.values())
    lanes = ops['v_readlane_b32'] + ops['v_writelane_b32']
    scr = sum(v for k, v in ops.items() if k.startswith('scratch_'))
    vmem = sum(v for k, v in ops.items() if k.startswith('global_'))
    print(f"{dn:44s} vgpr={g('next_free_vgpr'):>4s} sgpr={g('next_free_sgpr'):>4s} scratch={g('private_segment_fixed_size'):>4s} insts={tot:5d} lane-spill={lanes:4d} scratch-ops={scr:3d} global={vmem:3d} smem={sum(v for k,v in ops.items() if k.startswith('s_load')):3d} lds={sum(v for k,v in ops.items() if k.startswith('ds_')):3d}")
PY
