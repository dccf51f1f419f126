#!/usr/bin/env python3
"""Per-segment model of a kernel from the PMC summary tools/run_pmc.sh writes (one rocprofv3 --pmc run per counter
set) -> profiles/pmc_model.json, which bench.py scales to the live run.

usage: pmc_model.py <gpurun_out/pmc_TAG> <kernel short name, e.g. k_path_pool> <source note> [entry name]
       (entry name: the key in pmc_model.json, by default the kernel's short name; bench.py looks its workloads up by
       "<scene>" / "<scene>:book")

  fp64_flop_issued_per_unit  (ADD_F64 + MUL_F64 + 2 FMA_F64 wave-instructions) x 64 lanes / segments
  hbm_bytes_per_unit         (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 / segments   (gfx950 reports half the bytes of
                             wide coalesced reads; WRITE_SIZE as is)
  valu_busy_frac             SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
  mean_active_lanes          SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU
"""
import csv
import json
import os
import re
import sys

out_dir, kernel, note = sys.argv[1], sys.argv[2], sys.argv[3]
entry_name = sys.argv[4] if len(sys.argv) > 4 else kernel
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = [r for r in csv.DictReader(open(os.path.join(out_dir, "summary.csv"))) if r["kernel"].split("<")[0] == kernel]
if not rows:
    raise SystemExit(f"{kernel} not in {out_dir}/summary.csv")
r = max(rows, key=lambda x: float(x["SQ_INSTS_VALU"]))
log = open(os.path.join(out_dir, "sq1.log")).read()
segments = int(re.search(r"segments=(\d+)", log).group(1))
paths = int(re.search(r"paths=(\d+)", log).group(1))
f = lambda k: float(r[k])
fp64 = f("SQ_INSTS_VALU_ADD_F64") + f("SQ_INSTS_VALU_MUL_F64") + f("SQ_INSTS_VALU_FMA_F64")
flop = (f("SQ_INSTS_VALU_ADD_F64") + f("SQ_INSTS_VALU_MUL_F64") + 2 * f("SQ_INSTS_VALU_FMA_F64")) * 64
entry = {
    "source": note,
    "kernel_full_name": r["kernel"],
    "segments": segments,
    "paths": paths,
    "fp64_flop_issued_per_unit": flop / segments,
    "hbm_bytes_per_unit": (2 * f("FETCH_SIZE") + f("WRITE_SIZE")) * 1024 / segments,
    "fetch_bytes_per_unit": 2 * f("FETCH_SIZE") * 1024 / segments,
    "write_bytes_per_unit": f("WRITE_SIZE") * 1024 / segments,
    "valu_busy_frac": f("SQ_ACTIVE_INST_VALU") * 4 / (f("GRBM_GUI_ACTIVE") / 8 * 1024),
    "mean_active_lanes": f("SQ_THREAD_CYCLES_VALU") / f("SQ_ACTIVE_INST_VALU"),
    "valu_insts_per_unit": f("SQ_INSTS_VALU") * 64 / segments / 64,
    "valu_wave_insts_per_wave_segment": f("SQ_INSTS_VALU") / (segments / 64),
    "fp64_share_of_valu_insts": fp64 / f("SQ_INSTS_VALU"),
    "trans_f64_share_of_valu_insts": f("SQ_INSTS_VALU_TRANS_F64") / f("SQ_INSTS_VALU"),
    "int32_share_of_valu_insts": f("SQ_INSTS_VALU_INT32") / f("SQ_INSTS_VALU"),
    "wait_frac_of_wave_cycles": f("SQ_WAIT_ANY") / f("SQ_WAVE_CYCLES"),
    "l2_hit_frac": f("TCC_HIT_sum") / f("TCC_REQ_sum") if f("TCC_REQ_sum") else None,
    "kernel_ms_in_counter_pass": f("GRBM_GUI_ACTIVE") / 8 / 2.4e6,
}
path = os.path.join(ROOT, "profiles", "pmc_model.json")
model = json.load(open(path)) if os.path.exists(path) else {}
model[entry_name] = entry
json.dump(model, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps(entry, indent=1))
