#!/usr/bin/env python3
"""VALU-issue and lane-utilisation figures of the path_hip kernels from rocprofv3 SQ counters.

    python tools/pmc_valu.py <dir with *_counter_collection.csv> <prefix> [out.json]

Per kernel (summed over its launches in the profiled run; durations are the dispatch timestamps of the SAME run):
  valu_issue_frac = SQ_INSTS_VALU * 2 cycles / (1024 SIMD-32 * GPU cycles)      -- how busy the vector ALUs' issue slots were
  lane_util       = SQ_THREAD_CYCLES_VALU / (64 * SQ_INSTS_VALU)                -- active lanes per issued VALU instruction
  valu_frac       = SQ_THREAD_CYCLES_VALU / (256 * 4 * 32 lanes * GPU cycles)   -- useful lane-operations / lane-slots available
  wait / stall    = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY as fractions of SQ_WAVE_CYCLES (disjoint, sum ~ 1)
GPU cycles of a launch = its duration x the shader clock measured in the same pass (GRBM_GUI_ACTIVE / 8 XCDs / duration when
the counter is there, else 2.4 GHz).  Cross-check built in: avg_waves_per_simd must come out at the kernel's resident
waves per SIMD (k_mega: 3.0 = its launch bounds)."""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]; pre = sys.argv[2] if len(sys.argv) > 2 else ""
out = sys.argv[3] if len(sys.argv) > 3 else None
agg = collections.defaultdict(lambda: collections.defaultdict(float))       # kernel -> counter -> sum over launches
src = collections.defaultdict(dict)                                           # kernel -> counter -> file it came from
dur = collections.defaultdict(lambda: collections.defaultdict(dict))          # kernel -> file -> {dispatch: ns}
meta = {}
for f in sorted(glob.glob(os.path.join(d, pre + "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("k_"):
            continue
        name = r["Counter_Name"]
        if src[k].setdefault(name, f) != f:
            continue                                  # a counter listed in several passes: the first pass counts
        agg[k][name] += float(r["Counter_Value"])
        dur[k][f][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        meta[k] = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "lds_block": int(r["LDS_Block_Size"]), "scratch": int(r["Scratch_Size"])}
res = {}
for k, c in sorted(agg.items()):
    def base(counter):
        """(seconds, launches) of the pass that collected `counter`"""
        f = src[k].get(counter)
        if f is None:
            return 0.0, 0
        return sum(dur[k][f].values()) * 1e-9, len(dur[k][f])
    sec, launches = base("SQ_INSTS_VALU") if "SQ_INSTS_VALU" in c else base(next(iter(c)))
    gsec, _ = base("GRBM_GUI_ACTIVE")
    clock = c["GRBM_GUI_ACTIVE"] / gsec if gsec else 2.4e9
    if clock > 4e9:
        clock /= 8.0              # rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs (measured: 19.0 "GHz" = 8 x 2.38); the SQ counters are chip-wide sums
    cycles = sec * clock
    insts, thr = c.get("SQ_INSTS_VALU", 0.0), c.get("SQ_THREAD_CYCLES_VALU", 0.0)
    wave = c.get("SQ_WAVE_CYCLES", 0.0)
    e = {"launches": launches, "total_ms_in_profiled_run": round(sec * 1e3, 3), "avg_launch_us": round(sec * 1e6 / max(launches, 1), 2),
         "shader_clock_GHz": round(clock * 1e-9, 3), **meta[k]}
    if insts and cycles:
        e["valu_insts_per_launch"] = round(insts / launches, 1)
        e["valu_issue_frac"] = round(insts * 2.0 / (1024.0 * cycles), 4)
        e["lane_util"] = round(thr / (64.0 * insts), 4)
        e["valu_frac"] = round(thr / (256.0 * 4 * 32 * cycles), 4)
    if wave:
        wsec, _ = base("SQ_WAVE_CYCLES")
        e["wave_cycles_wait_frac"] = round(c.get("SQ_WAIT_ANY", 0.0) / wave, 4)
        e["wave_cycles_issue_stall_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / wave, 4)
        e["wave_cycles_active_frac"] = round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 4)
        e["avg_waves_per_simd"] = round(wave * 4.0 / (1024.0 * wsec * clock), 3) if wsec else None      # SQ_WAVE_CYCLES counts quad-cycles
    for name in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_BRANCH", "SQ_WAVES", "SQ_INSTS_SMEM"):
        if name in c:
            n = base(name)[1]
            e[name.lower() + "_per_launch"] = round(c[name] / max(n, 1), 1)
    res[k] = e
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitsuba_amd import _ffi as _ffi_id  # noqa: E402
res["build_id"] = _ffi_id.built_id(os.environ.get("PHIP_LIB"))       # the library the counters were taken on (bench.py flags a mismatch)
txt = json.dumps(res, indent=1)
print(txt)
if out:
    open(out, "w").write(txt + "\n")
