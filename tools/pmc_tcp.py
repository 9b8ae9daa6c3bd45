#!/usr/bin/env python3
"""Turns the vector-memory counter passes of tools/pmc_mem.sh into a per-kernel summary (-> profiles/r03_tcp_<workload>.json):

    python tools/pmc_tcp.py <dir> <tag> <out.json> [vmem_roof.json]

Per launch of every kernel: vector-memory wave instructions, L1 tag lookups (distinct 128-byte lines) per instruction, L1 / L2 hit
rates, the share of the launch the CU's texture-data unit is busy, and `issue_frac` = wave instructions x 16 clk (what one
dwordx4 wave instruction occupies the data-return path for whatever its active lanes -- measured by tools/vmem_roof.py: 64 B/clk/CU)
over the CU-cycles of the launch.  Durations come from the kernel trace of the same pass; the shader clock from GRBM_GUI_ACTIVE."""
import collections, csv, glob, json, os, sys

N_CU, N_XCD = 256, 8


def main():
    d, tag, out = sys.argv[1], sys.argv[2], sys.argv[3]
    ctr = collections.defaultdict(lambda: collections.defaultdict(float))     # kernel -> counter -> sum over launches
    launches = collections.defaultdict(lambda: collections.defaultdict(set))
    dur = collections.defaultdict(list)
    gui = collections.defaultdict(list)                                        # kernel -> [(dispatch key, GRBM_GUI_ACTIVE)]
    for f in sorted(glob.glob(os.path.join(d, tag + "_m*counter_collection.csv"))):
        key = os.path.basename(f).split("_counter")[0]
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            if not k.startswith("k_"):
                continue
            ctr[k][r["Counter_Name"]] += float(r["Counter_Value"]); launches[k][r["Counter_Name"]].add((key, r["Dispatch_Id"]))
    for f in sorted(glob.glob(os.path.join(d, tag + "_m*kernel_trace.csv"))):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            if k.startswith("k_"):
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    res = {}
    for k, c in ctr.items():
        per = {n: v / max(1, len(launches[k][n])) for n, v in c.items()}
        if not dur[k] or "TA_FLAT_WAVEFRONTS_sum" not in per:
            continue
        t = sum(dur[k]) / len(dur[k])
        clk = per.get("GRBM_GUI_ACTIVE", 0) / N_XCD                            # shader cycles of one launch (the counter sums the XCDs)
        cu_clk = clk * N_CU
        inst = per["TA_FLAT_WAVEFRONTS_sum"]
        lines = per.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0)
        r = {"launches": len(dur[k]), "avg_launch_ms": round(t * 1e3, 4), "shader_clock_GHz": round(clk / t / 1e9, 3) if t else None,
             "vmem_wave_instructions": inst, "clk_per_wave_instruction_per_cu": round(cu_clk / inst, 2) if inst else None,
             "l1_tag_lookups": lines, "lines_per_instruction": round(lines / inst, 2) if inst else None,
             "l1_miss_requests": per.get("TCP_TCC_READ_REQ_sum"), "l1_hit_rate": round(1 - per.get("TCP_TCC_READ_REQ_sum", 0) / lines, 4) if lines else None,
             "l2_requests": per.get("TCC_REQ_sum"), "l2_hit_rate": round(per.get("TCC_HIT_sum", 0) / per["TCC_REQ_sum"], 4) if per.get("TCC_REQ_sum") else None,
             "td_busy_frac": round(per.get("TD_TD_BUSY_sum", 0) / cu_clk, 4) if cu_clk else None,
             "ta_busy_frac": round(per.get("TA_TA_BUSY_sum", 0) / cu_clk, 4) if cu_clk else None,
             "tcp_pending_stall_frac": round(per.get("TCP_PENDING_STALL_CYCLES_sum", 0) / cu_clk, 4) if cu_clk else None,
             "avg_l2_round_trip_clk": round(per.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / per["TCP_TCC_READ_REQ_sum"], 1) if per.get("TCP_TCC_READ_REQ_sum") else None,
             "issue_frac": round(inst * 16.0 / cu_clk, 4) if cu_clk else None}
        res[k] = r
    roof = None
    if len(sys.argv) > 4 and os.path.exists(sys.argv[4]):
        roof = json.load(open(sys.argv[4])).get("peak_lane_random_16B")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mitsuba_amd import _ffi as _ffi_id
    json.dump({"build_id": _ffi_id.built_id(os.environ.get("PHIP_LIB")), "workload": tag, "source": "rocprofv3 --kernel-trace --pmc (tools/pmc_mem.sh), MI355X", "definition": __doc__, "vmem_roof_lane_loads_per_s": roof, "kernels": res},
              open(out, "w"), indent=1)
    for k, r in sorted(res.items(), key=lambda kv: -kv[1]["avg_launch_ms"] * kv[1]["launches"]):
        print(k, {x: r[x] for x in ("avg_launch_ms", "vmem_wave_instructions", "clk_per_wave_instruction_per_cu", "lines_per_instruction", "l1_hit_rate", "l2_hit_rate", "td_busy_frac", "issue_frac")})


if __name__ == "__main__":
    main()
