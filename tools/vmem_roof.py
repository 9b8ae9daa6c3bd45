"""The vector-memory roof of a CU under divergence (phip_debug_vmem_roof, mitsuba_amd/csrc/phip_debug.inl): lane-level 16-byte
requests per second and per clock and CU, by access pattern, working-set size and resident waves.  Writes a JSON summary
(-> profiles/r03_vmem_roof.json) that bench.py reads for roofline.vmem.

    python tools/vmem_roof.py out.json [quick]
"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitsuba_amd import _ffi

MODES = {0: "lane_random_16B", 1: "quad_shares_64B", 2: "8_lanes_share_128B", 3: "16_lanes_share_256B", 4: "lane_random_16B_32_of_64_lanes",
         5: "lane_random_16B_16_of_64_lanes", 6: "lane_random_80B_record_as_5_loads", 7: "lane_random_4B", 8: "lane_random_8B", 9: "coalesced_1KB",
         10: "lane_random_16B_8_of_64_lanes", 11: "lane_random_16B_4_of_64_lanes"}


def main():
    out = sys.argv[1]
    quick = len(sys.argv) > 2
    L = _ffi.lib()
    L.phip_debug_vmem_roof.argtypes = [C.c_int, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    n_cu, clock_ghz = 256, 2.4
    rows = []
    sizes = [("16KB_L1", 16 << 10), ("2MB_L2", 2 << 20), ("16MB_MALL", 16 << 20)]
    for name, size in sizes:
        for bpc in (4, 8):
            for mode in MODES:
                if quick and mode not in (0, 1, 4):
                    continue
                ms, loads = C.c_double(), C.c_double()
                iters = 2000 if mode != 6 else 400
                rc = L.phip_debug_vmem_roof(mode, size, bpc, iters, C.byref(ms), C.byref(loads))
                if rc != 0:
                    print("mode", mode, "failed:", _ffi.last_error()); continue
                rate = loads.value / (ms.value * 1e-3)
                row = {"pattern": MODES[mode], "mode": mode, "working_set": name, "waves_per_simd": bpc, "ms": round(ms.value, 4),
                       "lane_loads_per_s": rate, "lane_loads_per_clk_per_cu": round(rate / (n_cu * clock_ghz * 1e9), 4)}
                rows.append(row)
                print("%-36s %-10s %d waves/SIMD: %8.1f G lane-loads/s  %.3f per clk per CU" %
                      (MODES[mode], name, bpc, rate / 1e9, row["lane_loads_per_clk_per_cu"]))
    best = {}
    for r in rows:
        k = (r["pattern"], r["working_set"])
        if k not in best or r["lane_loads_per_s"] > best[k]["lane_loads_per_s"]:
            best[k] = r
    summary = {"device": "MI355X", "assumed_clock_GHz": clock_ghz, "n_cu": n_cu,
               "definition": "lane-level load requests (one active lane x one load instruction) per second, 4 independent loads per lane in flight",
               "peak_lane_random_16B": {ws: best[("lane_random_16B", ws)]["lane_loads_per_s"] for _, ws in [(0, s[0]) for s in sizes] if ("lane_random_16B", ws) in best},
               "rows": rows}
    json.dump(summary, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
