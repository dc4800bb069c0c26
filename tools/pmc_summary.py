"""Summarise rocprofv3 counter passes of the bench into the per-kernel table of profiles/r01_pmc_*.json.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
              SQ_VALU_MFMA_BUSY_CYCLES -d OUT/sq -o p --output-format csv -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d OUT/fetch -o p --output-format csv -- python3 bench.py ...   (own pass)
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d OUT/write -o p --output-format csv -- python3 bench.py ...   (own pass)
    python tools/pmc_summary.py [--workload "text"] OUT/sq/p_counter_collection.csv [OUT/fetch/... OUT/write/...] > summary.json

The summary carries `lib_srchash` = the source hash of the library that is built in this tree (cips_3dplusplus_amd/build.py's
stamp): bench.py replays HBM traffic from a summary only when that hash is the loaded library's.

Derived columns: clock = GRBM_GUI_ACTIVE / 8 XCDs / duration (meaningful for launches of >~20 us only); MFMA-pipe busy =
SQ_VALU_MFMA_BUSY_CYCLES / (clock cycles x 256 CUs x 4 SIMDs); wave fractions relative to SQ_WAVE_CYCLES; FETCH_SIZE (KB)
doubled per the gfx950 note of MI355X_MICROARCH.md (128-byte requests counted at 64 B), WRITE_SIZE as reported."""
import collections
import csv
import json
import re
import sys


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"^void ", "", k).split("(")[0]
        # one row per (kernel, grid): launches of one kernel on different shapes (the 512->512 chain GEMM and the half-grid
        # 512->256 exit) must not share a mean
        k = (k, int(r["Grid_Size"]))
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["_dur"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return agg


def mean(v):
    return sum(v) / len(v)


def lib_srchash():
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from cips_3dplusplus_amd import build
    with open(build.STAMP) as fh:
        return fh.read().strip()


def main(paths):
    workload = "bench.py defaults (1024^2, D=2, N=24, batch 1)"
    if paths and paths[0] == "--workload":
        workload, paths = paths[1], paths[2:]
    out = {}
    for p in paths:
        for k, v in load(p).items():
            e = out.setdefault(k, {"kernel": k[0], "grid": k[1]})
            e.setdefault("avg_us_under_pmc", round(mean(v["_dur"]) / 1e3, 1))
            e.setdefault("launches", len(v["_dur"]))
            if "GRBM_GUI_ACTIVE" in v:
                cyc = mean(v["GRBM_GUI_ACTIVE"]) / 8
                e["launches"] = len(v["GRBM_GUI_ACTIVE"])
                e["avg_us_under_pmc"] = round(mean(v["_dur"]) / 1e3, 1)
                e["clock_GHz"] = round(cyc / mean(v["_dur"]), 2)
                e["mfma_busy_frac_of_simd_cycles"] = round(mean(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / (cyc * 1024), 3)
                wc = mean(v["SQ_WAVE_CYCLES"])
                e["wave_cycles_wait_any"] = round(mean(v["SQ_WAIT_ANY"]) / wc, 2)
                e["wave_cycles_wait_inst"] = round(mean(v["SQ_WAIT_INST_ANY"]) / wc, 2)
                e["wave_cycles_active"] = round(mean(v["SQ_ACTIVE_INST_ANY"]) / wc, 2)
            if "FETCH_SIZE" in v:
                e["hbm_fetch_MB_x2"] = round(2 * mean(v["FETCH_SIZE"]) * 1024 / 1e6, 1)
            if "WRITE_SIZE" in v:
                e["hbm_write_MB"] = round(mean(v["WRITE_SIZE"]) * 1024 / 1e6, 1)
    ks = sorted((e for e in out.values() if e.get("avg_us_under_pmc", 0) >= 3.0), key=lambda e: -e["avg_us_under_pmc"])
    json.dump({"source": "tools/pmc_summary.py over rocprofv3 --kernel-trace --pmc passes of bench.py --steps 12 --warmup 3 "
                         "--no-cpu-baseline (SQ/GRBM set, FETCH_SIZE, WRITE_SIZE: three separate passes)",
               "workload": workload, "lib_srchash": lib_srchash(), "kernels": ks},
              sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
