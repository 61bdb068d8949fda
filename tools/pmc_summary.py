#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSVs (one directory per pass under <out>/pmc_*) per kernel: mean counter value per dispatch.

    python tools/pmc_summary.py gpurun_out/<tag>  ->  prints a table and writes <out>/pmc_summary.json

Derived figures (MI355X_MICROARCH.md: SQ_* wave counters are in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles; FETCH_SIZE
under-reports wide coalesced reads by 2x on gfx950 -- both the raw and the doubled figure are printed):
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * SQ_BUSY_CYCLES-per-CU) is approximated per dispatch as
              SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE-equivalent) when both are present; otherwise the raw counters are shown.
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    out = sys.argv[1]
    prefix = sys.argv[2] if len(sys.argv) > 2 else "pmc"      # directory prefix of the passes: pmc_* (fp32 line) or pmcs_* (split tier)
    # which kernels to keep: the frame loop's (default) or, with a third argument "train", the training step's as well
    keep = ("k_head", "k_torso", "k_frame", "k_cond")
    if len(sys.argv) > 3 and sys.argv[3] == "train":
        keep += ("k_field", "k_grid", "k_wgrad", "k_composite", "k_train", "k_march")
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for path in glob.glob(os.path.join(out, prefix + "_*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                k = k.replace("(anonymous namespace)::", "").split("(")[0]
                k = k[5:] if k.startswith("void ") else k
                k = k.replace("k_head_phase<false>", "k_head_phase").replace("k_head_phase<true>", "k_head_phase_fast").replace("k_head_phase<0>", "k_head_phase").replace("k_head_phase<1>", "k_head_phase_fast").replace("k_head_phase<2>", "k_head_phase_split")
                per[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                dur[(k, row["Dispatch_Id"], path)] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    summary = {}
    for k, ctrs in per.items():
        if not any(n in k for n in keep):
            continue
        d = {name: sum(v) / len(v) for name, v in ctrs.items()}
        d["dispatches"] = max(len(v) for v in ctrs.values())
        durs = [v for (kk, _, _), v in dur.items() if kk == k]
        d["mean_duration_us"] = sum(durs) / len(durs) / 1e3
        if "FETCH_SIZE" in d:
            d["fetch_MB_raw"] = d["FETCH_SIZE"] * 1024 / 1e6 if d["FETCH_SIZE"] < 1e9 else d["FETCH_SIZE"] / 1e6
            d["fetch_MB_x2"] = 2 * d["fetch_MB_raw"]
        if "WRITE_SIZE" in d:
            d["write_MB_raw"] = d["WRITE_SIZE"] * 1024 / 1e6 if d["WRITE_SIZE"] < 1e9 else d["WRITE_SIZE"] / 1e6
        if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d and d["TCC_HIT_sum"] + d["TCC_MISS_sum"] > 0:
            d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
        if "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"] > 0:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if n in d:
                    d[n + "/WAVE_CYCLES"] = d[n] / d["SQ_WAVE_CYCLES"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d and d["SQ_BUSY_CYCLES"] > 0:
            d["mfma_busy_over_sq_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CYCLES"]
        summary[k] = d
    # stamp the summary with the digest of the kernel sources it was collected on: bench.py marks `traffic_stale` when the tree has moved on
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from geneface_amd.csrc.build import source_digest
        summary["_source_digest"] = source_digest()
    except Exception as e:      # noqa: BLE001
        summary["_source_digest"] = None
        print("pmc_summary: no source digest:", e)
    for k, d in summary.items():
        if not isinstance(d, dict):
            continue
        print(f"== {k}")
        for n, v in sorted(d.items()):
            print(f"   {n:36s} {v:18.4f}")
    with open(os.path.join(out, "pmc_summary.json" if prefix == "pmc" else f"{prefix}_summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1)


if __name__ == "__main__":
    main()
