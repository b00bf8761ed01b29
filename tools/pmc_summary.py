"""Folds the counter_collection.csv files of tools/gpu_pmc.sh into one JSON: per hand-written kernel
the resources (VGPR / SGPR / LDS / workgroup) and the mean value of every counter over its dispatches.
usage: pmc_summary.py gpurun_out/pmc_* > profiles/pmc_hand_written_kernels.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

OURS = ("anonymous namespace", "evox")          # kernels of libevox_hip.so live in anonymous namespaces


def main(dirs):
    acc = defaultdict(lambda: defaultdict(list))
    res = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    name = row["Kernel_Name"]
                    if not any(tag in name for tag in OURS) or "at::" in name or "rocprim" in name:
                        continue
                    short = name.replace("(anonymous namespace)::", "").replace("void ", "")
                    short = short.split("(")[0][:72]
                    acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
                    res[short] = {"workgroup": int(row["Workgroup_Size"]), "lds_bytes": int(row["LDS_Block_Size"]),
                                  "vgpr": int(row["VGPR_Count"]), "agpr": int(row["Accum_VGPR_Count"]),
                                  "sgpr": int(row["SGPR_Count"]), "scratch": int(row["Scratch_Size"]),
                                  "grid": int(row["Grid_Size"])}
    out = {}
    for k in sorted(acc):
        counters = {c: sum(v) / len(v) for c, v in sorted(acc[k].items())}
        entry = dict(res[k])
        entry["dispatches"] = max(len(v) for v in acc[k].values())
        entry["counters"] = counters
        if "SQ_LDS_BANK_CONFLICT" in counters and counters.get("SQ_LDS_IDX_ACTIVE"):
            entry["lds_bank_conflict_ratio"] = counters["SQ_LDS_BANK_CONFLICT"] / counters["SQ_LDS_IDX_ACTIVE"]
        if "TCC_HIT_sum" in counters:
            tot = counters["TCC_HIT_sum"] + counters.get("TCC_MISS_sum", 0.0)
            entry["l2_hit_ratio"] = counters["TCC_HIT_sum"] / tot if tot else None
        if "SQ_ACTIVE_INST_VALU" in counters and counters.get("SQ_WAVE_CYCLES"):
            entry["valu_active_share_of_wave_cycles"] = counters["SQ_ACTIVE_INST_VALU"] / counters["SQ_WAVE_CYCLES"]
        out[k] = entry
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
