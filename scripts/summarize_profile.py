#!/usr/bin/env python3
"""Condenses a scripts/profile_gpu.sh output directory into one JSON summary.

    python scripts/summarize_profile.py gpurun_out/prof_x profiles/r01/x_summary.json

Per kernel: calls and average duration from the kernel trace, and every collected counter
averaged per launch.  HBM traffic = 2 x FETCH_SIZE + WRITE_SIZE (FETCH_SIZE on gfx950 tallies
128-byte requests at 64 bytes; both are reported in KiB by rocprofv3).
"""
import csv
import glob
import json
import os
import sys


def one_run_only(paths, where):
    """rocprofv3 names its files by process id, and gpurun merges a run's files into what is already
    there: two runs' files side by side would be condensed into one summary, the older winning or not
    by accident.  Delete the directory before profiling again."""
    if len(paths) > 1:
        raise SystemExit("%s holds the files of %d profiler runs (%s): delete it and profile again" %
                         (where, len(paths), ", ".join(os.path.basename(p) for p in paths)))


def main():
    src, dst = sys.argv[1], sys.argv[2]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mpc_amd.provenance import kernel_source_sha
    # the fingerprint of the kernel sources this profile was taken from (run this script on the
    # same tree that was profiled); bench.py only uses counter-derived figures of matching code
    out = {"source": os.path.basename(src.rstrip("/")), "kernel_source_sha": kernel_source_sha(), "kernels": {}}
    one_run_only(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), src)
    for d in glob.glob(os.path.join(src, "pmc_*")):
        one_run_only(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), d)
    for path in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
        for row in csv.DictReader(open(path)):
            name = row["Name"]
            if "mpcb::" not in name:
                continue
            k = out["kernels"].setdefault(short(name), {})
            k["calls"] = int(row["Calls"])
            k["avg_us"] = float(row["AverageNs"]) / 1e3
            k["min_us"] = float(row["MinNs"]) / 1e3
            k["max_us"] = float(row["MaxNs"]) / 1e3
    for path in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
        acc = {}
        meta = {}
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"]
            if "mpcb::" not in name:
                continue
            key = (short(name), row["Counter_Name"])
            s, n = acc.get(key, (0.0, set()))
            n.add(row["Dispatch_Id"])
            acc[key] = (s + float(row["Counter_Value"]), n)
            meta[short(name)] = {"vgpr": int(row["VGPR_Count"]), "agpr": int(row["Accum_VGPR_Count"]),
                                 "sgpr": int(row["SGPR_Count"]), "lds_block_bytes": int(row["LDS_Block_Size"]),
                                 "scratch_bytes": int(row["Scratch_Size"]), "grid": int(row["Grid_Size"]),
                                 "workgroup": int(row["Workgroup_Size"])}
        for (kname, cname), (s, n) in acc.items():
            k = out["kernels"].setdefault(kname, {})
            k.setdefault("counters_per_launch", {})[cname] = s / max(1, len(n))
            k["resources"] = meta[kname]
    for k in out["kernels"].values():
        c = k.get("counters_per_launch", {})
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            k["hbm_traffic_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        f = [c.get("SQ_INSTS_VALU_FMA_F64"), c.get("SQ_INSTS_VALU_ADD_F64"), c.get("SQ_INSTS_VALU_MUL_F64")]
        if None not in f:
            # wave-level instruction counts x 64 lanes; FMA = 2 flops
            k["executed_f64_flops_per_launch"] = 64.0 * (2.0 * f[0] + f[1] + f[2])
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


def short(name):
    name = name.replace("void ", "", 1).replace("mpcb::(anonymous namespace)::", "")
    return name.split("(")[0]


if __name__ == "__main__":
    main()
