#!/usr/bin/env python3
"""After scripts/profile_all.sh: writes the committed summaries of gpurun_out/<prefix>* into a
profiles/ directory (counter summaries, rocprofv3's kernel statistics, the decoupled workload's
two-queue timeline, compiler-reported resource usage of every kernel).

    python scripts/refresh_profiles.py profiles/r03 [workload,workload...] [--resources] [--prefix=r3_prof_]
"""
import csv
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOADS = ("coupled_vel", "coupled_jerk", "coupled_ref", "replanning", "decoupled")


def timeline(src, dst):
    path = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))[0]
    rows = [r for r in csv.DictReader(open(path)) if "mpcb::" in r["Kernel_Name"] and "ordered" not in r["Kernel_Name"]
            and "dispatch_order" not in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # two consecutive steps from the middle of the run (three first-stage launches per step: the two
    # sub-batches of the X axis and the Y axis, each followed by its second-stage launch)
    zs = [i for i, r in enumerate(rows) if "zspace_kernel" in r["Kernel_Name"]]
    first = zs[9]
    sel = rows[first:first + 12]
    t0 = int(sel[0]["Start_Timestamp"])
    with open(dst, "w") as f:
        f.write("rocprofv3 --kernel-trace of `bench.py --workload decoupled` (4,096 X/Y pairs): two consecutive steps.\n"
                "Two queues (high priority) carry the halves of the X axis, the third the Y axis.  Times in "
                "microseconds from the first kernel shown.\nThe Y axis' second-stage launch has nothing to do (no "
                "hand-overs): it only waits for a compute unit's LDS.\n")
        for r in sel:
            name = re.sub(r"^.*mpcb::\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"\(.*$", "", name).replace("void ", "")
            s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
            f.write("%-44s queue %s   start %8.1f  end %8.1f  duration %7.1f\n" % (name, r["Queue_Id"], s, e, e - s))


def resource_usage(dst):
    out = []
    for src in sorted(glob.glob(os.path.join(ROOT, "mpc_amd", "csrc", "*.hip"))):
        p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c",
                            "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", src],
                           capture_output=True, text=True, cwd=os.path.dirname(src))
        fields = []
        for line in p.stderr.splitlines():
            m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis=kernel-resource-usage\]", line)
            if m:
                fields.append(m.group(1).strip())
        if fields:
            out.append("== %s" % os.path.basename(src))
            out += fields
    open(dst, "w").write("\n".join(out) + "\n")


def main():
    dst = sys.argv[1]
    prefix = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--prefix=")), "r3_prof_")
    os.makedirs(dst, exist_ok=True)
    for wl in (sys.argv[2].split(",") if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else WORKLOADS):
        src = os.path.join(ROOT, "gpurun_out", prefix + wl)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "summarize_profile.py"), src,
                               os.path.join(dst, wl + "_summary.json")])
        stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
        open(os.path.join(dst, wl + "_kernel_stats.csv"), "w").write(open(stats).read())
    if os.path.isdir(os.path.join(ROOT, "gpurun_out", prefix + "decoupled")):
        timeline(os.path.join(ROOT, "gpurun_out", prefix + "decoupled"), os.path.join(dst, "decoupled_timeline.txt"))
    if "--resources" in sys.argv:
        resource_usage(os.path.join(dst, "kernel_resource_usage.txt"))


if __name__ == "__main__":
    main()
