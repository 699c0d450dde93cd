"""Prints the last kernels of a rocprofv3 --kernel-trace run (CSV directory given): queue, start, end, duration in us."""
import csv, glob, sys
p = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(p)) if "mpcb::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t0 = int(rows[-k]["Start_Timestamp"])
for r in rows[-k:]:
    n = r["Kernel_Name"]
    n = n[n.find("mpcb::(anonymous namespace)::") + 29:][:28]
    print("%-28s q%s %9.1f %9.1f %8.1f" % (n, r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3,
                                         (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
