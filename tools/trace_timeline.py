"""Print the dispatch timeline of a slice of a rocprofv3 --kernel-trace CSV (dev tool)."""
import csv, sys, glob
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 200
span_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 1.2
rows = list(csv.DictReader(open(path)))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
fp = [r for r in rows if "cbf_first_pass" in r["Kernel_Name"]]
print("first passes:", len(fp), "dispatches:", len(rows))
t0 = fp[min(skip, len(fp) - 1)]["s"]
for r in rows:
    if t0 <= r["s"] < t0 + span_ms * 1e6:
        n = r["Kernel_Name"].split("(")[0][-16:]
        print("%-16s q%s @%.3f dur %.3f grid %s" % (n, r["Queue_Id"], (r["s"] - t0) / 1e6, (r["e"] - r["s"]) / 1e6, r["Grid_Size_X"]))
