"""Sum one PMC counter per kernel from a rocprofv3 --pmc output directory (dev tool)."""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != sys.argv[2]:
            continue
        k = r["Kernel_Name"].split("(")[0]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items(), key=lambda x: -x[1][1])[:8]:
    print(f"{sys.argv[2]} {k:40s} launches {n:6d} sum {v/1e6:12.3f} GB (KB units) per launch {v/n/1e3:10.3f} MB")
