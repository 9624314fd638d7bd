"""Sum rocprofv3 --pmc counter CSVs under gpurun_out/<dir>/ per kernel and counter.
usage: python profiles/aggregate_sq.py gpurun_out/pmc_sq1 [gpurun_out/pmc_sq2 ...]"""
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: [0.0, 0])
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
            agg[(k, r["Counter_Name"])][0] += float(r["Counter_Value"])
            agg[(k, r["Counter_Name"])][1] += 1
ks = sorted({k for k, _ in agg})
for k in ks:
    print(k, {c: round(v[0] / v[1]) for (kk, c), v in sorted(agg.items()) if kk == k})
