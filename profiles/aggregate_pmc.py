import csv, glob, collections, json, sys
out = {}
for tag in ("fetch", "write"):
    fs = glob.glob("gpurun_out/pmc_%s/*/*counter_collection.csv" % tag)
    if not fs: continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        agg[(k, r["Counter_Name"])][0] += float(r["Counter_Value"])
        agg[(k, r["Counter_Name"])][1] += 1
    for (k, c), (v, n) in agg.items():
        out.setdefault(k, {})[c] = {"sum": v, "dispatches": n, "per_dispatch": v / n}
json.dump(out, open("gpurun_out/pmc_traffic_raw.json", "w"), indent=1)
for k, v in out.items():
    print(k[:60], {c: round(x["per_dispatch"]) for c, x in v.items()})
