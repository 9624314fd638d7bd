"""Per-launch duration and FETCH_SIZE of the long-read scan kernels from profiles/long_occupancy_experiment.sh"""
import csv, glob, sys, collections
for d in sorted(glob.glob("gpurun_out/occ_v*_pad*")):
    if not d.split("/")[-1].startswith("occ_") or d.endswith(".log"):
        continue
    dur = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "k_scan_long" in r["Kernel_Name"]:
                dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    fetch = collections.defaultdict(float); calls = collections.defaultdict(set)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_scan_long" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                k = r["Kernel_Name"].split("(")[0]
                fetch[k] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
    for k in dur:
        n = max(1, len(calls[k]))
        print(d.split("/")[-1], k.split("::")[-1], "launches", len(dur[k]), "avg ms %.3f" % (sum(dur[k]) / len(dur[k])),
              "FETCH_SIZE per launch %.2f GB (KB counter x 1024; x2 for 128-B requests: %.2f GB)" % (fetch[k] * 1024 / n / 1e9, 2 * fetch[k] * 1024 / n / 1e9))
