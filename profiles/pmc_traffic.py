#!/usr/bin/env python3
"""HBM-side traffic per kernel class and per step from two rocprofv3 counter runs (FETCH_SIZE, WRITE_SIZE; they
cannot share a run), as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE counts wide
(16 B per lane) streaming reads at half their bytes -> doubled; WRITE_SIZE as is.  The doubling is for THAT access
pattern only (round 4, the review's item 7): classes whose loads are 16-byte-per-lane streams (STREAMING below) are
doubled; for the others -- k_scan_short's 1-4-byte field loads, k_reduce's payload gathers, the splitter look-ups of
k_bs_assign -- nobody knows the factor, so they are reported as a RANGE: lower = FETCH_SIZE + WRITE_SIZE, upper =
2 x FETCH_SIZE + WRITE_SIZE.  step_bytes (what bench.py prints as roofline.traffic) is the upper end, step_bytes_lower
the lower.  Both counters are L2 <-> fabric requests: Infinity-Cache hits are included, so for working sets below
256 MiB even the lower end bounds HBM traffic from above.

usage: pmc_traffic.py WORKLOAD STEPS fetch_dir write_dir [traffic.json]
  STEPS = steps the profiled bench.py ran in all (warm-up + 1 probe + timed + 3 breakdown)
Updates traffic.json[WORKLOAD] = {class: bytes per STEP (all launches of the class), "step_bytes": ...,
"launches_per_step": ...}; per-launch figures are in traffic.json["detail"]."""
import collections
import csv
import glob
import json
import re
import sys

CLASS = [
    (r"k_scan_short|k_scan_long|k_scan_rest|k_scan_ops|k_close|k_cigar_tail", "k_scan"),
    (r"k_scatter", "k_scatter"), (r"k_hist", "k_hist"), (r"k_binscan", "k_binscan"),
    (r"k_bs_leaf", "k_bs_leaf"), (r"k_bs_assign", "k_bs_assign"), (r"k_bs_sample|k_bs_sort2", "k_bs_sample"), (r"k_bs_bounds", "k_bs_bounds"),
    (r"k_reduce<1>", "k_reduce_a"), (r"k_reduce", "k_reduce"),
    (r"k_count_heads", "k_count_heads"), (r"k_write_heads|k_init_rows", "k_write_heads"),
    (r"k_scan_sums|k_scan_top|k_scan_apply|k_scan_small", "scan_u32"),
    (r"k_emit_|k_finish_|k_motif|k_bins|k_rank_|k_publish|k_set_counters|fillBuffer|copyBuffer", "small"),
]


# classes whose global loads are 16 bytes per lane, streaming (the guide's case): k_scatter / k_bs_leaf / k_count_heads read
# items, k_hist reads sixteen digit bytes per lane, the scans of long CIGARs refill their rings in 16-byte loads
STREAMING = {"k_scatter", "k_hist", "k_bs_leaf", "k_count_heads", "k_bs_bounds"}
STREAMING_BY_WORKLOAD = {"L": {"k_scan"}}


def klass(name):
    for pat, c in CLASS:
        if re.search(pat, name):
            return c
    return "other"


def load(d, counter):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        raise SystemExit("no counter_collection.csv under " + d)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in fs:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = agg[klass(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return agg


def main():
    wl, steps, fdir, wdir = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    path = sys.argv[5] if len(sys.argv) > 5 else "profiles/traffic.json"
    fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    out, step_bytes, step_lower, launches = {}, 0.0, 0.0, 0
    detail = {}
    streaming = STREAMING | STREAMING_BY_WORKLOAD.get(wl, set())
    for c in sorted(set(fetch) | set(write)):
        fkb, fn = fetch.get(c, [0.0, 0])
        wkb, wn = write.get(c, [0.0, 0])
        n = max(fn, wn, 1)
        upper = fkb * 1024 * 2 + wkb * 1024          # FETCH_SIZE doubled (gfx950), both in KB
        lower = upper if c in streaming else fkb * 1024 + wkb * 1024
        out[c] = upper / steps
        detail[c] = {"launches_per_step": n / steps, "fetch_kb_per_launch": fkb / n, "write_kb_per_launch": wkb / n,
                     "bytes_per_step": upper / steps, "bytes_per_step_lower": lower / steps,
                     "loads": "16 B per lane, streaming (FETCH_SIZE doubled)" if c in streaming else "mixed (a range)"}
        step_bytes += upper / steps
        step_lower += lower / steps
        launches += n
    out["step_bytes"] = step_bytes
    out["step_bytes_lower"] = step_lower
    out["launches_per_step"] = launches / steps
    try:
        t = json.load(open(path))
    except Exception:
        t = {}
    # the kernel sources these counters were measured on: bench.py reports the traffic only while they are unchanged.
    # A file measured on other sources is started afresh (workloads measured on different builds must not mix).
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_hash
    stamp = kernel_source_hash()
    if t.get("kernel_sources_sha16") != stamp:
        t = {}
    t["kernel_sources_sha16"] = stamp
    t[wl] = out
    t.setdefault("detail", {})[wl] = detail
    t["note"] = ("per class: bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB counters; FETCH doubled per MI355X_MICROARCH.md, gfx950, which holds "
                 "for 16-byte-per-lane streaming loads: detail.<class>.loads says which classes those are; for the others "
                 "bytes_per_step_lower = FETCH_SIZE + WRITE_SIZE is the other end of the range); L2 <-> fabric including Infinity-Cache "
                 "hits; per STEP by kernel class (k_scan = scan kernels + their second stage + stripe closing); step_bytes / "
                 "step_bytes_lower = all launches of a step, upper / lower end")
    json.dump(t, open(path, "w"), indent=1)
    print(json.dumps({wl: out}, indent=1))


if __name__ == "__main__":
    main()
