#!/usr/bin/env python3
"""rocprofv3's rocpd database (-o NAME without --output-format csv) -> the per-kernel table that
`rocprofv3 --kernel-trace --stats` prints as CSV: name, calls, total / average / min / max duration (ns), share.
usage: db_kernel_stats.py results.db [out.csv]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*$", "", name)


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1
    lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append('"%s",%d,%d,%.1f,%.2f,%d,%d' % (k, a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot, a[2], a[3]))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    else:
        sys.stdout.write(out)


if __name__ == "__main__":
    main()
