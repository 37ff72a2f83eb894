"""Per-kernel, per-launch-geometry averages of a rocprofv3 --pmc run: python scripts/pmc_by_grid.py <dir> [kernel substring]"""
import collections, csv, glob, re, sys
d = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else ""
rows = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(\w+_kernel)', r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:30]
        if want and want not in k: continue
        key = (k, int(r["Grid_Size"]))
        rows[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(key, r["Counter_Name"])] += 1
        dur[(key, r["Counter_Name"])] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for key, v in sorted(rows.items()):
    c0 = next(iter(v)); n = cnt[(key, c0)]
    print("%s grid=%d  n=%d  avg %.1f us" % (key[0], key[1], n, dur[(key, c0)] / n / 1e3))
    for c, val in sorted(v.items()):
        print("   %-26s %14.6g" % (c, val / cnt[(key, c)]))
