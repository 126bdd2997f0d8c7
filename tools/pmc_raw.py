"""Raw per-(kernel, grid) averages of every counter in a rocprofv3 counter_collection directory.  Usage: pmc_raw.py DIR [--match substr] (default: gemm)"""
import csv, glob, os, re, sys
from collections import defaultdict
csv.field_size_limit(1 << 30)
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else "gemm"
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if match not in r["Kernel_Name"]:
            continue
        m = re.match(r"(?:void )?([A-Za-z0-9_]+)(<[^(]*>)?", r["Kernel_Name"])
        agg[(m.group(1) + (m.group(2) or ""), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (name, grid), c in sorted(agg.items()):
    print(f"{name} grid={grid}: " + "  ".join(f"{k}={sum(v) / len(v):.4g}" for k, v in sorted(c.items())))
