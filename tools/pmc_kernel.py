"""Mean PMC counter values per kernel from a rocprofv3 --pmc ... --output-format csv run.
python tools/pmc_kernel.py <dir> [substring of the kernel name]"""
import csv
import glob
import sys
from collections import defaultdict

d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if sub in k:
            acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} n={len(v):4d} mean={sum(v) / len(v):.6g}")
