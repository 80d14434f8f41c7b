"""Mean counter value per kernel launch from rocprofv3 --pmc csv output: tools/pmc_summary.py <dir>..."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            print(f"{k:60s} {c:28s} n={len(v):3d} mean={sum(v) / len(v):.4g}")
