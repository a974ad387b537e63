"""Sum rocprofv3 --pmc counter_collection CSVs per (kernel, counter).
    python tools/pmc_summary.py <dir with *_counter_collection.csv> > summary.csv"""
import csv
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0, 0.0])
for fn in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc[(name, r["Counter_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
print("kernel,counter,dispatch_rows,sum")
for (k, c), (n, v) in sorted(acc.items()):
    print('"%s",%s,%d,%s' % (k, c, n, repr(v) if v != int(v) else int(v)))
