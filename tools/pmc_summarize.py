#!/usr/bin/env python
"""Average PMC counter values per dispatch of kernels matching a substring, from rocprofv3 CSV output dirs."""
import csv, glob, sys, collections
root, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            per[(row["Dispatch_Id"], row["Kernel_Name"][:60])][row["Counter_Name"]] += float(row["Counter_Value"])
    for (d, k), cs in per.items():
        for c, v in cs.items():
            acc[(k, c)].append(v)
for (k, c), vs in sorted(acc.items()):
    print("%-62s %-28s n=%3d avg=%16.1f" % (k, c, len(vs), sum(vs) / len(vs)))
