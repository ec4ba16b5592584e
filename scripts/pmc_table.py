#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc passes (one directory per pass) to a per-kernel table of mean
counter values per dispatch.  usage: pmc_table.py gpurun_out/pmc2_*"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void\s+", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name)


def main(dirs):
    table = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            per_dispatch = defaultdict(float)
            names = {}
            with open(path) as f:
                for row in csv.DictReader(f):
                    key = (row["Dispatch_Id"], row["Counter_Name"])
                    per_dispatch[key] += float(row["Counter_Value"])
                    names[row["Dispatch_Id"]] = short(row["Kernel_Name"])
            for (disp, counter), v in per_dispatch.items():
                table[names[disp]][counter].append(v)
    counters = sorted({c for k in table.values() for c in k})
    print("| kernel | " + " | ".join(counters) + " |")
    print("|---|" + "---|" * len(counters))
    for k in sorted(table):
        if not k.startswith(("dsm", "conv", "deconv", "basicblock", "volume", "soft", "corr", "absmax", "spp", "box")):
            continue
        cells = []
        for c in counters:
            v = table[k].get(c)
            cells.append("%.4g" % (sum(v) / len(v)) if v else "")
        print("| `%s` | " % k + " | ".join(cells) + " |")


if __name__ == "__main__":
    main(sys.argv[1:])
