"""Per-kernel averages of every counter found in one or more rocprofv3 --pmc CSV directories.
    python tools/pmc_table.py --filter project_mfma gpurun_out/pmc_a gpurun_out/pmc_b"""
import argparse
import csv
import glob
import os
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--filter", default="")
    a = ap.parse_args()
    table = defaultdict(dict)
    for d in a.dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = defaultdict(lambda: defaultdict(float))
            for row in csv.DictReader(open(path)):
                k = row.get("Kernel_Name", "?").replace("void spv::(anonymous namespace)::", "").split("(")[0]
                if a.filter and a.filter not in k:
                    continue
                per[(k, row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
            for (k, c), v in per.items():
                table[k][c] = sum(v.values()) / len(v)
    for k in sorted(table):
        print(k)
        for c in sorted(table[k]):
            print("    %-28s %.4g" % (c, table[k][c]))


if __name__ == "__main__":
    main()
