"""Per-kernel table from rocprofv3 CSV output: average duration per dispatch (--kernel-trace pass)
and fabric bytes per dispatch (two separate --pmc passes: FETCH_SIZE, WRITE_SIZE), corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (KiB units; FETCH_SIZE counts 128-B
requests at 64 B, so the read side is doubled -- an upper bound for narrower accesses).

    python tools/pmc_kernels.py --stats-dir D1 --fetch-dir D2 --write-dir D3 --out profiles/r02_paths_pmc.md \
        --title "..." [--filter spv::]
"""
import argparse
import csv
import glob
import os
from collections import defaultdict


def find(dirname, pattern):
    hits = glob.glob(os.path.join(dirname, "**", pattern), recursive=True) if dirname else []
    return hits[0] if hits else None


def short(name):
    name = name.replace("void spv::(anonymous namespace)::", "").replace("spv::(anonymous namespace)::", "")
    return name.split("(")[0][:70]


def durations(stats_dir):
    path = find(stats_dir, "*kernel_trace.csv")
    agg = defaultdict(lambda: [0, 0.0])
    if path:
        for row in csv.DictReader(open(path)):
            k = short(row.get("Kernel_Name", "?"))
            agg[k][0] += 1
            agg[k][1] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
    return agg


def counter(pmc_dir, name):
    path = find(pmc_dir, "*counter_collection.csv")
    per = defaultdict(lambda: defaultdict(float))
    if path:
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") == name:
                per[short(row.get("Kernel_Name", "?"))][row.get("Dispatch_Id")] += float(row["Counter_Value"])
    return {k: sum(v.values()) / len(v) for k, v in per.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats-dir")
    ap.add_argument("--fetch-dir")
    ap.add_argument("--write-dir")
    ap.add_argument("--out", required=True)
    ap.add_argument("--title", default="rocprofv3 per-kernel summary")
    ap.add_argument("--filter", default="", help="keep kernels whose full name contains this")
    a = ap.parse_args()
    dur = durations(a.stats_dir)
    fetch = counter(a.fetch_dir, "FETCH_SIZE")
    write = counter(a.write_dir, "WRITE_SIZE")
    names = sorted(set(dur) | set(fetch) | set(write), key=lambda k: -dur.get(k, [0, 0.0])[1])
    with open(a.out, "w") as f:
        f.write("# %s\n\n" % a.title)
        f.write("Average per dispatch.  Fabric bytes: FETCH_SIZE x 1024 x 2 (gfx950: 128-B requests tallied at 64 B) "
                "and WRITE_SIZE x 1024, separate `--pmc` passes.\n\n")
        f.write("| kernel | dispatches | avg ms | fetched MiB | written MiB | (fetched + written) / avg ms, GB/s |\n|---|---|---|---|---|---|\n")
        for k in names:
            n, tot = dur.get(k, [0, 0.0])
            ms = tot / n if n else float("nan")
            fb = fetch.get(k, float("nan")) * 1024 * 2
            wb = write.get(k, float("nan")) * 1024
            rate = (fb + wb) / (ms * 1e-3) / 1e9 if n and ms > 0 else float("nan")
            f.write("| `%s` | %d | %.4f | %.1f | %.1f | %.0f |\n" % (k, n, ms, fb / 2**20, wb / 2**20, rate))
    print(open(a.out).read())


if __name__ == "__main__":
    main()
