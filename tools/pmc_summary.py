"""Turns rocprofv3 CSV output (kernel trace + separate --pmc passes) into the per-round
summaries committed under profiles/.

    python tools/pmc_summary.py --stats-dir gpurun_out/prof_stats --fetch-dir gpurun_out/prof_fetch \
        --write-dir gpurun_out/prof_write --tag r01 --xrows 262144 --yrows 262144 --dim 128

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are in KiB-units (x1024); on gfx950 FETCH_SIZE reports half of the bytes of a wide
coalesced streaming read, so the read side is doubled (upper bound for narrower accesses)."""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(dirname, pattern):
    hits = glob.glob(os.path.join(dirname, "**", pattern), recursive=True)
    return hits[0] if hits else None


def kernel_stats(stats_dir):
    path = find(stats_dir, "*kernel_trace.csv")
    if not path:
        return {}
    agg = defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(path)):
        name = row.get("Kernel_Name", "?")
        dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
        agg[name][0] += 1
        agg[name][1] += dur
    return {k: {"calls": v[0], "total_ms": v[1], "avg_ms": v[1] / v[0]} for k, v in agg.items()}


def counter_sum(pmc_dir, counter, kernel_substr):
    path = find(pmc_dir, "*counter_collection.csv")
    if not path:
        return None, 0
    per_dispatch = defaultdict(float)
    for row in csv.DictReader(open(path)):
        if row.get("Counter_Name") != counter or kernel_substr not in row.get("Kernel_Name", ""):
            continue
        per_dispatch[row.get("Dispatch_Id")] += float(row["Counter_Value"])
    if not per_dispatch:
        return None, 0
    return sum(per_dispatch.values()) / len(per_dispatch), len(per_dispatch)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats-dir")
    ap.add_argument("--fetch-dir")
    ap.add_argument("--write-dir")
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--kernel", default="l1k2_tile_kernel")
    ap.add_argument("--xrows", type=int, default=262144)
    ap.add_argument("--yrows", type=int, default=262144)
    ap.add_argument("--dim", type=int, default=128)
    a = ap.parse_args()
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    if a.stats_dir:
        stats = kernel_stats(a.stats_dir)
        rows = sorted(stats.items(), key=lambda kv: -kv[1]["total_ms"])
        total = sum(v["total_ms"] for _, v in rows) or 1.0
        with open(os.path.join(out_dir, "%s_kernel_stats.md" % a.tag), "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats summary (%s)\n\n" % a.tag)
            f.write("| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|\n")
            for name, v in rows:
                f.write("| `%s` | %d | %.3f | %.4f | %.1f |\n" % (name[:100], v["calls"], v["total_ms"], v["avg_ms"],
                                                               100 * v["total_ms"] / total))
    if a.fetch_dir or a.write_dir:
        fetch, nf = counter_sum(a.fetch_dir, "FETCH_SIZE", a.kernel) if a.fetch_dir else (None, 0)
        write, nw = counter_sum(a.write_dir, "WRITE_SIZE", a.kernel) if a.write_dir else (None, 0)
        rec = {"tag": a.tag, "kernel": a.kernel, "xrows": a.xrows, "yrows": a.yrows, "dim": a.dim,
               "FETCH_SIZE_raw_per_launch": fetch, "WRITE_SIZE_raw_per_launch": write,
               "dispatches": {"fetch": nf, "write": nw},
               "correction": "bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): gfx950 FETCH_SIZE counts 128-B "
                             "requests at 64 B (MI355X_MICROARCH.md, HBM)"}
        if fetch is not None and write is not None:
            rec["hbm_bytes_per_launch"] = 1024.0 * (2.0 * fetch + write)
        # one record per workload shape; bench.py picks the one matching its run
        path = os.path.join(out_dir, "l1k2_pmc.json")
        try:
            doc = json.load(open(path))
            records = doc.get("records", [doc])
        except Exception:
            records = []
        records = [r for r in records if (r.get("xrows"), r.get("yrows"), r.get("dim")) != (a.xrows, a.yrows, a.dim)]
        records.append(rec)
        json.dump({"records": records}, open(path, "w"), indent=1)
        print(json.dumps(rec))


if __name__ == "__main__":
    main()
