"""Condense a rocprofv3 output directory (--kernel-trace --stats, csv) into the
files kept under profiles/: the per-kernel stats table as rocprofv3 wrote it and
a short markdown summary.  With --pmc-dir, also folds FETCH_SIZE / WRITE_SIZE
passes (one counter per pass) into per-launch HBM bytes for one kernel.

usage: python tools/summarize_profile.py <rocprof dir> <out prefix> [--steps N]
"""
import argparse
import csv
import glob
import json
import os
import shutil


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    return hits[0] if hits else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("rocprof_dir")
    ap.add_argument("out_prefix")
    ap.add_argument("--title", default="")
    args = ap.parse_args()
    stats = find(args.rocprof_dir, "kernel_stats.csv")
    if stats is None:
        raise SystemExit("no kernel_stats.csv under %s" % args.rocprof_dir)
    shutil.copy(stats, args.out_prefix + "_kernel_stats.csv")
    rows = list(csv.DictReader(open(stats)))
    with open(args.out_prefix + "_summary.md", "w") as fh:
        fh.write("# %s\n\nrocprofv3 --kernel-trace --stats; one line per kernel, sorted by total time.\n\n" % args.title)
        fh.write("| kernel | calls | total ms | avg ms | min ms | max ms | % |\n|---|---|---|---|---|---|---|\n")
        for r in rows:
            if float(r["Percentage"]) < 0.05:
                continue
            fh.write("| `%s` | %s | %.2f | %.3f | %.3f | %.3f | %.2f |\n" % (
                r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6,
                float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6, float(r["Percentage"])))
    print(open(args.out_prefix + "_summary.md").read())


if __name__ == "__main__":
    main()
