"""rocprofv3 (ROCm 7) writes a rocpd SQLite database by default.  This turns one
into the plain files kept under profiles/:

  python tools/rocpd_export.py stats <results.db> <out prefix> [--title T]
      -> <prefix>_kernel_stats.csv (name, calls, total / average / min / max ns, %)
         <prefix>_summary.md
  python tools/rocpd_export.py pmc <results.db> <out.md> --kernel SUBSTR [--title T]
      -> per-launch sums of every collected counter for the kernels matching SUBSTR
"""
import argparse
import collections
import csv
import sqlite3


def stats(db, prefix, title):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), "
                       "max(end - start) from kernels group by name order by 3 desc").fetchall()
    total = float(sum(r[2] for r in rows)) or 1.0
    with open(prefix + "_kernel_stats.csv", "w", newline="") as fh:
        w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, calls, tot, avg, mn, mx in rows:
            w.writerow([name, calls, int(tot), float(avg), round(100.0 * tot / total, 2), int(mn), int(mx)])
    with open(prefix + "_summary.md", "w") as fh:
        fh.write("# %s\n\nrocprofv3 --kernel-trace --stats; one line per kernel, sorted by total time.\n\n" % title)
        fh.write("| kernel | calls | total ms | avg ms | min ms | max ms | % |\n|---|---|---|---|---|---|---|\n")
        for name, calls, tot, avg, mn, mx in rows:
            pct = 100.0 * tot / total
            if pct < 0.05:
                continue
            fh.write("| `%s` | %d | %.2f | %.3f | %.3f | %.3f | %.2f |\n" % (name[:110], calls, tot / 1e6, avg / 1e6,
                                                                            mn / 1e6, mx / 1e6, pct))
    print(open(prefix + "_summary.md").read())


def pmc(db, out, kernel, title):
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, dispatch_id, counter_name, value, end - start from counters_collection "
                       "where kernel_name like ?", ("%" + kernel + "%",)).fetchall()
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = {}
    names = set()
    for name, disp, cname, value, d in rows:
        per[disp][cname] += float(value)
        dur[disp] = d
        names.add(name)
    with open(out, "w") as fh:
        fh.write("# %s\n\nrocprofv3 --pmc (one pass, counters only); kernels matching `%s`: %s\n\n" % (
            title, kernel, "; ".join("`%s`" % n[:100] for n in sorted(names))))
        counters = sorted({c for d in per.values() for c in d})
        fh.write("| dispatch | duration ms (under the profiler) | " + " | ".join(counters) + " |\n")
        fh.write("|---|---|" + "---|" * len(counters) + "\n")
        for disp in sorted(per):
            fh.write("| %d | %.3f | " % (disp, dur[disp] / 1e6) + " | ".join("%.4g" % per[disp][c] for c in counters) + " |\n")
    print(open(out).read())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["stats", "pmc"])
    ap.add_argument("db")
    ap.add_argument("out")
    ap.add_argument("--kernel", default="")
    ap.add_argument("--title", default="")
    a = ap.parse_args()
    if a.mode == "stats":
        stats(a.db, a.out, a.title)
    else:
        pmc(a.db, a.out, a.kernel, a.title)


if __name__ == "__main__":
    main()
