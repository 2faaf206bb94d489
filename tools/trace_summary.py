"""Per-kernel count / mean / total of a rocprofv3 --kernel-trace run stored in
its SQLite output (the default format of rocprofv3 in ROCm 7.2).

    python tools/trace_summary.py <results.db> [--top 30]
"""
import argparse
import collections
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--top", type=int, default=30)
    ap.add_argument("--tail-frac", type=float, default=0.5,
                    help="fraction of the dispatches (from the end) used for the steady-state line")
    args = ap.parse_args()
    cur = sqlite3.connect(args.db).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    agg = collections.defaultdict(lambda: [0, 0])
    for name, s, e in rows:
        agg[name[:90]][0] += 1
        agg[name[:90]][1] += e - s
    print("%7s %10s %10s  kernel" % ("calls", "mean us", "total ms"))
    for name, (k, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print("%7d %10.2f %10.3f  %s" % (k, t / k / 1e3, t / 1e6, name))
    busy = sum(v[1] for v in agg.values())
    print("dispatches %d, kernel time %.3f ms, first-to-last span %.3f ms"
          % (len(rows), busy / 1e6, (rows[-1][2] - rows[0][1]) / 1e6))
    n0 = int(len(rows) * (1.0 - args.tail_frac))
    tail = rows[n0:]
    tb = sum(e - s for _, s, e in tail)
    span = tail[-1][2] - tail[0][1]
    print("last %d dispatches: busy %.3f ms of %.3f ms span (%.1f %%), %.2f us per dispatch slot"
          % (len(tail), tb / 1e6, span / 1e6, 100.0 * tb / span, span / len(tail) / 1e3))


if __name__ == "__main__":
    main()
