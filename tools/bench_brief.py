"""Print the few figures of a bench.py JSON line that are compared from run to run."""
import json
import sys

line = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
rd = d.get("roofline_dense", {})
print("epoch %.2f ms | spmm w256 %.2f ms (frac %.3f) | nn %.2f ms | tn %.2f ms" % (
    d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"],
    rd.get("nn", {}).get("avg_launch_ms", float("nan")), rd.get("tn", {}).get("avg_launch_ms", float("nan"))))
