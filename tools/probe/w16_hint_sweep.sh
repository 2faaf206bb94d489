set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3w
timeout -k 10 120 python tools/spmm_bench.py --scale 24 --width 16 --iters 7 > gpurun_out/r3w/w16_nohint.log 2>&1
for b in 0.5 1 2 4 8 32 176; do
  timeout -k 10 120 python tools/spmm_bench.py --scale 24 --width 16 --iters 7 --hint --hint-budget-mib $b > gpurun_out/r3w/w16_hint_$b.log 2>&1
done
grep -H "W=16\|hint" gpurun_out/r3w/w16_*.log
