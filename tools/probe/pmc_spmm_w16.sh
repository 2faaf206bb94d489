set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3w
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/r3w/p1 -- python3 $R/tools/spmm_bench.py --scale 24 --width 16 --iters 2 > $R/gpurun_out/r3w/p1.log 2>&1
echo done
