set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3g
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/r3g/p1 -- python3 $R/tools/spmm_bench.py --scale 24 --width 256 --iters 2 --hint --field-of-labels 100 > $R/gpurun_out/r3g/p1.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r3g/p2 -- python3 $R/tools/spmm_bench.py --scale 24 --width 256 --iters 2 --hint --field-of-labels 100 > $R/gpurun_out/r3g/p2.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum --output-format csv -d $R/gpurun_out/r3g/p3 -- python3 $R/tools/spmm_bench.py --scale 24 --width 256 --iters 2 --hint --field-of-labels 100 > $R/gpurun_out/r3g/p3.log 2>&1
cat $R/gpurun_out/r3g/p1.log
echo done
