set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3f
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3f/prof -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-rows 0 --cpu-epoch-scale 0 --planetoid-epochs 0 --f32-side-steps 0 --dense-side-steps 0 > $R/gpurun_out/r3f/line.json 2> $R/gpurun_out/r3f/err.log
ls $R/gpurun_out/r3f/prof/*/ | head
