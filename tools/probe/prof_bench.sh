# rocprofv3 kernel statistics of the default bench step and of the --labelled-field step (side legs off)
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3f
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r3f/prof $R/gpurun_out/r3f/prof_field
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3f/prof -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-rows 0 --cpu-epoch-scale 0 --planetoid-epochs 0 --f32-side-steps 0 --dense-side-steps 0 > $R/gpurun_out/r3f/line.json 2> $R/gpurun_out/r3f/err.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3f/prof_field -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-rows 0 --cpu-epoch-scale 0 --planetoid-epochs 0 --f32-side-steps 0 --dense-side-steps 0 --labelled-field > $R/gpurun_out/r3f/line_field.json 2> $R/gpurun_out/r3f/err_field.log
cd $R
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --labelled-field > gpurun_out/r3f/bench_field_main.json 2> gpurun_out/r3f/bench_field_main.err
echo done
