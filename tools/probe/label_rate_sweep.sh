# epoch time against the label rate: every row, dense gradients, the labelled-field step (both associations)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3s
for k in 1000 100 20 10 4 2; do
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --label-every $k --cpu-rows 0 --cpu-epoch-scale 0 --planetoid-epochs 0 --f32-side-steps 0 > gpurun_out/r3s/le_$k.json 2> gpurun_out/r3s/le_$k.err
done
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --edge-factor 16 --cpu-rows 0 --cpu-epoch-scale 0 --planetoid-epochs 0 --f32-side-steps 0 > gpurun_out/r3s/deg32.json 2> gpurun_out/r3s/deg32.err
echo done
