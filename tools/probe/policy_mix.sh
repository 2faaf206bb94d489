set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3r
cd /tmp && export TMPDIR=/tmp
P=$R/tools/probe/policy_mix_probe
for warm in 128 2048; do
  for aux in 0 1 2 3 16 17 18 19; do
    echo "== warm ${warm} MiB aux $aux" >> $R/gpurun_out/r3r/times.txt
    timeout -s KILL 60 $P $aux 2048 $warm 64 >> $R/gpurun_out/r3r/times.txt 2>&1
  done
done
for aux in 0 2 16 17 18 19; do
  timeout -s KILL 60 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $R/gpurun_out/r3r/pmc_$aux -- $P $aux 2048 128 64 > $R/gpurun_out/r3r/pmc_$aux.log 2>&1
done
echo done
