set -e
R=$GRAFT_REPO_ROOT
export GCNX_LIB=$R/gcn_amd/lib/variants/libgcnhip_xs.so
mkdir -p $R/gpurun_out/r3q
cd /tmp && export TMPDIR=/tmp
for m in rows slabs slabs_major; do
  timeout -s KILL 150 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/r3q/$m -- python3 $R/tools/probe/xcd_slab_pmc.py $m > $R/gpurun_out/r3q/$m.log 2>&1
done
echo done
