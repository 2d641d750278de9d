#!/bin/bash
# same-box A/B of bench.py over library variants / environment settings.  usage: scripts/gpu_ab.sh <tag> "<name>=<env assignments>" ...
# e.g. scripts/gpu_ab.sh r3b "base=BRDF_HIP_LIB=$PWD/brdf_amd/libbrdf_hip_base.so" "new=" "nochain=BRDF_HIP_DIF_CHAIN=1 BRDF_HIP_SPEC_JAC=0"
TAG=$1; shift
mkdir -p gpurun_out
WL=${WORKLOADS:-c2}
for wl in $WL; do
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  for rep in 1 2; do
    env $envs timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu --workload $wl $BENCH_ARGS > gpurun_out/${TAG}_${wl}_${name}_$rep.json 2> gpurun_out/${TAG}_${wl}_${name}_$rep.err; rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT $wl $name"; exit 1; fi
    python - <<PY
import json
try:
    d=json.load(open('gpurun_out/${TAG}_${wl}_${name}_$rep.json'))
    if 'bc_dif' in d:
        print('$wl $name rep$rep: dif us/launch %.1f passes %.0f ms/step %.4f frac %.3f | bc us/launch %.1f passes %.0f'%(d['roofline']['avg_launch_us'], d['config']['passes_per_fit'], d['ms_per_step'], d['roofline']['frac'], d['bc_dif']['avg_launch_us'], d['bc_dif']['passes_per_step']))
    else:
        print('$wl $name rep$rep: ms/step %.3f value %.4g sha %s'%(d['ms_per_step'], d['value'], str(d.get('result_sha256'))[:12]))
except Exception as e:
    print('$wl $name rep$rep: FAILED', e, open('gpurun_out/${TAG}_${wl}_${name}_$rep.err').read()[-500:])
PY
  done
done
done
