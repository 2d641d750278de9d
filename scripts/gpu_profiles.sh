#!/bin/bash
# one GPU call: the GPU test suite, the rocprofv3 evidence of the round (scripts/profile_round.sh, profile_c4.sh) and one
# bench line per BASELINE.json workload.  usage: bash scripts/gpu_profiles.sh <tag>
TAG=${1:-r02_b}
mkdir -p gpurun_out
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_$TAG.log 2>&1; rc=$?
tail -3 gpurun_out/pytest_$TAG.log
ok $rc || exit 1
bash scripts/profile_round.sh $TAG > gpurun_out/profile_round_$TAG.log 2>&1; rc=$?
tail -25 gpurun_out/profile_round_$TAG.log
ok $rc || exit 1
for wl in c2 c3; do
  timeout -k 10 300 python bench.py --workload $wl > gpurun_out/bench_${wl}_$TAG.json 2> gpurun_out/bench_${wl}_$TAG.err; rc=$?
  python -c "import json;d=json.load(open('gpurun_out/bench_${wl}_$TAG.json'));print('$wl', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['parity'], d['bc_dif']['ms_per_step'])"
  ok $rc || exit 1
done
for wl in c4 c5; do
  for en in dif bc_dif; do
    timeout -k 10 300 python bench.py --workload $wl --entry $en --steps 3 --warmup 1 > gpurun_out/bench_${wl}_${en}_$TAG.json 2> gpurun_out/bench_${wl}_${en}_$TAG.err; rc=$?
    python -c "import json;d=json.load(open('gpurun_out/bench_${wl}_${en}_$TAG.json'));print('$wl $en', d['value'], d['ms_per_step'], d['config']['fits_per_s'], d['config']['failed_fits'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
    ok $rc || exit 1
  done
done
bash scripts/profile_c4.sh $TAG c4 bc_dif > gpurun_out/profile_c4_bc_$TAG.log 2>&1; rc=$?; tail -8 gpurun_out/profile_c4_bc_$TAG.log | cut -c1-300
ok $rc || exit 1
bash scripts/profile_c4.sh $TAG c4 dif > gpurun_out/profile_c4_dif_$TAG.log 2>&1; rc=$?; tail -8 gpurun_out/profile_c4_dif_$TAG.log | cut -c1-300
ok $rc || exit 1
bash scripts/profile_c4.sh $TAG c5 dif > gpurun_out/profile_c5_dif_$TAG.log 2>&1; rc=$?; tail -8 gpurun_out/profile_c5_dif_$TAG.log | cut -c1-300
ok $rc || exit 1
bash scripts/profile_c4.sh $TAG c5 bc_dif > gpurun_out/profile_c5_bc_$TAG.log 2>&1; rc=$?; tail -8 gpurun_out/profile_c5_bc_$TAG.log | cut -c1-300
