#!/bin/bash
# one GPU call: the GPU test suite, then the A/B measurements (in-kernel stamps of two builds of the resident kernel,
# bench line).  Nothing runs after a step that was killed by its timeout.
mkdir -p gpurun_out
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
T=${1:-4}
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest$T.log 2>&1; rc=$?
tail -4 gpurun_out/pytest$T.log
ok $rc || exit 1
for v in a; do
  for rep in 8; do
    BRDF_STAMPS_WARD_ONLY=1 BRDF_HIP_RESIDENT_REPLICAS=$rep BRDF_HIP_LIB=$PWD/brdf_amd/libbrdf_hip_stamps_$v.so timeout -k 10 120 python scripts/gpu_stamps.py > gpurun_out/stamps${T}_${v}_r$rep.log 2>&1; rc=$?
    echo "== stamps $v replicas=$rep"; grep -v amdgpu.ids gpurun_out/stamps${T}_${v}_r$rep.log
    ok $rc || exit 1
  done
done
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu > gpurun_out/bench$T.json 2> gpurun_out/bench$T.err; rc=$?
python -c "import json;d=json.load(open('gpurun_out/bench$T.json'));print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['bc_dif']['ms_per_step'], d['bc_dif']['avg_launch_us'])"
ok $rc || exit 1
timeout -k 10 200 python tests/measure_capture.py > gpurun_out/capture$T.log 2>&1; tail -1 gpurun_out/capture$T.log | cut -c1-300
timeout -k 10 300 python scripts/gpu_lane.py 20 > gpurun_out/lane$T.log 2>&1; rc=$?
grep "lane_w1:\|wave_per_fit" gpurun_out/lane$T.log | cut -c1-120
ok $rc || exit 1
timeout -k 10 300 python scripts/gpu_batch.py > gpurun_out/batch$T.log 2>&1; grep -v amdgpu gpurun_out/batch$T.log
