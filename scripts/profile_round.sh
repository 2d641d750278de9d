#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box (run through gpurun); summaries go to gpurun_out/prof_$1/ and are
# copied from there into profiles/.  Counters are collected in passes of their own (--kernel-trace + --pmc only).
# usage: scripts/profile_round.sh <tag> [quick]
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # run <subdir> <rocprofv3 options...> -- <program...>
  local sub=$1; shift
  timeout -k 10 400 rocprofv3 --output-format csv -d $OUT/$sub "$@" > $OUT/$sub.log 2>&1 || { echo "FAILED: $sub"; tail -5 $OUT/$sub.log; exit 1; }
}
# the default bench run: configs[1] headline (both entry points) + the configs[2..4] sub-lines -> every fit kernel of the line
BENCH="python3 bench.py --steps 10 --warmup 2 --no-cpu"
run trace --kernel-trace --stats -- $BENCH
run pmc_fetch --kernel-trace --pmc FETCH_SIZE -- $BENCH
run pmc_write --kernel-trace --pmc WRITE_SIZE -- $BENCH
run pmc_valu --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- $BENCH
run cal_fetch --kernel-trace --pmc FETCH_SIZE -- python3 scripts/calib_traffic.py
run cal_write --kernel-trace --pmc WRITE_SIZE -- python3 scripts/calib_traffic.py
python3 scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
python3 scripts/summarize_profile.py $OUT --traffic-json $OUT/traffic.json $TAG >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
[ "$2" = quick ] && exit 0
# the kernels the default line does not reach: configs[3] through dlevmar_bc_dif, the lane-per-fit kernel, cosines, the capture loops
for spec in "c4_bc_dif:python3 bench.py --workload c4 --entry bc_dif --steps 2 --warmup 1 --no-cpu" \
            "c5_bc_dif:python3 bench.py --workload c5 --entry bc_dif --steps 2 --warmup 1 --no-cpu" \
            "lane_fit:python3 scripts/gpu_lane.py 18" "cosines:python3 tests/measure_cosines.py" "capture:python3 tests/measure_capture.py"; do
  name=${spec%%:*}; cmd=${spec#*:}
  mkdir -p $OUT/$name
  timeout -k 10 400 rocprofv3 --output-format csv -d $OUT/$name/trace --kernel-trace --stats -- $cmd > $OUT/$name.log 2>&1 || { echo "FAILED: $name"; tail -5 $OUT/$name.log; exit 1; }
  python3 scripts/summarize_profile.py $OUT/$name > $OUT/${name}_summary.txt 2>&1
  echo "== $name"; head -6 $OUT/${name}_summary.txt
done
