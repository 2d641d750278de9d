#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun); summaries go to gpurun_out/prof_$1/
# usage: scripts/profile_round.sh <tag>
set -o pipefail
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH="python3 bench.py --steps 10 --warmup 2 --no-cpu"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_trace.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/bench_fetch.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/bench_write.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- python3 scripts/calib_traffic.py > $OUT/cal_fetch.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -- python3 scripts/calib_traffic.py > $OUT/cal_write.log 2>&1 || exit 1
python3 scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
python3 scripts/summarize_profile.py $OUT --traffic-json $OUT/traffic.json $TAG >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
