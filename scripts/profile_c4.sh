#!/bin/bash
# rocprofv3 evidence for the multi-surfel workloads (SURVEY.md section 8(d) "GPU evidence"): kernel durations + HBM bytes
# usage: scripts/profile_c4.sh <tag> <c4|c5> <dif|bc_dif>        (summaries under gpurun_out/prof_<workload>_<entry>_<tag>/)
set -o pipefail
TAG=${1:-r02}
WL=${2:-c4}
ENTRY=${3:-dif}
OUT=gpurun_out/prof_${WL}_${ENTRY}_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# --no-cpu: under the profiler the bench must not start child processes (the CPU baseline workers)
BENCH="python3 bench.py --workload $WL --entry $ENTRY --steps 2 --warmup 1 --no-cpu"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/bench_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/bench_write.log 2>&1 || exit 1
python3 scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
head -12 $OUT/summary.txt
tail -1 $OUT/bench_trace.log | cut -c1-400
