#!/bin/bash
# rocprofv3 evidence for BASELINE.json configs[3] (65,536 surfels x 4,096 samples), SURVEY.md section 8(d) "GPU evidence"
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/prof_c4_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH="python3 bench.py --workload c4 --steps 2 --warmup 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/bench_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/bench_write.log 2>&1 || exit 1
python3 scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
head -30 $OUT/summary.txt
tail -1 $OUT/bench_trace.log | cut -c1-600
