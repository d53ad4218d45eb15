#!/bin/bash
# On the GPU box (through gpurun): profile THE DRIVER'S bench command and assemble profiles/<round> material in gpurun_out/.
# usage: tools/gpu_profile_driver_cmd.sh <tag>
set -u
tag=${1:-r03a}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --fused-steps 0 --no-measure-traffic"
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -o t -- python3 bench.py $ARGS > gpurun_out/${tag}_bench_under_trace.json 2> gpurun_out/${tag}_trace.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_w -o w -- python3 bench.py $ARGS > gpurun_out/${tag}_w.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_f -o f -- python3 bench.py $ARGS > gpurun_out/${tag}_f.log 2>&1
python3 tools/make_profile_summary.py ${tag} auto gpurun_out/${tag}_bench_driver_summary.json 20 3 2000 > gpurun_out/${tag}_summary.log 2>&1
find gpurun_out/${tag}_trace -name '*kernel_stats.csv' -exec cp {} gpurun_out/${tag}_kernel_stats.csv \;
# keep the merge-back small: the raw traces are tens of MB
rm -rf gpurun_out/${tag}_trace gpurun_out/${tag}_w gpurun_out/${tag}_f
python3 bench.py > gpurun_out/${tag}_bench_default.json 2>> gpurun_out/${tag}_bench.err
tail -c 1500 gpurun_out/${tag}_bench.json
tail -c 1600 gpurun_out/${tag}_summary.log
