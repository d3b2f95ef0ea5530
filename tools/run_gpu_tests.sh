#!/bin/bash
# Run on the GPU box: the GPU test suite with tools/abort_trace.c preloaded NEXT TO whatever the box preloads itself, so
# that an abort() raised on a native (non-Python) thread leaves its native backtrace in the log, and with --capture=sys:
# pytest's default fd capture swallows what the HIP / ROCr runtime prints to fd 2 before it aborts.
#   bash tools/run_gpu_tests.sh [log file] [pytest args...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LOG=${1:-$R/gpurun_out/gpu_tests.log}; shift || true
mkdir -p "$(dirname "$LOG")"
gcc -shared -fPIC -O1 -o /tmp/abort_trace.so "$R/tools/abort_trace.c" || exit 2
cd "$R"
LD_PRELOAD="${LD_PRELOAD:+$LD_PRELOAD }/tmp/abort_trace.so" timeout -k 10 900 python -X faulthandler -m pytest tests -x -q -m gpu -p no:cacheprovider --capture=sys "$@" > "$LOG" 2>&1
rc=$?
echo "gpu tests exit $rc"
# a GPU memory fault leaves a GPU core file in the working directory: which kernel, which address
for core in gpucore.*; do
    [ -f "$core" ] || continue
    timeout 120 rocgdb -q -batch -ex "info agents" -ex "info queues" -ex "info dispatches" -ex "info threads" -ex "bt" -c "$core" > "$LOG.gpucore.txt" 2>&1
    echo "GPU core $core analysed in $LOG.gpucore.txt"
    break
done
grep -v "^  File" "$LOG" | cut -c1-300 | tail -${TAIL:-8}
exit $rc
