#!/bin/bash
# one rocprofv3 --pmc pass per counter of tools/placement_counters.py (each pass its own process, its own hipMalloc draws)
#   tools/placement_counters.sh <out-dir> COUNTER ...
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for C in "$@"; do
    D="$OUT/pass_$C"; mkdir -p "$D"
    rocprofv3 --pmc "$C" --output-format csv -d "$D" -- python3 "$R/tools/placement_counters.py" "$D/marks.json" > "$D/log.txt" 2>&1 || tail -2 "$D/log.txt"
    tail -1 "$D/log.txt" | cut -c1-200
done
python3 "$R/tools/summarize_placement_counters.py" "$OUT" | tee "$OUT/summary.md"
