#!/bin/bash
# Run on the GPU box: tools/profile_gpu.sh for a list of "tag|bench args" entries (one rocprofv3 kernel-trace pass and two
# PMC passes each).  Usage: tools/profile_round.sh "von_mises_mixed|--workload von_mises_mixed" "von_mises_mixed_unpacked|--workload von_mises_mixed --history sparse" ...
# Summarise afterwards, per tag:  python tools/summarize_profile.py gpurun_out/prof/<tag> <round> <tag> ["extra bench args"]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for entry in "$@"; do
    tag=${entry%%|*}
    args=${entry#*|}
    echo "== $tag: $args"
    # shellcheck disable=SC2086
    bash "$R/tools/profile_gpu.sh" "$tag" $args --configs none
    for f in kt fetch write; do tail -n 1 "$R/gpurun_out/prof/$tag/$f.log"; done
    # keep what the summary needs, drop the bulky rest (the merge back is limited to 64 MiB)
    find "$R/gpurun_out/prof/$tag" -name "*_agent_info.csv" -delete
done
