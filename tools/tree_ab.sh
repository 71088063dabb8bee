#!/bin/bash
# Same-box A/B of two TREES (e.g. the previous round's tree in a git worktree against this one), alternated:
#   bash tools/tree_ab.sh <out under gpurun_out> <treeA> <treeB> [forward_time.py args...]
# Thin wrapper over tools/ab.py (failed runs are reported, never dropped).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; A=$2; B=$3; shift 3
exec python3 "$R/tools/ab.py" --out "gpurun_out/$OUT" --trees "$A $B" -- "$@"
