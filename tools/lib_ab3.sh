#!/bin/bash
# Same-box A/B of several builds of the library under THIS tree's host code, alternated:
#   bash tools/lib_ab3.sh <out under gpurun_out> "<lib1> <lib2> ..." [forward_time.py args...]   ("-" = in-tree)
# Thin wrapper over tools/ab.py (failed runs are reported, never dropped).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; LIBS=$2; shift 2
exec python3 "$R/tools/ab.py" --out "gpurun_out/$OUT" --libs "$LIBS" -- "$@"
