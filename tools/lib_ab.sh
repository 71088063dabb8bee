#!/bin/bash
# Same-box A/B of two builds of the library ("-" = in-tree), alternated ROUNDS times:
#   bash tools/lib_ab.sh <out file under gpurun_out> <libA> <libB> [forward_time.py args...]
# Thin wrapper over tools/ab.py, which keeps each run's stderr, reports a failed run as MISSING and
# exits non-zero when the variants end with unequal row counts.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; A=$2; B=$3; shift 3
exec python3 "$R/tools/ab.py" --out "gpurun_out/$OUT" --libs "$A $B" -- "$@"
