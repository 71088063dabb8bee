#!/bin/bash
# Same-box A/B of several builds of the library under THIS tree's host code, alternated:
#   bash tools/lib_ab3.sh <out under gpurun_out> "<lib1> <lib2> ..." [forward_time.py args...]   ("-" = in-tree)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; LIBS=$2; shift 2
ROUNDS=${ROUNDS:-2}
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for r in $(seq $ROUNDS); do
  for L in $LIBS; do
    if [ "$L" = "-" ]; then unset DDPM3D_LIB; else export DDPM3D_LIB=$R/$L; fi
    python3 $R/tools/forward_time.py --tag "$L" "$@" >> "$OUT" 2>/dev/null
  done
done
unset DDPM3D_LIB
python3 - "$OUT" <<'PY'
import json, sys, collections
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
by = collections.defaultdict(list)
for r in rows: by[r["tag"]].append(r)
for t, rs in by.items():
    print("%-36s ms/forward: %s" % (t, "  ".join("%.3f" % r["ms_per_forward"] for r in rs)))
fams = sorted({k for r in rows for k in r["families_ms"]})
for f in fams:
    print("  %-24s %s" % (f, "  ".join("%.3f" % (sum(r["families_ms"].get(f, 0) for r in rs) / len(rs)) for t, rs in by.items())))
PY
