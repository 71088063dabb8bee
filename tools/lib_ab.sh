#!/bin/bash
# Same-box A/B of two builds of the library, alternated ROUNDS times (boards of the pool differ by
# several per cent; only numbers from one box compare):
#   bash tools/lib_ab.sh <out file under gpurun_out> <libA> <libB> [forward_time.py args...]
# "-" = the in-tree library.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; A=$2; B=$3; shift 3
ROUNDS=${ROUNDS:-2}
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for r in $(seq $ROUNDS); do
  for L in "$A" "$B"; do
    if [ "$L" = "-" ]; then unset DDPM3D_LIB; else export DDPM3D_LIB=$R/$L; fi
    python3 $R/tools/forward_time.py --tag "round$r" "$@" >> "$OUT" 2>/dev/null
  done
done
unset DDPM3D_LIB
python3 - "$OUT" <<'PY'
import json, sys, collections
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
by = collections.defaultdict(list)
for r in rows: by[r["lib"]].append(r)
for lib, rs in by.items():
    print("%-44s ms/forward: %s" % (lib, "  ".join("%.3f" % r["ms_per_forward"] for r in rs)))
fams = sorted({k for r in rows for k in r["families_ms"]})
for f in fams:
    print("  %-28s %s" % (f, "   ".join("%s %.3f" % (lib.split("/")[-2] if "/" in lib else lib, sum(r["families_ms"].get(f, 0) for r in rs) / len(rs)) for lib, rs in by.items())))
PY
