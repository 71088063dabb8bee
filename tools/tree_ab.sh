#!/bin/bash
# Same-box A/B of two TREES (e.g. the round-2 tree in a git worktree against this one), alternated:
#   bash tools/tree_ab.sh <out under gpurun_out> <treeA> <treeB> [forward_time.py args...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; A=$2; B=$3; shift 3
ROUNDS=${ROUNDS:-2}
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for r in $(seq $ROUNDS); do
  for T in "$A" "$B"; do
    (cd $R/$T && python3 tools/forward_time.py --tag "$T" "$@" 2>/dev/null) >> "$OUT"
  done
done
python3 - "$OUT" <<'PY'
import json, sys, collections
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
by = collections.defaultdict(list)
for r in rows: by[r["tag"]].append(r)
for t, rs in by.items():
    print("%-24s ms/forward: %s" % (t, "  ".join("%.3f" % r["ms_per_forward"] for r in rs)))
fams = sorted({k for r in rows for k in r["families_ms"]})
for f in fams:
    print("  %-28s %s" % (f, "   ".join("%s %.3f" % (t, sum(r["families_ms"].get(f, 0) for r in rs) / len(rs)) for t, rs in by.items())))
PY
