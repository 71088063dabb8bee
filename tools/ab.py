#!/usr/bin/env python3
"""
Same-box A/B of several builds of the library (or of several trees), alternated ROUNDS times; boards of
the pool differ by several per cent, so only numbers from one box compare.

    python tools/ab.py --out gpurun_out/ab.txt --libs "- scratch/prev/libddpm3d.so" [--rounds 3] -- --precision bf16
    python tools/ab.py --out gpurun_out/ab.txt --trees "scratch/r03tree ." -- --precision f16x3

"-" = the in-tree library.  Every run is tools/forward_time.py in a child process; its stderr goes to
<out>.stderr, and a run that exits non-zero or prints no JSON line is reported as MISSING and makes this
script exit 1 AFTER the remaining runs: a summary never silently rests on fewer rows for one variant
(r03's lib_ab*.sh dropped such rows).
"""

import argparse
import collections
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--libs", default=None, help="space-separated library paths relative to the repo root ('-' = in-tree)")
    ap.add_argument("--trees", default=None, help="space-separated tree directories relative to the repo root")
    ap.add_argument("--flags", default=None, help="'|'-separated sets of extra forward_time.py flags, one variant each "
                                                  "(e.g. '|--graph'; an empty set = the defaults)")
    ap.add_argument("--rounds", type=int, default=int(os.environ.get("ROUNDS", "2")))
    ap.add_argument("rest", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    if sum(x is not None for x in (a.libs, a.trees, a.flags)) != 1:
        ap.error("give one of --libs, --trees, --flags")
    rest = a.rest[1:] if a.rest[:1] == ["--"] else a.rest
    variants = a.flags.split("|") if a.flags is not None else (a.libs or a.trees).split()
    out = os.path.join(ROOT, a.out) if not os.path.isabs(a.out) else a.out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    rows, missing = collections.defaultdict(list), []
    with open(out, "w") as fo, open(out + ".stderr", "w") as fe:
        for r in range(a.rounds):
            for v in variants:
                env = dict(os.environ)
                env.pop("DDPM3D_LIB", None)
                cwd = ROOT
                extra = []
                if a.flags is not None:
                    extra = v.split()
                elif a.libs is not None:
                    if v != "-":
                        env["DDPM3D_LIB"] = os.path.join(ROOT, v)
                        if not os.path.exists(env["DDPM3D_LIB"]):
                            missing.append((r, v, "no such library"))
                            continue
                else:
                    cwd = os.path.join(ROOT, v)
                fe.write("==== round %d %s\n" % (r, v))
                fe.flush()
                p = subprocess.run([sys.executable, os.path.join(cwd, "tools", "forward_time.py"), "--tag", (v or "default").replace("-", "_")]
                                   + extra + rest,
                                   cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=fe, text=True)
                line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
                if p.returncode != 0 or len(line) != 1:
                    missing.append((r, v, "exit %d, %d JSON lines" % (p.returncode, len(line))))
                    continue
                fo.write(line[0] + "\n")
                fo.flush()
                rows[v].append(json.loads(line[0]))
        lines = []
        for r, v, why in missing:
            lines.append("MISSING round %d variant %s: %s (see %s.stderr)" % (r, v, why, os.path.basename(out)))
        for v in variants:
            rs = rows[v]
            lines.append("%-44s ms/forward: %s" % (v or "default", "  ".join("%.3f" % x["ms_per_forward"] for x in rs) or "-"))
        fams = sorted({k for rs in rows.values() for x in rs for k in x["families_ms"]})
        for f in fams:
            lines.append("  %-28s %s" % (f, "   ".join(
                "%s %.3f" % (os.path.basename(os.path.dirname(v)) or v or "default",
                             sum(x["families_ms"].get(f, 0) for x in rows[v]) / max(1, len(rows[v]))) for v in variants)))
        counts = {v: len(rows[v]) for v in variants}
        if len(set(counts.values())) != 1:
            lines.append("UNEQUAL row counts per variant: %s -- the means above do not compare" % counts)
        fo.write("\n".join("# " + ln for ln in lines) + "\n")
    print("\n".join(lines))
    sys.exit(1 if missing or len(set(counts.values())) != 1 else 0)


if __name__ == "__main__":
    main()
