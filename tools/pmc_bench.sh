#!/bin/bash
# HBM-side traffic of the bench's conv kernels (rocprofv3 PMC), per launch and per kernel:
#   bash tools/pmc_bench.sh <outdir-under-gpurun_out> [bench.py args...]
# Two passes (FETCH_SIZE and WRITE_SIZE do not fit one), counters + kernel trace only, as
# MI355X_MICROARCH.md prescribes; FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of
# 16-B-per-lane reads at 64 B).  Writes <outdir>/traffic.json; copy it to
# profiles/rNN_pmc_traffic.json, from where bench.py fills roofline.traffic.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -- \
        python3 "$R/bench.py" --steps 1 --warmup 0 --ddpm-steps 2 --cpu-steps 0 --f32-steps 0 --probe-ms 0 --golden 0 "$@" > "$OUT/$c.log" 2>&1 || exit 1
done
python3 - "$OUT" <<'EOF'
import collections, csv, glob, json, re, sys
out = sys.argv[1]

def bench_name(k):
    """rocprof kernel name -> the name bench.py reports for that launch"""
    m = re.search(r"conv3d_wz_kernel<(\d+), (\d+), (\d+)", k)
    if m:   # Winograd-D form of the f16x3 / f16 / bf16 arithmetic = precisions 3 / 4 / 6; <MODE, IL, TX, TY>: the
            # engine's family tag says t8 for the 8-wide tiles (8x8x2, 8x4x4) and t4 for 4x4x8
        return "conv3d_p%d_k3_wn4_t%d" % ({0: 3, 1: 4, 2: 6}[int(m.group(1))], int(m.group(3)))
    m = re.search(r"conv3d_skinny_kernel<(\d+)", k)
    if m:
        return "conv3d_p%d_k3_skinny" % {0: 1, 1: 2, 2: 5}[int(m.group(1))]
    m = re.search(r"conv3d_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>", k)
    if m:
        prec, _pipe, ks, wn, _mt, txl, _tyl = map(int, m.groups())
        return f"conv3d_p{prec}_k{ks}_wn{wn}_t{1 << txl}"
    return None

res = collections.defaultdict(lambda: {"launches": 0})
for c, scale in (("FETCH_SIZE", 2.0 * 1024), ("WRITE_SIZE", 1024.0)):   # counters are in KiB
    f = glob.glob(f"{out}/{c}/*/*_counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            n = bench_name(r["Kernel_Name"])
            if n:
                acc[n].append(float(r["Counter_Value"]) * scale)
    for n, v in acc.items():
        res[n]["launches"] = len(v)
        res[n][c.lower() + "_bytes_per_launch"] = sum(v) / len(v)
for n, d in res.items():
    d["hbm_bytes_per_launch"] = d.get("fetch_size_bytes_per_launch", 0) + d.get("write_size_bytes_per_launch", 0)
sig = None
for line in open(f"{out}/FETCH_SIZE.log"):
    if line.startswith("{") and "traffic_signature" in line:
        sig = json.loads(line)["roofline"]["traffic_signature"]
json.dump({"signature": sig, "how": "tools/pmc_bench.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over "
                  "bench.py --steps 1 --warmup 0 --ddpm-steps 2; FETCH_SIZE x2 (gfx950), KiB -> bytes",
           "kernels": res}, open(f"{out}/traffic.json", "w"), indent=1)
for n, d in sorted(res.items()):
    print(n, d)
EOF
