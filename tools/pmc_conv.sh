#!/bin/bash
# PMC passes (rocprofv3) over tools/conv_microbench.py; run on the GPU box:
#   bash tools/pmc_conv.sh <outdir-under-gpurun_out> [microbench args...]
# Each counter set is its own run (counters + kernel trace only), as
# MI355X_MICROARCH.md "rocprofv3 PMC slots" prescribes.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_pass() {
    local tag=$1; shift
    local counters=$1; shift
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "$OUT/$tag" -- \
        python3 "$R/tools/conv_microbench.py" --iters 3 --warmup 1 "$@" > "$OUT/$tag.log" 2>&1
}
run_pass a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "$@"
run_pass b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL" "$@"
if [ "${PMC_MEM:-0}" = "1" ]; then
    run_pass c "FETCH_SIZE TCC_HIT_sum" "$@"
    run_pass d "WRITE_SIZE TCC_MISS_sum" "$@"
fi
python3 - "$OUT" <<'EOF'
import csv, glob, sys, collections
out = sys.argv[1]
for t in "abcd":
    cc = glob.glob(f"{out}/{t}/*/*_counter_collection.csv")
    kt = glob.glob(f"{out}/{t}/*/*_kernel_trace.csv")
    if not cc:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if "conv3d_" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt[0]))
            if "conv3d_" in r["Kernel_Name"] ] if kt else []
    print(f"pass {t}: kernel us (last) = {dur[-1] if dur else None}")
    for k, v in sorted(acc.items()):
        print(f"   {k:32s} {v[-1]:.4g}")
EOF
