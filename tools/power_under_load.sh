#!/bin/bash
# power / clock of the board while (a) the dominant conv kernel, (b) the register-only MFMA probe loop
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/power; mkdir -p $O; cd $R
sample() { # tag seconds
  for i in $(seq 1 $2); do
    echo "== $1 t=$i"; rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (edge|junction|hotspot)" | head -8
    sleep 0.5
  done
}
rocm-smi --showpower --showmaxpower --showclocks 2>&1 | head -30 > $O/idle.txt
echo "--- idle"; grep -E "Power|sclk" $O/idle.txt | head -6
python tools/conv_microbench.py --shape 64,64,64 --cin 128 --cout 128 --precision 3 --iters 20000 --warmup 10 > $O/conv_loop.txt 2>&1 &
P=$!; sleep 3; sample conv 6 > $O/power_conv.txt; wait $P; cat $O/conv_loop.txt | grep -v amdgpu
python tools/conv_microbench.py --shape 64,64,64 --cin 128 --cout 128 --precision 6 --iters 40000 --warmup 10 > $O/conv_bf16_loop.txt 2>&1 &
P=$!; sleep 3; sample convbf16 6 > $O/power_conv_bf16.txt; wait $P; cat $O/conv_bf16_loop.txt | grep -v amdgpu
python - > $O/probe_loop.txt 2>&1 <<'PY' &
import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "3d-denoising-diffusion-model_amd"))
import torch
from guided_diffusion import _hip as H
lib = H.load()
cus = torch.cuda.get_device_properties(0).multi_processor_count
blocks = 2 * cus
pout = torch.empty(blocks * 256, dtype=torch.float32, device="cuda"); pclk = torch.zeros(blocks * 2, dtype=torch.int64, device="cuda")
import time
t0 = time.time(); n = 0
while time.time() - t0 < 8.0:
    H.check(lib.ddpm3d_mfma_probe(H.PROBE_F16_32X32X16, 200000, blocks, H.ptr(pout), H.ptr(pclk), H.stream())); torch.cuda.synchronize(); n += 1
c = pclk.view(blocks, 2).double().cpu()
print("probe launches", n, "in-kernel clock GHz", float((c[:, 0] / c[:, 1]).median()) * 0.1)
PY
P=$!; sleep 3; sample probe 6 > $O/power_probe.txt; wait $P; cat $O/probe_loop.txt | grep -v amdgpu
for f in power_conv power_conv_bf16 power_probe; do echo "--- $f"; grep -E "Power|sclk" $O/$f.txt | sed 's/  */ /g' | sort | uniq -c | sort -rn | head -8; done
