#!/bin/bash
# Sweep the split-K factor (DDPM3D_KSPLIT override) over the published net's
# low-resolution conv shapes; prints one line per (shape, S).  Run on the GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
P=${1:-1}
for SHAPE in "64,16,16 128 256" "64,16,16 256 256" "64,16,16 512 256" "64,8,8 256 384" "64,8,8 384 384" "64,8,8 768 384" "64,4,4 384 512" "64,4,4 512 512" "64,4,4 1024 512" "64,32,32 128 128" "64,32,32 256 128"; do
    set -- $SHAPE
    for S in 0 1 2 3 4 6 8 12 16; do
        if [ "$S" = "0" ]; then unset DDPM3D_KSPLIT; TAG="auto"; else export DDPM3D_KSPLIT=$S; TAG="S=$S"; fi
        echo -n "$TAG  "
        python3 $R/tools/conv_microbench.py --precision $P --shape $1 --cin $2 --cout $3 --iters 10 2>&1 | tail -1
    done
done
