#!/bin/bash
# x384 tile-order experiment: group_m (tiles tall per XCD group) on the launches the 384 x 256 tiling serves
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05k
timeout 900 python -m pytest tests/test_gpu_flux.py tests/test_easycache.py -m gpu -q -s -k "controlnet or easycache" 2>&1 | grep -E "passed|failed|controlnet|Error" | tail -6
S="4608,21504,3072,gelu;75648,5120,5120,bias;75648,13824,5120,gelu"
for gm in 6 3 4 8 12 6; do KEY=gemm.x384 ARMS=2 TUNE=gemm.group_m=$gm SHAPES="$S" ROUNDS=3 REPS=24 timeout 600 python tools/gemm_x288_ab.py 2>&1 | grep shape | sed "s/^/group_m=$gm /" | cut -c1-200; done | tee gpurun_out/r05k/x384_group_m.log
