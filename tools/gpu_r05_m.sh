#!/bin/bash
# round 5: the one-wave-per-SIMD attention kernel (attn.w64) against the shipped one
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05m
SHAPES=${SHAPES:-flux,qwen,long} timeout 300 python tools/attn_w64_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05m/attn_w64_ab_${TAG:-a}.log | cut -c1-400
