#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05m
APEX_MI355_LIB=tools/ubench/bin/libapex_w64abl.so timeout 300 python tools/attn_w64_trace.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05m/attn_w64_trace_${TAG:-b}.log
