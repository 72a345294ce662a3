#!/bin/bash
# round 5, second session: where the attention loop's time goes per CLUSTER, and experiment bits A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05l
APEX_MI355_LIB=tools/ubench/bin/libapex_trace2.so ARMS=${TRACE_ARMS:-2:0,2:4} timeout 600 python tools/attn_cluster_trace.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05l/attn_cluster_trace_${TAG:-b}.log
ARMS=${AB_ARMS:-2:0,2:4} SHAPES=flux,qwen,long timeout 900 python tools/attn_dma_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05l/attn_xv_ab_${TAG:-b}.log
