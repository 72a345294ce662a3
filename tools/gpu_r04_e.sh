#!/bin/bash
# same-box A/B of two builds of the library on the Flux step (separate processes, alternating): previous commit vs this tree
mkdir -p gpurun_out; : > gpurun_out/r04_ab_epilogue_dispatch.log
for i in 1 2; do
  for l in prev new; do
    lib=""; [ $l = prev ] && lib=$PWD/tools/ubench/bin/libapex_prev.so
    echo "== $l (round $i)" >> gpurun_out/r04_ab_epilogue_dispatch.log
    APEX_MI355_LIB=$lib ARMS="base" STEPS=12 ROUNDS=3 CLK=1 timeout 600 python tools/flux_step_ab.py 2>&1 | tail -1 >> gpurun_out/r04_ab_epilogue_dispatch.log
  done
done
cat gpurun_out/r04_ab_epilogue_dispatch.log
