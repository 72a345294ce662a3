#!/bin/bash
# f32-storage text encoders + whole GPU suite with the rebuilt library
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_text.py -q -m gpu -s -k "f32_storage" 2>&1 | tail -40 > gpurun_out/r04_text_f32.log
cat gpurun_out/r04_text_f32.log
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r04_gpu_suite_tail_d.txt
cat gpurun_out/r04_gpu_suite_tail_d.txt
