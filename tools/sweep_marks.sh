#!/bin/bash
mkdir -p gpurun_out/j9
for marks in "8,12,16,24,32,48,64,80" "4,8,12,16,24,32,48,64,80" "6,10,16,24,32,48,64,80" "16,24,32,48,64,80" "8,16,24,32,40,48,56,64,72,80,90" "8,12,16,20,24,28,32,40,48,56,64,72,80,90" "3,6,9,12,16,24,32,48,64,80"; do
  PIK_PASSES=$marks timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --no-strict --no-pcie > "gpurun_out/j9/drv_marks_${marks//,/_}.json" 2>&1
done
for marks in "8,12,16,24,32,48,64,80" "4,8,12,16,24,32,48,64,80"; do
  PIK_PASSES=$marks timeout 200 python bench.py --cpu-sample 0 --no-strict --no-pcie > "gpurun_out/j9/def_marks_${marks//,/_}.json" 2>&1
done
