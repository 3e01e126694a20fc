#!/bin/bash
# What a background load does to the decoder BPTT (round 3): a synthetic load (csrc/xchg_bench.hip dev_load_k,
# engine knob TACO_DEV_LOAD = mode:workgroups:lds_bytes:reps:megabytes) takes the place of the weight-gradient flood, which
# TACO_FLUSH_AT=-1 moves behind the decoder.  mode 0 streaming reads, 1 dense MFMA chain, 2 packed-fp32 VALU chains,
# 3 MFMA at ~50 % duty, 4 fp32 atomic adds.   bash scripts/interfere_step.sh [C2|C5]
CFG=${1:-C5}
for cfg in "TACO_FLUSH_AT=-1" "TACO_FLUSH_AT=-1 TACO_DEV_LOAD=0:768:48000:13:84" "TACO_FLUSH_AT=-1 TACO_DEV_LOAD=3:128:100000:6000:0" \
           "TACO_FLUSH_AT=-1 TACO_DEV_LOAD=2:128:100000:8000:0" "TACO_FLUSH_AT=-1 TACO_DEV_LOAD=4:128:100000:300:3" \
           "TACO_FLUSH_AT=-1 TACO_DEV_LOAD=1:1:100000:8000:0"; do
  echo "== $CFG $cfg"
  env $cfg python scripts/dev_sections.py $CFG 2>&1 | grep -E "decoder bwd|sum"
done
