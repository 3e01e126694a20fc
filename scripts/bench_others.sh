#!/bin/bash
# bench.py on the per-GPU shapes of the other BASELINE configs (no CPU baseline, no train-loop legs): one JSON line per config
for c in C4 C5 C2x C2x2; do
  python bench.py --config $c --no-cpu-baseline --no-train-loop --no-families --steps 30 --warmup 5 2>/dev/null | tail -1
done
echo "--- C2x2 with the chunk pipeline forced on (three persistent launches that do not fit 256 CUs together)"
TACO_PIPE_OVERSUBSCRIBE=1 python bench.py --config C2x2 --no-cpu-baseline --no-train-loop --no-families --steps 30 --warmup 5 2>/dev/null | tail -1
