#!/bin/bash
# Experiment: stream priorities of the bench's pipelines (ORBX_BENCH_PRIORITIES), the number of pipelines, and the
# number of hardware queues the HIP runtime spreads its streams over (GPU_MAX_HW_QUEUES).  Arguments: the
# configurations to run, each as queues:streams:priorities (default: the table kept with the design notes).
B="--kfdb-frames 0 --frames-4k 0 --fed-frames 0 --no-cpu-baseline --ransac-pairs 0 --steps 40 --warmup 6"
CFGS=${@:-"1:2:0,0 2:2:0,0 3:2:0,0 4:2:0,0 8:2:0,0 3:3:0,0,0 4:3:0,0,0 8:3:0,0,0 4:3:-1,-1,-1 3:3:-1,-1,-1 4:4:-1,-1,-1,-1"}
for rep in 1 2; do
  for cfg in $CFGS; do
    q=${cfg%%:*}; rest=${cfg#*:}; n=${rest%%:*}; pr=${rest#*:}
    GPU_MAX_HW_QUEUES=$q ORBX_BENCH_PRIORITIES=$pr timeout -k 10 120 python3 bench.py --streams $n $B > gpurun_out/prio.json || exit 1
    python3 -c "
import json; b=json.load(open('gpurun_out/prio.json')); print('rep $rep queues $q streams $n prio $pr', round(b['value']), round(b['ms_per_step'],3))"
  done
done
