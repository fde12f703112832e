#!/bin/bash
# sweep the forward chunk geometry on the bench workload
for cfg in "4 24 1" "4 32 1" "8 32 1" "8 24 1" "16 32 1" "2 32 1"; do
  set -- $cfg
  HML_FWD_CHUNK=$1 HML_FWD_WARMUP=$2 HML_FWD_ROUNDS=$3 python bench.py --steps 100 --warmup 10 --breakdown --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernel_us_per_sweep']
print('L=$1 W=$2 R=$3  ms/step %.4f  fwd %.1f fix %.1f refits %d serial %d' % (d['ms_per_step'], k['forward'], k['forward_fix'], d['forward_refits'], d['forward_serial']))
"
done
