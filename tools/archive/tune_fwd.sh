#!/bin/bash
# sweep the forward chunk geometry (chunk length L, warm-up W, repair rounds R) on the bench workload
for cfg in "4 24 1" "4 16 1" "4 12 1" "4 8 1" "8 16 1" "2 12 1"; do
  set -- $cfg
  HML_FWD_CHUNK=$1 HML_FWD_WARMUP=$2 HML_FWD_ROUNDS=$3 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stream-leg 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('L=$1 W=$2 R=$3  ms/step %.4f  refits %d serial %d' % (d['ms_per_step'], d['forward_refits'], d['forward_serial']))
"
done
