#!/bin/bash
# forward chunk length L (HML_FWD_CHUNK) and warm-up W on the weakly compressed C5 trace: per-kernel-family times
for L in 4 16 32 64 128; do
  echo "== L=$L"
  HML_FWD_CHUNK=$L python bench.py --workload c5_2.5e8_depth_k5 --breakdown --no-cpu-baseline --no-stream-leg --no-two-chain-leg --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],2), 'ms/sweep', d['kernel_us_per_sweep'], 'refits', d['forward_refits'])"
done
