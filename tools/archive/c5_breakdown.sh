#!/bin/bash
# per-kernel-family times of settled sweeps on the weakly compressed C5 trace (2.5e8 read-depth positions, ~1.7e8 blocks)
python bench.py --workload c5_2.5e8_depth_k5 --breakdown --no-cpu-baseline --no-stream-leg --no-two-chain-leg --steps ${1:-20} --warmup ${2:-30} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],2), 'ms/sweep', '%.3e block-updates/s' % d['value'], d['kernel_us_per_sweep'], 'refits', d['forward_refits'], 'sweep_frac', round(d['roofline']['sweep_frac'],4))"
