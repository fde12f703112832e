#!/bin/bash
# wall clock of the command-line driver on small inputs (config 1: 10^5 values, 100 recorded sweeps)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
python3 - <<'P'
import numpy as np
rng = np.random.default_rng(1)
x = np.repeat(rng.integers(-1, 2, 50), 2000) + rng.normal(0, 0.2, 100000)
np.savetxt("/tmp/hml_c1.txt", x, fmt="%.6f")
P
for rep in 1 2 3; do
  s=$(date +%s.%N)
  "$ROOT/hammlet_amd/hammlet" -f /tmp/hml_c1.txt -a -s 3 -R 1 -i F 100 1 -w -o /tmp/hml_c1- .csv
  e=$(date +%s.%N)
  python3 -c "print('hammlet (MI355X) T=1e5, F 100 1: %.3f s' % ($e - $s))"
done
if [ -x "$ROOT/oracle/_ref/hammlet" ]; then
  s=$(date +%s.%N); "$ROOT/oracle/_ref/hammlet" -f /tmp/hml_c1.txt -a -s 3 -R 1 -i F 100 1 -w -o /tmp/hml_c1r- .csv; e=$(date +%s.%N)
  python3 -c "print('reference binary, same command: %.3f s' % ($e - $s))"
fi
python3 - <<'P'
import numpy as np
x = np.loadtxt("/tmp/hml_c1.txt").astype(np.float32); x.tofile("/tmp/hml_c1.f32")
P
for rep in 1 2; do
  s=$(date +%s.%N); "$ROOT/hammlet_amd/hammlet" -raw /tmp/hml_c1.f32 -a -s 3 -R 1 -i F 100 1 -w -o /tmp/hml_c1b- .csv; e=$(date +%s.%N)
  python3 -c "print('same through -raw (no text reader): %.3f s' % ($e - $s))"
  s=$(date +%s.%N); "$ROOT/hammlet_amd/hammlet" -raw /tmp/hml_c1.f32 -a -s 3 -R 1 -i F 1 0 -w -o /tmp/hml_c1c- .csv; e=$(date +%s.%N)
  python3 -c "print('-raw, one sweep: %.3f s' % ($e - $s))"
  s=$(date +%s.%N); "$ROOT/hammlet_amd/hammlet" -h > /dev/null; e=$(date +%s.%N)
  python3 -c "print('-h (no GPU): %.3f s' % ($e - $s))"
done
