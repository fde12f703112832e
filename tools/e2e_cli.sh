#!/bin/bash
# End-to-end wall clock of the command-line driver on a C3-sized text file (10^8 values): reading, construction,
# 1000 recorded-every-10th sweeps, writing the marginals.  Lines of the verbose log are stamped with elapsed seconds.
#   tools/e2e_cli.sh [N=100000000] [scheme...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-100000000}; shift || true
SCHEME=${@:-F 1000 10}
TXT=/tmp/hml_e2e_$N.txt
[ -f "$TXT" ] || HML_TEXT_KEEP=$TXT "$ROOT/tools/bin/text_bench" "$N" %.5f 1000 > /dev/null
ls -la "$TXT"
for rep in 1 2; do
  start=$(date +%s.%N)
  "$ROOT/hammlet_amd/hammlet" -f "$TXT" -a -s 5 -R 1 -i $SCHEME -w -v -o /tmp/hml_e2e_out- .csv | while IFS= read -r line; do
    printf '%8.3f  %s\n' "$(echo "$(date +%s.%N) - $start" | bc -l 2>/dev/null || python3 -c "import time;print(time.time()-$start)")" "$line"
  done
  end=$(date +%s.%N)
  python3 -c "print('total wall clock: %.3f s' % ($end - $start))"
done
ls -la /tmp/hml_e2e_out-*
wc -l /tmp/hml_e2e_out-marginals.csv
