#!/bin/bash
# per-boundary gaps of the strongly compressed sweep from a rocprofv3 kernel trace: tools/gap_trace.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/gaps
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python3 $R/bench.py --steps 400 --warmup 400 --no-cpu-baseline --no-stream-leg --no-two-chain-leg --no-uncompressed-leg --no-scheme-legs > $O/bench.json 2> $O/err.txt
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]) for r in csv.DictReader(open(f))), key=lambda t: t[0])
# steady state: the last 300 sweeps before the profiled table pass
idx = [i for i, r in enumerate(rows) if r[2] == "hml_k_params"]
lo, hi = idx[-520], idx[-220]
gap = collections.defaultdict(list); dur = collections.defaultdict(list)
for i in range(lo + 1, hi + 1):
    a, b = rows[i - 1], rows[i]
    gap[a[2] + " -> " + b[2]].append(b[0] - a[1])
    dur[b[2]].append(b[1] - b[0])
tot = 0
for k, v in gap.items():
    m = sorted(v)[len(v) // 2]; tot += m
    print("%-50s gap median %6.2f us  mean %6.2f  n %d" % (k, m / 1e3, sum(v) / len(v) / 1e3, len(v)))
td = 0
for k, v in dur.items():
    m = sorted(v)[len(v) // 2]; td += m
    print("%-50s dur median %6.2f us" % (k, m / 1e3))
print("sum of gaps %.2f us, sum of durations %.2f us, sweep %.2f us" % (tot / 1e3, td / 1e3, (rows[hi][1] - rows[lo][1]) / 300 / 1e3))
PY
rm -f $O/*kernel_trace.csv
