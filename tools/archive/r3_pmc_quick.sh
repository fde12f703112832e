#!/bin/bash
# instruction counters of the first trellis pass, one pass: tools/r3_pmc_quick.sh c3u|c5 [T]   (HML_LIBRARY, HML_TRELLIS_L passed through)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcq_$$
rm -rf $O && mkdir -p $O
export HML_TRELLIS_TUNE=0
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/a -o run -- python3 $R/tools/time_dense.py $1 4 ${2:-0} 12 > /dev/null 2> $O/a.err
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    if "trellis_rows" in k or "trellis_tile" in k:
        print(k, {c: "%.4g" % sorted(v)[len(v) // 2] for c, v in acc[k].items()})
PY
rm -rf $O
