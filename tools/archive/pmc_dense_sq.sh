#!/bin/bash
# SQ counters of the dense regime's kernels: tools/pmc_dense_sq.sh [c3u|c5]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sq_$1
rm -rf $O && mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/a -o run -- python3 $R/tools/time_dense.py $1 6 > /dev/null 2> $O/a.err
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/b -o run -- python3 $R/tools/time_dense.py $1 6 > /dev/null 2> $O/b.err
python3 - <<PY
import csv, glob, collections
for d in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in ("hml_k_trellis_tile<5>", "hml_k_counts_dense<5, false>", "hml_k_trellis_states<5>"):
        if k in acc:
            print(k, {c: "%.4g" % sorted(v)[len(v) // 2] for c, v in acc[k].items()})
PY
tail -n 3 $O/a.err
