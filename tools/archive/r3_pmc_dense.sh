#!/bin/bash
# counters of the dense regime's kernels (SQ issue/wait counters, then HBM traffic in passes of their own):
#   tools/r3_pmc_dense.sh c3u|c5 [tag]      (HML_LIBRARY / HML_TRELLIS_L / HML_TRELLIS_ROWS are passed through)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$1${2:+_$2}
rm -rf $O && mkdir -p $O
export HML_TRELLIS_TUNE=${HML_TRELLIS_TUNE:-0}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/a -o run -- python3 $R/tools/time_dense.py $1 6 0 24 > /dev/null 2> $O/a.err
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/b -o run -- python3 $R/tools/time_dense.py $1 6 0 24 > /dev/null 2> $O/b.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o run -- python3 $R/tools/time_dense.py $1 6 0 24 > /dev/null 2> $O/f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o run -- python3 $R/tools/time_dense.py $1 6 0 24 > /dev/null 2> $O/w.err
python3 - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for d in ("a", "b", "f", "w"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in acc:
        if any(t in k for t in ("trellis_rows", "trellis_tile", "counts_dense", "trellis_states", "compact_scan", "compact_scatter", "trellis_refit")):
            for c, v in acc[k].items():
                res[k][c] = sorted(v)[len(v) // 2]
for k, v in res.items():
    print(k, {c: "%.4g" % x for c, x in v.items()})
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        print("   HBM bytes per launch: raw %.3f GB, fetch doubled %.3f GB" % (1024 * (v["FETCH_SIZE"] + v["WRITE_SIZE"]) / 1e9, 1024 * (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) / 1e9))
json.dump(res, open("$O/summary.json", "w"), indent=1)
PY
tail -n 2 $O/a.err
