set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp REPS=2
O=$GRAFT_REPO_ROOT/gpurun_out/many_pmc
mkdir -p $O
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/p1 -o run -- python3 tools/multi_chain.py 16 300 c3_1e8_k5_dynamic attached > $O/p1.out 2> $O/p1.err || { tail -5 $O/p1.err; exit 1; }
timeout -k 10 500 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $O/p2 -o run -- python3 tools/multi_chain.py 16 300 c3_1e8_k5_dynamic attached > $O/p2.out 2> $O/p2.err || { tail -5 $O/p2.err; exit 1; }
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/many_pmc"
for p in ("p1", "p2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(O + "/" + p + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:44]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
    with open(O + "/" + p + "_summary.txt", "w") as out:
        for k in agg:
            if "hml_m_" in k:
                out.write(k + "\n")
                for c, v in sorted(agg[k].items()): out.write("   %-22s %.4g per launch (%d)\n" % (c, v / cnt[(k, c)], cnt[(k, c)]))
    print(open(O + "/" + p + "_summary.txt").read())
PY
find $O -name '*.csv' -size +4M -delete
cat $O/p1.out | tail -2
