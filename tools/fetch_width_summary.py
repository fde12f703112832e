"""FETCH_SIZE per launch and kernel from a rocprofv3 --pmc FETCH_SIZE CSV of tools/fetch_width_bench (the counter is in KB, summed over the
counter's instances per dispatch).  usage: python tools/fetch_width_summary.py <dir>"""
import csv, glob, sys, collections
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(files[0])):
    if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
        per[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k, d in per.items():
    v = sorted(d.values())
    med = v[len(v) // 2]
    print("%-40s %d launches, counter median %.1f KB = %.4f GB per launch = %.3f of the 1.0737 GB read" % (k[:40], len(v), med, med * 1024 / 1e9, med * 1024 / (1 << 30)))
