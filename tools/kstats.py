"""Print a rocprofv3 kernel_stats.csv (found under the given directory) as a compact table."""
import csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not files:
    sys.exit("no kernel_stats.csv under " + sys.argv[1])
for r in csv.DictReader(open(files[0])):
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(6), ("%.2f" % (float(r["AverageNs"]) / 1e3)).rjust(10),
          r["Percentage"].rjust(7), r["MinNs"].rjust(9), r["MaxNs"].rjust(9))
