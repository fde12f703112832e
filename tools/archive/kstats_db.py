"""Dump the kernel summary of a rocprofv3 results database (`rocprofv3 --kernel-trace --stats`, rocpd output) as CSV:
    python tools/kstats_db.py gpurun_out/prof_text/text_results.db > profiles/NAME.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print("Name,Calls,TotalDurationUs,AverageUs,Percentage")
for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    print('"%s",%d,%.3f,%.3f,%.2f' % (name, calls, total, avg, pct))
