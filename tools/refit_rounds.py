"""Per repair round of the weakly compressed sweep: durations of hml_k_trellis_refit / _verify_list launches from a rocprofv3
--kernel-trace CSV (launch order; four rounds a sweep).  usage: python tools/refit_rounds.py <dir with *kernel_trace.csv>"""
import csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(files[0])), key=lambda r: int(r["Start_Timestamp"]))
for name in ("hml_k_trellis_refit", "hml_k_trellis_verify_list", "hml_k_trellis_serial", "hml_k_trellis_verify<"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]]
    per = 4 if ("refit" in name or "verify_list" in name) else 1
    for k in range(per):
        v = sorted(d[k::per])
        if v:
            print("%-28s round %d: %4d launches, us min %.1f median %.1f p90 %.1f max %.1f, above 8 us: %d" % (
                name, k + 1, len(v), v[0], v[len(v) // 2], v[int(len(v) * 0.9)], v[-1], sum(1 for x in v if x > 8.0)))
