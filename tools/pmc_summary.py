"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into profiles/<name>.json.
usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> "<command>"
Counter unit: KB per dispatch.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B/lane)
coalesced streaming reads at 64 bytes, so for such kernels it is doubled (MI355X_MICROARCH.md, HBM section);
for gathers and narrow loads the factor is uncalibrated and both the raw and the doubled figure are given."""
import csv, glob, json, statistics, sys


def medians(d, counter):
    out = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            out.setdefault(name, []).append(float(r["Counter_Value"]))
    return {k: (statistics.median(v), len(v)) for k, v in out.items()}


fetch, write = medians(sys.argv[1], "FETCH_SIZE"), medians(sys.argv[2], "WRITE_SIZE")
kernels = {}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("hml_"):
        continue
    f, n = fetch.get(k, (0.0, 0))
    w, _ = write.get(k, (0.0, 0))
    kernels[k] = {"FETCH_SIZE_KB_median": f, "WRITE_SIZE_KB_median": w, "launches": n,
                  "bytes_raw": 1024 * (f + w), "bytes_fetch_doubled": 1024 * (2 * f + w)}
out = {"command": sys.argv[4], "kernels": kernels,
       "note": "separate passes per counter; unit KB per dispatch, medians over the launches of a kernel.  bytes_fetch_doubled "
               "applies the gfx950 correction for wide coalesced streaming reads (exact for hml_k_compact_scan); for the "
               "gather-dominated kernels the true figure lies between bytes_raw and bytes_fetch_doubled"}
for k, v in kernels.items():
    if k.startswith("hml_k_blocks_fused"):
        out["scan_kernel"] = dict(v, kernel=k, hbm_bytes_per_launch_corrected=v["bytes_fetch_doubled"])
    if k == "hml_k_compact_scan":
        out["float_scan_kernel"] = dict(v, kernel=k, hbm_bytes_per_launch_corrected=v["bytes_fetch_doubled"])
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("scan_kernel", "float_scan_kernel") if k in out}, indent=1))
