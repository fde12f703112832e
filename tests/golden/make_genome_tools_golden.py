#!/usr/bin/env python3
"""Golden outputs of the reference's genome-coordinate tools: runs oracle/_ref/mapLinesToGenome and
oracle/_ref/combineCounts (built by oracle/Makefile from /root/reference/src/tools/*.cpp with the reference's own
lib/gzstream) on small hand-made genomes and commits, per case, the command line, the tool's standard output, its
exit status and the text of the files it wrote (gzip output is stored uncompressed).  The input files are made by this
script too (tests/golden/genome_tools/inputs/).  Only runs where /root/reference exists.

    make -C oracle ref && python tests/golden/make_genome_tools_golden.py
"""
import gzip
import json
import os
import random
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(REPO, "oracle", "_ref")
OUT = os.path.join(HERE, "genome_tools")
INP = os.path.join(OUT, "inputs")


def write(path, text, gz=False):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    if gz:
        with gzip.GzipFile(path, "wb", mtime=0) as f:
            f.write(text.encode())
    else:
        with open(path, "w") as f:
            f.write(text)


def genome(prefix, refseqs, counts=None, pos_gz=True):
    """refseqs: [(name, [positions])]; files PREFIX-size.csv, PREFIX-pos.csv.gz (gzip or plain text under that
    name: zlib reads both), PREFIX-count.csv.gz when counts are given"""
    total, size, pos = 0, "", ""
    for name, ps in refseqs:
        total += len(ps)
        size += "%s\t%d\t%d\n" % (name, len(ps), total)
        pos += "".join("%d\n" % p for p in ps)
    write(os.path.join(INP, prefix + "-size.csv"), size)
    write(os.path.join(INP, prefix + "-pos.csv.gz"), pos, gz=pos_gz)
    if counts is not None:
        write(os.path.join(INP, prefix + "-count.csv.gz"), "".join("%s\n" % c for c in counts), gz=True)


def make_inputs():
    shutil.rmtree(OUT, ignore_errors=True)
    rng = random.Random(7)
    # genome A: three reference sequences with gaps
    a = [("chr1", [1, 2, 3, 4, 5, 9, 10, 11, 30, 31]), ("chr2", [5, 6, 7, 100, 101, 102, 103]), ("chrX", [1, 3, 5, 7, 9])]
    genome("gA", a)
    genome("gA_plainpos", a, pos_gz=False)
    # data files for genome A (22 positions)
    write(os.path.join(INP, "lines22.txt"), "".join("v%d\t%d\n" % (i, i * i) for i in range(22)))
    write(os.path.join(INP, "lines8_w3.txt"), "".join("w%d\n" % i for i in range(8)))            # 8 windows of 3 = 24 >= 22: last incomplete
    write(os.path.join(INP, "lines7_w3.txt"), "".join("w%d\n" % i for i in range(7)))            # 21 < 22: data ends before genome
    write(os.path.join(INP, "lines9_w3.txt"), "".join("w%d\n" % i for i in range(9)))            # one window too many
    write(os.path.join(INP, "rle.txt"), "4\ts0\t0.5\n6\ts1\t0.25\n1\ts2\n11\ts3\tx\ty\n")     # 22 positions as run lengths
    write(os.path.join(INP, "rle_w2.txt"), "2\ta\n3\tb\n6\tc\n")                                 # 11 windows of 2
    write(os.path.join(INP, "rle_zero.txt"), "4\ts0\n0\ts1\n")
    write(os.path.join(INP, "rle_notab.txt"), "22\n")
    write(os.path.join(INP, "rle_short.txt"), "4\ts0\n6\ts1\n")
    write(os.path.join(INP, "lines_noeol.txt"), "".join("v%d\n" % i for i in range(21)) + "last")
    write(os.path.join(INP, "empty.txt"), "")
    # genome B: longer, random gaps, for range merging
    b, p = [], 0
    for name, n in (("s1", 60), ("s2", 1), ("s3", 40)):
        ps = []
        for _ in range(n):
            p += rng.choice([1, 1, 1, 2, 5, 40])
            ps.append(p)
        b.append((name, ps))
        p = rng.randrange(3)
    genome("gB", b)
    write(os.path.join(INP, "rleB.txt"), "10\tA\n25\tB\n25\tC\n1\tD\n20\tE\n20\tF\n")
    write(os.path.join(INP, "linesB_w10.txt"), "".join("L%d\n" % i for i in range(11)))
    # count genomes for combineCounts
    genome("c1", [("chr1", [1, 2, 3, 10]), ("chr2", [4, 5])], counts=[5, 6, 7, 8, 1, 2])
    genome("c2", [("chr2", [5, 6, 4]), ("chr1", [10, 11]), ("chrM", [2])], counts=[10, 20, 30, 3, 4, 9])
    genome("c3", [("chr1", [3, 3, 1]), ("chr1", [2])], counts=[1, 1, 1, 100])                    # duplicates, refseq listed twice
    genome("c4", [("chrE", []), ("chr1", [7])], counts=[42])                                      # a refseq without entries
    genome("c5", [("chr1", [1, 2])], counts=["-3", "x7", " 12abc"])                               # negative, text, one count too many
    genome("c6", [("chr1", [5, 6, 7])], counts=[1])                                               # count file ends early
    big = [("r%d" % k, sorted(rng.sample(range(1, 5000), 300))) for k in range(3)]
    genome("c7", big, counts=[rng.randrange(0, 90) for _ in range(900)])
    big2 = [("r%d" % k, sorted(rng.sample(range(1, 5000), 250))) for k in (2, 0, 3)]
    genome("c8", big2, counts=[rng.randrange(0, 90) for _ in range(750)])
    write(os.path.join(INP, "alt-size.tsv"), "chr1\t2\t2\n")
    write(os.path.join(INP, "alt-p.gz"), "8\n9\n", gz=True)
    write(os.path.join(INP, "alt-c.gz"), "1\n1\n", gz=True)


MAP_CASES = {
    "per_position": ["-g", "gA", "-i", "lines22.txt"],
    "per_position_coordinates": ["-g", "gA", "-c", "-i", "lines22.txt"],
    "plain_text_pos_file": ["-g", "gA_plainpos", "-i", "lines22.txt"],
    "window3_last_incomplete": ["-g", "gA", "-w", "3", "-i", "lines8_w3.txt"],
    "window3_data_ends_first": ["-g", "gA", "-w", "3", "-i", "lines7_w3.txt"],
    "window3_data_too_long": ["-g", "gA", "-w", "3", "-i", "lines9_w3.txt"],
    "run_lengths": ["-g", "gA", "-b", "-i", "rle.txt"],
    "run_lengths_window2": ["-g", "gA", "-b", "-w", "2", "-i", "rle_w2.txt"],
    "run_lengths_ranges": ["-g", "gA", "-b", "-r", "-i", "rle.txt"],
    "run_lengths_ranges_reach1": ["-g", "gA", "-b", "-r", "1", "-i", "rle.txt"],
    "run_lengths_ranges_reach3_coordinates": ["-g", "gA", "-b", "-range", "3", "-coordinates", "-i", "rle.txt"],
    "ranges_per_line": ["-g", "gA", "-r", "-i", "lines22.txt"],
    "ranges_window3": ["-g", "gA", "-r", "-w", "3", "-i", "lines8_w3.txt"],
    "ranges_window3_ends_first": ["-g", "gA", "-r", "-w", "3", "-i", "lines7_w3.txt"],
    "ranges_genome_ends_first": ["-g", "gA", "-r", "-w", "3", "-i", "lines9_w3.txt"],
    "run_length_zero": ["-g", "gA", "-b", "-i", "rle_zero.txt"],
    "run_length_without_tab": ["-g", "gA", "-b", "-i", "rle_notab.txt"],
    "run_lengths_end_early": ["-g", "gA", "-b", "-i", "rle_short.txt"],
    "last_line_without_newline": ["-g", "gA", "-i", "lines_noeol.txt"],
    "empty_data": ["-g", "gA", "-i", "empty.txt"],
    "outfile_flag": ["-g", "gA", "-i", "lines22.txt", "-o", "@OUT@/named_output.txt"],
    "missing_genome": ["-g", "nothere", "-i", "lines22.txt"],
    "genomeB_run_lengths": ["-g", "gB", "-b", "-i", "rleB.txt"],
    "genomeB_ranges": ["-g", "gB", "-b", "-r", "-i", "rleB.txt"],
    "genomeB_ranges_reach2": ["-g", "gB", "-b", "-r", "2", "-i", "rleB.txt"],
    "genomeB_ranges_reach5_window10": ["-g", "gB", "-r", "5", "-w", "10", "-c", "-i", "linesB_w10.txt"],
    "help": ["-h"],
    "stdin": ["-g", "gA", "-w", "3"],   # data on standard input: lines8_w3.txt
}
MAP_STDIN = {"stdin": "lines8_w3.txt"}

COMBINE_CASES = {
    "add_two": ["-i", "+", "c1", "c2", "-o", "@OUT@/sum"],
    "subtract": ["-i", "+", "c1", "-", "c2", "-o", "@OUT@/diff"],
    "cancel_to_zero": ["-i", "+", "c1", "-", "c1", "-o", "@OUT@/zero"],
    "duplicates_and_repeated_refseq": ["-i", "+", "c3", "c1", "-o", "@OUT@/dup"],
    "refseq_without_entries": ["-i", "+", "c4", "-o", "@OUT@/e"],
    "odd_numbers": ["-i", "-", "c5", "-o", "@OUT@/odd"],
    "count_file_ends_early": ["-i", "+", "c6", "-o", "@OUT@/short"],
    "three_hundred_each": ["-i", "+", "c7", "c8", "-", "c7", "+", "c7", "-o", "@OUT@/big"],
    "other_suffixes": ["-i", "+", "alt", "-s", "-size.tsv", "-p", "-p.gz", "-c", "-c.gz", "-o", "@OUT@/alt"],
    "first_token_not_a_sign": ["-i", "c1", "-o", "@OUT@/x"],
    "missing_input": ["-i", "+", "nothere", "-o", "@OUT@/x"],
    "normalization_not_implemented": ["-i", "+", "c1", "-n", "c2", "-o", "@OUT@/x"],
    "no_out_prefix": ["-i", "+", "c1"],
    "help_needs_the_other_flags": ["-h", "-i", "+", "c1", "-o", "@OUT@/x"],
}


def read_outputs(d):
    files = {}
    for name in sorted(os.listdir(d)):
        raw = open(os.path.join(d, name), "rb").read()
        if raw[:2] == b"\x1f\x8b":
            raw = gzip.decompress(raw)
        files[name] = raw.decode("latin-1")
    return files


def run_cases(tool, cases, stdin_of=None):
    exe = os.path.join(REF, tool)
    if not os.path.exists(exe):
        raise SystemExit("reference tool missing: run `make -C oracle ref` in the build container")
    out = {}
    for name, argv in cases.items():
        with tempfile.TemporaryDirectory() as tmp:
            args = [a.replace("@OUT@", tmp) for a in argv]
            stdin = open(os.path.join(INP, stdin_of[name])) if stdin_of and name in stdin_of else subprocess.DEVNULL
            r = subprocess.run([exe] + args, cwd=INP, stdin=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            # an escaped exception aborts the reference (SIGABRT) with "what():  MESSAGE" on standard error
            msg = None
            for line in r.stderr.decode("latin-1").splitlines():
                if "what():" in line:
                    msg = line.split("what():", 1)[1].strip()
            out[name] = {"argv": argv, "stdin": stdin_of.get(name) if stdin_of else None, "ok": r.returncode == 0,
                         "stdout": r.stdout.decode("latin-1").replace(tmp, "@OUT@"), "message": msg, "files": read_outputs(tmp)}
            print("%-18s %-40s rc %4d  %5d bytes out  %s" % (tool, name, r.returncode, len(r.stdout), msg or ""))
    return out


def main():
    make_inputs()
    golden = {"mapLinesToGenome": run_cases("mapLinesToGenome", MAP_CASES, MAP_STDIN),
              "combineCounts": run_cases("combineCounts", COMBINE_CASES)}
    with open(os.path.join(OUT, "expected.json"), "w") as f:
        json.dump(golden, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
