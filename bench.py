#!/usr/bin/env python3
"""Benchmark of the Forward-Backward Gibbs sweep over wavelet-compressed blocks on MI355X.

A "step" is one Gibbs sweep (dynamic wavelet recompression, forward trellis, backward sampling,
count pass, conjugate resampling) of one chain over a synthetic piecewise-constant Gaussian trace
that is already resident in HBM.  Workload = BASELINE.json's headline configuration (10^8 positions,
5 states, dynamic blocks).  With N GPUs every rank runs an independent chain on the same trace
(weak scaling, no data-path collective); value = block updates of all ranks / max-over-ranks time.

    python bench.py --gpus 1 --steps 1000 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves

WORKLOADS = {
    # name: (T, K, levels, sigma, mean dwell, data seed)
    "c3_1e8_k5_dynamic": (100_000_000, 5, [-2, -1, 0, 1, 2], 0.3, 5000.0, 3),
    "c2_1e7_k5": (10_000_000, 5, [-2, -1, 0, 1, 2], 0.3, 5000.0, 2),
    "c1_1e5_k3": (100_000, 3, [-1, 0, 1], 0.2, 2000.0, 1),
    "c4_1e8_k10": (100_000_000, 10, [x - 4.5 for x in range(10)], 0.3, 5000.0, 4),
    # simulated WGS read depth (Poisson-lognormal, chr1 scale), 5-state CNV model: levels/sigma unused
    "c5_2.5e8_depth_k5": (250_000_000, 5, None, 0.15, 15.0, 5),
}


PMC_FILE = "profiles/round4_pmc_hbm_traffic.json"
# "c3u" = the uncompressed leg (same trace, every position its own block), collected with tools/time_dense.py
PMC_FILES = {"c3_1e8_k5_dynamic": PMC_FILE, "c3u": "profiles/round4_pmc_hbm_traffic_c3u.json",
             "c5_2.5e8_depth_k5": "profiles/round4_pmc_hbm_traffic_c5.json", "c4_1e8_k10": "profiles/round4_pmc_hbm_traffic_c4.json"}
DENSE_KERNEL = "hml_k_trellis_rows"   # first pass over the trellis of a weakly compressed sweep (hml_k_trellis_rows.h)
MIN_BRACKETS = 32                     # launches of the roofline kernel that are bracketed by events, whatever --steps is
LEG_MIN_STEPS = 1000                  # every auxiliary leg times at least this many sweeps, whatever --steps is (the headline keeps
                                      # the driver's --steps): a 20-sweep leg is a 1-2 ms region in which thread start-up and the first
                                      # launches dominate (round 3's driver line showed three chains at 0.4x one chain that way), and
                                      # with 200 the threaded chain legs still read 20-40 % low (18 ms regions); 1000 sweeps cost < 0.2 s a leg


def family_kernels(K):
    """kernel families timed by the library's event brackets (hml_profile_enable) -> the kernel INSTANTIATION each one launches
    in the default dynamic FB sweep, as rocprofv3 prints it (round 3 matched on the name up to '<' and took the first hit:
    hml_k_counts<5, false, false> - the mixture sweep's - for the FB sweep's hml_k_counts<5, true, false>)"""
    return {
        "blocks_compact": "hml_k_blocks_fused<%d>" % K,
        "forward": "hml_k_forward<%d>" % K,
        "backward_maps": "hml_k_backward_maps<%d>" % K,
        "backward_chain": "hml_k_backward_chain<%d>" % K,
        "counts": "hml_k_counts<%d, true, false>" % K,
        "params": "hml_k_params<%d>" % K,
    }


def dense_kernels(K, shared_sums):
    return {"blocks_compact": "hml_k_compact_scan_bits", "blocks_scatter": "hml_k_compact_scatter_bits",
            "trellis": "%s<%d, %s>" % (DENSE_KERNEL, K, "true" if shared_sums else "false"),
            "trellis_repair": "hml_k_trellis_verify + refit + serial", "backward_chain": "hml_k_trellis_super + chain + states",
            "counts": "hml_k_counts_dense<%d, false>" % K, "params": "hml_k_params<%d>" % K}


def pmc_table(workload):
    """Per-kernel memory traffic from the committed rocprofv3 --pmc passes of this same command (separate FETCH_SIZE and
    WRITE_SIZE passes, tools/pmc_summary.py): bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (the gfx950 correction of
    MI355X_MICROARCH.md for FETCH_SIZE).  Counters cannot be collected from inside the timed run, so the JSON line
    quotes the profile; Infinity-Cache hits are counted in FETCH_SIZE, so the figure is an upper bound on HBM bytes."""
    try:
        if workload not in PMC_FILES:
            return {}
        with open(os.path.join(REPO, PMC_FILES[workload])) as f:
            return json.load(f)["kernels"]
    except Exception:
        return {}


def pmc_traffic(workload, kernel, key="bytes_fetch_doubled"):
    """bytes per launch of `kernel`: 2 x FETCH_SIZE + WRITE_SIZE, or with key = "bytes_raw" FETCH_SIZE + WRITE_SIZE as
    counted.  `kernel` is the full instantiation ("hml_k_counts<5, true, false>"); a bare name matches only when the profile
    holds exactly ONE instantiation of it."""
    tab = pmc_table(workload)
    if kernel in tab:
        return tab[kernel][key]
    if "<" not in kernel:
        hits = [k for name, k in tab.items() if name.split("<")[0] == kernel]
        if len(hits) == 1:
            return hits[0][key]
    return None


def host_cpu():
    model, n = "unknown", os.cpu_count() or 0
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return model, n


def cpu_baseline(x, K, seed, budget_s=16.0, ref_sweeps=24, parallel=True):
    """The CPU restatement in reference mode (sequential mt19937, glibc math, pointer-jumping block
    enumeration), timed on this box's host cores; one thread like the reference."""
    from tests import oracle_lib as ol
    o = ol.OracleChain(K=K, seed=seed, rng=ol.RNG_MT, math=ol.MATH_LIBM, reduce=ol.REDUCE_REF)
    o.load(x)
    o.autoprior()
    o.init_model()
    o.set_record(marginals=False)
    # burn in a few sweeps so the timed ones see the compression level of a running chain
    t = o.time_sweeps("F", 3)
    per = max(t / 3, 1e-6)
    n = int(max(ref_sweeps, min(200, budget_s / per)))
    b0 = o.total_blocks()
    # (the first ref_sweeps sweeps of the chain are what the reference binary is asked for below: same seed, same chain)
    t_a = o.time_sweeps("F", ref_sweeps - 3)
    blocks_ref = o.total_blocks()
    t_b = o.time_sweeps("F", n - (ref_sweeps - 3)) if n > ref_sweeps - 3 else 0.0
    t = t_a + t_b
    blocks = o.total_blocks() - b0
    o.close()
    model, total = host_cpu()
    out = {"value": blocks / t, "unit": "block-updates/s", "cores": 1, "kind": "port", "host_cpu": model, "host_cores": total,
           "sample": "%d sweeps of the same %d-position trace (reference-mode CPU restatement, %.1f ms/sweep); one thread, "
                     "like the reference (src/main.cpp:108)" % (n, x.size, 1e3 * t / n)}
    try:
        ref, par = reference_binary_baseline(x, K, seed, ref_sweeps, blocks_ref, min(8, total) if parallel else 0)
        if ref:
            out["reference_binary"] = ref
        if par:
            out["chain_parallel"] = par
    except Exception as e:   # the prebuilt binary is optional on the GPU box
        out["reference_binary"] = {"error": str(e)[:200]}
    return out


def reference_binary_baseline(x, K, seed, sweeps, blocks, n_parallel=0):
    """The UNMODIFIED reference binary (oracle/_ref/hammlet, prebuilt in the build container from the reference's own
    main.cpp; absent -> None) on the WHOLE trace as text, a bounded number of sweeps: sweep time = wall(-i F n 0) -
    wall(-i F 0 0) (start-up and the reference's text parsing, reported separately, cancel); `blocks` = the block updates of
    those sweeps, from the restatement's run of the same chain (it reproduces the reference's files byte for byte, hence its
    block structures).  n_parallel > 0: SURVEY.md 8d's chain-parallel CPU figure - the reference is single-threaded
    (src/main.cpp:108), so N chains are N processes of it: the same two commands, N at a time (every process the same seed:
    identical work, so the aggregate is N x blocks over the slowest process's sweep time)."""
    import subprocess
    import tempfile
    exe = os.path.join(REPO, "oracle", "_ref", "hammlet")
    if not os.path.exists(exe):
        return None, None
    from tests import oracle_lib as ol
    lib = ol.load()
    with tempfile.TemporaryDirectory(dir=os.environ.get("HML_BENCH_TMP") or None) as tmp:
        txt = os.path.join(tmp, "trace.txt")
        import ctypes as C
        lib.orc_write_text.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_int]
        xc = np.ascontiguousarray(x, dtype=np.float32)
        if lib.orc_write_text(xc.ctypes.data, xc.size, txt.encode(), max(1, min(16, os.cpu_count() or 1))) != 0:
            raise RuntimeError("could not write the trace as text")

        def run(n, procs=1):
            t0 = time.perf_counter()
            ps = [subprocess.Popen([exe, "-f", txt, "-a", "-s", str(K), "-R", str(seed), "-i", "F", str(n), "0", "-w",
                                    "-o", os.path.join(tmp, "ref%d-" % i), ".csv"], stdout=subprocess.DEVNULL) for i in range(procs)]
            rcs = [q.wait() for q in ps]
            if any(rcs):
                raise RuntimeError("the reference binary failed")
            return time.perf_counter() - t0
        t_zero = run(0)
        t_n = run(sweeps)
        par = None
        if n_parallel > 1:
            p_zero = run(0, n_parallel)
            p_n = run(sweeps, n_parallel)
            pdt = max(p_n - p_zero, 1e-9)
            par = {"value": n_parallel * blocks / pdt, "unit": "block-updates/s", "cores": n_parallel, "kind": "reference",
                   "sample": "%d processes of the unmodified reference binary at once (min(8, host cores) independent chains, SURVEY.md 8d), each "
                             "on the whole %d-position trace, %d sweeps: %.1f ms per sweep-round (start-up + text parsing %.1f s, subtracted)"
                             % (n_parallel, x.size, sweeps, 1e3 * pdt / sweeps, p_zero)}
    dt = max(t_n - t_zero, 1e-9)
    return {"value": blocks / dt, "unit": "block-updates/s", "cores": 1, "kind": "reference", "sample_positions": int(x.size),
            "sample": "unmodified reference binary on the whole %d-position trace as text, %d sweeps: %.1f ms/sweep, "
                      "%.0f blocks/sweep (start-up + text parsing %.1f s, subtracted)" % (x.size, sweeps, 1e3 * dt / sweeps, blocks / sweeps, t_zero)}, par


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="c3_1e8_k5_dynamic", choices=sorted(WORKLOADS))
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="extra profiled pass: per-kernel-family times")
    ap.add_argument("--no-stream-leg", action="store_true", help="skip the second leg (float weight stream instead of the summary)")
    ap.add_argument("--no-two-chain-leg", action="store_true", help="skip the third leg (two and three chains sharing one GPU)")
    ap.add_argument("--no-scheme-legs", action="store_true", help="skip the recorded / config-2 mixture / config-2 static legs")
    ap.add_argument("--no-uncompressed-leg", action="store_true", help="skip the fourth leg (same trace, weights x 1e9: every position its own block)")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the bounded legs of BASELINE configs 4 and 5 (one chain of each on this GPU)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    # HML_BENCH_FORCE_DIST=1: take the multi-rank branches (process group over RCCL, communicator id by broadcast, pooling
    # through hml_pool_marginals) whatever the world size - what tests/test_gpu_bench_dist.py runs on a one-GPU box
    dist_mode = world > 1 or os.environ.get("HML_BENCH_FORCE_DIST") == "1"
    # HML_BENCH_NO_TORCH=1 (one rank; the counter passes of tools/round_profiles.sh): PyTorch is not loaded - rocprofv3 --pmc ended in a
    # segmentation fault inside the profiler with it in the process on round 5's boxes.  Chains synchronise their own streams
    # (hml_sync) in every timed region, so the single-rank timings do not need it; the device-memory figures of the chain legs do.
    no_torch = (not dist_mode) and os.environ.get("HML_BENCH_NO_TORCH") == "1"
    torch = None
    if not no_torch:
        import torch
    dist = None
    if dist_mode:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif torch is not None:
        torch.cuda.set_device(local_rank)

    import hammlet_amd
    T, K, levels, sigma, dwell, data_seed = WORKLOADS[args.workload]
    nthr = max(1, min(64, (os.cpu_count() or 8) // max(1, world)))
    if os.environ.get("HML_BENCH_THREADS"):   # (threads of the synthetic-trace generator)
        nthr = max(1, int(os.environ["HML_BENCH_THREADS"]))
    if levels is None:
        x = hammlet_amd.synth_depth(T, depth=dwell, ln_sigma=sigma, seed=data_seed, nthreads=nthr)
    else:
        x = hammlet_amd.synth_gauss(T, K, levels, sigma, dwell, data_seed, nthreads=nthr)

    def barrier():
        if dist is not None:
            dist.barrier()
        if torch is not None:
            torch.cuda.synchronize()

    def run_leg(weight_summary, profile_level, steps=None):
        """warm-up + K timed sweeps of a fresh chain; returns (chain, elapsed, blocks, stats0, stats1)"""
        steps = steps or args.steps
        ch = hammlet_amd.Chain(device=local_rank, seed=args.seed, chain_id=rank)
        ch.set_option("weight_keys", 1 if weight_summary else 0)
        ch.load(x)
        prior = ch.autoprior(0.2, 0.9)
        ch.set_model(K, prior)
        ch.sample_prior()
        ch.set_recording(marginals=False)
        ch.iterate("F", args.warmup, 0)
        ch.sync()
        s0 = ch.stats()
        ch.profile_enable(profile_level)   # HIP events around the dominant kernel only, on the chain's own stream
        barrier()
        t0 = time.perf_counter()
        ch.iterate("F", steps, 0)
        ch.sync()
        barrier()
        t1 = time.perf_counter()
        ch.profile_enable(0)
        s1 = ch.stats()
        return ch, t1 - t0, s1["block_updates"] - s0["block_updates"], s0, s1

    # headline leg: the library's default path - the block structure is recomputed in every sweep (dynamic blocks)
    # weakly compressed workloads (config 5: millions of blocks per sweep) spend their time in the first trellis pass
    # (hml_k_trellis_rows): the same level-1 brackets, every 32nd launch (bracketing every family of every sweep, as
    # rounds 1-2 did there, costs the timed region 3 %)
    dense_workload = levels is None
    FAMILY_KERNEL = family_kernels(K)
    DENSE_INST = dense_kernels(K, False)["trellis"]
    leg_steps = max(args.steps, LEG_MIN_STEPS)
    chain, elapsed, blocks, st0, st1 = run_leg(weight_summary=True, profile_level=1)
    roof_family = "trellis" if dense_workload else "blocks_compact"
    scan_ms, scan_n = chain.profile_get(roof_family)           # HIP events around the kernel's launches
    null_ms, null_n = chain.profile_get("event_null")          # empty brackets recorded right behind them
    brackets_timed = scan_n
    if scan_n < MIN_BRACKETS:
        # the timed region brackets every 32nd launch of the roofline kernel (a bracket costs ~5 us of stream time): with few
        # steps that is one sample or none.  A short pass OUTSIDE the timed region, same chain, same level, tops the sample
        # up to MIN_BRACKETS launches; kernel_avg_us is the average over all of them.
        per = 32
        chain.profile_enable(1)
        chain.iterate("F", (MIN_BRACKETS - scan_n) * per, 0)
        chain.sync()
        chain.profile_enable(0)
        scan_ms, scan_n = chain.profile_get(roof_family)
        null_ms, null_n = chain.profile_get("event_null")

    per_rank_ms = None
    if dist is not None:
        mine = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [1e3 * float(t.item()) / args.steps for t in every]
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        bb = torch.tensor([blocks], dtype=torch.int64, device="cuda")
        dist.all_reduce(bb, op=dist.ReduceOp.SUM)
        blocks_all = int(bb.item())
    else:
        blocks_all = blocks

    out = None
    if rank == 0:
        B_avg = blocks / max(1, args.steps)
        # an event pair measures ~3-5 us with nothing in between on this stack (marker packets + barrier): the
        # kernel's duration is the bracket minus the empty bracket (both reported; rocprofv3 agrees with the net)
        scan_raw_s = (scan_ms / max(1, scan_n)) * 1e-3
        null_s = (null_ms / max(1, null_n)) * 1e-3
        scan_avg_s = max(scan_raw_s - null_s, 1e-9)
        # The timed kernel is hml_k_blocks_fused: block starts from the weights, their order, block statistics,
        # emission terms.  It streams a one-byte-per-16-positions summary of the weights and opens only the groups
        # that can hold a block start (DESIGN.md "K4"), so it moves far fewer bytes than the 4 B/position weight stream
        # of SURVEY.md 8d.  `frac` prices it on the bytes the memory system really moved for it - the PMC counters of the
        # committed profile of this command (2 x FETCH_SIZE + WRITE_SIZE per launch) - and is therefore a fraction of
        # peak (<= 1); it is small because the kernel is bound by latency (three dependent memory round trips and one
        # inter-workgroup hand-off), not by bandwidth.  The algorithmic figure of SURVEY 8d (4 T + 20 B: weight stream +
        # one start and two integral-array gathers per block) is reported separately as what the launch REPLACES.
        algo_bytes = 4.0 * T + 20.0 * B_avg
        phys_bytes = T / 16.0 + B_avg * (64.0 + 128.0 + 12.0 + 8.0 * K)     # estimate, used when no profile is committed
        traffic = pmc_traffic(args.workload, FAMILY_KERNEL["blocks_compact"])
        if dense_workload:
            # hml_k_trellis_tile: per block a start and two integral-array gathers read, statistics and a 4-byte map written
            # (DESIGN.md 3a); the estimate used without a committed profile is what the counters showed on C3u (125 B/block)
            algo_bytes = B_avg * (4.0 + 16.0 + 8.0 + 4.0)
            phys_bytes = 50.0 * B_avg
            traffic = pmc_traffic(args.workload, DENSE_INST)
            if null_n == 0:               # (profile level 2 records no empty brackets: the usual 5.3 us)
                null_s = 5.3e-6
                scan_avg_s = max(scan_raw_s - null_s, 1e-9)
        moved = traffic if traffic else phys_bytes
        traffic_raw = pmc_traffic(args.workload, DENSE_INST if dense_workload else FAMILY_KERNEL["blocks_compact"], "bytes_raw")
        # `frac` is the LOWER bound: FETCH_SIZE + WRITE_SIZE as counted; `frac_upper` prices the fetches doubled (the gfx950
        # correction of the microarchitecture guide, calibrated on wide streaming reads - this kernel gathers, so the truth lies
        # between the two).  Without a committed profile both come from the estimate.
        moved_raw = traffic_raw if traffic_raw else moved
        achieved = moved_raw / scan_avg_s / 1e9
        achieved_upper = moved / scan_avg_s / 1e9
        sweep_bytes = 4.0 * T + B_avg * (36 + 8 * K)   # SURVEY.md 8d bytes_iter
        frac = achieved / HBM_PEAK_GBS
        frac_upper = achieved_upper / HBM_PEAK_GBS
        assert frac <= frac_upper <= 1.0, "a roofline fraction above 1 means the byte count is not what the kernel moves"
        out = {
            "metric": "block-updates/sec (Gibbs sweep) + HBM GB/s, 10^8 pos / 5 states",
            "value": blocks_all / elapsed,
            "unit": "block-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": args.workload, "positions": T, "states": K, "blocks_per_sweep": B_avg,
                       "compression": T / max(B_avg, 1.0), "block_structure": "dynamic", "chains": world,
                       "parallelism": "chain-parallel x%d" % world,
                       "cpu_baseline_sample": "port and reference binary: the same 10^8-position trace, a bounded number of sweeps"},
            "roofline": {"bound": "hbm", "kernel": DENSE_KERNEL + " (emission terms + forward filter + backward candidate maps)" if dense_workload
                         else "hml_k_blocks_fused (block scan + block statistics + emission terms)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac,
                         "achieved_upper": achieved_upper, "frac_upper": frac_upper,
                         "traffic": traffic, "traffic_source": PMC_FILES.get(args.workload) if traffic else None,
                         "bytes_priced": ("frac: pmc FETCH_SIZE + WRITE_SIZE per launch as counted (lower bound); frac_upper: 2*FETCH_SIZE + "
                                          "WRITE_SIZE (the guide's gfx950 correction; = `traffic`)") if traffic else "estimate (no committed profile)",
                         "traffic_raw": traffic_raw,
                         "kernel_avg_us": 1e6 * scan_avg_s, "kernel_bracket_us": 1e6 * scan_raw_s,
                         "empty_bracket_us": 1e6 * null_s, "launches": scan_n, "launches_in_timed_region": brackets_timed,
                         "launches_note": "HIP events on the chain's stream; launches beyond those of the timed region come from a "
                                          "short extra pass of the same chain outside it (at least %d in all)" % MIN_BRACKETS,
                         "limiter": "vector issue: about 530 VALU instructions per block-row, a third of them double precision (DESIGN.md 3a)" if dense_workload else
                                    "latency: three dependent memory round trips + one inter-workgroup hand-off per launch",
                         "profile_pair": None if dense_workload else
                                         "rocprofv3 adds 1.5-2 us to every dispatch of this kernel: the line that pairs with "
                                         "profiles/round5_kernel_stats_c3_bench.csv is profiles/round5_bench_c3_under_rocprof.json",
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "algorithmic_equivalent_gbs": algo_bytes / scan_avg_s / 1e9,
                         "algorithmic_speedup_vs_peak_float_stream": algo_bytes / scan_avg_s / 1e9 / HBM_PEAK_GBS,
                         "sweep_frac": sweep_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "sweep_frac_note": "SURVEY.md 8d: algorithmic bytes of a whole sweep (4 T + B (36 + 8 K)) / sweep time / 8 TB/s"},
            "forward_refits": st1["forward_refits"] - st0["forward_refits"],
            "forward_serial": st1["forward_serial"] - st0["forward_serial"],
        }
        assert out["roofline"]["sweep_frac"] <= 1.0

    # per-kernel table (not part of the timed region): every launch of 200 further sweeps bracketed by HIP events on the
    # chain's stream, next to the PMC traffic of the committed profile
    if rank == 0 and world == 1:
        fam_kernel = dict(FAMILY_KERNEL)
        n_extra = 200
        if dense_workload:
            fam_kernel = dense_kernels(K, False)
            n_extra = 10
        names = list(fam_kernel)
        before = {nm: chain.profile_get(nm) for nm in names + ["event_null"]}
        chain.profile_enable(2)
        chain.iterate("F", n_extra, 0)
        chain.sync()
        chain.profile_enable(0)
        table = {}
        for nm in names:
            ms, n = chain.profile_get(nm)
            dn = n - before[nm][1]
            if dn <= 0:
                continue
            us = 1e3 * (ms - before[nm][0]) / dn
            tr = pmc_traffic(args.workload, fam_kernel[nm])
            tr_raw = pmc_traffic(args.workload, fam_kernel[nm], "bytes_raw")
            row = {"bracket_us": round(us, 2), "launches_per_sweep": round(dn / n_extra, 2), "traffic": tr, "traffic_raw": tr_raw}
            net = max(us - 1e6 * null_s, 0.5)
            row["kernel_us"] = round(net, 2)
            if tr:   # frac: counter bytes as counted (lower bound), frac_upper: fetches doubled - as in `roofline`
                row["frac_upper"] = round(tr / (net * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                row["frac"] = round((tr_raw or tr) / (net * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                assert row["frac"] <= row["frac_upper"] <= 1.0
            table[fam_kernel[nm]] = row
        out["kernels"] = table

    if args.breakdown and rank == 0:
        names = ("blocks_compact", "blocks_scatter", "stats_emission", "forward", "backward_maps", "backward_chain", "counts", "params")
        before = {nm: chain.profile_get(nm) for nm in names}
        chain.profile_enable(2)
        chain.iterate("F", min(50, args.steps), 0)
        chain.sync()
        chain.profile_enable(0)
        fam = {}
        for nm in names:
            ms, n = chain.profile_get(nm)
            fam[nm] = round(1e3 * (ms - before[nm][0]) / max(1, n - before[nm][1]), 2)
        out["kernel_us_per_sweep"] = fam

    # what the other ranks cost this one: rank 0 repeats the timed region ALONE (the others wait at the barrier), so the line
    # carries the ratio the scaling curve should show if nothing but the shared host and power budget couples the chains
    if dist_mode:
        if rank == 0:
            torch.cuda.synchronize()
            ta = time.perf_counter()
            chain.iterate("F", args.steps, 0)
            chain.sync()
            alone = time.perf_counter() - ta
            out["per_rank_ms_per_step"] = per_rank_ms
            out["rank0_alone_ms_per_step"] = 1e3 * alone / args.steps
            out["efficiency_vs_rank0_alone"] = alone / elapsed
            out["efficiency_note"] = ("rank 0's time for the same number of sweeps with the other ranks idle (right after the timed region: "
                                      "an older chain, which is if anything faster) / the slowest rank's time in the timed region")
        barrier()

    # chain-parallel pooling (not timed): a few recorded sweeps, then the library's own collective - every rank joins
    # an RCCL communicator (hml_pool_create) and hml_pool_marginals relabels, pools and installs the marginals.  BOTH forms of
    # the collective run (round 5): the ranks' boundary lists through ncclAllGather (what the library picks for a strongly
    # compressed chain) on this rank's chain, and the dense int32 [K+1][T+1] payload through ncclAllReduce(sum) - the
    # gigabytes SURVEY.md section 5 prices - on a second chain attached to the same observations.
    if dist_mode:
        # (the headline above is complete by now: a pooling that fails - the collective has never run on more than one GPU in the
        # build's environment - is reported in the line, it does not take the line away)
        try:
            from hammlet_amd import chains
            pool = chains.make_pool(local_rank, always_broadcast=True)
            second = hammlet_amd.Chain(device=local_rank, seed=args.seed + 1000, chain_id=rank)
            second.attach(chain)
            second.set_model(K, second.autoprior(0.2, 0.9))
            second.sample_prior()
            pooling = {"transport": "RCCL inside libhammlet_hip.so (hml_pool_marginals): the ranks' boundary lists through ncclAllGather when they are "
                                    "at most an eighth of the dense int32 [K+1][T+1] payload, else that payload through ncclAllReduce(sum); "
                                    "both forms are forced here in turn (hml_pool_set_form)",
                       "dense_payload_bytes": 4 * ((K + 1) * (T + 1) + 1 + K), "ranks": world}
            n_rec = 2
            for form, name, ch in ((0, "default", chain), (1, "dense", second)):
                ch.set_recording(marginals=True)
                ch.iterate("F", 5 * n_rec, 5)
                ch.sync()
                pool.set_form(form)
                barrier()
                tp0 = time.perf_counter()
                seg, cnt, _ = chains.pooled_marginals(ch, pool)
                barrier()
                secs = time.perf_counter() - tp0
                # every position was recorded n_rec times by every rank: the pooled counts of every segment add up to ranks x recorded
                rows = cnt.sum(axis=1)
                assert int(seg.sum()) == T and int(rows.min()) == world * n_rec and int(rows.max()) == world * n_rec, \
                    "pooled marginals: row sums %d..%d, expected ranks x recorded = %d" % (int(rows.min()), int(rows.max()), world * n_rec)
                info, last = pool.info(), pool.last()
                pooling[name] = {"form": last["form"], "list_slot_segments": last["entries"], "collective_bytes": info["last_bytes"],
                                 "collective_ms": info["last_allreduce_ms"], "seconds_incl_export_and_install": secs,
                                 "pooled_segments": int(len(seg)), "counts_per_position": int(rows[0]),
                                 "algbw_GBps": info["last_bytes"] / max(info["last_allreduce_ms"], 1e-6) / 1e6}
                pooling["rccl_version"] = info["rccl_version"]
            # (keys of the first record at the top level as well: what round 4's line carried)
            d = pooling["default"]
            pooling.update({"form": d["form"], "list_slot_segments": d["list_slot_segments"], "all_reduce_bytes": d["collective_bytes"],
                            "all_reduce_ms": d["collective_ms"], "pooled_segments": d["pooled_segments"],
                            "counts_per_position": d["counts_per_position"]})
            if rank == 0:
                out["pooling"] = pooling
            second.close()
            pool.close()
        except Exception as e:
            if rank == 0:
                out["pooling"] = {"error": str(e)[:300]}

    # second leg: the same chain with the float weight stream (option weight_keys = 0): every sweep reads all T
    # float weights - the one genuinely bandwidth-bound kernel of the path, priced against the HBM roofline
    if not args.no_stream_leg and world == 1:
        chain.close()
        chain, el2, bl2, c0, c1 = run_leg(weight_summary=False, profile_level=1, steps=leg_steps)
        f_ms, f_n = chain.profile_get("blocks_compact")
        n_ms, n_n = chain.profile_get("event_null")
        if rank == 0 and f_n:
            f_s = max(f_ms / f_n - n_ms / max(1, n_n), 1e-6) * 1e-3
            f_bytes = 4.0 * T + 6.0 * (bl2 / max(1, leg_steps))
            out["float_stream"] = {"value": bl2 / el2, "unit": "block-updates/s", "steps": leg_steps, "ms_per_step": 1e3 * el2 / leg_steps,
                                   "roofline": {"bound": "hbm", "kernel": "hml_k_compact_scan (all T float weights)",
                                                "achieved": f_bytes / f_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                "frac": f_bytes / f_s / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic(args.workload, "hml_k_compact_scan"),
                                                "kernel_avg_us": 1e6 * f_s,
                                                "bytes_per_launch": f_bytes, "launches": f_n},
                                   "note": "same chain, same results; the default path above replaces this stream by the group summary"}

    # scheme legs: what the headline's plain sweeps leave out.  (a) C3's own scheme records every 10th sweep
    # (-i F n 10, marginals on): hml_k_record inside the timed region.  (b) BASELINE config 2: 10^7 positions, scheme
    # M 100 0 S P F n 10 - mixture sweeps on dynamic blocks, then FB sweeps on a FIXED block structure (static blocks).
    # (c) steady state: 1000 sweeps of a chain that has 64 behind it (the forward warm-up of a young chain has settled).
    if not args.no_scheme_legs and world == 1 and args.workload == "c3_1e8_k5_dynamic":
        chain.close()
        ch = hammlet_amd.Chain(device=local_rank, seed=args.seed, chain_id=rank)
        ch.load(x)
        ch.set_model(K, ch.autoprior(0.2, 0.9))
        ch.sample_prior()
        ch.set_recording(marginals=False)
        ch.iterate("F", 64, 0)
        ch.sync()
        r0 = ch.stats()
        barrier()
        t0 = time.perf_counter()
        ch.iterate("F", 1000, 0)
        ch.sync()
        barrier()
        t1 = time.perf_counter()
        r1 = ch.stats()
        bs = (r1["block_updates"] - r0["block_updates"]) / 1000.0
        out["steady_state"] = {"value": (r1["block_updates"] - r0["block_updates"]) / (t1 - t0), "unit": "block-updates/s", "steps": 1000,
                               "ms_per_step": 1e3 * (t1 - t0) / 1000, "forward_refits": r1["forward_refits"] - r0["forward_refits"],
                               "forward_warmup_rows": r1["forward_warmup"],
                               "sweep_frac": (4.0 * T + bs * (36 + 8 * K)) / ((t1 - t0) / 1000) / 1e9 / HBM_PEAK_GBS,
                               "note": "sweeps 64..1064 of the headline's chain: the settled rate (the headline at the driver's --steps 20 --warmup 5 "
                                       "times sweeps 5..25 of a fresh chain)"}
        ch.set_recording(marginals=True)
        ch.iterate("F", 10, 10)   # (the first recorded sweep allocates and clears the marginals' difference arrays - 2 GB, 45 ms: outside the timed region)
        ch.sync()
        r0 = ch.stats()
        barrier()
        t0 = time.perf_counter()
        ch.iterate("F", leg_steps, 10)
        ch.sync()
        barrier()
        t1 = time.perf_counter()
        r1 = ch.stats()
        out["recorded"] = {"value": (r1["block_updates"] - r0["block_updates"]) / (t1 - t0), "unit": "block-updates/s", "steps": leg_steps,
                           "ms_per_step": 1e3 * (t1 - t0) / leg_steps, "scheme": "F %d 10, marginals recorded" % leg_steps,
                           "recorded_sweeps": ch.recorded_sweeps(),
                           "note": "BASELINE config 3 AS SPECIFIED (SURVEY.md 8d: -i F 1000 10): the steady-state chain with every 10th sweep recorded "
                                   "into the marginals (hml_k_record inside the timed region); compare with steady_state, not with the young headline chain"}
        out["config"]["config_3_as_specified"] = "leg `recorded` (scheme F n 10, marginals on); the headline times plain sweeps"
        ch.close()
        T2, K2, lv2, sg2, dw2, ds2 = WORKLOADS["c2_1e7_k5"]
        x2 = hammlet_amd.synth_gauss(T2, K2, lv2, sg2, dw2, ds2, nthreads=nthr)
        ch = hammlet_amd.Chain(device=local_rank, seed=args.seed, chain_id=rank)
        ch.load(x2)
        ch.set_model(K2, ch.autoprior(0.2, 0.9))
        ch.sample_prior()
        ch.set_recording(marginals=True)
        ch.iterate("M", 20, 0)                      # (clocks and caches)
        ch.sync()
        m0 = ch.stats()
        barrier()
        t0 = time.perf_counter()
        ch.iterate("M", leg_steps, 0)
        ch.sync()
        barrier()
        t1 = time.perf_counter()
        m1 = ch.stats()
        ch.set_static_blocks()                      # S: createBlocks(theta) once
        ch.sample_prior()                           # P
        ch.iterate("F", max(args.warmup, 20), 0)
        ch.sync()
        f0 = ch.stats()
        barrier()
        t2 = time.perf_counter()
        ch.iterate("F", leg_steps, 10)
        ch.sync()
        barrier()
        t3 = time.perf_counter()
        f1 = ch.stats()
        out["c2_mixture"] = {"value": (m1["block_updates"] - m0["block_updates"]) / (t1 - t0), "unit": "block-updates/s", "steps": leg_steps,
                             "ms_per_step": 1e3 * (t1 - t0) / leg_steps, "scheme": "M %d 0 on the 10^7-position trace of config 2 (dynamic blocks)" % leg_steps}
        out["c2_static"] = {"value": (f1["block_updates"] - f0["block_updates"]) / (t3 - t2), "unit": "block-updates/s", "steps": leg_steps,
                            "ms_per_step": 1e3 * (t3 - t2) / leg_steps, "blocks_per_sweep": (f1["block_updates"] - f0["block_updates"]) / leg_steps,
                            "scheme": "M %d 0 S P F %d 10 (BASELINE config 2: fixed wavelet block structure), marginals recorded" % (leg_steps + 20, leg_steps),
                            "note": "latency-bound: 2 MB of block data per sweep (SURVEY.md 8d) - a launch-overhead-limited rate, not a bandwidth fraction"}
        ch.close()
        del x2
        chain = None
        # the reference-compatible mode on the headline's trace (option "compat": the reference's own mt19937 stream, libm
        # arithmetic, Kahan sums and size_t += float counts - its chain, bit for bit): a bounded sample of sweeps
        ch = hammlet_amd.Chain(device=local_rank, seed=args.seed, chain_id=rank)
        ch.set_option("compat", 1)
        ch.load(x)
        ch.set_model(K, ch.autoprior(0.2, 0.9))
        ch.sample_prior()
        ch.set_recording(marginals=False)
        ch.iterate("F", 64, 0)     # (a settled chain, like the steady_state leg: the count pass walks the states' lists side by side,
        ch.sync()                  #  and a young chain has nearly all blocks in one state)
        c0 = ch.stats()
        barrier()
        t0 = time.perf_counter()
        ch.iterate("F", 24, 0)
        ch.sync()
        barrier()
        t1 = time.perf_counter()
        c1 = ch.stats()
        ch.close()
        out["reference_compatible"] = {"value": (c1["block_updates"] - c0["block_updates"]) / (t1 - t0), "unit": "block-updates/s", "steps": 24,
                                       "ms_per_step": 1e3 * (t1 - t0) / 24, "blocks_per_sweep": (c1["block_updates"] - c0["block_updates"]) / 24,
                                       "chunks_rerun": c1["forward_refits"] - c0["forward_refits"],
                                       "note": "option compat = 1: the sweeps the reference's single thread computes, bit for bit (cpu_baseline.reference_binary "
                                               "times that thread on this trace); filter and backward draws in chunks that are checked against each other, "
                                               "count pass by state (hml_k_compat.h)"}

    # third leg: several independent chains of the same workload on ONE GPU (chain-parallel inside the GPU: a single chain is
    # latency-bound and leaves most of the machine idle).  Two / three chains each on its own stream and host thread; eight
    # chains attached to ONE construction (hml_attach_observations) and batched through hml_iterate_many.
    if not args.no_two_chain_leg and world == 1:
        import threading
        if chain is not None:
            chain.close()
        chain = None

        def several(n_chains):
            group = []
            for r in range(n_chains):
                ch = hammlet_amd.Chain(device=local_rank, seed=args.seed, chain_id=r)
                ch.load(x)
                ch.set_model(K, ch.autoprior(0.2, 0.9))
                ch.sample_prior()
                ch.set_recording(marginals=False)
                ch.iterate("F", max(args.warmup, 64), 0)   # (past the forward warm-up floor of a young chain: 64 sweeps)
                group.append(ch)
            for ch in group:
                ch.sync()
            b0 = [ch.stats()["block_updates"] for ch in group]
            # the threads exist and wait at a barrier BEFORE the clock starts
            gate = threading.Barrier(n_chains + 1)

            def run(ch):
                gate.wait()
                ch.iterate("F", leg_steps, 0)
                ch.sync()
            ths = [threading.Thread(target=run, args=(ch,)) for ch in group]
            for t in ths:
                t.start()
            barrier()
            t0 = time.perf_counter()
            gate.wait()
            for t in ths:
                t.join()
            barrier()
            t1 = time.perf_counter()
            b2 = sum(ch.stats()["block_updates"] - b for ch, b in zip(group, b0))
            for ch in group:
                ch.close()
            return b2 / (t1 - t0), 1e3 * (t1 - t0) / leg_steps
        v2, ms2 = several(2)
        out["two_chains_one_gpu"] = {"value": v2, "unit": "block-updates/s", "chains": 2, "steps": leg_steps, "ms_per_sweep_round": ms2,
                                     "note": "aggregate of two independent chains on one GPU, a host thread and a stream each; the headline value is one chain per GPU"}
        v3, ms3 = several(3)
        out["three_chains_one_gpu"] = {"value": v3, "unit": "block-updates/s", "chains": 3, "steps": leg_steps, "ms_per_sweep_round": ms3,
                                       "note": "three host threads, three streams (private constructions, scan + scatter launches)"}
        # eight (sixteen) chains attached to one construction, launched by one host thread (hml_iterate_many: the many-chain
        # block kernel of hml_k_blocks_fused_many.h; two groups of chains on a stream each)
        def attached(n_chains):
            free0 = torch.cuda.mem_get_info()[0] if torch is not None else 0
            group = []
            mem = []
            for r in range(n_chains):
                ch = hammlet_amd.Chain(device=local_rank, seed=args.seed, chain_id=r)
                if r == 0:
                    ch.load(x)
                else:
                    ch.attach(group[0])
                ch.set_model(K, ch.autoprior(0.2, 0.9))
                ch.sample_prior()
                ch.set_recording(marginals=False)
                ch.sync()
                group.append(ch)
                mem.append(free0 - torch.cuda.mem_get_info()[0] if torch is not None else 0)
            hammlet_amd.iterate_many(group, "F", max(args.warmup, 64), 0)
            for ch in group:
                ch.sync()
            b0 = [ch.stats()["block_updates"] for ch in group]
            barrier()
            t0 = time.perf_counter()
            hammlet_amd.iterate_many(group, "F", leg_steps, 0)
            for ch in group:
                ch.sync()
            barrier()
            t1 = time.perf_counter()
            bn = sum(ch.stats()["block_updates"] - b for ch, b in zip(group, b0))
            for ch in group:
                ch.close()
            return {"value": bn / (t1 - t0), "unit": "block-updates/s", "chains": n_chains, "steps": leg_steps,
                    "ms_per_sweep_round": 1e3 * (t1 - t0) / leg_steps,
                    "x_one_chain": (bn / (t1 - t0)) / out.get("steady_state", out)["value"],
                    "device_bytes_chain_1": mem[0], "device_bytes_per_further_chain": (mem[-1] - mem[0]) / (n_chains - 1.0)}
        out["eight_chains_one_gpu_batched"] = attached(8)
        out["eight_chains_one_gpu_batched"]["note"] = (
            "eight chains attached to ONE construction (hml_attach_observations), hml_iterate_many: block starts, statistics "
            "and emission terms of a group's chains from one pass over the shared trace (hml_m_blocks_fused), the other kernels "
            "once per group (the chain is the grid's second dimension); one host thread, two groups of four chains on a stream each")
        out["sixteen_chains_one_gpu_batched"] = attached(16)
        out["sixteen_chains_one_gpu_batched"]["note"] = "as above, two groups of eight chains"

    def dense_leg(xd, Kd, scale, pmc_key, shared_sums, n_timed):
        """a weakly compressed chain (millions of blocks per sweep: the fused trellis path, hml_k_trellis_rows.h): 64 burn-in
        sweeps (the warm-up policy settles, sweeps 48..57 measure the candidate chunk lengths), n_timed timed ones, then ten
        sweeps with every kernel family bracketed by events"""
        ch = hammlet_amd.Chain(device=local_rank, seed=args.seed, chain_id=rank)
        ch.load(xd)
        if scale != 1.0:
            ch.scale_weights(scale)
        ch.set_model(Kd, ch.autoprior(0.2, 0.9))
        ch.sample_prior()
        ch.set_recording(marginals=False)
        ch.iterate("F", 64, 0)
        ch.sync()
        u0 = ch.stats()
        barrier()
        t0 = time.perf_counter()
        ch.iterate("F", n_timed, 0)
        ch.sync()
        barrier()
        t1 = time.perf_counter()
        u1 = ch.stats()
        bu = u1["block_updates"] - u0["block_updates"]
        Bu = bu / n_timed
        fams = dense_kernels(Kd, shared_sums)
        ch.profile_enable(2)
        ch.iterate("F", 10, 0)
        ch.sync()
        ch.profile_enable(0)
        nb_ms, nb_n = ch.profile_get("event_null")
        nb_us = 1e3 * nb_ms / max(1, nb_n) if nb_n else 5.3
        dense_tab = {}
        for nm, kern in fams.items():
            ms, n = ch.profile_get(nm)
            if n:
                us = max(1e3 * ms / n - nb_us, 0.5)
                row = {"kernels": kern, "us_per_sweep": round(us, 1)}
                tr = pmc_traffic(pmc_key, kern)
                if tr:
                    row["traffic"] = tr
                    row["traffic_raw"] = pmc_traffic(pmc_key, kern, "bytes_raw")
                    row["frac_upper"] = round(tr / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                    row["frac"] = round(row["traffic_raw"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                    assert row["frac"] <= row["frac_upper"] <= 1.0
                dense_tab[nm] = row
        Td = ch.T
        ch.close()
        return {"value": bu / (t1 - t0), "unit": "block-updates/s", "steps": n_timed, "ms_per_step": 1e3 * (t1 - t0) / n_timed,
                "positions": Td, "states": Kd, "blocks_per_sweep": Bu, "forward_refits": u1["forward_refits"] - u0["forward_refits"],
                "forward_warmup_rows": u1["forward_warmup"],
                "sweep_frac": (4.0 * Td + Bu * (36 + 8 * Kd)) / ((t1 - t0) / n_timed) / 1e9 / HBM_PEAK_GBS,
                "kernels": dense_tab}

    # fourth leg: SURVEY 8d's stress case C3u - the same trace with the breakpoint weights multiplied by 1e9, so that
    # every position is its own block (B = T): the regime in which the trellis itself, not the block scan, is the load
    if not args.no_uncompressed_leg and world == 1 and args.workload == "c3_1e8_k5_dynamic":
        if chain is not None:
            chain.close()
        chain = None
        leg = dense_leg(x, K, 1e9, "c3u", True, max(30, min(100, args.steps)))
        leg["note"] = ("B = T: the fused trellis path (hml_k_trellis_rows.h) - bound by vector issue: SQ_INSTS_VALU of the first pass = about 530 "
                       "wavefront instructions per 64 blocks and warm-up row (profiles/round3_sq_counters_c3u.txt); DESIGN.md 3a")
        out["uncompressed_c3u"] = leg

    # fifth leg: the other single-GPU workloads of BASELINE.json, bounded (their full lines: python bench.py --workload ...)
    if not args.no_config_legs and world == 1 and args.workload == "c3_1e8_k5_dynamic":
        if chain is not None:
            chain.close()
        chain = None
        # config 4's workload on one GPU: 10^8 positions, 10 states, strongly compressed, dynamic blocks
        T4, K4, lv4, sg4, dw4, ds4 = WORKLOADS["c4_1e8_k10"]
        x4 = hammlet_amd.synth_gauss(T4, K4, lv4, sg4, dw4, ds4, nthreads=nthr)
        ch = hammlet_amd.Chain(device=local_rank, seed=args.seed, chain_id=rank)
        ch.load(x4)
        ch.set_model(K4, ch.autoprior(0.2, 0.9))
        ch.sample_prior()
        ch.set_recording(marginals=False)
        ch.iterate("F", 64, 0)
        ch.sync()
        r0 = ch.stats()
        barrier()
        t0 = time.perf_counter()
        ch.iterate("F", 400, 0)
        ch.sync()
        barrier()
        t1 = time.perf_counter()
        r1 = ch.stats()
        b4 = (r1["block_updates"] - r0["block_updates"]) / 400.0
        fk4 = family_kernels(K4)
        ch.profile_enable(2)
        ch.iterate("F", 100, 0)
        ch.sync()
        ch.profile_enable(0)
        tab4 = {}
        for nm, kern in fk4.items():
            ms, n = ch.profile_get(nm)
            if n:
                us = max(1e3 * ms / n - 5.3, 0.5)   # (level-2 brackets: minus the usual empty bracket)
                row = {"kernel_us": round(us, 2)}
                tr = pmc_traffic("c4_1e8_k10", kern)
                if tr:
                    row["traffic"] = tr
                    row["traffic_raw"] = pmc_traffic("c4_1e8_k10", kern, "bytes_raw")
                    row["frac_upper"] = round(tr / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                    row["frac"] = round(row["traffic_raw"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                tab4[kern] = row
        ch.close()
        del x4
        out["c4_1e8_k10"] = {"value": (r1["block_updates"] - r0["block_updates"]) / (t1 - t0), "unit": "block-updates/s", "steps": 400,
                             "ms_per_step": 1e3 * (t1 - t0) / 400, "positions": T4, "states": K4, "blocks_per_sweep": b4,
                             "forward_refits": r1["forward_refits"] - r0["forward_refits"],
                             "sweep_frac": (4.0 * T4 + b4 * (36 + 8 * K4)) / ((t1 - t0) / 400) / 1e9 / HBM_PEAK_GBS,
                             "kernels": tab4,
                             "note": "BASELINE config 4's workload on ONE GPU (one of its 8 chains): sweeps 64..464; latency-bound like the headline"}
        # more than 16 states (round 5, hml_k_wide.h / hml_k_wide_lanes.h: the number of states a run-time value, a chunk a lane): the headline's trace with 20
        # states in the model (more states than levels: states of tiny variance lower the threshold - 1.5 10^6 blocks per sweep)
        ch = hammlet_amd.Chain(device=local_rank, seed=args.seed, chain_id=rank)
        ch.load(x)
        ch.set_model(20, ch.autoprior(0.2, 0.9))
        ch.sample_prior()
        ch.set_recording(marginals=False)
        ch.iterate("F", 60, 0)
        ch.sync()
        w0 = ch.stats()
        barrier()
        t0 = time.perf_counter()
        ch.iterate("F", 100, 0)
        ch.sync()
        barrier()
        t1 = time.perf_counter()
        w1 = ch.stats()
        ch.close()
        out["twenty_states"] = {"value": (w1["block_updates"] - w0["block_updates"]) / (t1 - t0), "unit": "block-updates/s", "steps": 100,
                                "ms_per_step": 1e3 * (t1 - t0) / 100, "positions": T, "states": 20,
                                "blocks_per_sweep": (w1["block_updates"] - w0["block_updates"]) / 100.0,
                                "chunks_run_again": w1["forward_refits"] - w0["forward_refits"],
                                "note": "the default path for models of 17-64 states: filter and backward draws with a chunk a lane over chunk-transposed arrays "
                                        "(hml_k_wide_lanes.h), Philox uniforms, the count tree (DESIGN.md 3e); sweeps 60..160 of a chain on the headline's trace"}
        # config 5's workload on one GPU: 2.5 10^8 simulated read-depth positions, 5 states, weakly compressed
        T5, K5, _, sg5, dw5, ds5 = WORKLOADS["c5_2.5e8_depth_k5"]
        x5 = hammlet_amd.synth_depth(T5, depth=dw5, ln_sigma=sg5, seed=ds5, nthreads=nthr)
        leg = dense_leg(x5, K5, 1.0, "c5_2.5e8_depth_k5", False, 40)
        del x5
        leg["note"] = "BASELINE config 5's workload on ONE GPU (one of its 8 chains): simulated WGS read depth, the fused trellis path; sweeps 64..104"
        out["c5_2.5e8_depth_k5"] = leg

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(x, K, args.seed)
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        # the kernels SURVEY.md 8d and BASELINE.json's north star name, gathered into `roofline` from the legs that measured them
        roof = out["roofline"]
        if "float_stream" in out:   # 8d's literal roofline kernel: all T float weights streamed (bandwidth-bound; traffic = algorithmic bytes)
            roof["float_stream_frac"] = out["float_stream"]["roofline"]["frac"]
            roof["float_stream_kernel"] = out["float_stream"]["roofline"]["kernel"]

        def fwd_row(kern, us, workload_key):
            tr_up, tr_raw = pmc_traffic(workload_key, kern), pmc_traffic(workload_key, kern, "bytes_raw")
            row = {"kernel": kern, "kernel_us": us, "traffic": tr_up, "traffic_raw": tr_raw}
            if tr_raw:
                row["frac"] = tr_raw / (us * 1e-6) / 1e9 / HBM_PEAK_GBS
            if tr_up:
                row["frac_upper"] = tr_up / (us * 1e-6) / 1e9 / HBM_PEAK_GBS
            return row
        fwd = {}
        kf = family_kernels(K)["forward"]
        if kf in out.get("kernels", {}):
            fwd["strongly_compressed"] = fwd_row(kf, out["kernels"][kf]["kernel_us"], args.workload)
            fwd["strongly_compressed"]["workload"] = args.workload
        tr_leg = out.get("uncompressed_c3u", {}).get("kernels", {}).get("trellis")
        if tr_leg:
            fwd["uncompressed_c3u"] = fwd_row(tr_leg["kernels"], tr_leg["us_per_sweep"], "c3u")
            fwd["uncompressed_c3u"]["workload"] = "c3u (the same trace, every position its own block)"
        if dense_workload and DENSE_INST:
            fwd["this_workload"] = {"kernel": DENSE_INST, "kernel_us": roof["kernel_avg_us"], "frac": roof["frac"], "frac_upper": roof["frac_upper"]}
        if fwd:
            fwd["note"] = ("the forward-trellis kernel of the north star: hml_k_forward (speculative chunked filter) where sweeps are strongly "
                           "compressed - latency-bound - and hml_k_trellis_rows (emission terms + filter + candidate maps in one pass) where they "
                           "are not - bound by vector issue; frac = counter bytes as counted / duration / 8 TB/s, frac_upper with the fetches doubled")
            roof["forward_trellis"] = fwd
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
