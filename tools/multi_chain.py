"""Several independent chains on ONE GPU: aggregate block-updates/s - each chain on its own stream and host thread, or (mode
"many") all chains in one set of launches by one host thread (hml_iterate_many).
    python tools/multi_chain.py [chains] [sweeps] [workload] [threads|many|attached]
"attached" = mode "many" over contexts that share ONE construction (hml_attach_observations): the many-chain block kernel."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, hammlet_amd
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500
wl = sys.argv[3] if len(sys.argv) > 3 else "c3_1e8_k5_dynamic"
mode = sys.argv[4] if len(sys.argv) > 4 else "threads"
T, K, levels, sigma, dwell, data_seed = bench.WORKLOADS[wl]
x = hammlet_amd.synth_gauss(T, K, levels, sigma, dwell, data_seed, nthreads=8)
chains = []
for r in range(R):
    ch = hammlet_amd.Chain(device=0, seed=1, chain_id=r)
    if mode in ("attached", "attached2") and r > 0:
        ch.attach(chains[0])
    else:
        ch.load(x)
    ch.set_model(K, ch.autoprior(0.2, 0.9))
    ch.sample_prior()
    ch.set_recording(marginals=False)
    ch.iterate("F", 40, 0)
    chains.append(ch)
for ch in chains:
    ch.sync()
reps = int(os.environ.get("REPS", "5")) if mode in ("many", "attached", "attached2") else 1
GROUPS = int(os.environ.get("GROUPS", "2"))
times, blocks_all = [], []
for rep in range(reps):
    s0 = [ch.stats() for ch in chains]
    def run(ch):
        ch.iterate("F", n, 0)
        ch.sync()
    ths = [threading.Thread(target=run, args=(ch,)) for ch in chains]
    t0 = time.perf_counter()
    if mode in ("many", "attached"):
        hammlet_amd.iterate_many(chains, "F", n, 0)
        for ch in chains: ch.sync()
    elif mode == "attached2":   # GROUPS host threads, each batching its share of the chains (all attached to chain 0's construction)
        def run_group(g):
            hammlet_amd.iterate_many(g, "F", n, 0)
            for ch in g: ch.sync()
        gs = [chains[i::GROUPS] for i in range(GROUPS)]
        tg = [threading.Thread(target=run_group, args=(g,)) for g in gs]
        for t in tg: t.start()
        for t in tg: t.join()
    else:
        for t in ths: t.start()
        for t in ths: t.join()
    t1 = time.perf_counter()
    times.append(t1 - t0)
    if os.environ.get("SHOW_W"):
        st = [ch.stats() for ch in chains]
        print("   rep %d: %.4f ms | warm-up rows %s | refits %s" % (rep, 1e3 * (t1 - t0) / n, [s["forward_warmup"] for s in st],
                                                               [s["forward_refits"] - a["forward_refits"] for s, a in zip(st, s0)]))
    blocks_all.append(sum(ch.stats()["block_updates"] - s["block_updates"] for ch, s in zip(chains, s0)))
i = min(range(reps), key=lambda k: times[k])
dt, blocks = times[i], blocks_all[i]
print("[%s] " % mode + "%d chains x %d sweeps on one GPU: %.4f ms per sweep-round, %.3e block-updates/s aggregate (%.3e per chain)" % (
    R, n, 1e3 * dt / n, blocks / dt, blocks / dt / R) + ("   [best of %d: %s ms]" % (reps, " ".join("%.4f" % (1e3 * t / n) for t in times)) if reps > 1 else ""))
