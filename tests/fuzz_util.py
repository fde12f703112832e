"""Randomised differential run: GPU (through the C ABI) against the CPU checker in device mode on random configurations -
sizes, states, data dimensions, schemes, priors, weight multipliers, forward geometry, every switchable kernel path.
Shared by tools/fuzz_parity.py (thousands of configurations, by hand) and tests/test_gpu_fuzz.py (a bounded, fixed-seed
sample inside `-m gpu`).  Reference behaviour under test: the whole sweep, src/HMM.hpp:99-121."""
import os
import time

import numpy as np

from tests import oracle_lib as ol
from tests.test_gpu_parity import run_both

ENV_KEYS = ("HML_DENSE_MIN_BLOCKS", "HML_FWD_CHUNK_DENSE", "HML_TRELLIS_FUSED", "HML_TRELLIS_L", "HML_TRELLIS_ROWS", "HML_TRELLIS_CKPT", "HML_TRELLIS_REFIT_ROUNDS",
            "HML_STAGE_BITS", "HML_FWD_WARMUP", "HML_LATE_RESCALE", "HML_FWD_CHUNK",
            "HML_MANY_GROUPS", "HML_FUSED_MANY_SLOTS", "HML_MAX_BLOCKS", "HML_FWD_CHUNK_MANY", "HML_COMPAT_CHUNKS", "HML_COMPAT_WARMUP", "HML_WIDE", "HML_FM_SPLIT", "HML_FM_SPLIT_SUB",
            "HML_WIDE_L", "HML_WIDE_LANES", "HML_MID_MIN_BLOCKS")


def fuzz(hml, n_cfg, seed, log=None, many=False, compat=False, wide=False):
    """n_cfg random configurations; returns the number that ran identical (all, or an AssertionError names the first
    that differs).  The environment switches it sets are restored afterwards.  many: several chains through
    hml_iterate_many (_fuzz_many) instead of one through hml_iterate.  compat: the reference-compatible mode against the
    checker's REFERENCE mode (mt19937, libm, Kahan sums, size_t += float) - up to 64 states, chunk geometries that force
    wrong chunks.  wide: the default path's kernels for more than 16 states (hml_k_wide.h, hml_k_wide_lanes.h: the number of states at run
    time; a chunk a lane or a state a lane) against the checker's device mode - 2-64 states (HML_WIDE=1 sends models of up to 16 states there too), the
    same chunk geometries."""
    saved = {k: os.environ.get(k) for k in ENV_KEYS}
    try:
        if many:
            return _fuzz_many(hml, n_cfg, seed, log or (lambda *a, **k: None))
        if wide:
            os.environ["HML_WIDE"] = "1"
        return _fuzz(hml, n_cfg, seed, log or (lambda *a, **k: None), compat=compat, wide=wide)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _fuzz(hml, n_cfg, seed, log, compat=False, wide=False):
    rng = np.random.default_rng(seed)
    bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
    t_start = time.time()

    for it in range(n_cfg):
        D = int(rng.choice([1, 1, 1, 2, 2, 3]))
        if D == 1:
            P = None; K = int(rng.integers(2, 17))
            if (compat and rng.random() < 0.3) or (wide and rng.random() < 0.5):
                K = int(rng.choice([17, 20, 31, 32, 33, 48, 64]))
        else:
            P = int(rng.choice(([2, 3, 4, 5, 7] if (compat or wide) else [2, 3, 4]) if D == 2 else ([2, 3] if (compat or wide) else [2]))); K = P ** D
        T = int(rng.choice([17, 1000, 4097, 30000, 65535, 65537, 120000, 300000]))
        if K > 8 or D > 1:
            T = min(T, 120000)
        dense = bool(rng.random() < 0.3)
        mult = float(rng.choice([1.0, 1.0, 1.5, 1e9 if T <= 65537 else 1.0]))
        self_trans = bool(rng.random() < 0.8)
        kw = dict(t_off=float(rng.choice([0.5, 0.1, 1.0])), t_diag=float(rng.choice([0.5, 10.0])), pi_alpha=float(rng.choice([0.5, 2.0])),
                  e_var=float(rng.choice([0.2, 0.1])), e_p=float(rng.choice([0.9, 0.8])), self_trans=self_trans, weight_mult=mult)
        os.environ["HML_DENSE_MIN_BLOCKS"] = "500" if dense else str(1 << 22)
        os.environ["HML_FWD_CHUNK_DENSE"] = str(int(rng.choice([8, 16, 32])))
        # round 2's switches: fused trellis path and its chunk length, late rescale, forward chunk length
        os.environ["HML_TRELLIS_FUSED"] = str(int(rng.choice([1, 1, 1, 0])))
        os.environ["HML_TRELLIS_L"] = str(int(rng.choice([0, 32, 64, 96, 160, 256, 544, 1024])))
        # round 3's switches: first pass (rows | tile), checkpointed refits, flag staging of the dense scan, a short warm-up
        os.environ["HML_TRELLIS_ROWS"] = str(int(rng.choice([1, 1, 1, 0])))
        os.environ["HML_TRELLIS_CKPT"] = str(int(rng.choice([1, 1, 0])))
        os.environ["HML_TRELLIS_REFIT_ROUNDS"] = str((2, 2, 0, 1, 4, 3, 6)[(T + K) % 7])   # (no draw: the configurations of earlier rounds stay what they were)
        os.environ["HML_STAGE_BITS"] = str(int(rng.choice([1, 1, 0])))
        if rng.random() < 0.3: os.environ["HML_FWD_WARMUP"] = str(int(rng.choice([4, 8, 16])))
        else: os.environ.pop("HML_FWD_WARMUP", None)
        os.environ["HML_LATE_RESCALE"] = str(int(rng.choice([1, 1, 0])))
        os.environ["HML_FWD_CHUNK"] = str(int(rng.choice([4, 4, 1, 2, 8])))
        os.environ["HML_MID_MIN_BLOCKS"] = str(int(rng.choice([262144, 2000, 200])))   # (chunks of 8 from that many blocks on: hml_ctx.hpp)
        if compat or wide:   # chunks of the filter / backward draws: the default, the sequential form, many chunks with hardly any warm-up
            _setenv("HML_COMPAT_CHUNKS", rng.choice([None, None, 1, 7, 60, 500]))
            _setenv("HML_COMPAT_WARMUP", rng.choice([None, None, -1, 1, 4]))
            if wide:   # ... a chunk a lane (hml_k_wide_lanes.h; taken when no number of chunks is asked for): the chunk length, or a state a lane
                _setenv("HML_WIDE_L", rng.choice([None, None, 1, 2, 4, 8, 64]))
                _setenv("HML_WIDE_LANES", rng.choice([None, None, None, 0]))
            if K > 16:
                T = min(T, 65537)
        seed = int(rng.integers(0, 1 << 30))
        levels = min(K if D == 1 else P, 5)
        if rng.random() < 0.2 and D == 1:
            x = ol.synth_depth(T, seed=int(rng.integers(1, 100)))
        else:
            x = np.stack([ol.trace(T, levels, int(rng.integers(1, 1000)) + d) for d in range(D)], axis=1).reshape(-1)
        scheme = []
        for _ in range(int(rng.integers(1, 5))):
            tok = rng.choice(["F", "F", "M", "S", "D", "P"])
            scheme.append((str(tok), int(rng.integers(1, 9)), int(rng.integers(0, 3))) if tok in ("F", "M") else str(tok))
        if not isinstance(scheme[-1], tuple):   # the probes (blocks, states) describe the last SWEEP; end with one
            scheme.append(("F", int(rng.integers(1, 5)), 1))
        if compat:
            o = ol.OracleChain(K=K, seed=seed, rng=ol.RNG_MT, math=ol.MATH_LIBM, reduce=ol.REDUCE_REF, **kw)
        else:
            o = ol.OracleChain(K=K, seed=seed, rng=ol.RNG_CTR, math=ol.MATH_DEV, reduce=ol.REDUCE_DEV, **kw)
        g = hml.Chain(device=0, seed=seed)
        if compat:
            g.set_option("compat", 1)
        if D > 1:
            o.set_dimensions(D, P); g.set_dimensions(D, P)
        g.set_option("weight_keys", int(rng.choice([1, 1, 2, 0])))
        o.load(x); g.load(x)
        if mult != 1.0:
            g.scale_weights(mult)
        try:
            po = o.autoprior()
        except RuntimeError as err:
            # a degenerate draw (e.g. 17 positions in one block): the reference's own exception - the GPU must raise it too
            try:
                g.autoprior(kw["e_var"], kw["e_p"])
                raise AssertionError("configuration %d: the checker raised %r, the GPU did not" % (it, str(err)))
            except hml.HmlError as gerr:
                assert str(err) in str(gerr), (str(err), str(gerr))
            log("%3d ok  (both raise: %s)" % (it, err))
            g.close(); o.close()
            continue
        pg = g.autoprior(kw["e_var"], kw["e_p"])
        assert np.array_equal(bits(po), bits(pg)), ("autoprior", it)
        o.init_model()
        g.set_model(K, pg, kw["t_off"], kw["t_diag"], kw["pi_alpha"], self_trans)
        o.set_record(marginals=True)
        g._pending_prior = True
        run_both(o, g, scheme)
        ok = (np.array_equal(o.blocks(), g.blocks()) and np.array_equal(o.states(), g.states()) and np.array_equal(bits(o.theta()), bits(g.theta()))
              and hml.marginals_text(*g.marginals_rle()) == o.text("marginals"))
        desc = "%3d %s T=%d K=%d D=%d dense=%d mult=%g self=%d scheme=%s  B=%d  %.0f s" % (it, "ok " if ok else "DIFF", T, K, D, dense, mult, self_trans, scheme, len(g.blocks()) - 1, time.time() - t_start)
        log(desc)
        assert ok, desc
        g.close(); o.close()

    return n_cfg


def _setenv(name, value):
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = str(value)


def _fuzz_many(hml, n_cfg, seed, log):
    """Several chains of one trace through hml_iterate_many - attached to one construction (hml_attach_observations: the
    many-chain block kernel) or with private ones, in 1-4 groups of chains, tiles of several batches, tiny block capacities,
    weakly compressed chains that leave the batch - each against the checker's chain of the same (seed, chain) run alone."""
    rng = np.random.default_rng(seed)
    bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
    t_start = time.time()
    for it in range(n_cfg):
        K = int(rng.choice([2, 3, 4, 5, 5, 6, 8, 10, 13, 16]))
        T = int(rng.choice([1000, 4097, 30000, 65537, 120000, 300000]))
        if K > 8:
            T = min(T, 120000)
        n = int(rng.integers(2, 11))
        attached = bool(rng.random() < 0.75)
        mult = float(rng.choice([1.0, 1.0, 1.5]))
        self_trans = bool(rng.random() < 0.8)
        kw = dict(t_off=float(rng.choice([0.5, 0.1, 1.0])), t_diag=float(rng.choice([0.5, 10.0])), pi_alpha=float(rng.choice([0.5, 2.0])),
                  e_var=float(rng.choice([0.2, 0.1])), e_p=float(rng.choice([0.9, 0.8])), self_trans=self_trans, weight_mult=mult)
        env = {"HML_MANY_GROUPS": rng.choice([None, None, 1, 2, 3, 4]), "HML_FUSED_MANY_SLOTS": rng.choice([None, None, None, 2, 5]),
               "HML_MAX_BLOCKS": rng.choice([None, None, 64]), "HML_FWD_CHUNK_MANY": rng.choice([None, None, 4, 16]),
               "HML_DENSE_MIN_BLOCKS": rng.choice([1 << 22, 1 << 22, 1 << 22, 400]), "HML_LATE_RESCALE": rng.choice([1, 1, 0]),
               "HML_FWD_WARMUP": rng.choice([None, None, 4, 8]),
               # round 5: the block structure of attached chains in two launches (the default) or by the fused kernel; tiles of several batches
               "HML_FM_SPLIT": rng.choice([None, None, 1, 0]), "HML_FM_SPLIT_SUB": rng.choice([None, None, 2, 4])}
        for k, v in env.items():
            _setenv(k, v)
        seed_c = int(rng.integers(0, 1 << 30))
        x = ol.synth_depth(T, seed=int(rng.integers(1, 100))) if rng.random() < 0.15 else ol.trace(T, min(K, 5), int(rng.integers(1, 1000)))
        scheme = []
        for _ in range(int(rng.integers(1, 5))):
            tok = rng.choice(["F", "F", "F", "M", "S", "D", "P"])
            scheme.append((str(tok), int(rng.integers(1, 9)), int(rng.integers(0, 3))) if tok in ("F", "M") else str(tok))
        if not isinstance(scheme[-1], tuple):
            scheme.append(("F", int(rng.integers(1, 5)), 1))
        pairs = []
        weights_key = int(rng.choice([1, 1, 2, 0]))
        for chain in range(n):
            o = ol.OracleChain(K=K, seed=seed_c, chain=chain, rng=ol.RNG_CTR, math=ol.MATH_DEV, reduce=ol.REDUCE_DEV, **kw)
            g = hml.Chain(device=0, seed=seed_c, chain_id=chain)
            o.load(x)
            g.set_option("weight_keys", weights_key)
            if attached and chain > 0:
                g.attach(pairs[0][1])
            else:
                g.load(x)
                if mult != 1.0:
                    g.scale_weights(mult)
            po = o.autoprior()
            pg = g.autoprior(kw["e_var"], kw["e_p"])
            assert np.array_equal(bits(po), bits(pg)), ("autoprior", it, chain)
            o.init_model()
            g.set_model(K, pg, kw["t_off"], kw["t_diag"], kw["pi_alpha"], self_trans)
            o.set_record(marginals=True)
            pairs.append((o, g))
        gs = [g for _, g in pairs]
        pending = True
        for tok in scheme:
            if pending:
                for g in gs:
                    g.sample_prior()
                pending = False
            if tok in ("P", "S", "D"):
                for o, g in pairs:
                    o.token(tok)
                    if tok == "S":
                        g.set_static_blocks()
                    elif tok == "D":
                        g.set_dynamic(True)
                pending = tok == "P"
            else:
                m, iters, thin = tok
                for o, _ in pairs:
                    o.iterate(m, iters, thin)
                hml.iterate_many(gs, m, iters, thin)
        ok = True
        for chain, (o, g) in enumerate(pairs):
            g.sync()
            Ao, pio = o.transitions()
            Ag, pig = g.transitions()
            ok = ok and (np.array_equal(o.blocks(), g.blocks()) and np.array_equal(o.states(), g.states()) and np.array_equal(bits(o.theta()), bits(g.theta()))
                         and np.array_equal(bits(Ao), bits(Ag)) and np.array_equal(bits(pio), bits(pig))
                         and hml.marginals_text(*g.marginals_rle()) == o.text("marginals"))
            if not ok:
                break
        B = len(gs[0].blocks()) - 1
        desc = "%3d %s T=%d K=%d chains=%d attached=%d mult=%g self=%d env=%s scheme=%s  B=%d  %.0f s" % (
            it, "ok " if ok else "DIFF (chain %d)" % chain, T, K, n, attached, mult, self_trans,
            {k[4:]: (None if v is None else int(v)) for k, v in env.items()}, scheme, B, time.time() - t_start)
        log(desc)
        assert ok, desc
        for o, g in reversed(pairs):   # (attached contexts before their source - either order is allowed)
            g.close(); o.close()
    return n_cfg
