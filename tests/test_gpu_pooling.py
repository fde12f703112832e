"""Chain-parallel pooling through the C ABI on the GPU: the RCCL path with a one-rank communicator (RCCL
is initialised and ncclAllReduce runs on the box), `hml_allreduce_marginals` for one process driving several chains,
the export/install pair with an external sum, and the relabelling rule - each against the CPU checker's chains."""
import numpy as np
import pytest

from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu

T, K = 60000, 4


def checker_chain(x, chain, seed=21, sweeps=(24, 3), K=K):
    o = ol.OracleChain(K=K, seed=seed, chain=chain, rng=ol.RNG_CTR, math=ol.MATH_DEV, reduce=ol.REDUCE_DEV)
    o.load(x)
    o.autoprior()
    o.init_model()
    o.token("F")
    o.iterate("F", *sweeps)
    lens = [int(l.split("\t")[0]) for l in o.text("marginals").strip().split("\n")]
    bnd = np.zeros(T, np.int32)
    bnd[np.cumsum([0] + lens[:-1])] = 1
    return o.marginals_dense(), o.theta()[0::2].copy(), bnd


def gpu_chain(hml, x, chain, seed=21, sweeps=(24, 3), K=K):
    g = hml.Chain(device=0, seed=seed, chain_id=chain)
    g.load(x)
    g.set_model(K, g.autoprior(0.2, 0.9))
    g.sample_prior()
    g.iterate("F", *sweeps)
    g.sync()
    return g


def expected_pooled(x, chain_ids, K=K):
    from hammlet_amd import chains
    dense = np.zeros((K, T), np.int64)
    bnd = np.zeros(T, np.int64)
    perms = []
    for c in chain_ids:
        d, means, b = checker_chain(x, c, K=K)
        perm = chains.relabel_permutation(means)
        perms.append(perm)
        dense += d[perm]
        bnd += b
    starts = np.flatnonzero(bnd)
    seg = np.diff(np.append(starts, T))
    cols = int(np.flatnonzero(dense.any(axis=1)).max()) + 1
    return seg, dense[:cols, starts].T, perms


def test_one_rank_rccl_communicator_pools_through_the_c_entry_point(hml):
    """hml_pool_unique_id + hml_pool_create (ncclCommInitRank, one rank) + hml_pool_marginals (export, ncclAllReduce,
    install): the chain afterwards holds its own marginals under the common labels."""
    x = ol.trace(T, K, 3)
    g = gpu_chain(hml, x, 0)
    pool = hml.Pool(0, 0, 1, hml.Pool.unique_id())
    pool.set_form(1)   # the dense payload through ncclAllReduce (the boundary-list form: the test below)
    perm = pool.marginals(g)
    info = pool.info()
    assert info["n_ranks"] == 1 and info["rccl_version"] > 0
    assert info["last_bytes"] == 4 * ((K + 1) * (T + 1) + 1 + K)
    seg_e, cnt_e, perms = expected_pooled(x, [0])
    assert np.array_equal(perm, perms[0])
    seg, cnt = g.marginals_rle()
    assert np.array_equal(seg, seg_e) and np.array_equal(cnt, cnt_e)
    assert g.recorded_sweeps() == 8
    pool.close()
    g.close()


def test_boundary_list_form_gives_the_dense_forms_marginals(hml):
    """hml_pool_marginals' two collectives - the dense payload through ncclAllReduce, the ranks' boundary lists through
    ncclAllGather - leave the same pooled marginals; the default picks the lists for a strongly compressed chain (a few
    hundred segments in 60 000 positions) and the dense payload where every position is a segment of its own."""
    x = ol.trace(T, K, 3)
    res = {}
    for form in (1, 2, 0):
        g = gpu_chain(hml, x, 1)
        pool = hml.Pool(0, 0, 1, hml.Pool.unique_id())
        pool.set_form(form)
        perm = pool.marginals(g)
        res[form] = (perm, g.marginals_rle(), g.recorded_sweeps(), pool.last(), pool.info()["last_bytes"], g.max_segmentation())
        with pytest.raises(hml.HmlError):
            pool.marginals(g)              # pooled already
        pool.close()
        g.close()
    assert res[1][3]["form"] == "dense" and res[2][3]["form"] == "lists" and res[0][3]["form"] == "lists"
    assert res[2][4] * 8 < res[1][4]
    for form in (2, 0):
        assert np.array_equal(res[form][0], res[1][0])
        assert np.array_equal(res[form][1][0], res[1][1][0]) and np.array_equal(res[form][1][1], res[1][1][1])
        assert res[form][2] == res[1][2]
        assert np.array_equal(res[form][5][0], res[1][5][0]) and np.array_equal(res[form][5][1], res[1][5][1])
    # uncompressed: every position its own block and segment - the lists would be larger than the payload, the default stays dense
    g = hml.Chain(device=0, seed=21, chain_id=0)
    g.load(x)
    g.scale_weights(1e9)
    g.set_model(K, g.autoprior(0.2, 0.9))
    g.sample_prior()
    g.iterate("F", 6, 2)
    g.sync()
    pool = hml.Pool(0, 0, 1, hml.Pool.unique_id())
    pool.marginals(g)
    assert pool.last()["form"] == "dense"
    pool.close()
    g.close()


def test_allreduce_marginals_of_three_chains_on_one_device(hml):
    """hml_allreduce_marginals (what `hammlet -chains N` calls): chains that share a device are summed there, the sum
    goes through a one-device RCCL communicator, every context ends with the pooled marginals."""
    x = ol.trace(T, K, 3)
    hml.Chain(device=0).close()
    cs = [gpu_chain(hml, x, c) for c in (0, 1, 2)]
    hml.allreduce_marginals(cs)
    seg_e, cnt_e, _ = expected_pooled(x, [0, 1, 2])
    for g in cs:
        seg, cnt = g.marginals_rle()
        assert np.array_equal(seg, seg_e) and np.array_equal(cnt, cnt_e)
        assert g.recorded_sweeps() == 3 * 8
        assert np.all(cnt.sum(1) == 24)
    ln, st = cs[0].max_segmentation()
    assert int(ln.sum()) == T
    for g in cs:
        g.close()


def test_allreduce_marginals_of_chains_with_many_states(hml):
    """The same for a model of 20 states (round 5: more than 16 states on the default path, hml_k_wide.h): relabelling by ascending
    mean, payload, pooled run-length marginals and the arg-max segmentation take the number of states at run time."""
    K20 = 20
    x = ol.trace(T, 6, 3)
    cs = [gpu_chain(hml, x, c, sweeps=(12, 3), K=K20) for c in (0, 1)]
    hml.allreduce_marginals(cs)
    from hammlet_amd import chains
    dense = np.zeros((K20, T), np.int64)
    bnd = np.zeros(T, np.int64)
    for c in (0, 1):
        d, means, b = checker_chain(x, c, sweeps=(12, 3), K=K20)
        dense += d[chains.relabel_permutation(means)]
        bnd += b
    starts = np.flatnonzero(bnd)
    cols = int(np.flatnonzero(dense.any(axis=1)).max()) + 1
    for g in cs:
        seg, cnt = g.marginals_rle()
        assert np.array_equal(seg, np.diff(np.append(starts, T))) and np.array_equal(cnt, dense[:cols, starts].T)
        assert np.all(cnt.sum(1) == 2 * 4)
    ln, st = cs[0].max_segmentation()
    assert int(ln.sum()) == T and int(st.max()) < K20
    for g in cs:
        g.close()


def test_pooled_contexts_keep_their_permutation_and_refuse_a_second_pooling(hml):
    """hml_allreduce_marginals_perm hands out every chain's relabelling (chain i's own label of pooled state
    j), hml_pool_permutation keeps it with the context, and a context whose marginals are a pooled payload refuses to be
    exported or pooled again (the relabelling would be applied twice and the counts multiplied) and to record further
    sweeps into them (own labels into common labels)."""
    x = ol.trace(T, K, 3)
    cs = [gpu_chain(hml, x, c) for c in (0, 1)]
    assert all(np.array_equal(g.pool_permutation(), np.arange(K)) for g in cs)          # identity before any pooling
    perms = hml.allreduce_marginals(cs, with_perms=True)
    _, _, perms_e = expected_pooled(x, [0, 1])
    for i, g in enumerate(cs):
        assert np.array_equal(perms[i], perms_e[i])
        assert np.array_equal(g.pool_permutation(), perms_e[i])
    seg0, cnt0 = cs[0].marginals_rle()
    with pytest.raises(hml.HmlError, match="pooled already"):
        hml.allreduce_marginals(cs)
    with pytest.raises(hml.HmlError, match="pooled already"):
        pool = hml.Pool(0, 0, 1, hml.Pool.unique_id())
        try:
            pool.marginals(cs[0])
        finally:
            pool.close()
    with pytest.raises(hml.HmlError, match="cannot be recorded"):
        cs[0].iterate("F", 3, 1)
    cs[0].set_recording(marginals=False)
    cs[0].iterate("F", 3, 1)                     # sampling goes on; the pooled marginals stay as they are
    cs[0].sync()
    seg1, cnt1 = cs[0].marginals_rle()
    assert np.array_equal(seg0, seg1) and np.array_equal(cnt0, cnt1)
    for g in cs:
        g.close()


def test_export_external_sum_install_equals_the_library_path(hml):
    """hml_pool_export / hml_pool_install with the sum done by the caller (any transport may stand in between)"""
    import torch
    x = ol.trace(T, K, 3)
    a, b = gpu_chain(hml, x, 0), gpu_chain(hml, x, 1)
    n = a.pool_payload_size()
    assert n == (K + 1) * (T + 1) + 1 + K
    pa = torch.empty(n, dtype=torch.int32, device="cuda")
    pb = torch.empty(n, dtype=torch.int32, device="cuda")
    a.pool_export(pa.data_ptr())
    b.pool_export(pb.data_ptr())
    pa += pb
    torch.cuda.synchronize()
    a.pool_install(pa.data_ptr())
    seg_e, cnt_e, _ = expected_pooled(x, [0, 1])
    seg, cnt = a.marginals_rle()
    assert np.array_equal(seg, seg_e) and np.array_equal(cnt, cnt_e)
    # the tensor-level mirror used by the CPU (gloo) test reads the same payload
    from hammlet_amd import chains
    s2, c2, nrec = chains.payload_to_rle(pa, K, T)
    assert nrec == 16 and np.array_equal(s2.cpu().numpy(), seg_e) and np.array_equal(c2.cpu().numpy(), cnt_e)
    a.close()
    b.close()


def test_relabelling_of_shared_parameter_states_and_permutation_checks(hml):
    """`-s C 2 2`: four states over two parameters - the permutation has K entries (ascending tuples of mapped
    means); hml_marginals_dense_device rejects an array that is not a permutation of the K states."""
    import torch
    Tm = 30000
    x = np.stack([ol.trace(Tm, 2, 61 + d) for d in range(2)], axis=1).reshape(-1)
    g = hml.Chain(device=0, seed=19)
    g.set_dimensions(2, 2)
    g.load(x)
    g.set_model(4, g.autoprior(0.2, 0.9))
    g.sample_prior()
    g.iterate("F", 12, 2)
    g.sync()
    perm = g.relabel_permutation()
    assert sorted(perm.tolist()) == [0, 1, 2, 3]
    mu = g.theta()[0::2]
    tuples = [(mu[s % 2], mu[s // 2]) for s in perm]
    assert tuples == sorted(tuples)
    buf = torch.empty((5, Tm), dtype=torch.int32, device="cuda")
    g.marginals_dense_device(buf.data_ptr(), perm)
    assert int(buf[:4].sum(0).min().item()) == 6 and int(buf[:4].sum(0).max().item()) == 6
    with pytest.raises(hml.HmlError):
        g.marginals_dense_device(buf.data_ptr(), np.array([0, 1, 1, 3], np.int32))
    with pytest.raises(hml.HmlError):
        g.marginals_dense_device(buf.data_ptr(), np.array([0, 1, 2, 7], np.int32))
    a, b = g.block_stats()
    assert a.shape == (2, g.num_blocks()) and b.shape == a.shape       # one plane per data dimension
    # pooling a multivariate chain with itself through the one-process entry point
    hml.allreduce_marginals([g])
    seg, cnt = g.marginals_rle()
    assert int(seg.sum()) == Tm and np.all(cnt.sum(1) == 6)
    g.close()
