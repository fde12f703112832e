"""Chain-parallel pooling on CPU: world_size 2 over gloo.  Each rank runs its own chain (the CPU checker,
chain id = rank), relabels by ascending mean, builds the payload the library's hml_pool_export builds on the device
(relabelled difference arrays + boundary row + recorded sweeps + used states), all-reduces it and checks the pooled
result against the sum computed directly from both chains."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import oracle_lib as ol

T, K = 20000, 3
N_RECORDED = 15


def chain_dense(rank):
    x = ol.trace(T, K, 4)
    o = ol.OracleChain(K=K, seed=9, chain=rank, rng=ol.RNG_CTR, math=ol.MATH_DEV, reduce=ol.REDUCE_DEV)
    o.load(x)
    o.autoprior()
    o.init_model()
    o.iterate("F", 30, 2)
    dense = o.marginals_dense()                    # [K][T]
    means = o.theta()[0::2]
    # boundary row from the checker's run-length text
    lens = [int(l.split("\t")[0]) for l in o.text("marginals").strip().split("\n")]
    bnd = np.zeros(T, np.int32)
    bnd[np.cumsum([0] + lens[:-1])] = 1
    return dense, means, bnd


def worker(rank, world, port, out):
    from hammlet_amd import chains
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dense, means, bnd = chain_dense(rank)
    perm = chains.relabel_permutation(means)
    # the payload of hml_pool_export (relabelled difference arrays, boundary row, recorded sweeps, used states)
    t = torch.from_numpy(chains.payload_from_dense(dense, bnd, perm, N_RECORDED))
    chains.pool_payload(t)
    seg, cnt, n_rec = chains.payload_to_rle(t, K, T)
    # the boundary-list form (hml_pool_marginals' other collective: all-gather of the ranks' segment lists): same pooled arrays
    lists = chains.pool_lists(chains.list_from_dense(dense, bnd, perm, N_RECORDED), K)
    t2 = chains.payload_from_lists(lists, K, T)
    body, body2 = t[: (K + 1) * (T + 1)].view(K + 1, T + 1), t2[: (K + 1) * (T + 1)].view(K + 1, T + 1)
    assert torch.equal(body[:K], body2[:K]) and torch.equal(body[K] != 0, body2[K] != 0)
    assert torch.equal(t[(K + 1) * (T + 1):] != 0, t2[(K + 1) * (T + 1):] != 0) and int(t[(K + 1) * (T + 1)]) == int(t2[(K + 1) * (T + 1)])
    seg2, cnt2, n_rec2 = chains.payload_to_rle(t2, K, T)
    assert torch.equal(seg, seg2) and torch.equal(cnt, cnt2) and n_rec == n_rec2
    assert sum(l.numel() for l in lists) * 8 < t.numel()   # ... from a fraction of the bytes
    if rank == 0:
        np.savez(out, payload=t.numpy(), seg=seg.numpy(), cnt=cnt.numpy(), n_rec=n_rec)
    dist.barrier()
    dist.destroy_process_group()


def test_pooled_marginals_world2(tmp_path):
    from hammlet_amd import chains
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "pooled.npz")
    mp.spawn(worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    expect = np.zeros((K + 1, T), np.int64)
    for r in range(2):
        dense, means, bnd = chain_dense(r)
        perm = chains.relabel_permutation(means)
        assert np.all(np.diff(means[perm]) >= 0)
        expect[:K] += dense[perm]
        expect[K] += bnd
    body = got["payload"][: (K + 1) * (T + 1)].reshape(K + 1, T + 1)
    # prefix sums of the pooled difference rows are the sums of the relabelled dense counts
    assert np.array_equal(np.cumsum(body[:K, :T], axis=1), expect[:K])
    assert np.all(body[:K, T] == 0)
    assert int(got["n_rec"]) == 2 * N_RECORDED
    # pooled row sums = chains x recorded sweeps; segments tile [0, T) and are cut at the union of the chains' boundaries
    assert np.all(expect[:K].sum(0) == 2 * N_RECORDED)
    assert got["seg"].sum() == T
    starts = np.concatenate([[0], np.cumsum(got["seg"])[:-1]])
    assert np.array_equal(got["cnt"], expect[:K, starts].T)
    assert set(np.flatnonzero(expect[K])) == set(starts)


def test_payload_round_trip_and_unused_states():
    """payload_from_dense / payload_to_rle: a state no chain recorded is not printed (reference
    src/StateMarginals.hpp:300-303), an interior unused state is a zero column"""
    from hammlet_amd import chains
    dense = np.zeros((4, 12), np.int64)
    dense[0, :5] = 3
    dense[2, 5:] = 3
    bnd = np.zeros(12, np.int32)
    bnd[[0, 5, 9]] = 1
    t = torch.from_numpy(chains.payload_from_dense(dense, bnd, np.arange(4), 3))
    seg, cnt, n = chains.payload_to_rle(t, 4, 12)
    assert n == 3 and seg.tolist() == [5, 4, 3]
    assert cnt.tolist() == [[3, 0, 0], [0, 0, 3], [0, 0, 3]]


def test_max_segmentation_of_pooled_marginals_matches_the_tool():
    """chains.max_segmentation on run-length tensors against the restated reference tool (pinned on the reference
    tool's own outputs by tests/test_maxseg_cpu.py), on the committed golden marginals files"""
    import glob
    from hammlet_amd import chains
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    files = sorted(glob.glob(os.path.join(gold, "*", "marginals.csv"))) + sorted(glob.glob(os.path.join(gold, "maxseg", "*.marginals")))
    assert len(files) > 15
    for path in files:
        text = open(path).read()
        rows = [list(map(int, l.split())) for l in text.splitlines() if l.strip()]
        if not rows:
            continue
        width = max(len(r) for r in rows) - 1
        seg = torch.tensor([r[0] for r in rows], dtype=torch.int64)
        cnt = torch.tensor([r[1:] + [0] * (width - len(r) + 1) for r in rows], dtype=torch.int32).reshape(len(rows), width)
        ln, st = chains.max_segmentation(seg, cnt)
        lines = (["0\t0\n"] if int(st[0]) != 0 else []) + ["%d\t%d\n" % (int(a), int(b)) for a, b in zip(ln, st)]
        assert "".join(lines) == ol.max_segmentation_text(text), path


def test_relabel_permutation():
    from hammlet_amd import chains
    assert list(chains.relabel_permutation([0.5, -1.0, 2.0])) == [1, 0, 2]
    assert list(chains.relabel_permutation([1.0, 1.0, 0.0])) == [2, 0, 1]
    # "-s C 2 2": state s maps to parameters (s % 2, s / 2); tuples compared in dimension order
    mu = np.array([0.7, -0.3])
    tuples = np.array([[mu[s % 2], mu[s // 2]] for s in range(4)])
    assert list(chains.relabel_permutation(tuples)) == [3, 1, 2, 0]
