"""Chain-parallel execution: independent Gibbs chains, one per GPU (one process per GPU), and the single collective of
the design - a sum all-reduce that pools the chains' state marginals at the end.

The product path is the C ABI: `hml_pool_create` / `hml_pool_marginals` (RCCL over xGMI inside libhammlet_hip.so,
hammlet_amd.Pool here) or `hml_allreduce_marginals` for one process that drives several GPUs.  This module holds the
launcher glue around it - how the ranks of a torch.distributed job obtain the RCCL id - and a tensor-level mirror of the
payload algebra (`payload_from_dense`, `pool_payload`, `payload_to_rle`) that works on any backend, which is how the
world-size-2 gloo test on CPU covers the N > 1 semantics (relabel, sum, cut at the union of the boundaries).

The reference has nothing distributed (one process, one thread, src/main.cpp:108).  Pooling needs (1) a common
labelling of the states - every chain relabels its states by ascending emission mean, the idea of the reference's
bin/sortStates:1-6 - and (2) one all-reduce over the int32 payload [K+1][T+1] (+ K+1 words): K relabelled difference
arrays of the recorded marginals, one row that is non-zero at segment boundaries, the number of recorded sweeps and
which states were ever recorded (hammlet_amd/csrc/hml_k_pool.h).
"""
import numpy as np


def relabel_permutation(means):
    """perm[new] = old such that the states are ordered by ascending mean (ties keep their order); `means` may be
    [K] or, for "-s C P D" states, [K][D] tuples of mapped parameter means compared in dimension order"""
    means = np.asarray(means, np.float64)
    if means.ndim == 1:
        return np.argsort(means, kind="stable").astype(np.int32)
    return np.lexsort(tuple(means[:, d] for d in range(means.shape[1] - 1, -1, -1))).astype(np.int32)


def payload_from_dense(dense, boundary, perm, n_recorded):
    """The int32 payload of one chain from dense per-position counts [K][T], a 0/1 boundary row [T], the relabelling
    and the number of recorded sweeps - what hml_pool_export builds on the device from the difference arrays."""
    dense = np.asarray(dense, np.int64)
    K, T = dense.shape
    d = dense[np.asarray(perm)]
    diff = np.zeros((K + 1, T + 1), np.int64)
    diff[:K, 0] = d[:, 0]
    diff[:K, 1:T] = d[:, 1:] - d[:, :-1]
    diff[K, :T] = np.asarray(boundary) != 0
    diff[K, 0] = 1
    used = (d != 0).any(axis=1).astype(np.int64)
    return np.concatenate([diff.ravel(), [n_recorded], used]).astype(np.int32)


def pool_payload(payload, group=None):
    """Sum all-reduce of a payload tensor over the process group (in place).  Works on any backend: "nccl" (= RCCL on
    ROCm) for device tensors, "gloo" for the CPU tests.  (The product path does this inside the library.)"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(payload, op=dist.ReduceOp.SUM, group=group)
    return payload


def list_from_dense(dense, boundary, perm, n_recorded):
    """The boundary list of one chain (hml_k_pool_list_pack): header [M, recorded sweeps, used[0..K-1]], then for every
    marginal segment [position, deltas of the relabelled states] - the non-zero columns of payload_from_dense's arrays."""
    dense = np.asarray(dense, np.int64)
    K, T = dense.shape
    pay = payload_from_dense(dense, boundary, perm, n_recorded).astype(np.int64)
    body = pay[: (K + 1) * (T + 1)].reshape(K + 1, T + 1)
    pos = np.flatnonzero(body[K, :T])
    entries = np.concatenate([pos[:, None], body[:K, pos].T], axis=1)
    return np.concatenate([[len(pos), n_recorded], pay[(K + 1) * (T + 1) + 1:], entries.ravel()]).astype(np.int32)


def pool_lists(lst, K, group=None):
    """All-gather of the ranks' boundary lists in equally sized slots (the library: ncclAllGather; here any backend).
    Returns the list of all ranks' lists."""
    import torch
    import torch.distributed as dist
    t = torch.as_tensor(lst, dtype=torch.int32)
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return [t]
    world = dist.get_world_size(group)
    m = torch.tensor([int(t[0])], dtype=torch.int64)
    dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
    slot = 2 + K + int(m.item()) * (K + 1)
    mine = torch.zeros(slot, dtype=torch.int32)
    mine[: t.numel()] = t
    out = [torch.zeros(slot, dtype=torch.int32) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    return out


def payload_from_lists(lists, K, T):
    """The pooled dense payload from all ranks' boundary lists: every entry added into zeroed arrays
    (hml_k_pool_list_install) - equal to the sum of the ranks' dense payloads."""
    import torch
    body = torch.zeros((K + 1, T + 1), dtype=torch.int64)
    rec, used = 0, torch.zeros(K, dtype=torch.int64)
    for l in lists:
        l = torch.as_tensor(l).to(torch.int64)
        M = int(l[0])
        rec += int(l[1])
        used += l[2:2 + K]
        e = l[2 + K: 2 + K + M * (K + 1)].view(M, K + 1)
        body[:K].index_put_((torch.arange(K)[None, :].expand(M, K), e[:, :1].expand(M, K)), e[:, 1:], accumulate=True)
        body[K, e[:, 0]] += 1
    body[K, 0] = max(int(body[K, 0]), 1)
    return torch.cat([body.flatten(), torch.tensor([rec]), used]).to(torch.int32)


def payload_to_rle(payload, K, T):
    """Run-length form of a pooled payload: prefix sums of the difference rows, cut wherever the boundary row is
    non-zero; columns up to the highest state any chain recorded (reference src/StateMarginals.hpp:300-303).
    Returns (segment lengths [M], counts [M][columns], recorded sweeps)."""
    import torch
    body = payload[: (K + 1) * (T + 1)].view(K + 1, T + 1)
    tail = payload[(K + 1) * (T + 1):]
    flags = body[K, :T] != 0
    flags[0] = True
    starts = torch.nonzero(flags, as_tuple=False).flatten()
    ends = torch.cat([starts[1:], torch.tensor([T], device=payload.device, dtype=starts.dtype)])
    dense = torch.cumsum(body[:K, :T].to(torch.int64), dim=1)
    used = torch.nonzero(tail[1:1 + K] != 0, as_tuple=False).flatten()
    cols = int(used.max().item()) + 1 if used.numel() else 0
    counts = dense[:cols, starts].t().contiguous()
    return (ends - starts), counts, int(tail[0].item())


def max_segmentation(seg, cnt):
    """Maximum-posterior-margin segmentation of (pooled) run-length marginals - the reference's post-processing tool
    (reference src/tools/maxSegmentation.cpp:53-82) on tensors: arg-max state per segment (first maximum; state 0 for
    an all-zero row), adjacent segments of equal state merged.  Returns (run lengths, run states) on seg's device."""
    import torch
    if cnt.shape[1] == 0:
        state = torch.zeros(seg.shape[0], dtype=torch.int64, device=seg.device)
    else:
        best = cnt.max(dim=1, keepdim=True).values
        # first column that reaches the maximum; a row without a positive count gives column 0 like the tool's `>`
        first = (cnt == best).to(torch.int8).argmax(dim=1)
        state = torch.where(best.flatten() > 0, first, torch.zeros_like(first))
    keep = torch.ones_like(state, dtype=torch.bool)
    keep[1:] = state[1:] != state[:-1]
    starts = torch.nonzero(keep, as_tuple=False).flatten()
    csum = torch.cat([torch.zeros(1, dtype=seg.dtype, device=seg.device), torch.cumsum(seg, 0)])
    ends = torch.cat([starts[1:], torch.tensor([seg.shape[0]], device=seg.device, dtype=starts.dtype)])
    return csum[ends] - csum[starts], state[starts]


def make_pool(device, group=None, always_broadcast=False):
    """The RCCL communicator of this rank inside a torch.distributed job (one process per GPU): rank 0 creates the id,
    the job's own backend broadcasts its 128 bytes, every rank joins (ncclCommInitRank).  Without an initialised
    process group: a one-rank communicator.  always_broadcast: take the broadcast path in a one-rank group as well
    (rehearsal of the multi-rank path on one GPU)."""
    import torch.distributed as dist
    from .capi import Pool
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or always_broadcast):
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [Pool.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return Pool(device, rank, world, box[0])
    return Pool(device, 0, 1, Pool.unique_id())


def pooled_marginals(chain, pool):
    """Pools the recorded marginals of all chains of the communicator (hml_pool_marginals) and returns the pooled
    run-length marginals (segment lengths, counts) together with the permutation this chain applied."""
    perm = pool.marginals(chain)
    seg, cnt = chain.marginals_rle()
    return seg, cnt, perm
