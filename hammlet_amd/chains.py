"""Chain-parallel execution: independent Gibbs chains, one per GPU (one process per GPU), and the single
collective of the design - a sum all-reduce (RCCL over xGMI when the backend is "nccl") that pools the
chains' state marginals at the end.

The reference has nothing distributed (one process, one thread, src/main.cpp).  Chains never communicate
while sampling; pooling needs (1) a common labelling of the states - every chain relabels its states by
ascending emission mean, the idea of the reference's bin/sortStates:1-6 - and (2) one all-reduce over the
dense [K+1][T] int32 array (K count rows + one row that is non-zero at segment boundaries).
"""
import numpy as np


def relabel_permutation(means):
    """perm[new] = old such that the states are ordered by ascending mean (ties keep their order)."""
    means = np.asarray(means, np.float64)
    return np.argsort(means, kind="stable").astype(np.int32)


def pool_dense(dense, group=None):
    """Sum all-reduce of a [K+1][T] int32 tensor over the process group (in place).  Works on any backend:
    "nccl" (= RCCL on ROCm) for device tensors, "gloo" for the CPU tests."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(dense, op=dist.ReduceOp.SUM, group=group)
    return dense


def dense_to_rle(dense):
    """Run-length form of pooled dense counts: cut at every position whose boundary row is non-zero.
    Returns (segment lengths [M], counts [M][K]) as tensors on dense's device."""
    import torch
    K = dense.shape[0] - 1
    T = dense.shape[1]
    flags = dense[K] != 0
    flags[0] = True
    starts = torch.nonzero(flags, as_tuple=False).flatten()
    ends = torch.cat([starts[1:], torch.tensor([T], device=dense.device, dtype=starts.dtype)])
    counts = dense[:K, starts].t().contiguous()
    return (ends - starts), counts


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


def pooled_marginals(chain, group=None, device=None):
    """Relabel this chain's marginals by ascending mean, export them densely on the GPU, pool them over
    all chains of the process group and return (segment lengths, counts) of the pooled marginals."""
    import torch
    K, T = chain.K, chain.T
    theta = chain.theta()
    perm = relabel_permutation(theta[0::2])
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    dense = torch.empty((K + 1, T), dtype=torch.int32, device=dev)
    chain.marginals_dense_device(dense.data_ptr(), perm)
    pool_dense(dense, group)
    seg, cnt = dense_to_rle(dense)
    return seg, cnt, perm
