"""bench.py's multi-rank branches rehearsed on ONE GPU: `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`
with HML_BENCH_FORCE_DIST=1 takes the world > 1 code at world size 1 - torch.distributed over the "nccl" backend (RCCL), the
communicator id created on rank 0 and handed round by broadcast_object_list (hammlet_amd.chains.make_pool), the library's own
collective hml_pool_marginals (dlopen of librccl next to torch's copy in one process), the `pooling` record of the JSON line.
The scaling runs on 2 / 4 / 8 GPUs are the driver's; this is their dry run."""
import json
import os
import socket
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_bench_multi_rank_branches_on_one_gpu():
    env = dict(os.environ, HML_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
           "--workload", "c2_1e7_k5", "--no-stream-leg", "--no-two-chain-leg", "--no-uncompressed-leg", "--no-scheme-legs",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0
    p = line["pooling"]
    assert p["rccl_version"] > 0
    # a strongly compressed chain: the collective takes the ranks' boundary lists (header 2 + K, then 1 + K words per segment)
    assert p["form"] == "lists" and p["dense_payload_bytes"] == 4 * ((5 + 1) * (10_000_000 + 1) + 1 + 5)
    assert p["all_reduce_bytes"] == 4 * (2 + 5 + p["list_slot_segments"] * 6) and p["all_reduce_bytes"] * 8 <= p["dense_payload_bytes"]
    assert p["counts_per_position"] == 2                                           # two recorded sweeps (F 10 5) of one chain
    assert line["roofline"]["launches"] >= 32
    # round 5: BOTH forms of the collective run and are checked (pooled row sums = ranks x recorded, asserted inside bench.py):
    # the boundary lists on the rank's chain, the dense [K+1][T+1] payload through ncclAllReduce on a second, attached chain
    assert p["default"]["form"] == "lists" and p["dense"]["form"] == "dense"
    assert p["dense"]["collective_bytes"] == p["dense_payload_bytes"] and p["dense"]["counts_per_position"] == 2
    assert p["dense"]["collective_ms"] > 0 and p["dense"]["algbw_GBps"] > 0
    # per-rank times and the slowdown against rank 0 running alone
    assert len(line["per_rank_ms_per_step"]) == 1 and line["rank0_alone_ms_per_step"] > 0
    assert 0.3 < line["efficiency_vs_rank0_alone"] < 3.0
    # the roofline object: frac = counter bytes as counted (lower bound) <= frac_upper (fetches doubled) <= 1; the kernel the north
    # star names is in the parsed line
    r = line["roofline"]
    assert 0 < r["frac"] <= r["frac_upper"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert "forward_trellis" in r and r["forward_trellis"]["strongly_compressed"]["kernel"] == "hml_k_forward<5>"
