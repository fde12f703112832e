"""A bounded sample of the randomised differential run inside `-m gpu` (the thousands of configurations of
tools/fuzz_parity.py were builder-run only): 300 random configurations at a fixed seed - sizes around the structural
edges, 2-16 states, 1-3 data dimensions, random schemes of M / F / S / D / P tokens, priors, weight multipliers, read-depth
input, every switchable kernel path - GPU against the checker in device mode after each: blocks, state sequences,
parameter bits, marginals text; then 100 configurations of several chains through hml_iterate_many.  About 45 seconds."""
import pytest

from tests.fuzz_util import fuzz

pytestmark = pytest.mark.gpu


def test_bounded_fuzz_against_the_checker(hml):
    assert fuzz(hml, 300, 20261004) == 300


def test_bounded_fuzz_with_a_tiny_block_capacity(hml, monkeypatch):
    """The same differential run with every chain's per-block buffers sized for 64 blocks (HML_MAX_BLOCKS): enumerations that
    find more halt the chain on the device, the host grows the buffers and runs the skipped sweeps again (hml_settle) -
    at every kind of call (auto prior, explicit thresholds, static structures, prior draws, mixture / FB sweeps, both forward
    geometries) the results must be those of a chain with room from the start, i.e. the checker's."""
    monkeypatch.setenv("HML_MAX_BLOCKS", "64")
    assert fuzz(hml, 150, 4) == 150


def test_bounded_fuzz_of_batched_chains(hml):
    """Round 4's sweep of several chains: 2-10 chains of one trace through hml_iterate_many - attached to one construction
    (hml_attach_observations, the many-chain block kernel hml_m_blocks_fused) or with private ones, in 1-4 groups of
    chains on separate streams, tiles of several batches, block capacities of 64, weakly compressed chains that leave the
    batch, static block structures, prior draws between the calls - every chain against the checker's chain of the same
    (seed, chain) run alone: blocks, state sequences, parameter and transition bits, marginals text."""
    assert fuzz(hml, 100, 20261005, many=True) == 100


def test_bounded_fuzz_of_the_reference_compatible_mode(hml):
    """The reference-compatible mode (option "compat", hml_k_compat.h) against the checker's REFERENCE mode - sequential mt19937,
    libm arithmetic, Kahan sums, size_t += float counts, i.e. the reference binary's own chain (the checker in that mode is pinned
    on the binary's files, tests/test_oracle_golden.py) - on 120 random configurations: 2-64 states, 1-3 data dimensions, random
    schemes, weakly compressed input, and chunk geometries of the filter / backward draws from the sequential form to hundreds of
    chunks without warm-up, where chunks start wrong and run again."""
    assert fuzz(hml, 120, 20261006, compat=True) == 120


def test_bounded_fuzz_of_the_path_for_many_states(hml):
    """Round 5: the default path's kernels for models of more than 16 states (hml_k_wide.h, hml_k_wide_lanes.h - the number of states
    a run-time value; a chunk a lane over chunk-transposed arrays, or a state a lane; the reference takes any `-s K`,
    src/main.cpp:112-137) against the checker's DEVICE mode on 160 random configurations: 2-64 states (HML_WIDE=1 sends the small
    models there as well), 1-3 data dimensions (up to 7^2 = 49 states with shared parameters), random schemes of M / F / S / D / P
    tokens, read-depth input, and chunk geometries from the sequential form to chunks of one block without warm-up, where chunks
    start wrong and run again."""
    assert fuzz(hml, 160, 20261007, wide=True) == 160
