"""Randomised differential run of the GPU text reader against the checker's restatement of `while ( input >> v )`:
random token streams (plain numbers in many formats, ties, long digit strings, sub-normal and overflowing values,
glued and malformed tokens that end the extraction), random separators, random staging-buffer and feed sizes.
    python tools/fuzz_text.py [rounds=40] [seed=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hammlet_amd as hml
from tests import oracle_lib as ol
from tests.test_text_reader_cpu import random_tokens

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
SEPS = np.array([" ", "\n", "\t", "  ", "\r\n", " \n ", "\v", "\f"])
total = 0
for r in range(rounds):
    n = int(rng.choice([1, 50, 5000, 200000]))
    toks = random_tokens(rng, n)
    if rng.random() < 0.6:   # most rounds: no token that ends the extraction, so that the whole text is compared
        finite = np.isfinite(ol.parse_tokens(toks)[2])
        toks = [t for t, ok in zip(toks, finite) if ok and t not in ("5e", ".", "-", "abc", "1,5", "0x1p3", "nan", "1e39")]
    seps = SEPS[rng.integers(0, len(SEPS), len(toks))]
    text = ("".join(t + s for t, s in zip(toks, seps))).encode()
    if rng.random() < 0.3:
        text = text.rstrip()          # no blank after the last token
    want, stopped = ol.parse_text(text)
    chunk = int(rng.choice([0, 256, 1000, 4096, 65536, 1 << 20]))
    feed = int(rng.choice([0, 1, 7, 4096, 100000])) or None
    if feed == 1 and len(text) > 20000:
        feed = 4096
    longest = max((len(t) for t in toks), default=1)
    if chunk and chunk <= longest + 2:
        chunk = 4096
    got, info = hml.parse_text(text, chunk_bytes=chunk, feed_bytes=feed, with_info=True)
    ok = got.size == want.size and np.array_equal(got.view(np.uint32), want.view(np.uint32)) and info["stopped"] == stopped
    total += want.size
    print("%3d %s tokens=%d bytes=%d chunk=%d feed=%s values=%d stopped=%s irregular=%d" % (r, "ok " if ok else "DIFF", len(toks), len(text), chunk, feed, want.size, stopped, info["irregular_tokens"]), flush=True)
    if not ok:
        sys.exit(1)
print("all %d rounds identical (%d values)" % (rounds, total))
