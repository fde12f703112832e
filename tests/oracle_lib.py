"""TEST INFRASTRUCTURE: ctypes wrapper of the CPU checker oracle/liboracle.so (see oracle/hml_oracle.hpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")
_lib = None

RNG_MT, RNG_PHILOX_SEQ, RNG_CTR, RNG_MT_RESTATED = 0, 1, 2, 3
MATH_LIBM, MATH_DEV = 0, 1
REDUCE_REF, REDUCE_DEV = 0, 1

_P = C.c_void_p


def load():
    global _lib
    if _lib is not None:
        return _lib
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("oracle_capi.cpp", "hml_oracle.hpp")]
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.run(["make", "-C", ORACLE_DIR, "liboracle.so", "hammlet_oracle"], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(LIB)
    lib.orc_last_error.restype = C.c_char_p
    lib.orc_create.restype = _P
    lib.orc_create.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float,
                               C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int]
    lib.orc_destroy.argtypes = [_P]
    lib.orc_load.argtypes = [_P, _P, C.c_uint64, C.c_int]
    lib.orc_autoprior.argtypes = [_P, _P]
    lib.orc_set_prior.argtypes = [_P, _P]
    lib.orc_init_model.argtypes = [_P]
    lib.orc_token.argtypes = [_P, C.c_char]
    lib.orc_set_record.argtypes = [_P] + [C.c_int] * 5
    lib.orc_set_probes.argtypes = [_P, C.c_int]
    lib.orc_set_record_segments.argtypes = [_P, C.c_int]
    lib.orc_iterate.argtypes = [_P, C.c_char, C.c_uint64, C.c_uint64]
    lib.orc_enumerate_blocks.argtypes = [_P, C.c_float]
    for n in ("orc_T", "orc_nblocks", "orc_total_blocks", "orc_warn_uniform", "orc_n_recorded"):
        getattr(lib, n).restype = C.c_uint64
        getattr(lib, n).argtypes = [_P]
    lib.orc_sigma_hat.restype = C.c_double
    lib.orc_sigma_hat.argtypes = [_P]
    lib.orc_threshold.restype = C.c_float
    lib.orc_threshold.argtypes = [_P]
    for n in ("orc_get_coeffs", "orc_get_weights", "orc_get_blocks", "orc_get_states", "orc_get_theta", "orc_get_A",
              "orc_get_pi", "orc_get_loglik", "orc_get_forward_rows"):
        getattr(lib, n).argtypes = [_P, _P]
    lib.orc_get_integral.argtypes = [_P, _P, _P]
    lib.orc_get_block_stats.argtypes = [_P, _P, _P]
    lib.orc_eval_emission.argtypes = [_P, _P]
    lib.orc_set_params.argtypes = [_P, _P, _P, _P]
    lib.orc_get_counts.argtypes = [_P] * 6
    lib.orc_get_posterior.argtypes = [_P] * 4
    lib.orc_marginals_dense.argtypes = [_P, _P]
    lib.orc_text.restype = C.c_uint64
    lib.orc_text.argtypes = [_P, C.c_int, _P, C.c_uint64]
    lib.orc_philox.argtypes = [C.c_uint32] * 6 + [_P]
    for n in ("orc_expf_dev", "orc_expf_libm", "orc_logf_dev", "orc_logf_libm"):
        getattr(lib, n).argtypes = [_P, _P, C.c_uint64]
    for n in ("orc_powf_dev", "orc_powf_libm"):
        getattr(lib, n).argtypes = [_P, _P, _P, C.c_uint64]
    lib.orc_expf_mismatches.restype = C.c_uint64
    lib.orc_expf_mismatches.argtypes = [C.c_uint32, C.c_uint32, _P]
    for n in ("orc_check_gamma", "orc_check_normal"):
        getattr(lib, n).restype = C.c_uint64
        getattr(lib, n).argtypes = [C.c_uint32, C.c_uint64, C.c_float, C.c_float]
    lib.orc_check_categorical.restype = C.c_uint64
    lib.orc_check_categorical.argtypes = [C.c_uint32, C.c_uint64, C.c_int, C.c_int]
    lib.orc_synth_gauss.argtypes = [_P, _P, C.c_uint64, C.c_int, _P, C.c_float, C.c_double, C.c_uint64, C.c_int]
    lib.orc_debug_eval.argtypes = [C.c_int, _P, _P, _P, C.c_uint64, C.c_uint64]
    lib.orc_synth_depth.argtypes = [_P, _P, C.c_uint64, C.c_double, C.c_double, C.c_uint64, C.c_int]
    lib.orc_time_sweeps.restype = C.c_double
    lib.orc_time_sweeps.argtypes = [_P, C.c_char, C.c_uint64]
    _lib = lib
    return lib


def synth_gauss(T, K, mu, sigma, dwell, seed, nthreads=8):
    lib = load()
    x = np.empty(T, np.float32)
    mu = np.ascontiguousarray(mu, np.float32)
    lib.orc_synth_gauss(x.ctypes.data, None, T, K, mu.ctypes.data, sigma, dwell, seed, nthreads)
    return x


def synth_depth(T, depth=15.0, ln_sigma=0.15, seed=5, nthreads=8, with_states=False):
    lib = load()
    x = np.empty(T, np.float32)
    st = np.empty(T, np.int16) if with_states else None
    lib.orc_synth_depth(x.ctypes.data, st.ctypes.data if with_states else None, T, depth, ln_sigma, seed, nthreads)
    return (x, st) if with_states else x


def debug_eval(fn, a, b=None, seed=0):
    lib = load()
    a = np.ascontiguousarray(a, np.float32)
    out = np.empty_like(a)
    if b is not None:
        b = np.ascontiguousarray(b, np.float32)
    lib.orc_debug_eval(fn, a.ctypes.data, b.ctypes.data if b is not None else None, out.ctypes.data, a.size, seed)
    return out


# the synthetic configurations of SURVEY.md section 8d (levels, sigma, mean dwell)
LEVELS = {2: [-1, 1], 3: [-1, 0, 1], 4: [-1.5, -0.5, 0.5, 1.5], 5: [-2, -1, 0, 1, 2], 6: [-2.5, -1.5, -0.5, 0.5, 1.5, 2.5],
          10: [x - 4.5 for x in range(10)], 16: [x - 7.5 for x in range(16)]}
SIGMA = {2: 0.2, 3: 0.2, 4: 0.25, 5: 0.3, 6: 0.3, 10: 0.3, 16: 0.3}
DWELL = {2: 1000, 3: 2000, 4: 1000, 5: 5000, 6: 2000, 10: 5000, 16: 3000}


def trace(T, K, seed):
    return synth_gauss(T, K, LEVELS[K], SIGMA[K], DWELL[K], seed)


class OracleChain:
    """Mirror of hammlet_amd.Chain on the CPU checker."""

    def __init__(self, K=3, seed=0, chain=0, rng=RNG_MT, math=MATH_LIBM, reduce=REDUCE_REF, e_var=0.2, e_p=0.9,
                 t_off=0.5, t_diag=0.5, pi_alpha=0.5, self_trans=True, weight_mult=1.0):
        self.lib = load()
        self.K = K
        self.h = self.lib.orc_create(K, e_var, e_p, t_off, t_diag, pi_alpha, 1 if self_trans else 0, weight_mult, seed,
                                     chain, rng, math, reduce)
        if not self.h:
            raise RuntimeError(self.lib.orc_last_error().decode())
        self.T = 0
        self.D, self.P = 1, None

    def set_dimensions(self, D, P):
        """`-s C P D`: D interleaved data dimensions, P emission parameters shared by P**D states (before load)"""
        self.lib.orc_set_dims.argtypes = [_P, C.c_int, C.c_int]
        self._chk(self.lib.orc_set_dims(self.h, D, P))
        self.D, self.P = D, P

    def _chk(self, rc):
        if rc:
            raise RuntimeError(self.lib.orc_last_error().decode())

    def close(self):
        if self.h:
            self.lib.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def load(self, x, pointers=True):
        x = np.ascontiguousarray(x, np.float32)
        self.T = x.size // self.D
        self._chk(self.lib.orc_load(self.h, x.ctypes.data, x.size, 1 if pointers else 0))

    def autoprior(self):
        out = np.empty(4, np.float32)
        self._chk(self.lib.orc_autoprior(self.h, out.ctypes.data))
        return out

    def set_prior(self, p):
        p = np.ascontiguousarray(p, np.float32)
        self._chk(self.lib.orc_set_prior(self.h, p.ctypes.data))

    def init_model(self):
        self._chk(self.lib.orc_init_model(self.h))

    def token(self, t):
        self._chk(self.lib.orc_token(self.h, t.encode()))

    def set_record(self, marginals=True, sequences=False, blocks=False, params=False, compression=False, segments=False):
        self.lib.orc_set_record(self.h, int(marginals), int(sequences), int(blocks), int(params), int(compression))
        self.lib.orc_set_record_segments(self.h, int(segments))

    def set_probes(self, on=True):
        self.lib.orc_set_probes(self.h, int(on))

    def iterate(self, method, iters, thin=0):
        self._chk(self.lib.orc_iterate(self.h, method.encode(), iters, thin))

    def enumerate_blocks(self, thr):
        self._chk(self.lib.orc_enumerate_blocks(self.h, thr))

    def sigma_hat(self):
        return self.lib.orc_sigma_hat(self.h)

    def threshold(self):
        return self.lib.orc_threshold(self.h)

    def num_blocks(self):
        return self.lib.orc_nblocks(self.h)

    def _arr(self, fn, n, dtype):
        a = np.empty(n, dtype)
        getattr(self.lib, fn)(self.h, a.ctypes.data)
        return a

    def coeffs(self):
        return self._arr("orc_get_coeffs", self.T, np.float32)

    def weights(self):
        return self._arr("orc_get_weights", self.T, np.float32)

    def integral(self):
        a = np.empty(self.T + 1, np.float32)
        b = np.empty(self.T + 1, np.float32)
        self.lib.orc_get_integral(self.h, a.ctypes.data, b.ctypes.data)
        return a, b

    def blocks(self):
        return self._arr("orc_get_blocks", self.num_blocks() + 1, np.uint32)

    def block_stats(self):
        B = self.num_blocks()
        a = np.empty(B, np.float32)
        b = np.empty(B, np.float32)
        self.lib.orc_get_block_stats(self.h, a.ctypes.data, b.ctypes.data)
        return a, b

    def states(self):
        return self._arr("orc_get_states", self.num_blocks(), np.int16)

    def theta(self):
        return self._arr("orc_get_theta", 2 * (self.P or self.K), np.float32)

    def transitions(self):
        return self._arr("orc_get_A", self.K * self.K, np.float32).reshape(self.K, self.K), self._arr("orc_get_pi", self.K, np.float32)

    def set_params(self, mean_var, A, pi):
        mv = np.ascontiguousarray(mean_var, np.float32)
        A = np.ascontiguousarray(A, np.float32)
        pi = np.ascontiguousarray(pi, np.float32)
        self.lib.orc_set_params(self.h, mv.ctypes.data, A.ctypes.data, pi.ctypes.data)

    def loglik(self):
        return self._arr("orc_get_loglik", self.num_blocks() * self.K, np.float32).reshape(-1, self.K)

    def emission_terms(self):
        """innerProduct - N * logNormalizer of the current block list under the current theta (no self-transition term)"""
        E = np.empty((self.num_blocks(), self.K), np.float32)
        self.lib.orc_eval_emission(self.h, E.ctypes.data)
        return E

    def forward_rows(self):
        return self._arr("orc_get_forward_rows", (self.num_blocks() + 1) * self.K, np.float32).reshape(-1, self.K)

    def counts(self):
        K = self.K
        trans = np.empty((K, K), np.uint64)
        occ = np.empty(K, np.uint64)
        s = np.empty(K, np.float32)
        q = np.empty(K, np.float32)
        n = np.empty(K, np.uint64)
        self.lib.orc_get_counts(self.h, trans.ctypes.data, occ.ctypes.data, s.ctypes.data, q.ctypes.data, n.ctypes.data)
        return trans, occ, s, q, n

    def marginals_dense(self):
        out = np.empty((self.K, self.T), np.int32)
        self._chk(self.lib.orc_marginals_dense(self.h, out.ctypes.data))
        return out

    def text(self, which):
        idx = {"marginals": 0, "sequences": 1, "blocks": 2, "parameters": 3, "compression": 4, "segments": 5}[which]
        n = self.lib.orc_text(self.h, idx, None, 0)
        buf = C.create_string_buffer(n + 1)
        self.lib.orc_text(self.h, idx, buf, n)
        return buf.raw[:n].decode()

    def time_sweeps(self, method, iters):
        return self.lib.orc_time_sweeps(self.h, method.encode(), iters)

    def total_blocks(self):
        return self.lib.orc_total_blocks(self.h)


def parse_text(data):
    """The reference's reader restated (`while ( input >> v )`, reference src/wavelet.hpp:131): (values, stopped)."""
    lib = load()
    lib.orc_parse_text.restype = C.c_uint64
    lib.orc_parse_text.argtypes = [C.c_char_p, C.c_uint64, _P, C.c_uint64, C.POINTER(C.c_int)]
    data = bytes(data)
    cap = len(data) // 2 + 2
    out = np.empty(cap, np.float32)
    stopped = C.c_int()
    n = lib.orc_parse_text(data, len(data), out.ctypes.data, cap, C.byref(stopped))
    return out[:n].copy(), bool(stopped.value)


def parse_tokens(tokens):
    """hml_text.h's token converter compiled by gcc: (values, status, strtof values) for a list of ASCII tokens."""
    lib = load()
    lib.orc_parse_tokens.restype = None
    lib.orc_parse_tokens.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, _P, _P, _P]
    stride = max(len(t) for t in tokens) + 1
    blob = b"".join(t.encode().ljust(stride, b"\0") for t in tokens)
    n = len(tokens)
    out = np.empty(n, np.float32)
    st = np.empty(n, np.uint8)
    ref = np.empty(n, np.float32)
    lib.orc_parse_tokens(blob, n, stride, out.ctypes.data, st.ctypes.data, ref.ctypes.data)
    return out, st, ref


def max_segmentation_text(marginals_text):
    """The reference's post-processing tool restated (reference src/tools/maxSegmentation.cpp:53-82): per line the
    arg-max column (first maximum, strict `>` from 0), runs of equal arg-max merged, running state starting at 0."""
    out = []
    total, prev, best_i = 0, 0, 0
    for line in marginals_text.splitlines():
        f = line.split()
        rle = int(f[0]) if f else 0
        best, best_i = 0, 0
        for i, c in enumerate(f[1:]):
            if int(c) > best:
                best, best_i = int(c), i
        if best_i == prev:
            total += rle
        else:
            out.append("%d\t%d\n" % (total, prev))
            total, prev = rle, best_i
    out.append("%d\t%d\n" % (total, best_i))
    return "".join(out)
