"""ctypes binding of include/hml.h (libhammlet_hip.so) - the Python-side mirror of the C ABI.

There is no CPU fallback: constructing a Chain without the compiled gfx950 library or without a
GPU raises.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_lib = None


class HmlError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class HmlStats(C.Structure):
    _fields_ = [("sweeps", C.c_uint64), ("block_updates", C.c_uint64), ("uniform_fallbacks", C.c_uint64),
                ("forward_refits", C.c_uint64), ("forward_serial", C.c_uint64), ("forward_warmup", C.c_uint64),
                ("fused_fallbacks", C.c_uint64), ("buffer_growths", C.c_uint64), ("block_capacity", C.c_uint64)]


RECORD_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_void_p)

# name -> (restype, argtypes); every symbol include/hml.h declares
_P = C.c_void_p
SIGNATURES = {
    "hml_last_error": (C.c_char_p, []),
    "hml_abi_version": (C.c_uint32, []),
    "hml_device_arch": (C.c_char_p, []),
    "hml_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "hml_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_uint64, C.c_uint32, _P]),
    "hml_destroy": (None, [_P]),
    "hml_load_observations": (C.c_int, [_P, _P, C.c_uint64]),
    "hml_load_observations_device": (C.c_int, [_P, _P, C.c_uint64]),
    "hml_attach_observations": (C.c_int, [_P, _P]),
    "hml_text_open": (C.c_int, [C.POINTER(_P), C.c_int, C.c_uint64]),
    "hml_text_close": (None, [_P]),
    "hml_text_buffer": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "hml_text_commit": (C.c_int, [_P, C.c_uint64]),
    "hml_text_feed": (C.c_int, [_P, C.c_char_p, C.c_uint64]),
    "hml_text_reserve": (C.c_int, [_P, C.c_uint64]),
    "hml_text_finish": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "hml_text_values": (C.c_int, [_P, _P]),
    "hml_text_counters": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "hml_set_dimensions": (C.c_int, [_P, C.c_int, C.c_int]),
    "hml_get_dimensions": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "hml_set_weights": (C.c_int, [_P, _P, C.c_uint64]),
    "hml_noise_sigma": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "hml_scale_weights": (C.c_int, [_P, C.c_float]),
    "hml_autoprior": (C.c_int, [_P, C.c_float, C.c_float, _P]),
    "hml_set_model": (C.c_int, [_P, C.c_int, _P, C.c_float, C.c_float, C.c_float, C.c_int]),
    "hml_set_self_transitions": (C.c_int, [_P, C.c_int]),
    "hml_sample_prior": (C.c_int, [_P]),
    "hml_set_static_blocks": (C.c_int, [_P]),
    "hml_set_dynamic": (C.c_int, [_P, C.c_int]),
    "hml_create_blocks": (C.c_int, [_P, C.c_float]),
    "hml_iterate": (C.c_int, [_P, C.c_char, C.c_uint64, C.c_uint64]),
    "hml_iterate_many": (C.c_int, [_P, C.c_int, C.c_char, C.c_uint64, C.c_uint64]),
    "hml_set_recording": (C.c_int, [_P, C.c_int, RECORD_CB, _P]),
    "hml_set_option": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "hml_sync": (C.c_int, [_P]),
    "hml_get_num_blocks": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "hml_get_blocks": (C.c_int, [_P, _P]),
    "hml_get_block_stats": (C.c_int, [_P, _P, _P]),
    "hml_get_states": (C.c_int, [_P, _P]),
    "hml_get_theta": (C.c_int, [_P, _P]),
    "hml_get_transitions": (C.c_int, [_P, _P, _P]),
    "hml_set_parameters": (C.c_int, [_P, _P, _P, _P]),
    "hml_get_threshold": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "hml_enable_probes": (C.c_int, [_P, C.c_int]),
    "hml_get_block_loglik": (C.c_int, [_P, _P]),
    "hml_get_forward_rows": (C.c_int, [_P, _P]),
    "hml_get_counts": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "hml_get_weights": (C.c_int, [_P, _P]),
    "hml_get_coefficients": (C.c_int, [_P, _P]),
    "hml_get_integral_array": (C.c_int, [_P, _P, _P]),
    "hml_marginals_rle": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_int), _P, _P]),
    "hml_max_segmentation": (C.c_int, [_P, C.POINTER(C.c_uint64), _P, _P]),
    "hml_marginals_dense_device": (C.c_int, [_P, _P, _P]),
    "hml_recorded_sweeps": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "hml_categorical_draw": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_uint32)]),
    "hml_relabel_permutation": (C.c_int, [_P, _P]),
    "hml_pool_payload_size": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "hml_pool_export": (C.c_int, [_P, _P, _P]),
    "hml_pool_install": (C.c_int, [_P, _P]),
    "hml_pool_unique_id": (C.c_int, [_P]),
    "hml_pool_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.c_int, _P]),
    "hml_pool_destroy": (None, [_P]),
    "hml_pool_marginals": (C.c_int, [_P, _P, _P]),
    "hml_pool_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "hml_pool_set_form": (C.c_int, [_P, C.c_int]),
    "hml_pool_last": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "hml_allreduce_marginals": (C.c_int, [_P, C.c_int]),
    "hml_allreduce_marginals_perm": (C.c_int, [_P, C.c_int, _P]),
    "hml_pool_permutation": (C.c_int, [_P, _P]),
    "hml_get_stats": (C.c_int, [_P, C.POINTER(HmlStats)]),
    "hml_profile_enable": (C.c_int, [_P, C.c_int]),
    "hml_profile_get": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "hml_debug_eval": (C.c_int, [C.c_int, C.c_int, _P, _P, _P, C.c_uint64, C.c_uint64]),
    "hml_synth_depth": (C.c_int, [_P, _P, C.c_uint64, C.c_double, C.c_double, C.c_uint64, C.c_int]),
    "hml_synth_gauss": (C.c_int, [_P, _P, C.c_uint64, C.c_int, _P, C.c_float, C.c_double, C.c_uint64, C.c_int]),
}


ABI_VERSION = 3   # hml_abi_version() of include/hml.h this mirror was written against


def load_library(path=None):
    """dlopen libhammlet_hip.so and attach the signatures of include/hml.h.  Fails loudly."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("HML_LIBRARY") or _build.LIB_PATH   # HML_LIBRARY: A/B timing of two builds
    if not os.path.exists(path):
        raise HmlError(-1, "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.hml_abi_version() != ABI_VERSION:
        raise HmlError(-1, "%s has ABI version %d, this package expects %d: rebuild the library" % (path, lib.hml_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise HmlError(rc, load_library().hml_last_error().decode())


def parse_text(source, device=0, chunk_bytes=0, feed_bytes=None, with_info=False):
    """The float32 values that the reference's reader (`while ( input >> v )`, reference src/wavelet.hpp:131)
    extracts from whitespace-separated decimal text, converted on the GPU.  `source`: bytes or a file path.
    Returns the array (and, with_info, a dict: stopped, bytes, irregular_tokens, host_chunks)."""
    lib = load_library()
    h = _P()
    _check(lib.hml_text_open(C.byref(h), device, chunk_bytes))
    try:
        if isinstance(source, (bytes, bytearray, memoryview)):
            data = bytes(source)
            step = feed_bytes or max(len(data), 1)
            for i in range(0, len(data), step):
                part = data[i:i + step]
                _check(lib.hml_text_feed(h, part, len(part)))
        else:
            with open(source, "rb", buffering=0) as f:
                while True:
                    buf, cap = _P(), C.c_uint64()
                    _check(lib.hml_text_buffer(h, C.byref(buf), C.byref(cap)))
                    view = (C.c_char * cap.value).from_address(buf.value)
                    n = f.readinto(view)
                    if not n:
                        break
                    _check(lib.hml_text_commit(h, n))
        n, stopped = C.c_uint64(), C.c_int()
        _check(lib.hml_text_finish(h, C.byref(n), C.byref(stopped)))
        out = np.empty(n.value, np.float32)
        _check(lib.hml_text_values(h, out.ctypes.data))
        if not with_info:
            return out
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib.hml_text_counters(h, C.byref(a), C.byref(b), C.byref(c)))
        return out, {"stopped": bool(stopped.value), "bytes": a.value, "irregular_tokens": b.value, "host_chunks": c.value}
    finally:
        lib.hml_text_close(h)


def synth_gauss(T, K, mu, sigma, dwell, seed, nthreads=8, with_states=False):
    lib = load_library()
    x = np.empty(T, np.float32)
    mu = np.ascontiguousarray(mu, np.float32)
    st = np.empty(T, np.int16) if with_states else None
    _check(lib.hml_synth_gauss(x.ctypes.data, st.ctypes.data if with_states else None, T, K, mu.ctypes.data, sigma,
                               dwell, seed, nthreads))
    return (x, st) if with_states else x


def synth_depth(T, depth=15.0, ln_sigma=0.15, seed=5, nthreads=8, with_states=False):
    lib = load_library()
    x = np.empty(T, np.float32)
    st = np.empty(T, np.int16) if with_states else None
    _check(lib.hml_synth_depth(x.ctypes.data, st.ctypes.data if with_states else None, T, depth, ln_sigma, seed, nthreads))
    return (x, st) if with_states else x


def debug_eval(fn, a, b=None, seed=0, device=0):
    lib = load_library()
    a = np.ascontiguousarray(a, np.float32)
    out = np.empty_like(a)
    if b is not None:
        b = np.ascontiguousarray(b, np.float32)
    _check(lib.hml_debug_eval(device, fn, a.ctypes.data, b.ctypes.data if b is not None else None, out.ctypes.data, a.size, seed))
    return out


class Chain:
    """One Gibbs chain on one GPU: the Python mirror of hml_ctx (see include/hml.h for the reference
    interfaces each call stands for)."""

    def __init__(self, device=0, seed=0, chain_id=0, stream=None):
        self.lib = load_library()
        h = _P()
        _check(self.lib.hml_create(C.byref(h), device, seed, chain_id, stream))
        self.h = h
        self.K = None
        self.T = None
        self.D = 1          # data dimensions / emission parameters ("-s C P D"); P = None: P = K
        self.P = None
        self._cb = None

    def close(self):
        if self.h:
            self.lib.hml_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- construction -------------------------------------------------------------------
    def set_dimensions(self, D, P):
        """D data dimensions (their values follow each other in x), P emission parameters shared by P**D states"""
        _check(self.lib.hml_set_dimensions(self.h, D, P))
        self.D, self.P = D, P

    def load(self, x):
        x = np.ascontiguousarray(x, np.float32)
        self.T = int(x.size) // self.D
        _check(self.lib.hml_load_observations(self.h, x.ctypes.data, x.size))

    def load_device(self, ptr, T):
        self.T = int(T)
        _check(self.lib.hml_load_observations_device(self.h, ptr, T))

    def attach(self, source):
        """hml_attach_observations: share `source`'s construction (same device) instead of loading a copy"""
        _check(self.lib.hml_attach_observations(self.h, source.h))
        self.T, self.D, self.P = source.T, source.D, source.P

    def noise_sigma(self):
        v = C.c_double()
        _check(self.lib.hml_noise_sigma(self.h, C.byref(v)))
        return v.value

    def scale_weights(self, m):
        _check(self.lib.hml_scale_weights(self.h, m))

    def set_weights(self, w):
        w = np.ascontiguousarray(w, np.float32)
        _check(self.lib.hml_set_weights(self.h, w.ctypes.data, w.size))

    def autoprior(self, s2=0.2, p=0.9):
        out = np.empty(4, np.float32)
        _check(self.lib.hml_autoprior(self.h, s2, p, out.ctypes.data))
        return out

    def set_model(self, K, nig4, a_off=0.5, a_diag=0.5, pi_alpha=0.5, self_trans=True):
        nig4 = np.ascontiguousarray(nig4, np.float32)
        self.K = K
        _check(self.lib.hml_set_model(self.h, K, nig4.ctypes.data, a_off, a_diag, pi_alpha, 1 if self_trans else 0))

    def sample_prior(self):
        _check(self.lib.hml_sample_prior(self.h))

    def set_self_transitions(self, on=True):
        _check(self.lib.hml_set_self_transitions(self.h, 1 if on else 0))

    def set_static_blocks(self):
        _check(self.lib.hml_set_static_blocks(self.h))

    def set_dynamic(self, on=True):
        _check(self.lib.hml_set_dynamic(self.h, 1 if on else 0))

    def create_blocks(self, thr):
        _check(self.lib.hml_create_blocks(self.h, thr))

    def iterate(self, method, iterations, thinning=0):
        _check(self.lib.hml_iterate(self.h, method.encode()[0:1], iterations, thinning))

    def set_recording(self, marginals=True, callback=None):
        if callback is not None:
            def tramp(_ctx, sweep, _user):
                callback(self, sweep)
            self._cb = RECORD_CB(tramp)
        else:
            self._cb = C.cast(None, RECORD_CB)
        _check(self.lib.hml_set_recording(self.h, 1 if marginals else 0, self._cb, None))

    def set_option(self, name, value):
        _check(self.lib.hml_set_option(self.h, name.encode(), int(value)))

    def sync(self):
        _check(self.lib.hml_sync(self.h))

    # ---- probes -------------------------------------------------------------------------
    def num_blocks(self):
        v = C.c_uint64()
        _check(self.lib.hml_get_num_blocks(self.h, C.byref(v)))
        return v.value

    def blocks(self):
        B = self.num_blocks()
        s = np.empty(B + 1, np.uint32)
        _check(self.lib.hml_get_blocks(self.h, s.ctypes.data))
        return s

    def block_stats(self):
        """(sum x, sum x^2) of every block; with D > 1 data dimensions arrays of shape [D][B]"""
        B = self.num_blocks()
        a = np.empty((self.D, B), np.float32)
        b = np.empty((self.D, B), np.float32)
        _check(self.lib.hml_get_block_stats(self.h, a.ctypes.data, b.ctypes.data))
        return (a[0], b[0]) if self.D == 1 else (a, b)

    def states(self):
        B = self.num_blocks()
        q = np.empty(B, np.int16)
        _check(self.lib.hml_get_states(self.h, q.ctypes.data))
        return q

    def theta(self):
        t = np.empty(2 * (self.P or self.K), np.float32)
        _check(self.lib.hml_get_theta(self.h, t.ctypes.data))
        return t

    def transitions(self):
        A = np.empty((self.K, self.K), np.float32)
        pi = np.empty(self.K, np.float32)
        _check(self.lib.hml_get_transitions(self.h, A.ctypes.data, pi.ctypes.data))
        return A, pi

    def set_parameters(self, mean_var, A, pi):
        mv = np.ascontiguousarray(mean_var, np.float32)
        A = np.ascontiguousarray(A, np.float32)
        pi = np.ascontiguousarray(pi, np.float32)
        _check(self.lib.hml_set_parameters(self.h, mv.ctypes.data, A.ctypes.data, pi.ctypes.data))

    def threshold(self):
        v = C.c_float()
        _check(self.lib.hml_get_threshold(self.h, C.byref(v)))
        return v.value

    def enable_probes(self, on=True):
        _check(self.lib.hml_enable_probes(self.h, 1 if on else 0))

    def block_loglik(self):
        B = self.num_blocks()
        E = np.empty((B, self.K), np.float32)
        _check(self.lib.hml_get_block_loglik(self.h, E.ctypes.data))
        return E

    def forward_rows(self):
        B = self.num_blocks()
        r = np.empty((B + 1, self.K), np.float32)
        _check(self.lib.hml_get_forward_rows(self.h, r.ctypes.data))
        return r

    def counts(self):
        K = self.K
        trans = np.empty((K, K), np.uint64)
        occ = np.empty(K, np.uint64)
        s = np.empty(K, np.float32)
        q = np.empty(K, np.float32)
        n = np.empty(K, np.uint64)
        _check(self.lib.hml_get_counts(self.h, trans.ctypes.data, occ.ctypes.data, s.ctypes.data, q.ctypes.data, n.ctypes.data))
        return trans, occ, s, q, n

    def weights(self):
        w = np.empty(self.T, np.float32)
        _check(self.lib.hml_get_weights(self.h, w.ctypes.data))
        return w

    def coefficients(self):
        w = np.empty(self.T, np.float32)
        _check(self.lib.hml_get_coefficients(self.h, w.ctypes.data))
        return w

    def integral_array(self):
        a = np.empty(self.T + 1, np.float32)
        b = np.empty(self.T + 1, np.float32)
        _check(self.lib.hml_get_integral_array(self.h, a.ctypes.data, b.ctypes.data))
        return a, b

    # ---- results ------------------------------------------------------------------------
    def recorded_sweeps(self):
        v = C.c_uint64()
        _check(self.lib.hml_recorded_sweeps(self.h, C.byref(v)))
        return v.value

    def marginals_rle(self):
        n = C.c_uint64()
        k = C.c_int()
        _check(self.lib.hml_marginals_rle(self.h, C.byref(n), C.byref(k), None, None))
        seg = np.empty(n.value, np.uint64)
        cnt = np.empty((n.value, max(k.value, 0)), np.int32)
        _check(self.lib.hml_marginals_rle(self.h, C.byref(n), C.byref(k), seg.ctypes.data, cnt.ctypes.data if k.value else None))
        return seg, cnt

    def max_segmentation(self):
        """(run lengths, arg-max state per run) of the recorded marginals, merged (maxSegmentation.cpp:53-82)"""
        n = C.c_uint64()
        _check(self.lib.hml_max_segmentation(self.h, C.byref(n), None, None))
        ln = np.empty(n.value, np.uint64)
        st = np.empty(n.value, np.int32)
        _check(self.lib.hml_max_segmentation(self.h, C.byref(n), ln.ctypes.data, st.ctypes.data))
        return ln, st

    def marginals_dense_device(self, out_ptr, perm=None):
        p = None
        if perm is not None:
            perm = np.ascontiguousarray(perm, np.int32)
            p = perm.ctypes.data
        _check(self.lib.hml_marginals_dense_device(self.h, out_ptr, p))

    # ---- chain-parallel pooling ---------------------------------------------------------
    def relabel_permutation(self):
        perm = np.empty(self.K, np.int32)
        _check(self.lib.hml_relabel_permutation(self.h, perm.ctypes.data))
        return perm

    def pool_payload_size(self):
        n = C.c_uint64()
        _check(self.lib.hml_pool_payload_size(self.h, C.byref(n)))
        return n.value

    def pool_export(self, payload_ptr):
        perm = np.empty(self.K, np.int32)
        _check(self.lib.hml_pool_export(self.h, payload_ptr, perm.ctypes.data))
        return perm

    def pool_permutation(self):
        """perm[j] = this chain's label of pooled state j (identity before any pooling)"""
        perm = np.zeros(self.K, np.int32)
        _check(self.lib.hml_pool_permutation(self.h, perm.ctypes.data))
        return perm

    def pool_install(self, payload_ptr):
        _check(self.lib.hml_pool_install(self.h, payload_ptr))

    def categorical_draw(self, weights):
        w = np.ascontiguousarray(weights, np.float32)
        out = C.c_uint32()
        _check(self.lib.hml_categorical_draw(self.h, w.ctypes.data, w.size, C.byref(out)))
        return out.value

    def stats(self):
        s = HmlStats()
        _check(self.lib.hml_get_stats(self.h, C.byref(s)))
        return {f: getattr(s, f) for f, _ in HmlStats._fields_}

    def profile_enable(self, level=2):
        _check(self.lib.hml_profile_enable(self.h, int(level)))

    def profile_get(self, name):
        ms = C.c_double()
        n = C.c_uint64()
        _check(self.lib.hml_profile_get(self.h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


def _prefer_torch_rccl():
    """A Python process that also runs PyTorch must not hold two RCCL libraries (torch bundles one; the second copy ends the
    process in `double free or corruption` at exit): point the library's dlopen at torch's copy before its first pooling
    call, without importing torch.  No torch, or HML_RCCL_LIBRARY set by the user: nothing happens."""
    if os.environ.get("HML_RCCL_LIBRARY"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so") if spec and spec.origin else None
        if cand and os.path.exists(cand):
            os.environ["HML_RCCL_LIBRARY"] = cand
            # ... and let torch bring up its runtime FIRST: an RCCL communicator created before `import torch` has been seen
            # to end the process in the same way when it exits (the order of the two libraries' exit handlers)
            import torch  # noqa: F401
    except Exception:
        pass


class Pool:
    """RCCL communicator of the chain-parallel pooling (hml_pool_* of include/hml.h): one rank per process and GPU."""

    def __init__(self, device, rank, n_ranks, unique_id):
        _prefer_torch_rccl()
        self.lib = load_library()
        h = _P()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _check(self.lib.hml_pool_create(C.byref(h), device, rank, n_ranks, C.cast(buf, _P)))
        self.h = h

    @staticmethod
    def unique_id():
        """ncclGetUniqueId: made by rank 0, handed to every rank by the launcher"""
        _prefer_torch_rccl()
        lib = load_library()
        buf = C.create_string_buffer(128)
        _check(lib.hml_pool_unique_id(C.cast(buf, _P)))
        return buf.raw

    def marginals(self, chain):
        """export + ncclAllReduce(sum, int32) + install: afterwards `chain` holds the pooled marginals"""
        perm = np.empty(chain.K, np.int32)
        _check(self.lib.hml_pool_marginals(self.h, chain.h, perm.ctypes.data))
        return perm

    def info(self):
        r, n, ms, b, v = C.c_int(), C.c_int(), C.c_double(), C.c_uint64(), C.c_int()
        _check(self.lib.hml_pool_info(self.h, C.byref(r), C.byref(n), C.byref(ms), C.byref(b), C.byref(v)))
        return {"rank": r.value, "n_ranks": n.value, "last_allreduce_ms": ms.value, "last_bytes": b.value, "rccl_version": v.value}

    def set_form(self, form):
        """0: dense payload or boundary lists, whichever is smaller (default); 1: dense (ncclAllReduce); 2: lists (ncclAllGather)"""
        _check(self.lib.hml_pool_set_form(self.h, form))

    def last(self):
        f, e = C.c_int(), C.c_uint64()
        _check(self.lib.hml_pool_last(self.h, C.byref(f), C.byref(e)))
        return {"form": {1: "dense", 2: "lists"}.get(f.value, "none"), "entries": e.value}

    def close(self):
        if self.h:
            self.lib.hml_pool_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def allreduce_marginals(chains, with_perms=False):
    """hml_allreduce_marginals(_perm): one process driving several chains (possibly on several GPUs).  with_perms: returns
    perms[i][j] = chain i's own label of pooled state j."""
    _prefer_torch_rccl()
    lib = load_library()
    arr = (_P * len(chains))(*[c.h for c in chains])
    if not with_perms:
        _check(lib.hml_allreduce_marginals(C.cast(arr, _P), len(chains)))
        return None
    perms = np.zeros((len(chains), chains[0].K), np.int32)
    _check(lib.hml_allreduce_marginals_perm(C.cast(arr, _P), len(chains), perms.ctypes.data))
    return perms


def iterate_many(chains, method, iterations, thinning=0):
    """hml_iterate_many: `iterations` sweeps of every chain; chains of one GPU and one shape share their launches"""
    lib = load_library()
    arr = (_P * len(chains))(*[c.h for c in chains])
    _check(lib.hml_iterate_many(C.cast(arr, _P), len(chains), method.encode(), iterations, thinning))


def marginals_text(seg, cnt):
    """StateMarginals::save format (reference src/StateMarginals.hpp:268-310)."""
    lines = []
    for i in range(len(seg)):
        lines.append("\t".join([str(int(seg[i]))] + [str(int(v)) for v in cnt[i]]))
    return "\n".join(lines) + "\n"
