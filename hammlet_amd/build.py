"""Build helpers: compile the gfx950 shared library and the command-line driver."""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libhammlet_hip.so")
CLI_PATH = os.path.join(PKG_DIR, "hammlet")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X kernels cannot be built")
    return exe


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _sources():
    out = []
    for root in (CSRC, os.path.join(REPO_DIR, "include")):
        for dp, _, fns in os.walk(root):
            out += [os.path.join(dp, f) for f in fns if f.endswith((".h", ".hpp", ".hip", ".cpp"))]
    return out


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -shared -o hammlet_amd/libhammlet_hip.so"""
    if force or _newer(LIB_PATH, _sources()):
        cmd = [_hipcc()] + HIPCC_FLAGS + ["-shared", "-o", LIB_PATH, os.path.join(CSRC, "hml_capi.hip")]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE if not verbose else None, text=True)
        if r.returncode != 0:
            errs = [l for l in (r.stderr or "").splitlines() if "error" in l or "note:" in l]
            raise RuntimeError("hipcc failed:\n" + "\n".join(errs[:40]))
    return LIB_PATH


def build_cli(force=False, verbose=False):
    """The `hammlet` command-line driver (host C++ over the C ABI)."""
    src = os.path.join(CSRC, "host", "hammlet_main.cpp")
    if not os.path.exists(src):
        return None
    if force or _newer(CLI_PATH, _sources()):
        build_library(force=False, verbose=verbose)
        cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", CLI_PATH, src, "-I", os.path.join(REPO_DIR, "include"),
               "-L", PKG_DIR, "-lhammlet_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return CLI_PATH
