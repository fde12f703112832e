"""Build helpers: compile the gfx950 shared library and the command-line driver."""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libhammlet_hip.so")
CLI_PATH = os.path.join(PKG_DIR, "hammlet")
TOOL_PATH = os.path.join(PKG_DIR, "maxSegmentation")
SORT_TOOL_PATH = os.path.join(PKG_DIR, "sortStates")
AVG_TOOL_PATH = os.path.join(PKG_DIR, "avg")
GENOME_TOOLS = ("mapLinesToGenome", "combineCounts")   # host tools over zlib (csrc/host/*_main.cpp)

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
            if os.path.basename(dp) == "build":
                continue
            out += [os.path.join(dp, f) for f in fns if f.endswith((".h", ".hpp", ".hip", ".cpp"))]
    return out


OBJ_DIR = os.path.join(CSRC, "build")


def _deps(depfile, fallback):
    """prerequisites recorded by `hipcc -MMD` for one object (all sources if there is no record yet)"""
    try:
        with open(depfile) as f:
            text = f.read().replace("\\\n", " ")
        deps = [d for d in text.split(":", 1)[1].split() if os.path.exists(d)]
        return deps or fallback
    except (OSError, IndexError):
        return fallback


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE if not verbose else None, text=True)
    if r.returncode != 0:
        errs = [l for l in (r.stderr or "").splitlines() if "error" in l or "note:" in l]
        raise RuntimeError("hipcc failed:\n" + "\n".join(errs[:40]))


K_STATES = range(2, 17)   # numbers of states the kernels are compiled for (HML_MAX_K = 16)


def _objects():
    """(source, object name, extra flags) of every object of the library: each csrc/*.hip once - except hml_sweep.hip, the sweep
    for K states, which is compiled once per number of states (-DHML_TU_K=k): fifteen objects with their own code objects,
    built in parallel and loaded on demand (csrc/hml_capi_shared.hpp, top)."""
    out = []
    for u in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
        if u == "hml_sweep.hip":
            out += [(u, "hml_sweep_k%d.o" % k, ["-DHML_TU_K=%d" % k]) for k in K_STATES]
        else:
            out.append((u, u[:-4] + ".o", []))
    return out


def build_library(force=False, verbose=False, jobs=None):
    """hammlet_amd/csrc/*.hip -> objects (hipcc --offload-arch=gfx950 -c, in parallel) -> hammlet_amd/libhammlet_hip.so"""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo, objs = [], []
    fallback = _sources()
    for u, o, extra in _objects():
        src, obj = os.path.join(CSRC, u), os.path.join(OBJ_DIR, o)
        dep = obj[:-2] + ".d"
        objs.append(obj)
        if force or _newer(obj, _deps(dep, fallback)):
            todo.append([_hipcc()] + HIPCC_FLAGS + extra + ["-MMD", "-MF", dep, "-c", "-o", obj, src])
    relink = force or bool(todo) or not os.path.exists(LIB_PATH)
    if todo:
        # (each hipcc is single-threaded; memory: ~1.5 GB per per-K object)
        with ThreadPoolExecutor(max_workers=jobs or max(1, min(8, os.cpu_count() or 1))) as ex:
            list(ex.map(lambda cmd: _run(cmd, verbose), todo))
    # objects of an earlier layout (one object for all numbers of states) must not be linked in
    if relink or _newer(LIB_PATH, objs):
        _run([_hipcc()] + HIPCC_FLAGS + ["-shared", "-o", LIB_PATH] + objs, verbose)
    return LIB_PATH


def build_cli(force=False, verbose=False):
    """The `hammlet` command-line driver (host C++ over the C ABI)."""
    src = os.path.join(CSRC, "host", "hammlet_main.cpp")
    if not os.path.exists(src):
        return None
    if force or _newer(CLI_PATH, _sources()):
        build_library(force=False, verbose=verbose)
        cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", "-o", CLI_PATH, src, "-I", os.path.join(REPO_DIR, "include"),
               "-L", PKG_DIR, "-lhammlet_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    # post-processing tool (file in, text out; no GPU)
    tsrc = os.path.join(CSRC, "host", "maxSegmentation_main.cpp")
    if force or _newer(TOOL_PATH, [tsrc, os.path.join(REPO_DIR, "include", "hammlet", "Parser.hpp")]):
        cmd = ["g++", "-O2", "-std=c++17", "-o", TOOL_PATH, tsrc, "-I", os.path.join(REPO_DIR, "include")]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    asrc = os.path.join(CSRC, "host", "avg_main.cpp")
    if force or _newer(AVG_TOOL_PATH, [asrc, LIB_PATH]):
        cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", AVG_TOOL_PATH, asrc, "-I", os.path.join(REPO_DIR, "include"),
               "-L", PKG_DIR, "-lhammlet_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    for tool in GENOME_TOOLS:
        gsrc = os.path.join(CSRC, "host", tool + "_main.cpp")
        gout = os.path.join(PKG_DIR, tool)
        if force or _newer(gout, [gsrc, os.path.join(CSRC, "host", "gz_lines.hpp"), os.path.join(REPO_DIR, "include", "hammlet", "Parser.hpp")]):
            cmd = ["g++", "-O2", "-std=c++17", "-o", gout, gsrc, "-I", os.path.join(REPO_DIR, "include"), "-lz"]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
    ssrc = os.path.join(CSRC, "host", "sortStates_main.cpp")
    if force or _newer(SORT_TOOL_PATH, [ssrc]):
        cmd = ["g++", "-O2", "-std=c++17", "-o", SORT_TOOL_PATH, ssrc]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return CLI_PATH
