"""Builds libdwg_hip.so (all HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

    python dreamwaltz-g_amd/build.py [--force]

Objects are cached by mtime under csrc/_obj/; the shared library lands next to the sources so that it
travels with the repo snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
import concurrent.futures
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(CSRC, "libdwg_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall",
         "-Wno-unused-function", "-Wno-unused-variable", "-Wno-misleading-indentation"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") or f.endswith(".cpp"))


def _headers_mtime():
    inc = os.path.join(HERE, "..", "include")
    m = 0.0
    for d in (CSRC, inc):
        for f in os.listdir(d):
            if f.endswith(".h"):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


def _src_mtime(sp):
    """mtime of a source and of the .hip files it re-includes (gemm_f16.hip / attention_f16.hip compile their bf16 twins again with
    DWG_*_F16_TU defined)."""
    m = os.path.getmtime(sp)
    for inc in re.findall(r'#include "([^"]+\.hip)"', open(sp).read()):
        m = max(m, os.path.getmtime(os.path.join(CSRC, inc)))
    return m


def _compile(src, force):
    obj = os.path.join(OBJ, src + ".o")
    sp = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= _src_mtime(sp)
            and os.path.getmtime(obj) >= _headers_mtime()):
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-x", "hip", "-c", sp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    _write_resources(src, r.stderr)
    return obj, True


_RES = re.compile(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|"
                  r"LDS Size \[bytes/block\]): +(\S+)")


def _write_resources(src, remarks):
    """Per-kernel registers / scratch / occupancy as the compiler reports them -> _obj/<src>.resources.json.  A kernel that
    silently starts spilling (or whose accumulators get demoted to scratch) shows up here and in tests/test_cabi.py."""
    kernels, cur = {}, None
    for line in remarks.splitlines():
        m = _RES.search(line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = kernels.setdefault(v, {})
        elif cur is not None:
            cur[k.split(" [")[0]] = int(v)
    with open(os.path.join(OBJ, src + ".resources.json"), "w") as f:
        json.dump(kernels, f, indent=1, sort_keys=True)


def resources():
    """{source: {mangled kernel: {VGPRs, ScratchSize, ...}}} of the last build."""
    out = {}
    for f in sorted(os.listdir(OBJ)) if os.path.isdir(OBJ) else []:
        if f.endswith(".resources.json"):
            out[f[:-len(".resources.json")]] = json.load(open(os.path.join(OBJ, f)))
    return out


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in res]
    changed = any(c for _, c in res)
    if changed or force or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[dwg build] linked", LIB)
    elif verbose:
        print("[dwg build] up to date:", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
