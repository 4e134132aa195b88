"""Recipe: build the REFERENCE's own grid-encoder backend for gfx950 into oracle/_ref/ (checker for SURVEY row L10 / boundary B2).

TEST INFRASTRUCTURE.  Sources are compiled from where they lie, /root/reference/core/nerf/gridencoder/src/{gridencoder.cu, bindings.cpp,
gridencoder.h} -- nothing of them is copied into the repository.  The reference builds this extension with torch.utils.cpp_extension
(core/nerf/gridencoder/backend.py:22-31); on ROCm that toolchain translates the .cu through torch's own hipify step before hipcc sees
it, IN PLACE next to the source.  This recipe therefore runs the reference's own `load()` call on a scratch copy of the three files under
oracle/_ref/_gridenc_build/ (deleted afterwards): cpp_extension.load(PYTORCH_ROCM_ARCH=gfx950) -> oracle/_ref/_gridencoder_ref.so.
No stand-in headers, no edits to the sources or to their translation.

    python oracle/build_ref_gridencoder.py          (needs /root/reference: runs in the build container; the GPU box uses the prebuilt .so)

oracle/_ref/ is git-ignored (it still travels to the GPU box with gpurun).  tests/test_gridencoder_ref_gpu.py loads the module and
compares csrc/gridenc.hip with it -- IF it exists.

STATUS on this image (ROCm 7.2.0, torch 2.10.0+rocm7.0): the build FAILS at the device compile of the translated gridencoder.cu, so
oracle/_ref/_gridencoder_ref.so is NOT produced and row L10 stays pinned only by the build's own restatement:
  * gridencoder.cu:328 `atomicAdd((__half2*)&grad_grid[index + c], v)` -- HIP has no atomicAdd overload for __half2 (only
    unsafeAtomicAdd, hip/amd_detail/amd_hip_fp16.h:882); the half2 branch sits behind a RUN-time `if (std::is_same<scalar_t, at::Half>...)`,
    so it is compiled for every scalar type, float included;
  * gridencoder.cu:327 `(__half)(w * grad_cur[c])` -- needs the half conversions the reference re-enables for nvcc with
    -U__CUDA_NO_HALF_CONVERSIONS__ (backend.py:9-12); the HIP spelling of those flags is passed below, which cures this one only.
Making the first one compile would need a hand-written atomicAdd(__half2*, __half2) -- a stand-in for something the toolchain lacks, which
this recipe does not do.  The failure (first compiler errors) is recorded in oracle/_ref/gridencoder_ref.unbuildable.txt and the
attempt is not repeated until the sources or this recipe change."""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/core/nerf/gridencoder/src"
OUT = os.path.join(HERE, "_ref")
SO = os.path.join(OUT, "_gridencoder_ref.so")
NAME = "_gridencoder_ref"
FAILED = os.path.join(OUT, "gridencoder_ref.unbuildable.txt")


def build(force=False, verbose=False):
    if not os.path.isdir(REF_SRC):
        if os.path.exists(SO):
            print("[oracle/_ref] %s not present (GPU box): using the prebuilt %s" % (REF_SRC, SO))
            return SO
        print("[oracle/_ref] %s not present (GPU box) and no prebuilt %s (the build container could not produce one: see this file's header)"
              % (REF_SRC, os.path.basename(SO)))
        return None
    srcs = [os.path.join(REF_SRC, f) for f in ("gridencoder.cu", "bindings.cpp", "gridencoder.h")]
    newest = max(os.path.getmtime(x) for x in srcs + [os.path.abspath(__file__)])
    if not force and os.path.exists(SO) and os.path.getmtime(SO) >= newest:
        return SO
    if not force and os.path.exists(FAILED) and os.path.getmtime(FAILED) >= newest:
        print("[oracle/_ref]", open(FAILED).readline().strip())
        return None
    os.makedirs(OUT, exist_ok=True)
    os.environ["PYTORCH_ROCM_ARCH"] = "gfx950"
    os.environ.setdefault("MAX_JOBS", "4")
    from torch.utils import cpp_extension
    tmp = os.path.join(OUT, "_gridenc_build")
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp)
    # The reference's own build call (backend.py:22-31: cpp_extension.load on gridencoder.cu + bindings.cpp with -O3 -std=c++17) on a scratch
    # copy of the three files under oracle/_ref/ -- on ROCm `load()` translates the .cu with torch's hipify IN PLACE next to the source,
    # which must not happen inside /root/reference.  The scratch copy is deleted again below.
    for f in srcs:
        shutil.copyfile(f, os.path.join(tmp, os.path.basename(f)))
    bdir = os.path.join(tmp, "ninja")
    os.makedirs(bdir, exist_ok=True)
    # the reference's flags (backend.py:7-16); -U__CUDA_NO_HALF_* spelled for HIP (torch's ROCm build defines the __HIP_NO_HALF_* twins)
    hip_flags = ["-O3", "-std=c++17", "-U__HIP_NO_HALF_OPERATORS__", "-U__HIP_NO_HALF_CONVERSIONS__"]
    try:
        cpp_extension.load(name=NAME, sources=[os.path.join(tmp, "gridencoder.cu"), os.path.join(tmp, "bindings.cpp")], build_directory=bdir,
                           extra_cflags=["-O3", "-std=c++17"], extra_cuda_cflags=hip_flags, verbose=False, is_python_module=False)
    except Exception as e:      # noqa: BLE001  (RuntimeError from ninja; the compiler output is in its message)
        errs, seen = [], set()
        for l in str(e).splitlines():               # ninja's captured compiler output: the distinct errors, once each
            if " error: " in l and l.split(" error: ")[1] not in seen:
                seen.add(l.split(" error: ")[1]); errs.append(l.replace(tmp + os.sep, ""))
        errs = errs[:6]
        with open(FAILED, "w") as f:
            f.write("reference grid-encoder backend: UNBUILDABLE on this image without stand-ins (see oracle/build_ref_gridencoder.py header)\n")
            f.write("sources mtime %.0f recipe mtime %.0f\n" % (max(os.path.getmtime(x) for x in srcs), os.path.getmtime(os.path.abspath(__file__))))
            f.write("\n".join(errs) + "\n")
        shutil.rmtree(tmp, ignore_errors=True)
        print("[oracle/_ref] reference grid encoder does not build here:", errs[0] if errs else str(e)[-300:])
        return None
    built = os.path.join(bdir, NAME + ".so")
    assert os.path.exists(built), os.listdir(bdir)
    shutil.copy2(built, SO)
    shutil.rmtree(tmp, ignore_errors=True)      # only the shared object stays (and travels)
    if os.path.exists(FAILED):
        os.unlink(FAILED)
    print("[oracle/_ref] built", SO)
    return SO


def load():
    """The reference backend as a Python module exposing grid_encode_forward / grid_encode_backward (bindings.cpp:5-9); None if the
    shared object was never built (the caller skips / fails as it sees fit)."""
    if not os.path.exists(SO):
        return None
    import importlib.util
    import torch  # noqa: F401  (the extension links against libtorch)
    spec = importlib.util.spec_from_file_location(NAME, SO)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
