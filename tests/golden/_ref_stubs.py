"""Capture-time stand-ins that let the importable parts of /root/reference import in this container.
Used ONLY by tests/golden/capture_golden*.py, which run here (the reference is absent on the GPU box); never shipped as
product code and never imported by tests.

Two kinds, kept apart on purpose:
  * INERT packages (cv2, nvdiffrast, igl, open3d, trimesh, pyrallis, plyfile, human_body_prior, xformers, diffusers,
    jaxtyping, imageio, loguru, ...): any attribute is a do-nothing class.  They satisfy `import` statements and type
    annotations; no arithmetic of a captured function goes through them.
  * ARITHMETIC stand-ins for the two third-party libraries whose functions the captured code does call:
    smplx.lbs (4 functions) and pytorch3d.transforms (4 functions) -> oracle.animate restatements.  Fixtures captured through
    them are tagged "stub_dependent": they pin the reference's IN-REPO algebra, not those third-party functions.
"""
import importlib.abc
import importlib.machinery
import sys
import types

INERT_TOP = {"cv2", "nvdiffrast", "igl", "open3d", "trimesh", "pyrallis", "plyfile", "human_body_prior", "xformers", "diffusers",
             "jaxtyping", "imageio", "loguru", "kiui", "tensorboardX", "tensorboard", "mediapipe", "lpips", "clip", "pymeshlab",
             "xatlas", "skimage", "matplotlib", "diff_gaussian_rasterization", "torch_scatter", "smplx", "pytorch3d", "rembg",
             "torchmetrics", "wandb", "torchvision", "easydict", "omegaconf", "peft", "transformers", "huggingface_hub"}


class _InertMeta(type):
    def __getattr__(cls, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _make_inert(cls.__name__ + "." + k)

    def __getitem__(cls, k):
        return cls

    def __or__(cls, other):
        return cls

    def __ror__(cls, other):
        return cls


def _make_inert(name):
    def _init(self, *a, **k):
        pass

    def _getattr(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return lambda *a, **kw: None
    return _InertMeta(name, (), {"__init__": _init, "__getattr__": _getattr, "__call__": lambda self, *a, **k: None})


class _InertModule(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        v = _make_inert(self.__name__ + "." + k)
        setattr(self, k, v)
        return v


class _InertFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in INERT_TOP and fullname not in sys.modules:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _InertModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def install(oa):
    """oa = oracle.animate (the arithmetic stand-ins come from it).  Returns the fake SMPLX base class."""
    if not any(isinstance(f, _InertFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _InertFinder())
    import smplx  # noqa: F401  (inert)
    import smplx.lbs as lbs

    class SMPL:  # noqa
        pass

    class SMPLX(SMPL):  # noqa
        pass
    smplx.SMPL, smplx.SMPLX = SMPL, SMPLX
    lbs.blend_shapes = oa.blend_shapes
    lbs.vertices2joints = oa.vertices2joints
    lbs.batch_rodrigues = lambda r, dtype=None: oa.batch_rodrigues(r)
    lbs.batch_rigid_transform = lambda rm, j, parents, dtype=None: oa.batch_rigid_transform(rm, j, parents)
    import pytorch3d.transforms as p3t
    p3t.quaternion_to_matrix = oa.quaternion_to_matrix
    p3t.matrix_to_quaternion = oa.matrix_to_quaternion
    p3t.quaternion_multiply = oa.quaternion_multiply
    p3t.standardize_quaternion = oa.standardize_quaternion
    import loguru

    class _L:
        def __getattr__(self, k):
            return lambda *a, **k2: None
    loguru.logger = _L()
    return SMPLX
