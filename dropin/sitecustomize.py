"""Picked up by the interpreter at start-up when <repo>/dropin is on PYTHONPATH: with DWG_BIND=1 it installs the post-import hooks of
dwg_bind, so that `python main.py ...` of the UNEDITED reference runs its hot path on the HIP kernels.  Without the
variable this file does nothing of its own -- and in either case it then hands over to the NEXT `sitecustomize` on sys.path (a virtual
environment's, the distribution's, a coverage hook's), which this directory would otherwise shadow.  `import dwg_bind; dwg_bind.install()`
from the program itself is the equivalent that needs no start-up file."""
import os
import sys


def _chain():
    """Run the sitecustomize module this one shadows, if any (first match on sys.path outside this directory)."""
    import importlib.machinery
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    for d in sys.path:
        if not d or os.path.abspath(d) == here:
            continue
        try:
            spec = importlib.machinery.PathFinder.find_spec("sitecustomize", [d])
        except Exception:   # noqa: BLE001
            spec = None
        if spec is not None and spec.origin and os.path.abspath(spec.origin) != os.path.abspath(__file__):
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            return


if os.environ.get("DWG_BIND") == "1":
    try:
        import dwg_bind
        dwg_bind.install()
    except Exception as e:      # noqa: BLE001  (never break interpreter start-up; the bind is reported, loudly, once)
        sys.stderr.write("dwg_bind: NOT installed (%s: %s)\n" % (type(e).__name__, e))
try:
    _chain()
except Exception as e:          # noqa: BLE001
    sys.stderr.write("dropin/sitecustomize: the shadowed sitecustomize failed (%s: %s)\n" % (type(e).__name__, e))
