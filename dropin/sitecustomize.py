"""Picked up by the interpreter at start-up when <repo>/dropin is on PYTHONPATH: with DWG_BIND=1 it installs the post-import hooks of
dwg_bind, so that `python main.py ...` of the UNEDITED reference runs its hot path on the HIP kernels.  Without the
variable this file does nothing."""
import os

if os.environ.get("DWG_BIND") == "1":
    try:
        import dwg_bind
        dwg_bind.install()
    except Exception as e:      # noqa: BLE001  (never break interpreter start-up; the bind is reported, loudly, once)
        import sys
        sys.stderr.write("dwg_bind: NOT installed (%s: %s)\n" % (type(e).__name__, e))
