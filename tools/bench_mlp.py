"""Times the two per-Gaussian MLP chains of the avatar (static 32-64-64-4, deformation 32(+63)-64-64-64-64-10) alone: forward under
inference_mode (the c5 / c1 frames) and forward + backward (the training step), at the row counts of c5 / c3 / c1.
usage: python tools/bench_mlp.py [rows ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dwg_import  # noqa: F401
from dreamwaltz_g_amd.mlp import MLP, DeformNetwork
from dreamwaltz_g_amd import _lib


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    rows = [int(a) for a in sys.argv[1:]] or [300000, 100000, 10000]
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    static = MLP(32, 4, 64, 3, bias=True).to(dev)
    deform = DeformNetwork().to(dev)
    pose = torch.randn(1, 63, device=dev)
    for M in rows:
        x = torch.randn(M, 32, device=dev) * 0.1
        with torch.inference_mode():
            ts = timed(lambda: static(x))
            td = timed(lambda: deform(x, pose))
        xg = x.clone().requires_grad_(True)

        def fb_static():
            static(xg).sum().backward()

        def fb_deform():
            o = deform(xg, pose)
            sum(t.sum() for t in o if torch.is_tensor(t)).backward() if isinstance(o, (tuple, list)) else o.sum().backward()
        tfs, tfd = timed(fb_static, 10, 3), timed(fb_deform, 10, 3)
        fl_s, fl_d = 2 * (32 * 64 + 64 * 64 + 64 * 4) * M, 2 * (32 * 64 + 3 * 64 * 64 + 64 * 10) * M
        ks = {}
        for name, fn in (("static", lambda: static(x)), ("deform", lambda: deform(x, pose))):       # the kernel alone (HIP events around the launch)
            _lib.prof_enable(True)
            with torch.inference_mode():
                for _ in range(10):
                    fn()
            torch.cuda.synchronize()
            t = _lib.prof_table().get("mlp_chain_fwd", (1, 0.0))
            _lib.prof_enable(False)
            ks[name] = t[1] / max(t[0], 1) * 1e3
        print("M=%7d  kernel alone: static %.1f us  deform %.1f us" % (M, ks["static"], ks["deform"]))
        print("M=%7d  static fwd %7.1f us (%5.1f TF/s)  deform fwd %7.1f us (%5.1f TF/s)   fwd+bwd (host-paced) static %7.1f us  deform %7.1f us"
              % (M, ts, fl_s / ts / 1e6, td, fl_d / td / 1e6, tfs, tfd))


if __name__ == "__main__":
    main()
