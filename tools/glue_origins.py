"""Where the torch ("glue") ops of the avatar-side step come from: one eager c2 step under a TorchDispatchMode that records, for every aten op
that launches device work (everything but views / metadata), the innermost dreamwaltz_g_amd source line on the Python stack (forward) or
"backward" (ops issued by autograd nodes).   python tools/glue_origins.py [gaussians=50000] [res=512]"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from torch.utils._python_dispatch import TorchDispatchMode
from dreamwaltz_g_amd import sds_step

G = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
res = int(sys.argv[2]) if len(sys.argv) > 2 else 512
GUIDE = len(sys.argv) > 3 and sys.argv[3] == "1"
torch.cuda.set_stream(torch.cuda.Stream())
st = sds_step.SDSStep(n_gaussians=G, res=res, device="cuda", guidance=GUIDE)
for _ in range(3):
    st.run()
torch.cuda.synchronize()
VIEWS = ("view", "reshape", "expand", "slice", "select", "permute", "transpose", "t.", "unsqueeze", "squeeze", "alias", "detach", "as_strided",
         "_unsafe_view", "unbind", "split", "narrow", "empty", "size", "stride", "is_", "_local_scalar", "lift_fresh", "sym_", "result_type", "_reshape_alias",
         "new_empty", "record_stream", "set_", "resize_")
by = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        if not any(name.startswith(v) for v in VIEWS):
            where = "backward / no package frame"
            for fr in reversed(traceback.extract_stack()):
                if "dreamwaltz" in fr.filename and "tools" not in fr.filename:
                    where = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
                    break
            dev = [a.device.type for a in list(args) + list((kwargs or {}).values()) if torch.is_tensor(a)]
            by[(name, where, "cuda" if "cuda" in dev else ("cpu" if dev else "-"))] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    st.run()
torch.cuda.synchronize()
print("aten ops of one eager c2 step that touch data (views / metadata excluded): %d" % sum(by.values()))
for (n, w, d), c in sorted(by.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print("  %3d  %-5s %-32s %s" % (c, d, n, w))
