"""Pair counts of the c2 step over a long run: eager (per-step count) and under the whole-step graph (count + truncation counter per replay)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import sds_step

dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
st = sds_step.SDSStep(n_gaussians=50000, res=512, device=dev, guidance=False, async_pair_count=True)
ks = []
for i in range(N):
    st.run()
    torch.cuda.synchronize()
    ks.append(st.num_pairs[0])
print("eager: K first %d last %d min %d max %d redone %d" % (ks[0], ks[-1], min(ks), max(ks), st.trainer.redone_frames))
print("eager K every 25:", ks[::25])
st2 = sds_step.SDSStep(n_gaussians=50000, res=512, device=dev, guidance=False, async_pair_count=True)
r = st2.graphed()
g = r.graph
print("graph cap", g._state.cap)
ks, tr = [], []
for i in range(N):
    r.step()
    torch.cuda.synchronize()
    ks.append(int(g._state.host[0])); tr.append((int(g._state.truncated_host[0]), int(g._state.host[1])))
print("graph: K first %d last %d min %d max %d" % (ks[0], ks[-1], min(ks), max(ks)))
print("graph K every 25:", ks[::25])
print("truncated counter / last flag every 25:", tr[::25])
first = next((i for i, t in enumerate(tr) if t[0] or t[1]), None)
print("first flagged replay:", first, tr[first] if first is not None else None, ks[first] if first is not None else None)
# the bench's way: no synchronisation between replays
st3 = sds_step.SDSStep(n_gaussians=50000, res=512, device=dev, guidance=False, async_pair_count=True)
r = st3.graphed()
g = r.graph
for rnd in range(4):
    for i in range(110):
        r.step()
    torch.cuda.synchronize()
    print("async round %d: K %d flag %d truncated %d cap %d" % (rnd, int(g._state.host[0]), int(g._state.host[1]), int(g._state.truncated_host[0]), g._state.cap))
print("check:", g.check())
