"""Which branch topologies does hipGraph stream capture accept?  Each pattern runs in its own process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAT = sys.argv[1] if len(sys.argv) > 1 else None
if PAT is None:
    for p in ("A", "B", "C", "D", "E"):
        r = subprocess.run([sys.executable, __file__, p], capture_output=True, text=True)
        print(p, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1])
    sys.exit(0)
sys.path.insert(0, ROOT)
import ctypes, torch
import dwg_import  # noqa
from dreamwaltz_g_amd import sd15, _lib
torch.cuda.set_stream(torch.cuda.Stream())
p = sd15.Plan(torch.device("cuda"))
L = _lib.lib()
pp = lambda t: ctypes.c_void_p(t.data_ptr())
n = 1 << 16
def op(a, b, y):
    p.add_call(L.dwg_add_bf16, n, pp(a), pp(b), pp(y))
x = p.buf(n); x.fill_(1.0)
bufs = [p.buf(n) for _ in range(40)]
if PAT == "A":      # same side stream forked/joined repeatedly
    cur = x
    for i in range(5):
        p.fork(4)
        with p.on_branch(4):
            op(cur, x, bufs[2 * i])
        op(cur, x, bufs[2 * i + 1])
        p.join(4)
        nxt = bufs[20 + i]; op(bufs[2 * i], bufs[2 * i + 1], nxt); cur = nxt
elif PAT == "B":    # one fork at the start, many ops on the side, several joins into main
    p.fork(2)
    cur = x
    for i in range(5):
        with p.on_branch(2):
            op(x, x, bufs[i])
        op(cur, x, bufs[10 + i])
        p.join(2)
        nxt = bufs[20 + i]; op(bufs[i], bufs[10 + i], nxt); cur = nxt
elif PAT == "C":    # 3 forked from 0, joined into 1; 1 joined into 0
    p.fork(1); p.fork(3)
    with p.on_branch(3):
        op(x, x, bufs[0])
    with p.on_branch(1):
        op(x, x, bufs[1])
        p.join(3)
        op(bufs[0], bufs[1], bufs[2])
    op(x, x, bufs[3])
    p.join(1)
    op(bufs[2], bufs[3], bufs[4]); cur = bufs[4]
elif PAT == "D":    # side stream 5 forked from side stream 1
    p.fork(1)
    with p.on_branch(1):
        op(x, x, bufs[0])
        p.fork(5)
        with p.on_branch(5):
            op(bufs[0], x, bufs[1])
        op(bufs[0], x, bufs[2])
        p.join(5)
        op(bufs[1], bufs[2], bufs[3])
    op(x, x, bufs[4])
    p.join(1)
    op(bufs[3], bufs[4], bufs[5]); cur = bufs[5]
elif PAT == "E":    # join of a branch twice without new work in between
    p.fork(2)
    with p.on_branch(2):
        op(x, x, bufs[0])
    p.join(2); op(bufs[0], x, bufs[1]); p.join(2); op(bufs[1], x, bufs[2]); cur = bufs[2]
p.capture()
p.run(); torch.cuda.synchronize()
print("ok", float(cur.float().mean()))
