"""Worker of tests/test_multiview_gpu.py: the real SDSStep (HIP kernels) for a few multi-view steps, one process per rank.

    python tests/multiview_worker.py <out_prefix> <views> <steps>       (RANK / WORLD_SIZE / MASTER_* from the launcher, or one process)

Every rank runs on cuda:0 over gloo (a 1-GPU box; RCCL refuses two ranks on one device).  Writes <out_prefix>_rank<r>.pt with the flat
parameter and gradient buffers after the last step."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dwg_import  # noqa: E402,F401
from dreamwaltz_g_amd import sds_step  # noqa: E402


def main():
    out, views, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(0)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    torch.cuda.set_stream(torch.cuda.Stream(device="cuda:0"))
    step = sds_step.SDSStep(n_gaussians=12000, res=128, device="cuda:0", rank=rank, world=world, guidance=False, dist=dist, views=views,
                            iters=1000)
    assert step.my_views == list(range(rank, views, world))
    for _ in range(steps):
        step.run()
    torch.cuda.synchronize()
    b = step.optimizers.buffers
    torch.save({"flat": b.flat.cpu(), "grad": b.grad.cpu(), "views": step.my_views, "redone": step.trainer.redone_frames,
                "grad_scale": step.optimizers["avatar"].grad_scale}, "%s_rank%d.pt" % (out, rank))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
