"""Worker of tests/test_multiview_gpu.py: the real SDSStep (HIP kernels) for a few multi-view steps, one process per rank.

    python tests/multiview_worker.py <out_prefix> <views> <steps>       (RANK / WORLD_SIZE / MASTER_* from the launcher, or one process)

Every rank runs on cuda:0 over gloo (a 1-GPU box; RCCL refuses two ranks on one device) -- or, with DWG_WORKER_BACKEND=nccl on a box with
at least WORLD_SIZE GPUs, rank r on cuda:r over RCCL / xGMI (the path bench.py --gpus N takes).  Writes <out_prefix>_rank<r>.pt with the flat
parameter and gradient buffers after the last step (and the exchange step's mean duration)."""
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
    backend = os.environ.get("DWG_WORKER_BACKEND", "gloo")
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    step = sds_step.SDSStep(n_gaussians=12000, res=128, device=dev, rank=rank, world=world, guidance=False, dist=dist, views=views,
                            iters=1000)
    assert step.my_views == list(range(rank, views, world))
    for _ in range(steps):
        step.run()
    torch.cuda.synchronize()
    b = step.optimizers.buffers
    torch.save({"flat": b.flat.cpu(), "grad": b.grad.cpu(), "views": step.my_views, "redone": step.trainer.redone_frames,
                "grad_scale": step.optimizers["avatar"].grad_scale,
                "allreduce_ms": step.trainer.allreduce_ms if world > 1 else None, "backend": backend if world > 1 else None,
                "device": dev}, "%s_rank%d.pt" % (out, rank))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
