"""Workload for PMC passes on the grid-encoder backward alone: 100 000 points, the avatar's encoder configuration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dwg_import  # noqa
from dreamwaltz_g_amd import gridencoder

torch.manual_seed(0)
enc = gridencoder.GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048).cuda()
x = (torch.rand(100000, 3, device="cuda") * 2 - 1) * torch.tensor([0.4, 0.9, 0.2], device="cuda")
g = torch.randn(100000, 32, device="cuda")
for _ in range(4):
    enc.embeddings.grad = None
    y = enc(x, bound=1.0)
    y.backward(g)
torch.cuda.synchronize()
print("ok", float(enc.embeddings.grad.abs().sum()))
