"""CPU tests of the host-side logic above the C-ABI: weight re-layouts the kernels rely on, mesh adjacency, optimizer
bookkeeping.  No GPU, no HIP launches (the oracle here is plain torch on CPU)."""
import torch
import torch.nn.functional as F

import dwg_import  # noqa: F401
from dreamwaltz_g_amd import meshbind as mb
from dreamwaltz_g_amd import sd15
from dreamwaltz_g_amd import optim
from oracle import animate as oa


def test_stride2_dgrad_parity_class_weights_reproduce_the_input_gradient():
    """Weights.conv_dgrad_s2: the four tap subsets, applied as stride-1 convolutions of dy with pad (KH-1, KW-1) on the top /
    left and interleaved by output parity, equal the autograd input gradient of F.pad(x,(0,1,0,1)) + conv3x3 stride 2
    (diffusers Downsample2D with padding 0, as used by the VAE encoder)."""
    g = torch.Generator().manual_seed(0)
    B, Cx, Cy, H = 2, 16, 24, 10
    x = torch.randn(B, Cx, H, H, generator=g, dtype=torch.float64).requires_grad_(True)
    w = torch.randn(Cy, Cx, 3, 3, generator=g, dtype=torch.float64)
    y = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, stride=2)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (gx,) = torch.autograd.grad(y, x, dy)
    W = sd15.Weights({"c.weight": w.float()}, torch.device("cpu"))
    out = torch.zeros_like(gx)
    for py in (0, 1):
        for px in (0, 1):
            ws = W.conv_dgrad_s2("c", py, px).double()                   # [Cx(pad 8), KH', KW', Cy(pad 8)]
            assert ws.shape[0] == Cx and ws.shape[3] == Cy and ws.shape[1] == 2 - py and ws.shape[2] == 2 - px
            wt = ws.permute(0, 3, 1, 2)                                  # conv2d layout [out=Cx, in=Cy, KH', KW']
            d = F.pad(dy, (ws.shape[2] - 1, 0, ws.shape[1] - 1, 0))
            out[:, :, py::2, px::2] = F.conv2d(d, wt)
    # bf16 storage of the weights: relative error ~ 2^-9 per tap
    assert (out - gx).abs().max() / gx.abs().max() < 2e-2


def test_geglu_weight_interleave():
    """Weights.lin_geglu: rows come out in blocks of 64 = [32 hidden | the matching 32 gate] rows."""
    g = torch.Generator().manual_seed(1)
    C = 64
    w = torch.randn(8 * C, C, generator=g); b = torch.randn(8 * C, generator=g)
    W = sd15.Weights({"p.weight": w, "p.bias": b}, torch.device("cpu"))
    wi, bi = W.lin_geglu("p")
    Fh = 4 * C
    for q in range(Fh // 32):
        assert torch.equal(bi[64 * q:64 * q + 32], b[32 * q:32 * q + 32])
        assert torch.equal(bi[64 * q + 32:64 * q + 64], b[Fh + 32 * q:Fh + 32 * q + 32])
        assert torch.equal(wi[64 * q + 32].float(), w[Fh + 32 * q].bfloat16().float())
    # the fused epilogue computes hidden * gelu(gate) for output column 32 q + i from rows (64 q + i, 64 q + 32 + i)
    x = torch.randn(5, C, generator=g)
    ref = (x @ w.bfloat16().float().t() + b)                              # same bf16-rounded weights: isolates the permutation
    ref = ref[:, :Fh] * F.gelu(ref[:, Fh:])
    y = x @ wi.float().t() + bi
    y = y.view(5, -1, 2, 32)
    got = (y[:, :, 0] * F.gelu(y[:, :, 1])).reshape(5, Fh)
    assert (got - ref).abs().max() < 1e-4


def test_vertex_face_csr_matches_index_add_normals():
    g = torch.Generator().manual_seed(2)
    V = 60
    tri = torch.randint(0, V, (150, 3), generator=g)
    tri = tri[(tri[:, 0] != tri[:, 1]) & (tri[:, 1] != tri[:, 2]) & (tri[:, 0] != tri[:, 2])]
    off, faces = mb.build_vertex_face_csr(tri, V)
    assert off[0] == 0 and off[-1] == 3 * tri.shape[0] and off.dtype == torch.int32
    verts = torch.randn(V, 3, generator=g, dtype=torch.float64)
    vn_ref, fn = oa.compute_normal(verts, tri)
    acc = torch.zeros(V, 3, dtype=torch.float64)
    for v in range(V):
        for e in range(int(off[v]), int(off[v + 1])):
            acc[v] += fn[int(faces[e])]
    deflt = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
    acc = torch.where((acc * acc).sum(-1, keepdim=True) > 1e-20, acc, deflt)
    assert (oa.safe_normalize(acc) - vn_ref).abs().max() < 1e-12


def test_flat_optimizer_learning_rate_schedule_bookkeeping():
    """GaussianOptimizer.update_learning_rate semantics (gaussian_optimizer.py:130-141) on a view of the flat-buffer optimizer."""
    a = torch.nn.Parameter(torch.zeros(10, 3)); b = torch.nn.Parameter(torch.zeros(10, 3)); c = torch.nn.Parameter(torch.zeros(7))
    spec = optim.AdamSpec([dict(params=[a], lr=1.6e-4, name="positions"), dict(params=[b], lr=2.5e-3, name="scales"),
                           dict(params=[c], lr=1e-3, name="quaternions")], eps=1e-15,
                          gaussian=dict(iterations=10000, position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01,
                                        position_lr_max_steps=20000, scaling_lr=2.5e-3))
    opts = optim.build_flat_optimizers({"avatar": spec}, torch.device("cpu"))
    opt = opts["avatar"]
    lr = opt.update_learning_rate(spatial_scale=1.04, iteration=0)
    g = opt.param_groups
    assert abs(g[0]["lr"] - 1.6e-4 * 1.04) < 1e-12 and abs(g[1]["lr"] - 2.5e-3 * 1.04) < 1e-12    # lr_delay_steps = 0: no ease-in
    assert g[2]["lr"] == 1e-3 and lr == 2.5e-3                  # the reference returns the last group rate it touched
    opt.update_learning_rate(spatial_scale=1.04, iteration=20000)
    assert abs(g[0]["lr"] - 1.6e-6 * 1.04) < 1e-15
    # parameters and gradients are views of the flat buffers (16-byte aligned slices)
    buf = opts.buffers
    assert a.data.data_ptr() == buf.flat.data_ptr() and a.grad.data_ptr() == buf.grad.data_ptr()
    assert (b.data.data_ptr() - buf.flat.data_ptr()) % 16 == 0 and (c.data.data_ptr() - buf.flat.data_ptr()) % 16 == 0
    sd = opt.state_dict()
    opt.load_state_dict(sd)
    assert sd["exp_avg"].numel() == opt.end - opt.start


def test_reference_checkpoint_keys_load_into_the_native_avatar():
    """A reference-shaped `model` state dict (Scene.state_dict(): avatar under `avatar.` and `avatars.0.`, trainer.py:238-259)
    with a DIFFERENT Gaussian count loads into dreamwaltz_g_amd.avatar.DreamWaltzG: per-Gaussian parameters are resized
    (gaussian_model.py:58-85) and every network / mesh-binding tensor is copied by name."""
    from dreamwaltz_g_amd import avatar as av, synth
    g = torch.Generator().manual_seed(3)
    body = synth.synthetic_body(seed=0)
    glbs = av.GeneralLinearBlendSkinning(body)
    N0, N1 = 50, 80
    mk = lambda n: dict(p=torch.randn(n, 3, generator=g) * 0.3, s=torch.rand(n, 3, generator=g) * 0.01 + 0.002,  # noqa: E731
                        q=torch.randn(n, 4, generator=g), w=torch.softmax(torch.randn(n, 55, generator=g), -1))
    a0 = mk(N0)
    vi = torch.arange(30); tri = torch.tensor([[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, 10, 11]])
    mesh = {"hands": av.MeshBindingGaussianModel(body["v_template"][vi], tri, vi)}
    cnl = dict(body_pose=torch.zeros(1, 63), global_orient=torch.zeros(1, 3), left_hand_pose=torch.zeros(1, 45),
               right_hand_pose=torch.zeros(1, 45), expression=torch.zeros(1, 100))
    a = av.DreamWaltzG(glbs, a0["p"], a0["s"], a0["q"], a0["w"], cnl, mesh)
    # reference-shaped checkpoint: same names, other Gaussian count, plus keys the native module has no use for
    src = {k: torch.randn(v.shape, generator=g) for k, v in a.state_dict().items() if v.is_floating_point()}
    b1 = mk(N1)
    src.update({"_positions": b1["p"], "_scales": torch.log(b1["s"]), "_quaternions": b1["q"], "_lbs_weights": b1["w"]})
    ckpt = {"avatar." + k: v for k, v in src.items()}
    ckpt.update({"avatars.0." + k: v for k, v in src.items()})
    ckpt["avatar.nearest_triangles_buffer"] = torch.zeros(N1, dtype=torch.long)
    ckpt["background.some_weight"] = torch.zeros(3)
    loaded, unknown, missing = a.load_reference_state_dict(ckpt)
    assert a._positions.shape == (N1, 3) and a._lbs_weights.shape == (N1, 55) and not a._lbs_weights.requires_grad
    assert torch.equal(a._positions.data, b1["p"]) and torch.equal(a._scales.data, torch.log(b1["s"]))
    assert torch.equal(a.nerf_encoder.embeddings.data, src["nerf_encoder.embeddings"])
    assert torch.equal(a.mesh_binding_gaussians["hands"]._bary_coords.data, src["mesh_binding_gaussians.hands._bary_coords"])
    assert unknown == ["nearest_triangles_buffer"]
    assert "nerf_opacity_and_color_net.net.0.weight" in loaded and "nerf_scale_and_quaternion_net.gaussian_warp.weight" in loaded
    assert all(not k.startswith("_") for k in missing)      # only (integer) buffers of the native module may be absent
    # derived state is rebuilt after a load: encoder host offsets, skeleton subset cache, canonical caches
    assert a.nerf_encoder._host_offsets_py is None and a.lbs_model._subsets == [] and a._canonical_cache is None
    # the key names of the REFERENCE classes (state_dict of a reference DreamWaltzG built by tests/golden/capture_golden_r2.py):
    # every one has a counterpart here, except the integer index buffers we rebuild and `lbs_model.*` tensors we name differently
    import numpy as np, os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_golden_r2.npz"))
    own = set(a.state_dict().keys())
    rename = {"lbs_model.shapedirs": "lbs_model.shapedirs_all", "lbs_model.expr_dirs": "lbs_model.shapedirs_all"}
    absent = [k for k in (str(x) for x in G["sd.animate.state_dict_keys"]) if rename.get(k, k) not in own]
    assert all(k.startswith("lbs_model.") or k.endswith("points_to_triangles") for k in absent), absent


def test_trainer_checkpoint_round_trip_has_the_reference_layout(tmp_path):
    """SDSTrainer.save_checkpoint / load_checkpoint (mirror of trainer.py:188-259): file name, top-level keys, the avatar under
    `avatar.` with the reference's parameter names, optimizer state, rolling window, resize on load."""
    from dreamwaltz_g_amd import avatar as av, configs, scene as sc, synth, trainer as tr
    body = synth.synthetic_body(V=64, J=55, seed=0)
    g = torch.Generator().manual_seed(0)

    def make(n):
        glbs = av.GeneralLinearBlendSkinning(body)
        vi = torch.arange(12); tri = torch.tensor([[0, 1, 2], [3, 4, 5]])
        mesh = {"hands": av.MeshBindingGaussianModel(body["v_template"][vi], tri, vi)}
        return av.DreamWaltzG(glbs, torch.rand(n, 3, generator=g), torch.rand(n, 3, generator=g) + 0.1, torch.randn(n, 4, generator=g),
                              torch.rand(n, 55, generator=g), {}, mesh)
    cfg = configs.TrainConfig(); cfg.device = "cpu"
    a = make(9)
    opts = a.get_optimizer(cfg)
    t = tr.SDSTrainer(cfg, sc.Scene(cfg, a), None, opts)
    t.train_step_index = 7
    opts["avatar"].t = 7; opts.buffers.m.fill_(0.25)
    path = t.save_checkpoint(str(tmp_path), full=True)
    assert path.endswith("step_000007.pth")
    d = torch.load(path, weights_only=False)
    assert set(d.keys()) == {"train_step", "checkpoints", "optimizers", "scaler", "model"} and d["train_step"] == 7
    assert d["checkpoints"] == ["step_000007.pth"] and len(d["optimizers"]) == len(opts)
    assert {"avatar._positions", "avatar._scales", "avatar._quaternions", "avatar._lbs_weights", "avatar._betas", "avatar.nerf_bound",
            "avatar.nerf_encoder.embeddings", "avatar.mesh_binding_gaussians.hands._bary_coords"} <= set(d["model"].keys())
    # load into a scene with another Gaussian count: per-Gaussian parameters are resized first (gaussian_model.py:58-85)
    b = make(5)
    t2 = tr.SDSTrainer(cfg, sc.Scene(cfg, b), None, None)
    t2.load_checkpoint(path, model_only=True)
    assert b._positions.shape == (9, 3) and torch.equal(b._positions.data, d["model"]["avatar._positions"]) and t2.train_step_index == 7
    # same count: optimizer state comes back too
    c = make(9)
    o3 = c.get_optimizer(cfg)
    t3 = tr.SDSTrainer(cfg, sc.Scene(cfg, c), None, o3)
    t3.load_checkpoint(path)
    assert o3["avatar"].t == 7 and float(o3.buffers.m[:27].min()) == 0.25
    assert torch.equal(c._positions.data, d["model"]["avatar._positions"])
    # rolling window of checkpoint files
    for step in (8, 9):
        t.train_step_index = step
        t.save_checkpoint(str(tmp_path), full=False, max_keep_ckpts=2)
    import os
    assert sorted(os.listdir(tmp_path)) == ["step_000008.pth", "step_000009.pth"]


def test_morton_order_is_a_permutation_that_keeps_neighbours_together():
    from dreamwaltz_g_amd.rasterizer import morton_order
    g = torch.Generator().manual_seed(3)
    p = torch.rand(5000, 3, generator=g) * torch.tensor([2.0, 0.5, 1.0]) - 1.0
    o = morton_order(p)
    assert o.dtype == torch.int32 and o.is_contiguous() and sorted(o.tolist()) == list(range(5000))
    step_sorted = (p[o.long()][1:] - p[o.long()][:-1]).norm(dim=1).mean()
    step_index = (p[1:] - p[:-1]).norm(dim=1).mean()
    assert step_sorted < 0.25 * step_index
    # the code is the bit interleave of the quantised coordinates (x lowest)
    q = ((p - p.amin(0)) / (p.amax(0) - p.amin(0)) * 1023).long().clamp(0, 1023)
    code = torch.zeros(5000, dtype=torch.long)
    for b in range(10):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    assert torch.equal(code[o.long()], torch.sort(code).values)
    assert sorted(morton_order(torch.zeros(7, 3)).tolist()) == list(range(7))            # degenerate box: still a permutation
    try:
        morton_order(p, bits=11)
        assert False
    except ValueError:
        pass


def test_renderer_refreshes_its_binning_order_on_schedule():
    from dreamwaltz_g_amd.renderer import GaussianRenderer
    r = GaussianRenderer(reorder_every=3)
    p = torch.rand(100, 3)
    a = r._visit_order_for(p)
    assert r._visit_order_for(p) is a and r._visit_order_for(p) is a
    b = r._visit_order_for(p.flip(0))                     # 4th frame: recomputed from the positions it is given
    assert b is not a and not torch.equal(a, b)
    c = r._visit_order_for(torch.rand(50, 3))              # another Gaussian count: its own entry
    assert c.numel() == 50 and r._visit_order_for(p.flip(0)) is b
    assert GaussianRenderer(reorder_every=0).reorder_every == 0


def test_frozen_pair_capacity_never_waits():
    from dreamwaltz_g_amd.rasterizer import PairCapacity
    st = PairCapacity()
    st.seed(1000)
    cap = st.cap
    st.frozen, st.pending, st.event = True, True, None      # a frame inside a captured graph: no event to wait on
    st.resolve()
    assert st.cap == cap and st.pending
    st.frozen, st.pending = False, False
    assert st.consume_overflow() is False


def test_flat_optimizer_leaves_groups_without_a_gradient_alone_like_torch_adam():
    """torch.optim.Adam skips a parameter whose `grad is None` (it took no part in this step's backward): parameter, moments and step count
    stay (/root/reference/core/gaussian/gaussian_optimizer.py:93 builds exactly that optimizer; trainer.py:861-890 zero_grad()s to None).
    The flat-buffer optimizers reproduce it from the participation the backward recorded -- one group's gradient withheld on alternating
    steps, against torch.optim.Adam step for step.  (The fused HIP Adam launch is restated in torch: no GPU here.)"""
    import torch
    from dreamwaltz_g_amd import optim
    from tests.test_distributed_cpu import _cpu_adam_launch
    saved = optim.FlatOptimizer._launch
    optim.FlatOptimizer._launch = _cpu_adam_launch
    try:
        g = torch.Generator().manual_seed(0)
        a = torch.nn.Parameter(torch.randn(6, 3, generator=g)); b = torch.nn.Parameter(torch.randn(5, generator=g))
        c = torch.nn.Parameter(torch.randn(4, generator=g))
        ra, rb, rc = (torch.nn.Parameter(t.detach().clone()) for t in (a, b, c))
        ref = torch.optim.Adam([dict(params=[ra], lr=1e-2), dict(params=[rb], lr=3e-3), dict(params=[rc], lr=1e-3)], eps=1e-15)
        opts = optim.build_flat_optimizers({"avatar": optim.AdamSpec([dict(params=[a], lr=1e-2), dict(params=[b], lr=3e-3)], eps=1e-15),
                                            "lbs": optim.AdamSpec([dict(params=[c], lr=1e-3)], eps=1e-15)}, torch.device("cpu"))

        def loss_of(pa, pb, pc, it):
            x = torch.linspace(0.1, 1.0, 3) * (it + 1)
            out = (pa @ x).pow(2).sum()
            if it % 2 == 0:
                out = out + (pb * pb).sum() * 0.5 + pc.sum() * 0.0      # c takes part with an exactly ZERO gradient: it IS stepped
            return out                                                  # odd steps: b and c take no part at all
        for it in range(6):
            for o in opts.values():
                o.zero_grad()
            ref.zero_grad(set_to_none=True)
            loss_of(a, b, c, it).backward()
            loss_of(ra, rb, rc, it).backward()
            assert (rb.grad is None) == (it % 2 == 1)
            for o in opts.values():
                o.step()
            ref.step()
            for p, r in ((a, ra), (b, rb), (c, rc)):
                assert torch.allclose(p.data, r.data, rtol=1e-6, atol=1e-7), (it, float((p.data - r.data).abs().max()))
        assert opts["avatar"].param_groups[0]["t"] == 6 and opts["avatar"].param_groups[1]["t"] == 3 and opts["lbs"].param_groups[0]["t"] == 3
        # gradients written by hand (no backward ran, nothing recorded): every group steps, as before
        for o in opts.values():
            o.zero_grad()
        before = b.data.clone()
        b.grad.fill_(1.0)
        for o in opts.values():
            o.step()
        assert not torch.equal(before, b.data) and opts["avatar"].param_groups[1]["t"] == 4
    finally:
        optim.FlatOptimizer._launch = saved


def test_mlp_gradients_go_in_place_only_into_genuine_flat_gradient_slices():
    """mlp._flat_slice decides whether the fused MLP backward may ADD a parameter's gradient straight into `p.grad` (no AccumulateGrad
    launch): only for a leaf Parameter marked by optim.FlatBuffers whose .grad is the contiguous fp32 slice of its own shape and whose
    values the kernels read in place; everything else goes through autograd as usual."""
    from dreamwaltz_g_amd import gridencoder, mlp
    w, b = torch.nn.Parameter(torch.randn(8, 16)), torch.nn.Parameter(torch.randn(8))
    other = torch.nn.Parameter(torch.randn(8, 16))
    flat = optim.FlatBuffers([w, b], torch.device("cpu"))
    assert mlp._flat_slice(w, w) is flat and mlp._flat_slice(b, None) is flat
    assert mlp._flat_slice(other, other) is None                                   # not part of a flat buffer
    assert mlp._flat_slice(None, None) is None                                     # a layer without a bias
    assert mlp._flat_slice(w, w.detach().clone()) is None                          # the kernels read a COPY (dtype / layout conversion)
    cat = torch.cat([w, other], 0)
    assert mlp._flat_slice(cat, cat) is None                                       # a concatenation of parameters is not a leaf
    g = w.grad
    w.grad = None
    assert mlp._flat_slice(w, w) is None                                           # zero_grad(set_to_none=True)-style callers
    w.grad = g.t().contiguous().t()                                                # same shape, not contiguous: not the flat slice
    assert mlp._flat_slice(w, w) is None
    w.grad = torch.zeros_like(g)                                                   # same shape, contiguous, but REBOUND by the user: the optimizer never reads it
    assert mlp._flat_slice(w, w) is None
    w.grad = g
    assert mlp._flat_slice(w, w) is flat
    assert mlp._flat_slice(w, w, needed=False) is None                             # autograd did not ask for this gradient (torch.autograd.grad of another input)
    w.requires_grad_(False)
    assert mlp._flat_slice(w, w) is None                                           # a frozen parameter keeps its slice untouched
    w.requires_grad_(True)
    assert mlp._flat_slice(w, w) is flat
    flat.release([w, b])
    assert mlp._flat_slice(w, w) is None                                           # buffers replaced (densifier resize): no stale target
    # concurrent backwards (the views of a batched step on their own streams) opt out through the grid encoder's switch
    assert mlp._inplace_allowed()
    with gridencoder.table_grad_inplace(False):
        assert not mlp._inplace_allowed()
    assert mlp._inplace_allowed()


def test_parameter_epoch_moves_with_every_write_behind_autograds_back():
    """optim.PARAM_EPOCH: the fused Adam writes parameters through a raw pointer, so no tensor version counter moves -- caches of
    parameter-derived values (avatar.DreamWaltzG._frozen_key, mlp.DeformNetwork's concatenated heads) key on this epoch instead.  It must
    move on every optimizer step, eager (`step`) or captured (`prepare_step`: the host half of a replayed step)."""
    import torch
    from dreamwaltz_g_amd import optim
    from tests.test_distributed_cpu import _cpu_adam_launch
    saved = optim.FlatOptimizer._launch
    optim.FlatOptimizer._launch = _cpu_adam_launch
    try:
        a = torch.nn.Parameter(torch.randn(4, 3))
        opts = optim.build_flat_optimizers({"avatar": optim.AdamSpec([dict(params=[a], lr=1e-2)], eps=1e-15)}, torch.device("cpu"))
        o = opts["avatar"]
        v0, e0 = a._version, optim.PARAM_EPOCH[0]
        o.zero_grad()
        (a * a).sum().backward()
        o.step()
        assert optim.PARAM_EPOCH[0] == e0 + 1
        hyper = torch.zeros(64, 4)
        o.prepare_step(hyper, 0)
        assert optim.PARAM_EPOCH[0] == e0 + 2
    finally:
        optim.FlatOptimizer._launch = saved
