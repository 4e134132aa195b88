"""player.GraphedAnimation (config c5's frame as one captured hipGraph) against the eager Scene.forward on the same poses."""
import pytest
import torch

import dwg_import  # noqa: F401

pytestmark = pytest.mark.gpu


def _scene(G, res, dev):
    from dreamwaltz_g_amd import camera, configs, scene as sc, sds_step
    cfg = configs.TrainConfig(); cfg.device = str(dev); cfg.render.bg_color = (0.5, 0.5, 0.5)
    avatar, _, _ = sds_step.build_synthetic_avatar(G, dev, seed=0)
    scene = sc.Scene(cfg, avatar, async_pair_count=True).to(dev).eval()
    data = camera.make_camera(radius=2.0, azimuth=20.0, elevation=80.0, fovy=55.0, height=res, width=res, device=dev)
    return scene, data


def test_graphed_frames_equal_eager_frames():
    from dreamwaltz_g_amd import player, synth
    dev = torch.device("cuda:0")
    scene, data = _scene(20000, 256, dev)
    poses = [synth.random_smpl_inputs(seed=i, device=dev) for i in range(6)]
    eager = []
    with torch.inference_mode():
        for p in poses:
            o = scene.forward(data, smpl_observed_inputs=p, use_densifier=False, bg_mode=None)
            eager.append({k: o[k].clone() for k in ("image", "alpha", "depth")})
    pl = player.GraphedAnimation(scene, data, poses[0], warmup_poses=poses[:3])
    for p, e in zip(poses, eager):
        o = pl.replay(p)
        assert pl.check()
        K, Kref = pl.last_num_pairs
        assert K > 0 and Kref > 0
        for k in ("image", "alpha", "depth"):
            assert torch.equal(o[k], e[k]), k                  # same kernels, same inputs, deterministic forward
    # eager frames through the SAME Scene while the graph is alive (training between previews): the renderer replaces its binning-order
    # tensor every `reorder_every` frames and keeps its own pair bookkeeping; the graph holds its own references and its own frozen state
    scene.renderer.reorder_every = 2
    with torch.inference_mode():
        for p in poses[:5]:
            scene.forward(data, smpl_observed_inputs=p, use_densifier=False, bg_mode=None)
    junk = [torch.randint(0, 2 ** 31 - 1, (20000,), dtype=torch.int32, device=dev) for _ in range(8)]     # recycle freed blocks, if any
    assert not scene.renderer.pair_state(dev, 256, 256).frozen
    o = pl.replay(poses[2])
    assert pl.check() and torch.equal(o["image"], eager[2]["image"])
    del junk
    # a capacity too small for the frame: the replay flags it, a recapture repairs it
    pl._state.cap = 1024
    pl.graph, pl.outputs = None, None
    pl._capture_frozen()
    pl.replay(poses[4])
    assert not pl.check()
    pl.recapture(poses[:3])
    o = pl.replay(poses[4])
    assert pl.check() and torch.equal(o["image"], eager[4]["image"])
    pl.close()
    with torch.inference_mode():                              # the eager path works again afterwards, with its own bookkeeping
        o = scene.forward(data, smpl_observed_inputs=poses[5], use_densifier=False, bg_mode=None)
    assert torch.equal(o["image"], eager[5]["image"])
    assert not scene.renderer.consume_overflow()


def test_player_needs_the_async_pair_count():
    from dreamwaltz_g_amd import camera, configs, player, scene as sc, sds_step, synth
    dev = torch.device("cuda:0")
    cfg = configs.TrainConfig(); cfg.device = str(dev)
    avatar, _, _ = sds_step.build_synthetic_avatar(2000, dev, seed=0)
    scene = sc.Scene(cfg, avatar, async_pair_count=False).to(dev).eval()
    data = camera.make_camera(height=64, width=64, device=dev)
    with pytest.raises(ValueError):
        player.GraphedAnimation(scene, data, synth.random_smpl_inputs(seed=0, device=dev))


@pytest.mark.parametrize("bg_mode", [None, "white"])
def test_frames_per_launch_playback_equals_frame_by_frame(bg_mode):
    """Scene.forward_frames: F pose frames animated one by one, rasterized by ONE launch chain (include/dwg_raster.h dwg_raster_frames) == the
    reference's frame-by-frame evaluation loop (trainer.py:1019-1150: Scene.forward per pose under inference mode), every output bit."""
    from dreamwaltz_g_amd import synth
    dev = torch.device("cuda:0")
    scene, data = _scene(20000, 256, dev)
    poses = [synth.random_smpl_inputs(seed=10 + i, device=dev) for i in range(5)]
    with torch.inference_mode():
        single = [scene.forward(data, smpl_observed_inputs=p, use_densifier=False, bg_mode=bg_mode) for p in poses]
        single = [{k: o[k].clone() for k in ("image", "image_fg", "alpha", "depth")} for o in single]
        for F in (1, 3, 5):
            batch = scene.forward_frames(data, poses[:F], bg_mode=bg_mode)
            assert batch["image"].shape == (F, 256, 256, 3)
            for f in range(F):
                for k in ("image", "image_fg", "alpha", "depth"):
                    assert torch.equal(batch[k][f:f + 1], single[f][k]), (F, f, k)
            hdr = scene.renderer.last_frames_headers.cpu()
            assert (hdr[:, 0] > 0).all() and (hdr[:, 1] == 0).all()           # pairs counted, nothing truncated
        # a camera per frame (the reference's evaluation loader yields one with every pose)
        from dreamwaltz_g_amd import camera
        cams = [camera.make_camera(radius=2.0 + 0.1 * f, azimuth=20.0 + 40.0 * f, elevation=80.0 - 5.0 * f, fovy=55.0, height=256, width=256, device=dev)
                for f in range(3)]
        own = [scene.forward(cams[f], smpl_observed_inputs=poses[f], use_densifier=False, bg_mode=bg_mode)["image"].clone() for f in range(3)]
        batch = scene.forward_frames(cams, poses[:3], bg_mode=bg_mode)
        for f in range(3):
            assert torch.equal(batch["image"][f:f + 1], own[f]), f
        assert not torch.equal(own[0], own[1])
        with pytest.raises(ValueError):
            scene.forward_frames(cams[:2], poses[:3])
    with pytest.raises(RuntimeError):
        g = scene.avatar_forward(smpl_observed_inputs=poses[0])              # gradients enabled: not the playback path
        scene.renderer.render_frames(data, [g])


def test_frozen_avatar_playback_keeps_the_pose_independent_part_and_notices_every_parameter_change():
    """Scene.forward_frames(frozen_avatar=True): the canonical encoding and the colour / opacity network run once, not per frame -- same
    bits as frame-by-frame `forward`; an optimizer step (which writes the flat buffer through a raw pointer), an in-place edit of a
    parameter, and `invalidate_caches()` each make the next frame recompute."""
    from dreamwaltz_g_amd import _lib, optim, synth
    dev = torch.device("cuda:0")
    scene, data = _scene(20000, 256, dev)
    av = scene.avatar
    poses = [synth.random_smpl_inputs(seed=30 + i, device=dev) for i in range(4)]

    def reference():
        with torch.inference_mode():
            return [scene.forward(data, smpl_observed_inputs=p, use_densifier=False, bg_mode=None)["image"].clone() for p in poses]

    def played():
        _lib.prof_enable(True)
        with torch.inference_mode():
            out = scene.forward_frames(data, poses, frozen_avatar=True)["image"].clone()
        torch.cuda.synchronize()
        t = _lib.prof_table(); _lib.prof_enable(False)
        return out, t.get("grid_fwd", (0, 0.0))[0]

    ref = reference()
    out, enc_launches = played()
    assert enc_launches == 1                                            # four frames, ONE encoder pass
    assert all(torch.equal(out[f:f + 1], ref[f]) for f in range(4))
    out, enc_launches = played()
    assert enc_launches == 0 and all(torch.equal(out[f:f + 1], ref[f]) for f in range(4))      # kept across calls
    # (a) an in-place edit of a parameter (version counter)
    with torch.no_grad():
        av.nerf_opacity_and_color_net.net[0].bias.add_(0.05)
    ref = reference()
    out, enc_launches = played()
    assert enc_launches == 1 and all(torch.equal(out[f:f + 1], ref[f]) for f in range(4))
    # (b) a write behind autograd's back, announced the way the fused optimizers announce theirs
    pos = av._positions
    alias = torch.empty(0, device=dev).set_(pos.untyped_storage(), pos.storage_offset(), pos.shape, pos.stride())   # own version counter
    v0 = pos._version
    alias.mul_(1.01)
    assert pos._version == v0                                           # the parameter's own counter did not notice
    optim.PARAM_EPOCH[0] += 1
    ref = reference()
    out, enc_launches = played()
    assert enc_launches == 1 and all(torch.equal(out[f:f + 1], ref[f]) for f in range(4))
    # (c) explicit invalidation (checkpoint load, densification)
    av.invalidate_caches()
    out, enc_launches = played()
    assert enc_launches == 1 and all(torch.equal(out[f:f + 1], ref[f]) for f in range(4))
    # not in use outside forward_frames(frozen_avatar=True)
    assert av.frozen_playback is False
