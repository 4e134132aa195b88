"""oracle/condition.py against the golden vectors captured from the imported reference (tests/golden/capture_golden_condition.py ->
reference_golden_r2_condition.npz): projection, invisibility, per-group occlusion thresholds, the rows handed to the drawing code.
The drawing half has no reference output to compare with (OpenCV absent: parity unpinned) -- its restated primitives are checked
against their own definitions."""
import os

import numpy as np
import pytest

from oracle import condition as oc

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_golden_r2_condition.npz"))
CASES = ["front", "side", "wide", "side_ignore_body"]


@pytest.mark.parametrize("name", CASES)
def test_pose_rows_match_the_reference_export_pose(name):
    p = "cond.%s." % name
    W, H = [int(v) for v in G[p + "size"]]
    rows = oc.pose_keypoints(G["cond.keypoints"], G["cond.vertices"], G["cond.triangles"], G[p + "extrinsic"], G[p + "intrinsics"], W, H,
                             ignore_body_self_occlusion=name.endswith("ignore_body"))
    g = G[p + "rows"]
    assert np.array_equal(np.isnan(rows[:, 0]), np.isnan(g[:, 0]))
    m = ~np.isnan(g[:, 0])
    assert m.sum() > 60 and (~m).sum() > 10                         # both outcomes are exercised
    assert np.abs(rows[m] - g[m]).max() < 1e-12
    assert list(G[p + "draw_kwargs"]) == ["draw_body=True", "draw_face=True", "draw_hand=True", "flip_LR=False"]


@pytest.mark.parametrize("name", CASES)
def test_occlusion_rule_and_size_adjusted_intrinsics(name):
    p = "cond.%s." % name
    E = G[p + "extrinsic"]
    center = (np.linalg.inv(E[:3, :3]) @ (-E[:3, 3:4])).reshape(3)
    occ, tfar = oc.occlusion(center, G["cond.keypoints"], G[p + "t_hit"].astype(np.float64), oc.keypoint_groups(),
                             ignore_body_self_occlusion=name.endswith("ignore_body"))
    assert np.array_equal(occ, G[p + "occluded"]) and np.abs(tfar - G[p + "t_far"]).max() < 1e-12
    W, H = [int(v) for v in G[p + "size"]]
    assert np.abs(oc.adjust_intrinsics_size(G[p + "intrinsics_raw"], W, H) - G[p + "intrinsics"]).max() == 0.0


def test_keypoint_groups_are_the_reference_index_lists():
    g = oc.keypoint_groups()
    assert sorted(np.nonzero(g == 2)[0]) == sorted(G["cond.face_indices"])
    assert sorted(np.nonzero(g == 1)[0]) == sorted(G["cond.hand_indices"])
    assert sorted(np.nonzero(g == 0)[0]) == sorted(G["cond.body_indices"])


def test_ignoring_body_self_occlusion_only_ever_restores_body_keypoints():
    a, b = G["cond.side.rows"], G["cond.side_ignore_body.rows"]
    restored = np.isnan(a[:, 0]) & ~np.isnan(b[:, 0])
    assert restored.sum() >= 1 and set(np.nonzero(restored)[0]) <= set(G["cond.body_indices"])
    assert not (np.isnan(b[:, 0]) & ~np.isnan(a[:, 0])).any()


def test_drawing_primitives_against_their_definitions():
    # midpoint circle: symmetric, radius reached on the axes, area between the inscribed diamond and the bounding square
    for r in range(1, 13):
        hw = oc.circle_halfwidths(r)
        assert hw[0] == r and hw[r] >= 0 and (np.diff(hw) <= 0).all()
        area = int((2 * hw + 1).sum() * 2 - (2 * hw[0] + 1))
        assert 2 * r * r < area <= (2 * r + 1) ** 2
    assert list(oc.circle_halfwidths(4)) == [4, 3, 3, 2, 0]
    # ellipse polygon: closed, integer, within the axis-aligned extent of the rotated ellipse (+1 for rounding), hull rows filled
    poly = oc.ellipse2poly(100, 80, 40, 4, 30)
    assert poly.dtype == np.int64 and (np.abs(poly - np.array([100, 80])).max(0) <= np.array([37, 25])).all()
    m = oc.convex_poly_mask(poly, 200, 200)
    area = 3.141592653589793 * 40 * 4
    assert all(m[y, x] for x, y in poly) and m[80, 100] and area < m.sum() < 1.25 * area        # ellipse + its rounded boundary
    ys, xs = np.nonzero(m)
    assert abs(xs.mean() - 100) < 0.5 and abs(ys.mean() - 80) < 0.5                              # centred
    assert np.array_equal(m[80 - 30:80 + 31, 100 - 45:100 + 46], m[80 - 30:80 + 31, 100 - 45:100 + 46][::-1, ::-1])   # point-symmetric
    # thick segment: the 5-pixel plus at a zero-length segment of thickness 2, a 3-wide band along a horizontal one
    assert oc.thick_line_mask(10, 10, 10, 10, 2, 32, 32).sum() == 5
    band = oc.thick_line_mask(5, 10, 20, 10, 2, 32, 32)
    assert band[9:12, 5:21].all() and band.sum() == 3 * 16 + 2
    # blend: uint8, half-to-even on exact ties, identity when old == colour
    cv = np.array([[[10, 20, 255]]], dtype=np.uint8)
    oc.add_weighted_inplace(cv, np.ones((1, 1), dtype=bool), (10, 20, 255))
    assert cv.tolist() == [[[10, 20, 255]]]
    assert oc.hsv_edge_color(0) == (255, 0, 0) and oc.hsv_edge_color(10) == (0, 255, 255) and len({oc.hsv_edge_color(e) for e in range(20)}) == 20


def test_draw_order_later_primitives_cover_earlier_ones():
    rows = G["cond.front.rows"]
    full = oc.draw_poses(rows, 512, 512)
    body_only = oc.draw_poses(rows, 512, 512, draw_hand=False, draw_face=False)
    face_px = (full == 255).all(2)
    assert face_px.sum() > 200                                        # white face landmarks on top of everything
    assert ((full != body_only).any(2) & ~face_px).sum() > 100        # hands drawn over the body
    flipped = oc.draw_poses(rows, 512, 512, flip_LR=True)
    assert (flipped != full).any()
    small = oc.draw_poses(G["cond.wide.rows"], 256, 384)
    assert small.shape == (256, 384, 3) and oc.draw_sizes(256, 384) == [2, 2, 2, 1, 1]
