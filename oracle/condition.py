"""CPU restatement (numpy) of the reference's OpenPose condition-image generator -- TEST INFRASTRUCTURE ONLY (tests/, smoke(),
bench.py's cpu_baseline); the product (dreamwaltz-g_amd/condition.py + csrc/condition.hip) never imports it.

Follows /root/reference:
  core/human/smpl_condition.py:191-235  SMPL2Condition.export_pose          -> pose_keypoints()
  core/human/smpl_condition.py:82-143   OcclusionCulling                    -> occlusion()
  core/human/smpl_condition.py:20-79    to_controlnet_pose                  -> (x / W, y / H, dist) rows, NaN row = missing keypoint
  core/human/open_pose.py:48-141        draw_bodypose                       -> _draw_body()
  core/human/open_pose.py:144-218       draw_handpose                       -> _draw_hand()
  core/human/open_pose.py:221-246       draw_facepose                       -> _draw_face()
  core/human/open_pose.py:279-333       adaptive_draw_poses                 -> draw_poses()
  utils/point3d.py:32-76, utils/se3.py:45-60, data/camera/utils.py:233-242  projection helpers

PARITY PINNING.  The numpy half of the reference (projection, invisibility, occlusion thresholds per keypoint group, the
normalised keypoint rows handed to the drawing code) is pinned by tests/golden/reference_golden_r2_condition.npz, captured from
the imported reference modules (tests/golden/capture_golden_condition.py).  Two third-party pieces are ABSENT from this image and
restated from their published behaviour -- PARITY UNPINNED there:
  * open3d 0.17 `RaycastingScene.cast_rays` (nearest hit distance along a ray)  -> ray_cast(): Moeller-Trumbore over all triangles;
  * OpenCV 4.x drawing (`cv2.circle` filled, `cv2.ellipse2Poly` + `cv2.fillConvexPoly`, `cv2.line` with thickness, `cv2.addWeighted`)
    -> circle_halfwidths() restates drawing.cpp's midpoint `Circle()`; ellipse2poly() restates `ellipse2Poly` (integer-degree sine
    table, cvRound, duplicate removal); the polygon fill paints, per row, round(x_left) .. round(x_right) of the exact boundary
    crossings (fillConvexPoly's left / right edge walk without its fixed-point increments, and without the Bresenham outline it
    draws first); the thick line is "every pixel within thickness / 2 of the segment" -- OpenCV's own output can differ from these
    by boundary pixels.
"""
import math

import numpy as np

EPS = 0.01                                   # open_pose.py:13

BODY_LIMBS = [(2, 3), (2, 6), (3, 4), (4, 5), (6, 7), (7, 8), (2, 9), (9, 10), (10, 11), (2, 12), (12, 13), (13, 14), (2, 1), (1, 15),
              (15, 17), (1, 16), (16, 18)]                                                     # open_pose.py:64-70 (1-based)
BODY_COLORS = [(255, 0, 0), (255, 85, 0), (255, 170, 0), (255, 255, 0), (170, 255, 0), (85, 255, 0), (0, 255, 0), (0, 255, 85),
               (0, 255, 170), (0, 255, 255), (0, 170, 255), (0, 85, 255), (0, 0, 255), (85, 0, 255), (170, 0, 255), (255, 0, 255),
               (255, 0, 170), (255, 0, 85)]                                                    # open_pose.py:103-109
BODY_FLIP = [0, 1, 5, 6, 7, 2, 3, 4, 11, 12, 13, 8, 9, 10, 15, 14, 17, 16]                     # open_pose.py:92-101
HAND_EDGES = [(0, 1), (1, 2), (2, 3), (3, 4), (0, 5), (5, 6), (6, 7), (7, 8), (0, 9), (9, 10), (10, 11), (11, 12), (0, 13), (13, 14),
              (14, 15), (15, 16), (0, 17), (17, 18), (18, 19), (19, 20)]                       # open_pose.py:172-173
N_BODY, N_HAND, N_FACE = 18, 21, 68
N_KEYPOINTS = N_BODY + 2 * N_HAND + N_FACE      # 128 (smpl_condition.py:22)


# ----------------------------------------------------------------------------------------------------------------------
# keypoints: projection, visibility, occlusion
# ----------------------------------------------------------------------------------------------------------------------
def keypoint_groups(smpl_type="smplx"):
    """smpl_condition.py:83-94 -> per-keypoint group id: 0 body, 1 hand, 2 face."""
    if smpl_type != "smplx":
        raise NotImplementedError
    g = np.zeros(N_KEYPOINTS, dtype=np.uint8)
    g[[0, 14, 15, 16, 17]] = 2
    g[18 + 2 * N_HAND:] = 2
    g[18:18 + 2 * N_HAND] = 1
    return g


def adjust_intrinsics_size(intrinsics, width, height):
    """data/camera/utils.py:233-242 (on a copy)."""
    k = np.array(intrinsics, dtype=np.float64, copy=True)
    w_raw, h_raw = k[0, 2] * 2, k[1, 2] * 2
    k[0, 2] = width / 2; k[1, 2] = height / 2
    k[0, 0] *= width / w_raw; k[1, 1] *= height / h_raw
    return k


def ray_cast(origin, directions, vertices, triangles):
    """Nearest positive hit distance of rays origin + t * direction with the triangle mesh (inf = no hit).  Stand-in for open3d's
    RaycastingScene.cast_rays()['t_hit'] (float32 rays, smpl_condition.py:124-125)."""
    o = np.asarray(origin, dtype=np.float64).reshape(3)
    d = np.asarray(directions, dtype=np.float32).astype(np.float64)        # the reference casts float32 rays
    v = np.asarray(vertices, dtype=np.float64)
    tri = np.asarray(triangles)
    v0, v1, v2 = v[tri[:, 0]], v[tri[:, 1]], v[tri[:, 2]]
    e1, e2 = v1 - v0, v2 - v0
    out = np.full(d.shape[0], np.inf)
    tv = o[None, :] - v0
    for i in range(d.shape[0]):
        p = np.cross(d[i][None, :], e2)
        det = (e1 * p).sum(1)
        ok = np.abs(det) > 1e-12
        inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
        u = (tv * p).sum(1) * inv
        q = np.cross(tv, e1)
        w = (q * d[i][None, :]).sum(1) * inv
        t = (q * e2).sum(1) * inv
        hit = ok & (u >= 0) & (w >= 0) & (u + w <= 1) & (t > 0)
        if hit.any():
            out[i] = t[hit].min()
    return out


def occlusion(center, keypoints, t_hit, groups, thres_body=0.2, thres_face=0.02, thres_hand=0.2, ignore_body_self_occlusion=False):
    """smpl_condition.py:96-143 for one person: occluded[K] (bool), t_far[K].  With ONE person in the scene every hit carries that
    person's geometry id, so `ignore_body_self_occlusion` (the shipped default, configs/__init__.py:445) clears the body group
    wherever the ray hit anything -- and where it hit nothing the keypoint is not occluded anyway: body keypoints are never culled."""
    c = np.asarray(center, dtype=np.float64).reshape(1, 3)
    t_far = np.linalg.norm(np.asarray(keypoints, dtype=np.float64) - c, axis=1)
    thr = np.choose(groups, [thres_body, thres_hand, thres_face])
    occ = (t_far - t_hit) > thr
    if ignore_body_self_occlusion:
        occ = occ & (groups != 0)
    return occ, t_far


def pose_keypoints(keypoints, vertices, triangles, extrinsic, intrinsics, width, height, use_occlusion_culling=True,
                   smpl_type="smplx", ignore_body_self_occlusion=False):
    """export_pose up to the call of the drawing code (smpl_condition.py:191-224 + to_controlnet_pose :20-79), one person:
    rows (x / W, y / H, dist); a NaN row is a keypoint the drawing code receives as None.  `intrinsics` already size-adjusted."""
    kp = np.asarray(keypoints, dtype=np.float64).reshape(-1, 3)
    ext = np.asarray(extrinsic, dtype=np.float64)
    R, T = ext[:3, :3], ext[:3, 3:4]
    cam = (R @ kp.T + T).T                                    # transform_keypoints_to_novelview with identity source view
    cam[cam[:, 2] < 0] = np.nan                               # smpl_condition.py:208-209
    K = np.asarray(intrinsics, dtype=np.float64)
    h = K @ cam.T
    img = (h[:2] / h[2]).T                                    # project_camera3d_to_2d
    dist = -np.ones(kp.shape[0])
    if use_occlusion_culling:
        center = (np.linalg.inv(R) @ (-T)).reshape(3)
        d = kp - center[None, :]
        d = d / np.linalg.norm(d, axis=1, keepdims=True)
        t_hit = ray_cast(center, d, vertices, triangles)
        occ, dist = occlusion(center, kp, t_hit, keypoint_groups(smpl_type), ignore_body_self_occlusion=ignore_body_self_occlusion)
        img[occ] = np.nan
    W, H = K[0, 2] * 2, K[1, 2] * 2                           # to_controlnet_pose :27
    out = np.stack([img[:, 0] / float(W), img[:, 1] / float(H), dist], axis=1)
    out[np.isnan(img).any(1)] = np.nan
    return out


# ----------------------------------------------------------------------------------------------------------------------
# OpenCV drawing primitives, restated
# ----------------------------------------------------------------------------------------------------------------------
def cv_round(x):
    return int(np.rint(x))          # cvRound: round half to even


def circle_halfwidths(radius):
    """Filled cv2.circle as the union of the horizontal spans drawing.cpp's Circle() emits: hw[dy] = largest |dx| painted on the
    rows cy +- dy (dy = 0..radius), -1 = row untouched."""
    hw = -np.ones(radius + 1, dtype=np.int32)
    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1
    while dx >= dy:
        hw[dy] = max(hw[dy], dx)              # rows cy +- dy span cx +- dx
        hw[dx] = max(hw[dx], dy)              # rows cy +- dx span cx +- dy
        dy += 1
        err += plus
        plus += 2
        mask = -1 if err > 0 else 0           # (err <= 0) - 1
        err -= minus & mask
        dx += mask
        minus -= mask & 2
    return hw


_SIN = np.sin(np.radians(np.arange(0, 451, dtype=np.float64))).astype(np.float32)      # drawing.cpp SinTable (float, integer degrees)


def ellipse2poly(cx, cy, a, b, angle):
    """cv2.ellipse2Poly((cx, cy), (a, b), angle, 0, 360, 1): integer vertices, consecutive duplicates removed."""
    while angle < 0:
        angle += 360
    while angle > 360:
        angle -= 360
    alpha, beta = float(_SIN[450 - angle]), float(_SIN[angle])          # cos, sin of the rotation
    pts, prev = [], None
    for i in range(0, 361):
        x = a * float(_SIN[450 - i]); y = b * float(_SIN[i])
        p = (cv_round(cx + x * alpha - y * beta), cv_round(cy + x * beta + y * alpha))
        if p != prev:
            pts.append(p); prev = p
    if len(pts) == 1:
        pts = [(cx, cy), (cx, cy)]
    return np.asarray(pts, dtype=np.int64)


def convex_poly_mask(poly, H, W):
    """cv2.fillConvexPoly's scan conversion, restated: per image row y the polygon's boundary is cut at y (exact rational x per
    edge; both end points of an edge lying IN the row) and the pixels round(x_leftmost) .. round(x_rightmost) are painted, with
    round(x) = floor(x + 1/2) (drawing.cpp adds XY_ONE/2 before the shift).  OpenCV walks a left and a right edge with a
    fixed-point increment per row, so it can differ by a boundary pixel on long edges."""
    m = np.zeros((H, W), dtype=bool)
    ys = np.arange(H, dtype=np.int64)
    lo = np.full(H, np.iinfo(np.int64).max); hi = np.full(H, np.iinfo(np.int64).min)
    a = poly.astype(np.int64); b = np.roll(a, -1, axis=0)
    for (x1, y1), (x2, y2) in zip(a, b):
        if y1 == y2:
            if 0 <= y1 < H:
                lo[y1] = min(lo[y1], x1, x2); hi[y1] = max(hi[y1], x1, x2)
            continue
        ya, yb = max(min(y1, y2), 0), min(max(y1, y2), H - 1)
        if ya > yb:
            continue
        yy = ys[ya:yb + 1]
        den = y2 - y1
        num = x1 * den + (x2 - x1) * (yy - y1)
        if den < 0:
            den, num = -den, -num
        xr = np.floor_divide(2 * num + den, 2 * den)
        lo[ya:yb + 1] = np.minimum(lo[ya:yb + 1], xr); hi[ya:yb + 1] = np.maximum(hi[ya:yb + 1], xr)
    for y in range(H):
        if lo[y] <= hi[y]:
            xa, xb = max(int(lo[y]), 0), min(int(hi[y]), W - 1)
            if xa <= xb:
                m[y, xa:xb + 1] = True
    return m


def circle_mask(cx, cy, radius, H, W):
    hw = circle_halfwidths(radius)
    m = np.zeros((H, W), dtype=bool)
    for dy in range(radius + 1):
        if hw[dy] < 0:
            continue
        for y in (cy - dy, cy + dy):
            if 0 <= y < H:
                xa, xb = max(cx - hw[dy], 0), min(cx + hw[dy], W - 1)
                if xa <= xb:
                    m[y, xa:xb + 1] = True
    return m


def thick_line_mask(x1, y1, x2, y2, thickness, H, W):
    """Pixels p with 4 * dist(p, segment)^2 <= thickness^2 (integer arithmetic)."""
    r = (thickness + 1) // 2
    xa, xb = max(min(x1, x2) - r, 0), min(max(x1, x2) + r, W - 1)
    ya, yb = max(min(y1, y2) - r, 0), min(max(y1, y2) + r, H - 1)
    m = np.zeros((H, W), dtype=bool)
    if xa > xb or ya > yb:
        return m
    ys, xs = np.mgrid[ya:yb + 1, xa:xb + 1].astype(np.int64)
    abx, aby = x2 - x1, y2 - y1
    apx, apy = xs - x1, ys - y1
    L2 = abx * abx + aby * aby
    tn = apx * abx + apy * aby
    d_a = apx * apx + apy * apy
    d_b = (xs - x2) ** 2 + (ys - y2) ** 2
    t2 = thickness * thickness
    inside = np.where(tn <= 0, 4 * d_a <= t2, np.where(tn >= L2, 4 * d_b <= t2, 4 * (d_a * L2 - tn * tn) <= t2 * L2))
    m[ya:yb + 1, xa:xb + 1] = inside
    return m


def add_weighted_inplace(canvas, mask, color):
    """canvas = cv2.addWeighted(canvas, 0.4, painted copy, 0.6, 0) where the copy is `color` under `mask` (open_pose.py:135-138):
    float32 arithmetic, round half to even."""
    c = canvas[mask].astype(np.float32)
    col = np.asarray(color, dtype=np.float32)[None, :]
    canvas[mask] = np.clip(np.rint(c * np.float32(0.4) + col * np.float32(0.6)), 0, 255).astype(np.uint8)


def hsv_edge_color(ie, n=20):
    """matplotlib.colors.hsv_to_rgb([ie / n, 1, 1]) * 255, saturate_cast<uchar> (cvRound) as cv2 does with a float Scalar."""
    h = ie / float(n)
    i = int(h * 6.0); f = h * 6.0 - i
    p, q, t = 0.0, 1.0 - f, f
    r, g, b = [(1.0, t, p), (q, 1.0, p), (p, 1.0, t), (p, q, 1.0), (t, p, 1.0), (1.0, p, q)][i % 6]
    return tuple(int(np.clip(np.rint(255.0 * v), 0, 255)) for v in (r, g, b))


def _kp(row):
    return None if np.isnan(row[0]) or np.isnan(row[1]) else (float(row[0]), float(row[1]))


def _draw_body(canvas, rows, radius, stickwidth, flip_LR):
    H, W, _ = canvas.shape
    kps = [_kp(r) for r in rows]
    if flip_LR:
        kps = [kps[i] for i in BODY_FLIP]
    for kp, color in zip(kps, BODY_COLORS):
        if kp is None:
            continue
        x, y = int(kp[0] * W), int(kp[1] * H)
        if x > EPS and y > EPS:
            canvas[circle_mask(x, y, radius, H, W)] = color
    for (i1, i2), color in zip(BODY_LIMBS, BODY_COLORS):
        k1, k2 = kps[i1 - 1], kps[i2 - 1]
        if k1 is None or k2 is None:
            continue
        Y = np.array([k1[0], k2[0]]) * float(W)
        X = np.array([k1[1], k2[1]]) * float(H)
        mX, mY = np.mean(X), np.mean(Y)
        length = ((X[0] - X[1]) ** 2 + (Y[0] - Y[1]) ** 2) ** 0.5
        angle = math.degrees(math.atan2(X[0] - X[1], Y[0] - Y[1]))
        poly = ellipse2poly(int(mY), int(mX), int(length / 2), stickwidth, int(angle))
        add_weighted_inplace(canvas, convex_poly_mask(poly, H, W), color)


def _draw_hand(canvas, rows, radius, thickness):
    H, W, _ = canvas.shape
    kps = [_kp(r) for r in rows]
    for kp in kps:
        if kp is None:
            continue
        x, y = int(kp[0] * W), int(kp[1] * H)
        if x > EPS and y > EPS:
            canvas[circle_mask(x, y, radius, H, W)] = (0, 0, 255)
    for ie, (e1, e2) in enumerate(HAND_EDGES):
        k1, k2 = kps[e1], kps[e2]
        if k1 is None or k2 is None:
            continue
        x1, y1, x2, y2 = int(k1[0] * W), int(k1[1] * H), int(k2[0] * W), int(k2[1] * H)
        if x1 > EPS and y1 > EPS and x2 > EPS and y2 > EPS:
            # cv2.line on the canvas, then the 0.4 / 0.6 blend with a copy holding the same line: the line colour itself
            canvas[thick_line_mask(x1, y1, x2, y2, thickness, H, W)] = hsv_edge_color(ie)


def _draw_face(canvas, rows, radius):
    H, W, _ = canvas.shape
    for r in rows:
        kp = _kp(r)
        if kp is None:
            continue
        x, y = int(kp[0] * W), int(kp[1] * H)
        if x > EPS and y > EPS:
            canvas[circle_mask(x, y, radius, H, W)] = (255, 255, 255)


def draw_sizes(H, W):
    """adaptive_draw_poses' size rule (open_pose.py:303-315) -> body radius, stick width, hand radius, hand thickness, face radius."""
    s = [4, 4, 4, 2, 3]
    if H != 512 or W != 512:
        r = (H + W) / 2.0 / 512.0
        s = [max(int(v * r), 1) for v in s]
    return s


def draw_poses(rows, H, W, draw_body=True, draw_hand=True, draw_face=True, flip_LR=False):
    """adaptive_draw_poses for one person, hand_dist_thres=None (what export_pose passes) -> uint8 [H, W, 3] RGB."""
    rows = np.asarray(rows, dtype=np.float64)
    canvas = np.zeros((H, W, 3), dtype=np.uint8)
    br, bs, hr, ht, fr = draw_sizes(H, W)
    if draw_body:
        _draw_body(canvas, rows[:N_BODY], br, bs, flip_LR)
    if draw_hand and rows.shape[0] > N_BODY:
        _draw_hand(canvas, rows[N_BODY:N_BODY + N_HAND], hr, ht)
        _draw_hand(canvas, rows[N_BODY + N_HAND:N_BODY + 2 * N_HAND], hr, ht)
    if draw_face and rows.shape[0] > N_BODY + 2 * N_HAND:
        _draw_face(canvas, rows[N_BODY + 2 * N_HAND:N_KEYPOINTS], fr)
    return canvas


def export_pose(keypoints, vertices, triangles, extrinsic, intrinsics, width, height, **kw):
    """SMPL2Condition.__call__(condition_type='pose') for one person -> uint8 [H, W, 3]."""
    flags = {k: kw.pop(k) for k in ("draw_body", "draw_hand", "draw_face", "flip_LR") if k in kw}
    rows = pose_keypoints(keypoints, vertices, triangles, extrinsic, adjust_intrinsics_size(intrinsics, width, height), width, height,
                          **kw)
    return draw_poses(rows, height, width, **flags)
