"""RigidTransform with the surface the reference reads (/root/reference/core/human/inverse_lbs.py:15-260), bound to the HIP
kernels for the two calls that touch every Gaussian (boundary B3):

    .transform_points(points, weights=[N,J])                                  -> lbs.hip k_blend_fwd / k_blend_bwd
    .transform_quaternions(q, weights=[N,J], flip_rotation_axis=True)         -> same kernels (row-flipped quaternion path)

Everything else (compose, inverse, weight, index, squeeze, the `indices=` gathers, the 'matrix' / 'quaternion' rotation modes) is
small 4x4 algebra on at most [V,4,4] tensors and stays thin element-wise torch code -- written without `@` so that no BLAS
library kernel is launched for a 4x4 product.  Semantics kept on purpose (SURVEY checklist): `inverse()` overwrites the last row
of its SOURCE in place (Q7), `squeeze()` mutates and returns self (Q7), blended rotations are not re-orthonormalised (Q2), rows
1,2 are negated before and after the blend in the flip path (Q3), `_inverse_transform_points` is a general 3x3 inverse (Q8).
"""
from typing import Optional

import torch
from torch import Tensor

from . import lbs as lbs_ops


def mm4(a: Tensor, b: Tensor) -> Tensor:
    """a @ b for [...,k,k] operands as broadcast multiply + sum (element-wise kernels only)."""
    return (a.unsqueeze(-1) * b.unsqueeze(-3)).sum(-2)


def quaternion_to_matrix(q: Tensor) -> Tensor:
    """pytorch3d.transforms.quaternion_to_matrix: real-first, scaled by 2/|q|^2 (valid for non-unit q)."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def _sqrt_positive_part(x: Tensor) -> Tensor:
    return torch.where(x > 0, torch.sqrt(torch.clamp(x, min=1e-38)), torch.zeros_like(x))


def matrix_to_quaternion(matrix: Tensor) -> Tensor:
    """pytorch3d.transforms.matrix_to_quaternion: four candidates, arg-max of the |q| estimates, divide by 2 max(q_abs, 0.1)."""
    m = matrix.reshape(matrix.shape[:-2] + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m, -1)
    q_abs = _sqrt_positive_part(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22,
                                             1.0 - m00 - m11 + m22], dim=-1))
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    cand = cand / (2.0 * q_abs[..., None].clamp_min(0.1))
    idx = q_abs.argmax(dim=-1)
    return torch.gather(cand, -2, idx[..., None, None].expand(idx.shape + (1, 4))).squeeze(-2)


def standardize_quaternion(q: Tensor) -> Tensor:
    return torch.where(q[..., 0:1] < 0, -q, q)


def quaternion_multiply(a: Tensor, b: Tensor) -> Tensor:
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    o = torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), -1)
    return standardize_quaternion(o)


class RigidTransform:
    def __init__(self, R: Optional[Tensor] = None, T: Optional[Tensor] = None, SE3: Optional[Tensor] = None):
        if SE3 is None:
            SE3 = self._build_SE3(R, T)
        self.SE3 = SE3
        self.R = SE3[..., :3, :3]
        self.T = SE3[..., :3, 3]

    @staticmethod
    def _build_SE3(R: Optional[Tensor], T: Optional[Tensor]) -> Tensor:
        ref = R if R is not None else T
        lead = R.shape[:-2] if R is not None else T.shape[:-1]
        SE3 = torch.eye(4, dtype=ref.dtype, device=ref.device).expand(*lead, 4, 4).clone()
        if R is not None:
            SE3[..., :3, :3] = R
        if T is not None:
            SE3[..., :3, 3] = T
        return SE3

    # -- algebra ----------------------------------------------------------------------------------------------------
    @staticmethod
    def _invert_transformation(SE3: Tensor) -> Tensor:
        SE3[..., 3, :] = torch.tensor([0, 0, 0, 1], dtype=SE3.dtype, device=SE3.device)      # in place on the source (Q7)
        Rt = SE3[..., :3, :3].transpose(-1, -2)
        out = torch.zeros_like(SE3)
        out[..., :3, :3] = Rt
        out[..., :3, 3] = -(Rt * SE3[..., :3, 3].unsqueeze(-2)).sum(-1)
        out[..., 3, 3] = 1.0
        return out

    def inverse(self) -> "RigidTransform":
        return RigidTransform(SE3=self._invert_transformation(self.SE3))

    def compose(self, *others: "RigidTransform") -> "RigidTransform":
        SE3 = self.SE3.clone()
        for other in others:
            if not isinstance(other, RigidTransform):
                raise ValueError("Only possible to compose RigidTransform objects; got %s" % type(other))
            SE3 = mm4(other.SE3, SE3)
        return RigidTransform(SE3=SE3)

    @staticmethod
    def correct_rotation_matrices(R: Tensor) -> Tensor:
        Q, _ = torch.linalg.qr(R)
        return Q * torch.sign(torch.det(Q)).unsqueeze(-1).unsqueeze(-1)

    def index(self, indices: Tensor) -> "RigidTransform":
        return RigidTransform(SE3=self.SE3[indices])

    def weight(self, weights: Tensor, qr_correct: bool = False) -> "RigidTransform":
        if qr_correct:
            R = torch.einsum('nj,jkl->nkl', weights, self.R)
            T = torch.einsum('nj,jk->nk', weights, self.T)
            return RigidTransform(R=self.correct_rotation_matrices(R), T=T)
        return RigidTransform(SE3=torch.einsum('nj,jkl->nkl', weights, self.SE3))

    @staticmethod
    def _transform_points(pts: Tensor, R: Tensor, T: Tensor) -> Tensor:
        return (R * pts.unsqueeze(-2)).sum(-1) + T

    @staticmethod
    def _inverse_transform_points(pts: Tensor, R: Tensor, T: Tensor) -> Tensor:
        return (torch.inverse(R) * (pts - T).unsqueeze(-2)).sum(-1)

    # -- the two per-Gaussian calls ------------------------------------------------------------------------------------
    def transform_points(self, points: Tensor, indices: Optional[Tensor] = None, weights: Optional[Tensor] = None) -> Tensor:
        assert indices is None or weights is None
        if weights is not None and self.SE3.dim() == 3 and points.is_cuda:
            return lbs_ops.lbs_blend(self.SE3, weights, points, None, normalize_weights=False)     # HIP: k_blend_fwd
        R, T = self.R, self.T
        if indices is not None:
            R, T = R[indices], T[indices]
        if weights is not None:
            R = torch.einsum('nj,jkl->nkl', weights, R)
            T = torch.einsum('nj,jk->nk', weights, T)
        return self._transform_points(points, R, T)

    def transform_quaternions(self, quaternions: Tensor, indices: Optional[Tensor] = None, weights: Optional[Tensor] = None,
                              rotation_mode: str = 'quaternion', flip_rotation_axis: bool = False) -> Tensor:
        assert indices is None or weights is None
        if flip_rotation_axis and weights is not None and self.SE3.dim() == 3 and quaternions.is_cuda:
            return lbs_ops.lbs_blend_quaternions(self.SE3, weights, quaternions)                  # HIP: k_blend_fwd (Q2, Q3)
        R = self.R
        if indices is not None:
            R = self.R[indices]
        if weights is not None:
            R = torch.einsum('nj,jkl->nkl', weights, self.R)
        if flip_rotation_axis:
            flip = torch.tensor([1.0, -1.0, -1.0], dtype=R.dtype, device=R.device)[None, :, None]
            return matrix_to_quaternion(mm4(R, quaternion_to_matrix(quaternions) * flip) * flip)
        if rotation_mode == 'matrix':
            return matrix_to_quaternion(mm4(R, quaternion_to_matrix(quaternions)))
        elif rotation_mode == 'quaternion':
            return quaternion_multiply(matrix_to_quaternion(R), quaternions)
        assert 0, rotation_mode

    def __repr__(self) -> str:
        return f"SE3: {self.SE3},\r\nR: {self.R},\r\nT: {self.T}"

    def squeeze(self, dim=0):
        self.SE3 = self.SE3.squeeze(dim=dim)
        self.R = self.R.squeeze(dim=dim)
        self.T = self.T.squeeze(dim=dim)
        return self
