"""The fields of the reference's TrainConfig (/root/reference/configs/__init__.py) that the hot path reads, with the reference's
default values (cross-checked against `TrainConfig()` of the imported reference by tests/golden/capture_golden_r2.py: opt.cfg,
text.cfg and the flag list printed in DESIGN.md).  Any object with the same attributes -- the reference's own pyrallis
dataclasses included -- can be passed wherever a `cfg` is expected."""
from dataclasses import dataclass, field
from typing import Optional, Tuple


@dataclass
class RenderConfig:
    sh_levels: int = 4
    bg_color: Tuple[float, float, float] = (0.0, 0.0, 0.0)       # the GS recipes pass (0.5, 0.5, 0.5): train_w_expr.sh:68,81,94
    n_gaussians_per_triangle: int = 6
    position_lr_init: float = 0.00016
    position_lr_final: float = 0.0000016
    feature_lr: float = 0.0125
    opacity_lr: float = 0.01
    scaling_lr: float = 0.0025
    rotation_lr: float = 0.001
    lbs_lr: float = 0.0001
    betas_lr: float = 0.01
    init_scale: float = 0.001
    max_scale: float = 0.01
    init_offset: float = 0.01
    learn_positions: bool = True
    learn_scales: bool = True
    learn_quaternions: bool = True
    learn_lbs_weights: bool = False
    learn_hand_betas: bool = False
    learn_face_betas: bool = False
    learn_mesh_bary_coords: bool = True
    learn_mesh_vertex_coords: bool = False
    learn_mesh_scales: bool = True
    use_joint_shape_offsets: bool = False
    use_vertex_shape_offsets: bool = False
    use_vertex_pose_offsets: bool = False
    use_non_rigid_offsets: bool = True
    use_non_rigid_scales: bool = True
    use_non_rigid_rotations: bool = False
    non_rigid_scale_mode: str = 'add'
    non_rigid_rotation_mode: str = 'add'
    use_nerf_encoded_position: bool = True
    render_mesh_binding_3d_gaussians_only: bool = False
    render_unconstrained_3d_gaussians_only: bool = False
    use_zero_scales: bool = False
    use_constant_colors: Optional[Tuple[float, float, float]] = None
    use_constant_opacities: Optional[float] = None
    use_fixed_n_gaussians: Optional[int] = None
    avatar_transl: Optional[str] = None
    avatar_scale: Optional[str] = None
    use_densifier: bool = False                      # configs/__init__.py:159-171 (off in every shipped recipe)
    densify_from_iter: Optional[int] = None
    densify_until_iter: Optional[int] = None
    densify_grad_threshold: float = 100              # 0.0002 for MSE, 100 for SDS
    densify_disable_clone: bool = False
    densify_disable_split: bool = False
    densify_disable_prune: bool = False
    densify_disable_reset: bool = True
    enable_grad_prune: bool = False
    always_animate: bool = True
    spatial_scale: Optional[float] = None


@dataclass
class OptimConfig:
    iters: int = 5000
    fp16: bool = False


@dataclass
class NeRFConfig:
    lr: float = 0.001
    bound: float = 2.0


@dataclass
class GuideConfig:
    guidance_scale: float = 50.0
    guidance_adjust: str = 'constant'
    sds_loss_type: str = 'sds'
    sds_weight_type: str = 'sjc'
    use_negative_text: bool = True
    min_timestep: float = 0.02
    max_timestep: float = 0.98
    time_sampling: str = 'uniform'
    input_interpolate: bool = True
    controlnet_scale: float = 1.0
    lambda_guidance: float = 1.0
    pgc_clip_rgb: float = -1
    pgc_suppress_type: int = 0
    grad_rgb_clip: bool = False
    grad_rgb_norm: bool = False
    grad_rgb_clip_scale: float = 3.0
    grad_latent_clip: bool = False
    grad_latent_clip_scale: float = 3.0
    grad_latent_norm: bool = False
    grad_latent_nan_to_num: bool = False
    grad_viz: bool = False
    text: str = ""


@dataclass
class PromptConfig:
    text_augmentation: bool = True
    text_augmentation_mode: str = 'dreamwaltz-g'
    angle_front: float = 90.0
    angle_overhead: float = 60.0
    scene: str = 'canonical'
    # condition image (configs/__init__.py:414,441-446)
    smpl_type: str = 'smplx'
    use_occlusion_culling: bool = True
    draw_body_keypoints: bool = True
    draw_hand_keypoints: bool = True
    draw_face_landmarks: bool = False
    ignore_body_self_occlusion: bool = True
    openpose_left_right_flip: bool = False


@dataclass
class TrainConfig:
    render: RenderConfig = field(default_factory=RenderConfig)
    optim: OptimConfig = field(default_factory=OptimConfig)
    nerf: NeRFConfig = field(default_factory=NeRFConfig)
    guide: GuideConfig = field(default_factory=GuideConfig)
    prompt: PromptConfig = field(default_factory=PromptConfig)
    stage: str = 'gs'
    device: str = 'cuda'
