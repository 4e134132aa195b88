"""View-dependent prompt selection of the SDS step: mirror of TextAugmentation
(/root/reference/core/guidance/text.py:36-154).  Host-side logic (one azimuth / elevation pair per step, batch size 1: the
reference's own chained comparisons on tensors are only valid there -- checklist Q11); the text ENCODER (CLIP) is outside the hot
path (embeddings are computed once at start-up, text.py:13-33)."""
from typing import List, Optional, Tuple

import torch


class TextAugmentation:
    def __init__(self, text: str, cfg) -> None:
        self.mode = cfg.text_augmentation_mode
        self.azimuth_range, self.elevation_range = self.get_angle_ranges(cfg.angle_front, cfg.angle_overhead)
        self.texts = self.get_view_augmented_texts(text)
        if self.mode in ('dreamwaltz', 'dreamwaltz-g'):
            texts, part2index = self.get_body_part_augmented_texts(text, len(self.texts))
            self.texts.extend(texts)
            self.part2index = part2index
        else:
            self.part2index = None

    @staticmethod
    def get_angle_ranges(angle_front, angle_overhead) -> Tuple[List[float], List[float]]:
        assert 0 <= angle_front <= 180
        azimuth_range = sorted([angle_front / 2, 180 - angle_front / 2, 180 + angle_front / 2, 360 - angle_front / 2])
        assert 0 <= angle_overhead <= 90
        elevation_range = sorted([angle_overhead, 180 - angle_overhead])
        return azimuth_range, elevation_range

    def get_view_augmented_texts(self, text: str) -> list:
        if self.mode == 'prefix':
            names = ['front view of {t}', 'side view of {t}', 'backside view of {t}', 'side view of {t}', 'overhead view of {t}',
                     'bottom view of {t}']
        elif self.mode == 'suffix':
            names = ['{t}, front view', '{t}, side view', '{t}, back view', '{t}, side view', '{t}, overhead view', '{t}, bottom view']
        elif self.mode == 'dreamwaltz':
            names = ['front view of {t}', 'side view of {t}', 'back view of {t}', 'side view of {t}', 'overhead view of {t}',
                     'bottom view of {t}']
        elif self.mode == 'dreamwaltz-g':
            names = ['front view of {t}', 'left side view of {t}', 'back view of {t}', 'right side view of {t}', 'overhead view of {t}',
                     'bottom view of {t}']
        else:
            raise NotImplementedError(f'{self.mode}')
        return [n.format(t=text) for n in names]

    @staticmethod
    def get_body_part_augmented_texts(text: str, start_idx: int):
        parts = ['head', 'face', 'arm_left', 'arm_right', 'hand_left', 'hand_right', 'foot_left', 'foot_right']
        names = ['head', 'face', 'left arm', 'right arm', 'left hand', 'right hand', 'left foot', 'right foot']
        return [f'{n} of {text}' for n in names], {p: start_idx + i for i, p in enumerate(parts)}

    def __call__(self, azim, elev, part: Optional[str] = None) -> torch.Tensor:
        """azim, elev: one-element tensors (or floats), degrees.  front 0 / side-left 1 / back 2 / side-right 3 by azimuth,
        overridden by overhead 4 / bottom 5 by elevation (text.py:125-154)."""
        a = float(azim.reshape(-1)[0]) if torch.is_tensor(azim) else float(azim)
        e = float(elev.reshape(-1)[0]) if torch.is_tensor(elev) else float(elev)
        az, el = self.azimuth_range, self.elevation_range
        res = 0
        if az[0] <= a < az[1]:
            res = 1
        elif az[1] <= a < az[2]:
            res = 2
        elif az[2] <= a < az[3]:
            res = 3
        if e < el[0]:
            res = 4
        if e > el[1]:
            res = 5
        return torch.tensor([res], dtype=torch.long)
