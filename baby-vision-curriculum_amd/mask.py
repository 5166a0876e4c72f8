"""Host-side mask samplers with the reference's interface (pretraining/generative/mask.py).

numpy on the host, exactly like the reference: the step consumes the result as a bool tensor
(pretrain_videomae.py:294-298).  An optional ``rng`` makes a run reproducible; with ``rng=None`` the
global numpy RNG is used, as in the reference (which never seeds it).
"""
import numpy as np


class TubeMaskingGenerator:
    """Same spatial mask in every temporal slot (reference mask.py:3-24)."""

    def __init__(self, input_size, mask_ratio, rng=None):
        self.frames, self.height, self.width = input_size
        self.num_patches_per_frame = self.height * self.width
        self.total_patches = self.frames * self.num_patches_per_frame
        self.num_masks_per_frame = int(mask_ratio * self.num_patches_per_frame)
        self.total_masks = self.frames * self.num_masks_per_frame
        self.rng = rng

    def __repr__(self):
        return "Maks: total patches {}, mask patches {}".format(self.total_patches, self.total_masks)

    def __call__(self):
        per_frame = np.hstack([np.zeros(self.num_patches_per_frame - self.num_masks_per_frame),
                               np.ones(self.num_masks_per_frame)])
        (self.rng if self.rng is not None else np.random).shuffle(per_frame)
        return np.tile(per_frame, (self.frames, 1)).flatten()


class RandomMaskingGenerator:
    """Independent mask over all tokens (reference mask.py:26-46)."""

    def __init__(self, input_size, mask_ratio, rng=None):
        if not isinstance(input_size, tuple):
            input_size = (input_size,) * 3
        self.frames, self.height, self.width = input_size
        self.num_patches = self.frames * self.height * self.width
        self.num_mask = int(mask_ratio * self.num_patches)
        self.rng = rng

    def __repr__(self):
        return "Maks: total patches {}, mask patches {}".format(self.num_patches, self.num_mask)

    def __call__(self):
        mask = np.hstack([np.zeros(self.num_patches - self.num_mask), np.ones(self.num_mask)])
        (self.rng if self.rng is not None else np.random).shuffle(mask)
        return mask
