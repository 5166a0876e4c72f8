"""Host-side VideoMAE mask samplers with the interface of pretraining/generative/mask.py: ``Generator((T, H, W), ratio)()``
returns a flat float64 vector of T*H*W zeros (visible) and ones (masked), which the step turns into a bool tensor
(pretrain_videomae.py:294-298).

The draws follow the reference's random stream (one in-place ``shuffle`` of a "visible first, masked last" vector per call)
so that a seeded run reproduces its masks; ``rng`` selects a private ``numpy.random.RandomState`` instead of the global one
the reference uses.  Fixture: tests/golden/tube_mask.json.
"""
import numpy as np


def _shuffled_flags(count, masked, rng):
    """`count` flags, the last `masked` of them set, permuted in place by the chosen generator."""
    flags = (np.arange(count) >= count - masked).astype(np.float64)
    (np.random if rng is None else rng).shuffle(flags)
    return flags


class TubeMaskingGenerator:
    """One spatial mask per clip, repeated over every temporal slot (a "tube")."""

    def __init__(self, input_size, mask_ratio, rng=None):
        slots, rows, cols = input_size
        self.slots, self.per_slot = slots, rows * cols
        self.masked_per_slot = int(mask_ratio * self.per_slot)
        self.rng = rng

    @property
    def total_patches(self):
        return self.slots * self.per_slot

    @property
    def total_masks(self):
        return self.slots * self.masked_per_slot

    def __repr__(self):
        return f"TubeMaskingGenerator({self.total_masks} of {self.total_patches} tokens masked)"

    def __call__(self):
        spatial = _shuffled_flags(self.per_slot, self.masked_per_slot, self.rng)
        return np.broadcast_to(spatial, (self.slots, self.per_slot)).reshape(-1).copy()


class RandomMaskingGenerator:
    """Independent mask over all T*H*W tokens."""

    def __init__(self, input_size, mask_ratio, rng=None):
        dims = input_size if isinstance(input_size, tuple) else (input_size,) * 3
        self.num_patches = int(np.prod(dims))
        self.num_mask = int(mask_ratio * self.num_patches)
        self.rng = rng

    def __repr__(self):
        return f"RandomMaskingGenerator({self.num_mask} of {self.num_patches} tokens masked)"

    def __call__(self):
        return _shuffled_flags(self.num_patches, self.num_mask, self.rng)
