"""Host-side multi-block mask sampler for the JEPA step, with the interface of pretraining/predictive/mask.py
(``MaskCollator(...)`` used as the DataLoader ``collate_fn``, ``update_masks``; pretrain_jepa.py:226-235,329-340).

It has to reproduce the reference's random stream to reproduce its masks (fixture: tests/golden/jepa.json,
``mask_collator``): per call one generator seeded with a counter shared by all loader workers yields the target-block and
context-block shapes (one uniform draw each); block corners then come from the GLOBAL torch generator, row before column,
target blocks of a sample first, then its context blocks, redrawing while a block keeps too few tokens.  Everything else
(how rectangles, keep-lists and the batch tensors are built) is this package's own.
"""
import math
from logging import getLogger
from multiprocessing import Value

import torch
from torch.utils.data import default_collate

logger = getLogger()


def update_masks(masks, image_size, patch_size, num_frames, tubelet_size, isencoder=False):
    """Spatial token indices -> indices into the (T, H, W) token grid: context tokens live in temporal slot 0, target tokens
    in the last slot.  Shifts the tensors in place (the caller keeps using the same list), like the reference."""
    per_slot = (image_size // patch_size) ** 2
    slot = 0 if isencoder else num_frames // tubelet_size - 1
    for m in masks:
        m += slot * per_slot
    return masks


class MaskCollator(object):
    def __init__(self, input_size=(224, 224), patch_size=16, enc_mask_scale=(0.2, 0.8), pred_mask_scale=(0.2, 0.8),
                 aspect_ratio=(0.3, 3.0), nenc=1, npred=2, min_keep=4, allow_overlap=False):
        size = input_size if isinstance(input_size, tuple) else (input_size, input_size)
        self.patch_size = patch_size
        self.height, self.width = size[0] // patch_size, size[1] // patch_size
        self.enc_mask_scale, self.pred_mask_scale, self.aspect_ratio = enc_mask_scale, pred_mask_scale, aspect_ratio
        self.nenc, self.npred, self.min_keep, self.allow_overlap = nenc, npred, min_keep, allow_overlap
        self._itr_counter = Value('i', -1)          # lives in shared memory: one sequence of seeds for all workers

    def step(self):
        with self._itr_counter.get_lock():
            self._itr_counter.value += 1
            return self._itr_counter.value

    # ---- block shapes: ONE uniform number fixes both the area fraction and the aspect ratio of a block
    def _block_shape(self, gen, area_range, ratio_range):
        u = torch.rand(1, generator=gen).item()
        area = int(self.height * self.width * (area_range[0] + u * (area_range[1] - area_range[0])))
        ratio = ratio_range[0] + u * (ratio_range[1] - ratio_range[0])
        rows = min(int(round(math.sqrt(area * ratio))), self.height - 1)
        cols = min(int(round(math.sqrt(area / ratio))), self.width - 1)
        return rows, cols

    # ---- one block: corner from the global generator, kept tokens = the rectangle minus the forbidden cells
    def _place_block(self, shape, forbidden=()):
        rows, cols = shape
        allowed = list(forbidden)                    # 0/1 grids the block may not leave (complements of the target blocks)
        misses = 0
        while True:
            top = int(torch.randint(0, self.height - rows, (1,)))
            left = int(torch.randint(0, self.width - cols, (1,)))
            grid = torch.zeros((self.height, self.width), dtype=torch.bool)
            grid[top:top + rows, left:left + cols] = True
            for region in allowed:
                grid &= region
            kept = torch.nonzero(grid.reshape(-1)).reshape(-1)
            if kept.numel() > self.min_keep:
                outside = torch.ones((self.height, self.width), dtype=torch.bool)
                outside[top:top + rows, left:left + cols] = False
                return kept, outside
            misses += 1
            if misses % 20 == 0 and allowed:          # 20 failed corners: stop honouring the most recent target block
                allowed.pop()
                logger.warning("MaskCollator: no valid context block, relaxing the overlap constraint (%d regions left)", len(allowed))

    def __call__(self, batch):
        collated = default_collate(batch)
        gen = torch.Generator()
        gen.manual_seed(self.step())
        target_shape = self._block_shape(gen, self.pred_mask_scale, self.aspect_ratio)
        context_shape = self._block_shape(gen, self.enc_mask_scale, (1.0, 1.0))

        targets, contexts = [], []                    # per sample: list of index tensors
        for _ in range(len(batch)):
            placed = [self._place_block(target_shape) for _ in range(self.npred)]
            targets.append([kept for kept, _ in placed])
            forbidden = () if self.allow_overlap else [outside for _, outside in placed]
            contexts.append([self._place_block(context_shape, forbidden)[0] for _ in range(self.nenc)])

        def to_batches(per_sample, count):
            keep = min(t.numel() for sample in per_sample for t in sample)      # every row is cut to the shortest list
            return [torch.stack([sample[k][:keep] for sample in per_sample]) for k in range(count)]

        return collated, to_batches(contexts, self.nenc), to_batches(targets, self.npred)
