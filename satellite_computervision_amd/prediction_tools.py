"""Overlap-tile (sliding window) inference: drop-in for the chip helpers of the reference's
utils/prediction_tools.py:87-156, with the per-chip batch-1 `m.predict` loop replaced by
batched device inference.

Index arithmetic is restated exactly (including the exclusive range stop that skips a chip
ending on the image edge and the never-predicted buff//2 border), so
outputs are identical to the reference's for the same model outputs.
"""
import numpy as np


def generate_chip_indices(arr, buff=128, kernel=256):
    """utils/prediction_tools.py:87-109.  Returns [(y, x)] upper-left corners of the kernel-sized
    centres; arr is (H, W, C)."""
    H, W, C = arr.shape
    side = buff + kernel
    x_buff = y_buff = buff // 2
    y_indices = list(range(y_buff, H - side, kernel))
    x_indices = list(range(x_buff, W - side, kernel))
    return [(y_index, x_index) for y_index in y_indices for x_index in x_indices]


def extract_chips(arr, buff=128, kernel=256):
    """utils/prediction_tools.py:111-131 (as coded: the (y, x) tuples are unpacked as `x, y`, :127)."""
    x_buff = y_buff = buff // 2
    chips = []
    for x, y in generate_chip_indices(arr, buff, kernel):
        chips.append(arr[y - y_buff:y + kernel + y_buff, x - x_buff:x + kernel + x_buff, :])
    return chips


def predict_chips(arr, chip_indices, template, m, kernel=256, buff=128, batch_size=16, channel=0):
    """utils/prediction_tools.py:133-156: predict every (kernel+buff)^2 chip and accumulate the centre
    kernel^2 of one output channel into `template` (+=).

    Differences from the reference, by design: chips are predicted `batch_size` at a time on the
    device instead of one `m.predict` per chip; a model with list outputs ([probs, classes],
    get_unet_model) contributes its first output (the reference indexes the list as if it were an
    array, Appendix B Q11); `channel` selects the class probability written (reference: 0)."""
    y_buff = x_buff = buff // 2
    idx = list(chip_indices)
    for s in range(0, len(idx), batch_size):
        part = idx[s:s + batch_size]
        chips = np.stack([arr[y - y_buff:y + kernel + y_buff, x - x_buff:x + kernel + x_buff, :] for y, x in part])
        preds = m.predict(chips, batch_size=len(part), verbose=0)
        if isinstance(preds, (list, tuple)):
            preds = preds[0]
        for k, (y, x) in enumerate(part):
            template[y:y + kernel, x:x + kernel] += preds[k, y_buff:(kernel + y_buff), x_buff:(kernel + x_buff), channel]
    return template


def predict_chips_sharded(arr, chip_indices, template, m, kernel=256, buff=128, batch_size=16, channel=0):
    """Multi-GPU form of `predict_chips` (one process per GPU): the chip list of `generate_chip_indices` is split
    round-robin over the ranks, every rank predicts its share into a zero template -- no collective on the data path, the chips
    are independent units (utils/prediction_tools.py:147-154 loops over them one by one) -- and the per-rank templates are
    summed once (disjoint centres, so the sum equals the single-process result).  Every rank returns the full template.
    Without an initialised process group it is `predict_chips`."""
    import torch
    from . import parallel
    rank, world = (parallel.dist.get_rank(), parallel.dist.get_world_size()) if parallel.dist.is_initialized() else (0, 1)
    if world == 1:
        return predict_chips(arr, chip_indices, template, m, kernel, buff, batch_size, channel)
    mine = parallel.shard_list(list(chip_indices), rank, world)
    part = predict_chips(arr, mine, np.zeros_like(template), m, kernel, buff, batch_size, channel)
    dev = 'cuda' if parallel.dist.get_backend() == 'nccl' else 'cpu'
    t = torch.from_numpy(np.ascontiguousarray(part)).to(dev)
    parallel.reduce_templates(t)
    template += t.cpu().numpy().astype(template.dtype)
    return template
