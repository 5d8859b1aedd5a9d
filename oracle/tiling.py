"""Restatement of the reference's overlap-tile inference helpers.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PINNED against the
reference's own function bodies executed in the build container
(tests/golden/tiling_reference.npz) and SURVEY.md Appendix D.

Follows /root/reference/utils/prediction_tools.py:
  generate_chip_indices :87-109
  extract_chips         :111-131
  predict_chips         :133-156
"""
import numpy as np


def generate_chip_indices(arr, buff=128, kernel=256):
    """(y, x) upper-left corners of the kernel-sized centres (:101-109).

    Grid starts at buff//2, steps by `kernel`, and the range stop
    H - (buff + kernel) is exclusive, so a chip that would end exactly on the
    image edge is skipped (Appendix B Q12).
    """
    H, W, C = arr.shape
    side = buff + kernel
    half = buff // 2
    ys = list(range(half, H - side, kernel))
    xs = list(range(half, W - side, kernel))
    return [(y, x) for y in ys for x in xs]


def extract_chips(arr, buff=128, kernel=256):
    """(:120-131).  The reference unpacks the (y, x) tuples as `x, y` (:127), i.e.
    slices rows with the x index and columns with the y index; restated as coded."""
    half = buff // 2
    chips = []
    for x, y in generate_chip_indices(arr, buff, kernel):
        chips.append(arr[y - half:y + kernel + half, x - half:x + kernel + half, :])
    return chips


def predict_chips(arr, chip_indices, template, predict_fn, kernel=256, buff=128):
    """(:145-156) serial batch-1 predict of each (kernel+buff)^2 chip; the centre
    kernel^2 of channel 0 is accumulated (+=) into `template`."""
    half = buff // 2
    for y, x in chip_indices:
        chip = arr[y - half:y + kernel + half, x - half:x + kernel + half, :]
        preds = predict_fn(np.array([chip]))
        template[y:y + kernel, x:x + kernel] += preds[0, half:kernel + half, half:kernel + half, 0]
    return template
