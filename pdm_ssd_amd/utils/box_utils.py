"""Box helpers the point head needs (boxes are rows [x, y, z, dx, dy, dz, heading, ...], centre-based)."""
import torch


def enlarge_box3d(boxes3d, extra_width=(0, 0, 0)):
    """/root/reference/pcdet/utils/box_utils.py:187-201 — a copy of the boxes with `extra_width` added to the three
    sizes (the ring between a box and its enlarged twin is the head's 'ignore' zone)."""
    large = boxes3d.clone()
    large[:, 3:6] += boxes3d.new_tensor(extra_width)[None, :]
    return large
