"""Software pipeline over consecutive batches: while the feature half of batch i (ball query, fused
SA/FP kernels, PDM neck) runs on the main stream, the coordinate-only sampling chain of batch i+1
(FPS + gather for the four SA layers) runs on a side HIP stream.

The coordinate-only chain also holds the ball-query indices of every SA scale and the three-NN indices and
weights of every FP module (none of them reads a feature), so the feature path is MLP kernels only.

Why: FPS is one workgroup per cloud — at bs=32 it occupies 32 of the 256 CUs for milliseconds and is
the longest dependency chain of the step, while everything else is wide.  Nothing in the sampling
chain depends on features, so it is hoisted one batch ahead; every step still does one full batch of
every kind of work (the sampling it consumes was produced by the previous step).
The PDM neck, which reads only SA outputs, is likewise started on its own stream as soon as the SA stack
is done and runs beside the feature-propagation layers.
Works eagerly and under hipGraph capture (fork/join of the two streams inside the captured region;
the hand-over buffers are static).

depth=2 splits the sampling chain over two batches in flight: while batch i's features run, level 1 of
batch i+2 (the long 16384 -> 4096 FPS) runs on one side stream and levels 2..L of batch i+1 (which start
from the level-1 result handed over by the previous step) on another.  The longest serial chain beside
the main stream is then the level-1 FPS alone instead of the whole chain; one more batch is in flight
(latency +1 step), each step still does one full batch of every kind of work.
"""
import torch

from . import _native


def _flat(d):
    """Tensors of a coordinate_levels() dict in a fixed order."""
    out = list(d['sampled_xyz'])
    for lvl in d['ball_idx']:
        out.extend(lvl)
    for idx, w in d['fp_interp']:
        out.extend((idx, w))
    return out


def _merge(a, b):
    return {k: list(a[k]) + list(b[k]) for k in ('sampled_xyz', 'ball_idx', 'fp_interp')}


class PipelinedHotPath:
    def __init__(self, backbone, neck=None, depth=1):
        assert depth in (1, 2)
        self.backbone = backbone
        self.neck = neck
        self.depth = depth
        self.side = torch.cuda.Stream()
        self.neck_stream = torch.cuda.Stream()
        # depth 2: the short tail of the next batch's chain runs on the neck's stream ahead of the neck (which cannot
        # start before the SA stack is done anyway) — hipGraph replay starts a fourth parallel branch late
        self.side2 = self.neck_stream if depth == 2 else None
        self.cur = None   # static hand-over buffers: coordinate-only results of the batch about to be processed
        self.half = None  # depth 2: level-1 results of the batch after that

    @staticmethod
    def _xyz(points, batch_size):
        return points[:, 1:4].contiguous().view(batch_size, -1, 3)

    @staticmethod
    def _clone(d):
        return {'sampled_xyz': [t.clone() for t in d['sampled_xyz']],
                'ball_idx': [[t.clone() for t in lvl] for lvl in d['ball_idx']],
                'fp_interp': [(i.clone(), w.clone()) for i, w in d['fp_interp']]}

    @torch.no_grad()
    def prime(self, points, batch_size, points_next=None):
        """Coordinate-only chain for the first batch (not overlapped with anything); depth 2 also needs level 1 of
        the second batch (`points_next`, default: the same points)."""
        nlev = len(self.backbone.SA_modules)
        self.cur = self._clone(self.backbone.coordinate_levels(self._xyz(points, batch_size), 0, nlev))
        if self.depth == 2:
            nxt = points if points_next is None else points_next
            self.half = self._clone(self.backbone.coordinate_levels(self._xyz(nxt, batch_size), 0, 1))

    @torch.no_grad()
    def step(self, points_cur, points_next, batch_size, extra=None, points_next2=None):
        """Features of `points_cur` (its coordinate-only results are in self.cur) || coordinate-only chain of
        `points_next` (depth 2: || levels 2..L of `points_next` || level 1 of `points_next2`).
        The coordinate-only chain = FPS + gather, ball-query indices and three-NN weights of every level: nothing in
        it reads a feature, so only the MLP kernels (and the neck) stay on the feature path."""
        assert self.cur is not None, "call prime() first"
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)
        nlev = len(self.backbone.SA_modules)
        if self.depth == 2:
            assert points_next2 is not None and self.half is not None
            self.side2.wait_stream(main)
            with torch.cuda.stream(self.side):      # the long one: 16384 -> 4096 FPS of the batch after next
                nxt_half = self.backbone.coordinate_levels(self._xyz(points_next2, batch_size), 0, 1)
            with torch.cuda.stream(self.side2):     # the tail of the next batch's chain
                nxt = _merge(self.half, self.backbone.coordinate_levels(self.half['sampled_xyz'][0], 1, nlev))
        else:
            with torch.cuda.stream(self.side):
                nxt = self.backbone.coordinate_levels(self._xyz(points_next, batch_size), 0, nlev)
        bd = {'batch_size': batch_size, 'points': points_cur}
        bd.update(self.cur)
        if extra:
            bd.update(extra)
        if self.neck is not None:
            # the neck reads only SA outputs: run it beside the FP layers
            def start_neck(d):
                self.neck_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.neck_stream):
                    self.neck(d)
            bd['after_sa_hook'] = start_neck
        bd = self.backbone(bd)
        if self.neck is not None:
            main.wait_stream(self.neck_stream)
        main.wait_stream(self.side)
        if self.depth == 2:
            main.wait_stream(self.side2)
        # hand-over into the static buffers (addresses must not change under hipGraph replay): one multi-tensor copy
        _native.copy_many(_flat(self.cur), _flat(nxt))
        if self.depth == 2:   # a second launch: self.half is a SOURCE of the first one (level 1 of `nxt`)
            _native.copy_many(_flat(self.half), _flat(nxt_half))
        return bd
