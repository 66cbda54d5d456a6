"""Software pipeline over consecutive batches: while the feature half of batch i (ball query, fused
SA/FP kernels, PDM neck) runs on the main stream, the coordinate-only sampling chain of batch i+1
(FPS + gather for the four SA layers) runs on a side HIP stream.

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


class PipelinedHotPath:
    def __init__(self, backbone, neck=None, depth=1):
        assert depth in (1, 2)
        self.backbone = backbone
        self.neck = neck
        self.depth = depth
        self.side = torch.cuda.Stream()
        self.side2 = torch.cuda.Stream() if depth == 2 else None
        self.neck_stream = torch.cuda.Stream()
        self.cur = None   # static hand-over buffers: sampled xyz of the batch about to be processed
        self.half = None  # depth 2: level-1 sampled xyz of the batch after that

    @staticmethod
    def _xyz(points, batch_size):
        return points[:, 1:4].contiguous().view(batch_size, -1, 3)

    @torch.no_grad()
    def prime(self, points, batch_size, points_next=None):
        """Sampling chain for the first batch (not overlapped with anything); depth 2 also needs level 1 of
        the second batch (`points_next`, default: the same points)."""
        self.cur = [t.clone() for t in self.backbone.sample_chain(self._xyz(points, batch_size))]
        if self.depth == 2:
            nxt = points if points_next is None else points_next
            self.half = self.backbone.sample_levels(self._xyz(nxt, batch_size), 0, 1)[0].clone()

    @torch.no_grad()
    def step(self, points_cur, points_next, batch_size, extra=None, points_next2=None):
        """Features of `points_cur` (its sampling is in self.cur) || sampling of `points_next`
        (depth 2: || levels 2..L of `points_next` || level 1 of `points_next2`)."""
        assert self.cur is not None, "call prime() first"
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)
        nlev = len(self.backbone.SA_modules)
        if self.depth == 2:
            assert points_next2 is not None and self.half is not None
            self.side2.wait_stream(main)
            with torch.cuda.stream(self.side):      # the long one: 16384 -> 4096 of the batch after next
                nxt_half = self.backbone.sample_levels(self._xyz(points_next2, batch_size), 0, 1)[0]
            with torch.cuda.stream(self.side2):     # the tail of the next batch's chain
                nxt = [self.half] + self.backbone.sample_levels(self.half, 1, nlev)
        else:
            with torch.cuda.stream(self.side):
                nxt = self.backbone.sample_chain(self._xyz(points_next, batch_size))
        bd = {'batch_size': batch_size, 'points': points_cur, 'sampled_xyz': self.cur}
        if extra:
            bd.update(extra)
        if self.neck is not None:
            # the neck reads only SA outputs: run it (atomic-bound scatter) beside the FP layers (MFMA-bound)
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
        for c, n in zip(self.cur, nxt):
            c.copy_(n)
        if self.depth == 2:
            self.half.copy_(nxt_half)
        return bd
