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
"""
import torch


class PipelinedHotPath:
    def __init__(self, backbone, neck=None):
        self.backbone = backbone
        self.neck = neck
        self.side = torch.cuda.Stream()
        self.neck_stream = torch.cuda.Stream()
        self.cur = None  # static hand-over buffers: sampled xyz of the batch about to be processed

    @staticmethod
    def _xyz(points, batch_size):
        return points[:, 1:4].contiguous().view(batch_size, -1, 3)

    @torch.no_grad()
    def prime(self, points, batch_size):
        """Sampling chain for the first batch (not overlapped with anything)."""
        self.cur = [t.clone() for t in self.backbone.sample_chain(self._xyz(points, batch_size))]

    @torch.no_grad()
    def step(self, points_cur, points_next, batch_size, extra=None):
        """Features of `points_cur` (its sampling is in self.cur) || sampling of `points_next`."""
        assert self.cur is not None, "call prime() first"
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)
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
        for c, n in zip(self.cur, nxt):
            c.copy_(n)
        return bd
