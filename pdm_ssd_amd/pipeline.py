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

depth >= 3 additionally cuts the level-1 FPS itself into S = depth - 1 resumable segments
(pdm_furthest_point_sampling_jobs): one launch per step runs segment s of batch i+1+S-s for every s side by side
(S x B workgroups instead of B), the rest of the chain of batch i+1 (gather, ball query + compaction, three-NN,
levels 2..L) runs on a third stream.  The longest serial piece beside the feature path is then 1/S of the level-1
FPS; S + 2 batches are in flight, each step still does one full batch of every kind of work.  With the SA kernels
running over compacted neighbour lists the feature path is ~2 ms at bs=32 and the step is bound by the total work
of the three branches, no longer by the FPS dependency chain.
"""
import torch

from . import _native
from .pointnet2_batch import pointnet2_utils


def _flat(d):
    """Tensors of a coordinate_levels() dict in a fixed order."""
    out = list(d['sampled_xyz'])
    for lvl in d['ball_idx']:
        for q in lvl:   # a dense (B,M,ns) index tensor or a compacted (pack, meta) pair
            out.extend(q if isinstance(q, tuple) else (q,))
    for idx, w in d['fp_interp']:
        out.extend((idx, w))
    return out


def _live(d):
    """Per tensor of _flat(d): None, or (device-side row count, bytes per row) for the compacted neighbour lists — their
    buffers are sized for the worst case (every slot distinct) and mostly empty on sparse clouds."""
    out = [None] * len(d['sampled_xyz'])
    for lvl in d['ball_idx']:
        for q in lvl:
            out.extend([(q[1][6:7], 8), None] if isinstance(q, tuple) else [None])
    out.extend([None] * (2 * len(d['fp_interp'])))
    return out


def _merge(a, b):
    return {k: list(a[k]) + list(b[k]) for k in ('sampled_xyz', 'ball_idx', 'fp_interp')}


def overlapping_stream(device, tries=8):
    """A side stream whose kernels really run beside the current stream's (e.g. for the next batch's sampling chain under a
    training step: bench.py's train loop).  HIP spreads a process's streams over a few hardware queues; in a process that had
    already created a dozen streams a fresh torch stream landed on the queue of the main stream and the sampling chain ran
    in line (28.4 ms per training step against 25.6, same kernel durations; a high-priority stream made it 48 ms).  Candidates are tried with two spin kernels: side by side they take one spin,
    in line two."""
    main = torch.cuda.current_stream(device)
    spin = 2_000_000      # ~1 ms
    def both(side):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(device)
        e0.record(main)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            torch.cuda._sleep(spin)
        torch.cuda._sleep(spin)
        main.wait_stream(side)
        e1.record(main)
        torch.cuda.synchronize(device)
        return e0.elapsed_time(e1)
    torch.cuda._sleep(spin); torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main); torch.cuda._sleep(spin); e1.record(main); torch.cuda.synchronize(device)
    one = e0.elapsed_time(e1)
    best, best_ms = None, None
    for _ in range(tries):
        cand = torch.cuda.Stream(device)
        ms = min(both(cand), both(cand))
        if best is None or ms < best_ms:
            best, best_ms = cand, ms
        if ms < 1.4 * one:
            break
    return best


class PipelinedHotPath:
    def __init__(self, backbone, neck=None, depth=1, dense_head=None, point_head=None):
        """dense_head / point_head: the two halves of the hybrid head (detector slots DENSE_HEAD / POINT_HEAD).  The
        heat-map head reads only the neck's grid and runs behind it on the neck's stream; the point head reads the
        backbone's point features and runs on the main stream after the FP layers."""
        assert depth in (1, 2, 3, 4, 5)
        self.backbone = backbone
        self.neck = neck
        self.dense_head = dense_head
        self.point_head = point_head
        # True: the heat-map head runs behind the neck on the neck's stream, i.e. beside the FP layers — two MFMA-bound
        # branches sharing the chip; False: on the main stream behind the point head (the memory-bound neck still
        # overlaps the FP layers).  Same step time either way (measured); False keeps each MFMA kernel's launch
        # duration what it is alone, which is what the bench's in-situ roofline reads.
        self.heads_overlap = False
        self.depth = depth
        self.side = torch.cuda.Stream()
        self.neck_stream = torch.cuda.Stream()
        # depth 2: the short tail of the next batch's chain runs on the neck's stream ahead of the neck (which cannot
        # start before the SA stack is done anyway) — hipGraph replay starts a fourth parallel branch late
        self.side2 = self.neck_stream if depth == 2 else None
        # depth >= 3: a third side stream for the rest of the coordinate chain.  That makes four concurrent branches
        # (main, neck, side, side3) — replaying a capture with a fifth one segfaults inside the ROCm 7.2 runtime.
        self.side3 = torch.cuda.Stream() if depth >= 3 else None
        self.cur = None   # static hand-over buffers: coordinate-only results of the batch about to be processed
        self.half = None  # depth 2: level-1 results of the batch after that
        self.nseg = depth - 1 if depth >= 3 else 0
        self.l1idx = None  # depth >= 3: complete level-1 FPS indices of the next batch
        self.seg = None    # depth >= 3: seg[s] = (temp, idx) of the batch that has finished s segments, s = 1..S-1

    def _bounds(self):
        m = self.backbone.SA_modules[0].npoint
        return [1 + (m - 1) * s // self.nseg for s in range(self.nseg + 1)]   # iterations [1, m) in S near-equal parts

    @staticmethod
    def _fresh(xyz, m):
        return (torch.full(xyz.shape[:2], 1e10, dtype=torch.float32, device=xyz.device),
                torch.empty((xyz.shape[0], m), dtype=torch.int32, device=xyz.device))

    @torch.no_grad()
    def prime_segmented(self, points_list, batch_size):
        """depth >= 3: points_list = the first S + 1 batches [0 .. S].  Runs (not overlapped) the whole chain of batch 0,
        the complete level-1 FPS of batch 1 and the first S - s segments of batch s + 1 ... so that step() finds
        every batch at the stage it expects."""
        S, nlev = self.nseg, len(self.backbone.SA_modules)
        assert S >= 2 and len(points_list) >= S + 1
        m = self.backbone.SA_modules[0].npoint
        bounds = self._bounds()
        self.cur = self._clone(self.backbone.coordinate_levels(self._xyz(points_list[0], batch_size), 0, nlev))
        self.seg = [None] * S
        # batch 1 + S - s has finished s segments when the first step starts (s = S: complete -> l1idx)
        for s_done in range(1, S + 1):
            xyz = self._xyz(points_list[1 + S - s_done], batch_size)
            temp, idx = self._fresh(xyz, m)
            for s in range(s_done):
                pointnet2_utils.fps_segments([(xyz, temp, idx, bounds[s], bounds[s + 1])], m)
            if s_done == S:
                self.l1idx = idx
            else:
                self.seg[s_done] = (temp, idx)

    def _step_segmented(self, points_cur, points_ahead, batch_size, extra):
        S, nlev = self.nseg, len(self.backbone.SA_modules)
        assert self.l1idx is not None and len(points_ahead) == S + 1, "prime_segmented() first; points_ahead = batches i+1 .. i+1+S"
        main = torch.cuda.current_stream()
        m = self.backbone.SA_modules[0].npoint
        bounds = self._bounds()
        # (the FPS segments are forked at the START of the step, beside the SA and FP stacks, whose kernels they slow by taking CU
        #  slots — FP stack 1.26 ms of kernels in 1.86 ms; forked behind the SA stack the step took 4.61 ms, behind the FP stack,
        #  i.e. beside the head kernels, 4.71 ms, against 4.51-4.52)
        self.side.wait_stream(main)
        self.side3.wait_stream(main)
        with torch.cuda.stream(self.side):
            # one launch: segment s of batch i+1+S-s, s = 0 .. S-1 (points_ahead[k] = batch i+1+k)
            xyz0 = self._xyz(points_ahead[S], batch_size)
            fresh = self._fresh(xyz0, m)
            jobs = [(xyz0, fresh[0], fresh[1], bounds[0], bounds[1])]
            for s in range(1, S):
                jobs.append((self._xyz(points_ahead[S - s], batch_size), self.seg[s][0], self.seg[s][1], bounds[s], bounds[s + 1]))
            ws = pointnet2_utils.fps_segments(jobs, m)
            self._fps_ws = (ws, xyz0.shape[0], xyz0.shape[1])   # N > 16384: status words, see check_sampling()
        with torch.cuda.stream(self.side3):   # everything of batch i+1 behind its level-1 FPS
            # (its own stream: measured 2.77 ms/step against 2.93 on the neck's stream ahead of the neck, bs=32)
            nxt = self.backbone.coordinate_levels(self._xyz(points_ahead[0], batch_size), 0, nlev, first_idx=self.l1idx)
        bd = self._features(points_cur, batch_size, extra)
        main.wait_stream(self.side3)
        main.wait_stream(self.side)
        # (the hand-over only needs the backbone and the neck to be done with this batch's coordinate results; run on side3
        #  behind the FP stack, beside the head kernels, it made the step LONGER: 4.64 against 4.53 ms — the replayed graph
        #  then joins one more branch in front of the point head)
        _native.copy_many(_flat(self.cur), _flat(nxt), _live(nxt))
        # shift the segment states one stage on, last stage first (each launch's sources are the next one's targets)
        _native.copy_many([self.l1idx], [self.seg[S - 1][1]])
        for s in range(S - 1, 1, -1):
            _native.copy_many(list(self.seg[s]), list(self.seg[s - 1]))
        _native.copy_many(list(self.seg[1]), list(fresh))
        return bd

    def check_sampling(self):
        """Clouds of more than 16384 points are sampled by cooperating workgroups with bounded waits; raise if one
        of them gave up in the last (possibly graph-replayed) step.  Synchronises; call outside timed regions."""
        ws, b, n = getattr(self, "_fps_ws", ((), 0, 0))
        for w in ws:
            _native.fps_check_workspace(w, b, n)
        _native.fps_check()

    # -- the same step as three separately launchable parts (depth >= 3), e.g. one hipGraph each on its own stream:
    #    coordinates (level-1 FPS segments || rest of batch i+1's chain), features (+ neck), hand-over.
    #    The caller orders them: hand-over after both others; the next step's parts after the hand-over.
    @torch.no_grad()
    def part_coordinates(self, points_ahead, batch_size):
        S, nlev = self.nseg, len(self.backbone.SA_modules)
        assert self.l1idx is not None and len(points_ahead) == S + 1
        cur = torch.cuda.current_stream()
        m = self.backbone.SA_modules[0].npoint
        bounds = self._bounds()
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            xyz0 = self._xyz(points_ahead[S], batch_size)
            fresh = self._fresh(xyz0, m)
            jobs = [(xyz0, fresh[0], fresh[1], bounds[0], bounds[1])]
            for s in range(1, S):
                jobs.append((self._xyz(points_ahead[S - s], batch_size), self.seg[s][0], self.seg[s][1], bounds[s], bounds[s + 1]))
            pointnet2_utils.fps_segments(jobs, m)
        nxt = self.backbone.coordinate_levels(self._xyz(points_ahead[0], batch_size), 0, nlev, first_idx=self.l1idx)
        cur.wait_stream(self.side)
        self._pending = (nxt, fresh)

    @torch.no_grad()
    def part_features(self, points_cur, batch_size, extra=None):
        return self._features(points_cur, batch_size, extra)

    @torch.no_grad()
    def part_handover(self):
        S = self.nseg
        nxt, fresh = self._pending
        _native.copy_many(_flat(self.cur), _flat(nxt), _live(nxt))
        _native.copy_many([self.l1idx], [self.seg[S - 1][1]])
        for s in range(S - 1, 1, -1):
            _native.copy_many(list(self.seg[s]), list(self.seg[s - 1]))
        _native.copy_many(list(self.seg[1]), list(fresh))

    def _features(self, points_cur, batch_size, extra):
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
                    if self.dense_head is not None and self.heads_overlap:
                        self.dense_head(d)
            bd['after_sa_hook'] = start_neck
        if self.point_head is not None and hasattr(self.point_head, 'wants_deferred_fp') and self.point_head.wants_deferred_fp():
            bd['defer_last_fp'] = True      # the backbone's last FP module runs inside the point head's launch
        bd = self.backbone(bd)
        if self.point_head is not None:
            bd = self.point_head(bd)
        owed = bd.pop('point_features_deferred', None)
        if owed is not None:
            owed.materialize()
        if self.neck is not None:
            torch.cuda.current_stream().wait_stream(self.neck_stream)
            if self.dense_head is not None and not self.heads_overlap:
                bd = self.dense_head(bd)
        return bd

    @staticmethod
    def _xyz(points, batch_size):
        return points[:, 1:4].contiguous().view(batch_size, -1, 3)

    @staticmethod
    def _clone(d):
        return {'sampled_xyz': [t.clone() for t in d['sampled_xyz']],
                'ball_idx': [[tuple(u.clone() for u in t) if isinstance(t, tuple) else t.clone() for t in lvl]
                             for lvl in d['ball_idx']],
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
    def step(self, points_cur, points_next, batch_size, extra=None, points_next2=None, points_ahead=None):
        """Features of `points_cur` (its coordinate-only results are in self.cur) || coordinate-only chain of
        `points_next` (depth 2: || levels 2..L of `points_next` || level 1 of `points_next2`).
        The coordinate-only chain = FPS + gather, ball-query indices and three-NN weights of every level: nothing in
        it reads a feature, so only the MLP kernels (and the neck) stay on the feature path."""
        assert self.cur is not None, "call prime() first"
        if self.nseg:
            return self._step_segmented(points_cur, points_ahead, batch_size, extra)
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
        bd = self._features(points_cur, batch_size, extra)
        main.wait_stream(self.side)
        if self.depth == 2:
            main.wait_stream(self.side2)
        # hand-over into the static buffers (addresses must not change under hipGraph replay): one multi-tensor copy
        _native.copy_many(_flat(self.cur), _flat(nxt), _live(nxt))
        if self.depth == 2:   # a second launch: self.half is a SOURCE of the first one (level 1 of `nxt`)
            _native.copy_many(_flat(self.half), _flat(nxt_half))
        return bd
