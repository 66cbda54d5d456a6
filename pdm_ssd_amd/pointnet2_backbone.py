"""PointNet2MSG backbone on the HIP operators.

Host-side restatement of /root/reference/pcdet/models/backbones_3d/pointnet2_backbone.py:9-94: same
constructor (model_cfg, input_channels, **kwargs), same `.num_point_features`, same batch_dict keys in
and out, same state_dict keys (SA_modules.{i}.mlps..., FP_modules.{i}.mlp...).  model_cfg may be an
EasyDict (as in OpenPCDet) or a plain nested dict with the same key names.

Differences: the per-sample point count check (:72-76) is one bincount instead of B host syncs, and
the intermediate sets are additionally published as batch_dict['sa_xyz'] / ['sa_features'] for the
PDM neck (extra keys; nothing upstream reads them).
"""
import torch
import torch.nn as nn

from .pointnet2_batch import pointnet2_modules, pointnet2_utils


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class PointNet2MSG(nn.Module):
    def __init__(self, model_cfg, input_channels, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        sa_cfg = _get(model_cfg, 'SA_CONFIG')
        npoints, radius, nsample = _get(sa_cfg, 'NPOINTS'), _get(sa_cfg, 'RADIUS'), _get(sa_cfg, 'NSAMPLE')
        mlps_cfg = _get(sa_cfg, 'MLPS')
        use_xyz = _get(sa_cfg, 'USE_XYZ', True)
        fp_mlps = _get(model_cfg, 'FP_MLPS')

        self.SA_modules = nn.ModuleList()
        channel_in = input_channels - 3
        self.num_points_each_layer = []
        skip_channel_list = [input_channels - 3]
        channel_out = channel_in
        for k in range(len(npoints)):
            # fresh lists per layer: the SA constructor mutates mlps[i][0] (ref pointnet2_modules.py:86-88)
            mlps = [[channel_in] + list(spec) for spec in mlps_cfg[k]]
            channel_out = sum(spec[-1] for spec in mlps)
            self.SA_modules.append(pointnet2_modules.PointnetSAModuleMSG(
                npoint=npoints[k], radii=list(radius[k]), nsamples=list(nsample[k]), mlps=mlps,
                use_xyz=use_xyz))
            skip_channel_list.append(channel_out)
            channel_in = channel_out

        self.FP_modules = nn.ModuleList()
        for k in range(len(fp_mlps)):
            pre_channel = fp_mlps[k + 1][-1] if k + 1 < len(fp_mlps) else channel_out
            self.FP_modules.append(pointnet2_modules.PointnetFPModule(
                mlp=[pre_channel + skip_channel_list[k]] + list(fp_mlps[k])))
        self.num_point_features = fp_mlps[0][-1]

    @staticmethod
    def break_up_pc(pc):
        batch_idx = pc[:, 0]
        xyz = pc[:, 1:4].contiguous()
        features = pc[:, 4:].contiguous() if pc.size(-1) > 4 else None
        return batch_idx, xyz, features

    @torch.no_grad()
    def sample_chain(self, xyz):
        """The coordinate-only half of the SA stack: FPS + gather for every layer, xyz (B,N,3) ->
        [new_xyz_1 .. new_xyz_L].  It depends on no feature, so a caller may run it ahead of time (e.g. on
        a side stream for the NEXT batch, see pdm_ssd_amd/pipeline.py) and hand the result to forward()
        as batch_dict['sampled_xyz']."""
        return self.sample_levels(xyz, 0, len(self.SA_modules))

    @torch.no_grad()
    def coordinate_levels(self, xyz_in, first, last, first_idx=None):
        """Everything levels [first, last) compute from coordinates alone, starting from `xyz_in` = the input cloud of
        level `first`: sampled xyz (FPS + gather), the ball-query indices of every scale (compacted, see
        fused.sa_pack), and the three-NN (idx, weight) of the FP module that interpolates level k+1 back onto level k.
        Lists indexed by level.  first_idx = FPS indices of level `first` computed elsewhere (pipeline.py)."""
        out = {'sampled_xyz': [], 'ball_idx': [], 'fp_interp': []}
        src = xyz_in
        # a level's sampled set is the known set of its three_nn AND the source of the next level's ball queries: one
        # search grid serves all three.  Nothing in this block rewrites a point set (xyz_in is only read).
        with pointnet2_utils.shared_search_grids():
            for k in range(first, last):
                sa = self.SA_modules[k]
                new_xyz = sa.sample(src) if (k > first or first_idx is None) else sa.sample_from_idx(src, first_idx)
                out['sampled_xyz'].append(new_xyz)
                out['ball_idx'].append(sa.query(src, new_xyz))
                out['fp_interp'].append(pointnet2_modules.PointnetFPModule.interpolation(src.contiguous(), new_xyz))
                src = new_xyz
        return out

    @torch.no_grad()
    def autotune_hoisting(self, points, batch_size, reps=3):
        """Pick, per SA layer, between the hoisted first layer (W1[:, features] applied once to every SOURCE point,
        fused.cached_pre_packs) and the plain form (applied to the compacted neighbour rows): with sparse
        neighbourhoods the compacted lists hold fewer rows than the source set and hoisting is the more expensive
        form.  Both forms of every level are run on one representative batch (`points` as for forward()) and timed
        with events; sets `sa.use_pre` to the faster one and returns the measurements.  Call it at set-up time."""
        xyz = points[:, 1:4].contiguous().view(batch_size, -1, 3)
        feats = points[:, 4:].contiguous().view(batch_size, -1, points.shape[1] - 4).permute(0, 2, 1).contiguous()
        decisions = []
        for sa in self.SA_modules:
            new_xyz = sa.sample(xyz)
            idx_list = sa.query(xyz, new_xyz)
            cin = feats.shape[1]
            timing = {}
            for form in ((True, False) if cin >= pointnet2_modules.PRE_MIN_CIN else (False,)):
                sa.use_pre = form
                out = sa(xyz, feats, new_xyz=new_xyz, idx_list=idx_list)[1]   # also builds the weight packs
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    out = sa(xyz, feats, new_xyz=new_xyz, idx_list=idx_list)[1]
                e1.record()
                torch.cuda.synchronize()
                timing[form] = e0.elapsed_time(e1) / reps
            sa.use_pre = min(timing, key=timing.get)
            decisions.append({'use_pre': bool(sa.use_pre), 'ms_hoisted': timing.get(True), 'ms_plain': timing.get(False)})
            if not sa.use_pre and True in timing:
                out = sa(xyz, feats, new_xyz=new_xyz, idx_list=idx_list)[1]
            xyz, feats = new_xyz, out.contiguous()
        return decisions

    @torch.no_grad()
    def sample_levels(self, xyz, first, last):
        """Levels [first, last) of the sampling chain, starting from `xyz` = the input of level `first`."""
        out = []
        for sa in list(self.SA_modules)[first:last]:
            xyz = sa.sample(xyz)
            out.append(xyz)
        return out

    def forward(self, batch_dict):
        """batch_dict['points'] (sum N, 1+3+C) with column 0 = sample index -> adds
        'point_features' (B*N, C_out) and 'point_coords' (B*N, 4).
        Optional batch_dict['sampled_xyz'] = sample_chain(xyz) computed earlier."""
        batch_size = batch_dict['batch_size']
        points = batch_dict['points']
        batch_idx, xyz, features = self.break_up_pc(points)
        if not batch_dict.get('points_per_sample_checked', False):
            counts = torch.bincount(batch_idx.long(), minlength=batch_size)
            assert counts.min() == counts.max(), 'PointNet2MSG needs the same point count in every sample'
            batch_dict['points_per_sample_checked'] = True   # later modules (the point head's targets) need not re-check
        xyz = xyz.view(batch_size, -1, 3)
        if features is not None:
            features = features.view(batch_size, -1, features.shape[-1]).permute(0, 2, 1).contiguous()

        l_xyz, l_features = [xyz], [features]
        presampled = batch_dict.get('sampled_xyz', None)
        ball_idx = batch_dict.get('ball_idx', None)      # optional, from coordinate_levels()
        fp_interp = batch_dict.get('fp_interp', None)
        with pointnet2_utils.shared_search_grids():
            return self._forward_levels(batch_dict, batch_idx, l_xyz, l_features, presampled, ball_idx, fp_interp)

    def _forward_levels(self, batch_dict, batch_idx, l_xyz, l_features, presampled, ball_idx, fp_interp):
        for k, sa in enumerate(self.SA_modules):
            li_xyz, li_features = sa(l_xyz[-1], l_features[-1],
                                     new_xyz=None if presampled is None else presampled[k],
                                     idx_list=None if ball_idx is None else ball_idx[k])
            l_xyz.append(li_xyz)
            l_features.append(li_features)
        batch_dict['sa_xyz'] = list(l_xyz)
        batch_dict['sa_features'] = list(l_features)
        after_sa = batch_dict.get('after_sa_hook', None)
        if after_sa is not None:
            # everything the PDM neck reads exists now; a caller may start it on another stream so it runs
            # beside the feature-propagation layers (pdm_ssd_amd/pipeline.py)
            after_sa(batch_dict)

        # batch_dict['defer_last_fp']: the caller (a detector whose point head can take it: pdm_fp_head_fused) runs the LAST FP
        # module's final step itself; point_features is then storage not yet written, batch_dict['point_features_deferred']
        # the object that fills it (fused.DeferredFP).  The caller owns that call.
        deferred = [] if batch_dict.get('defer_last_fp') else None
        for i in range(-1, -(len(self.FP_modules) + 1), -1):
            # FP module i interpolates level len+i onto level len+i-1: fp_interp is indexed by the coarser level - 1
            last = i == -len(self.FP_modules)
            l_features[i - 1] = self.FP_modules[i](l_xyz[i - 1], l_xyz[i], l_features[i - 1], l_features[i],
                                                   interp=None if fp_interp is None else fp_interp[len(l_xyz) + i - 1],
                                                   **({'defer': deferred} if (last and deferred is not None) else {}))

        point_features = l_features[0].permute(0, 2, 1).contiguous()  # (B, N, C)
        if deferred:
            if point_features.data_ptr() == deferred[0].out_pm.data_ptr():
                batch_dict['point_features_deferred'] = deferred[0]
            else:       # (the rows were copied on the way: fill them now)
                deferred[0].materialize()
                point_features = l_features[0].permute(0, 2, 1).contiguous()
        batch_dict['point_features'] = point_features.view(-1, point_features.shape[-1])
        batch_dict['point_coords'] = torch.cat((batch_idx[:, None].float(), l_xyz[0].view(-1, 3)), dim=1)
        return batch_dict


POINTRCNN_MSG_CFG = {
    # upstream OpenPCDet tools/cfgs/kitti_models/pointrcnn.yaml BACKBONE_3D (yaml absent from the snapshot,
    # SURVEY.md section 8); shapes BASELINE.json's 16384-point configs are quoted on.
    'NAME': 'PointNet2MSG',
    'SA_CONFIG': {
        'NPOINTS': [4096, 1024, 256, 64],
        'RADIUS': [[0.1, 0.5], [0.5, 1.0], [1.0, 2.0], [2.0, 4.0]],
        'NSAMPLE': [[16, 32], [16, 32], [16, 32], [16, 32]],
        'MLPS': [[[16, 16, 32], [32, 32, 64]],
                 [[64, 64, 128], [64, 96, 128]],
                 [[128, 196, 256], [128, 196, 256]],
                 [[256, 256, 512], [256, 384, 512]]],
    },
    'FP_MLPS': [[128, 128], [256, 256], [512, 512], [512, 512]],
}
