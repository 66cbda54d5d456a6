"""PDM-SSD: PointNet2MSG backbone -> PDM neck (MAP_TO_BEV slot) -> hybrid head (BEV heat-map head in the DENSE_HEAD
slot + point box head in the POINT_HEAD slot).

The reference snapshot holds no detector class or YAML for it; the class follows the reference's
single-stage point detectors: forward loop and training contract `({'loss': ...}, tb_dict, disp_dict)` of
/root/reference/pcdet/models/detectors/point_rcnn.py:9-30, losses summed over the heads that exist.
"""
import os

import torch

from .detector3d_template import Detector3DTemplate

# 1: in TRAINING the dense (heat-map) half of the hybrid head — forward, targets, loss and, because autograd runs a node's backward
# on the stream of its forward, its whole backward — runs on a stream of its own beside the point half: two independent chains of
# HBM-bound kernels (each launch leaves ramps and tails a second chain fills).  Same kernels, same values.  Measured 18.39 -> 18.13 ms
# per step at bs = 32 (A/B twice on one box).  Opt-in (0 by default): under DistributedDataParallel the gradients of the dense head
# would be written on this stream while the reducer orders its all-reduce behind the stream of the LAST gradient only.
BRANCH_STREAM = os.environ.get("PDM_TRAIN_BRANCH_STREAM", "0") == "1"
_branch_streams = {}


def _branch_stream(device):
    key = torch.device(device).index
    if key not in _branch_streams:
        from ..pipeline import overlapping_stream
        _branch_streams[key] = overlapping_stream(device)       # a stream on a hardware queue of its own (probed once)
    return _branch_streams[key]


class PDMSSD(Detector3DTemplate):
    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()
        self._dense_loss = None

    def forward(self, batch_dict):
        # the point head can run the backbone's last FP module inside its own launch (pdm_fp_head_fused)
        head = getattr(self, 'point_head', None)
        if head is not None and hasattr(head, 'wants_deferred_fp') and head.wants_deferred_fp():
            batch_dict['defer_last_fp'] = True
        from .. import fused_bn
        self._dense_loss = None
        dense = getattr(self, 'dense_head', None)
        with fused_bn.counter_scope():      # the BatchNorm step counters of every stack: one multi-tensor add
            for cur_module in self.module_list:
                sf = batch_dict.get('spatial_features')
                if (BRANCH_STREAM and self.training and cur_module is dense and getattr(self, 'point_head', None) is not None
                        and sf is not None and sf.is_cuda and not torch.cuda.is_current_stream_capturing()):
                    main, side = torch.cuda.current_stream(sf.device), _branch_stream(sf.device)
                    side.wait_stream(main)
                    sf.record_stream(side)            # allocated on the main stream, read (forward and backward) on the side stream
                    with torch.cuda.stream(side):
                        batch_dict = cur_module(batch_dict)
                        self._dense_loss = (side, *self.dense_head.get_loss({}))
                    continue
                batch_dict = cur_module(batch_dict)
        owed = batch_dict.pop('point_features_deferred', None)
        if owed is not None:            # (no module took it)
            owed.materialize()
        if self.training:
            loss, tb_dict, disp_dict = self.get_training_loss()
            return {'loss': loss}, tb_dict, disp_dict
        pred_dicts, recall_dicts = self.post_processing(batch_dict)
        return pred_dicts, recall_dicts

    def get_training_loss(self):
        disp_dict, tb_dict, loss = {}, {}, 0
        if self.point_head is not None:
            loss_point, tb_dict = self.point_head.get_loss(tb_dict)
            loss = loss + loss_point
        if self.dense_head is not None:       # auxiliary phase of the hybrid head: the scene heat-map
            if self._dense_loss is not None:  # formed on the branch stream (BRANCH_STREAM): join it here
                side, loss_hm, tb_hm = self._dense_loss
                self._dense_loss = None
                main = torch.cuda.current_stream(loss_hm.device)
                main.wait_stream(side)
                loss_hm.record_stream(main)
                for v in tb_hm.values():
                    if torch.is_tensor(v):
                        v.record_stream(main)
                tb_dict.update(tb_hm)
            else:
                loss_hm, tb_dict = self.dense_head.get_loss(tb_dict)
            loss = loss + loss_hm
        return loss, tb_dict, disp_dict
