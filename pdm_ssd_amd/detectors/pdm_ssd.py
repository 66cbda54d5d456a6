"""PDM-SSD: PointNet2MSG backbone -> PDM neck (MAP_TO_BEV slot) -> hybrid head (BEV heat-map head in the DENSE_HEAD
slot + point box head in the POINT_HEAD slot).

The reference snapshot holds no detector class or YAML for it; the class follows the reference's
single-stage point detectors: forward loop and training contract `({'loss': ...}, tb_dict, disp_dict)` of
/root/reference/pcdet/models/detectors/point_rcnn.py:9-30, losses summed over the heads that exist.
"""
from .detector3d_template import Detector3DTemplate


class PDMSSD(Detector3DTemplate):
    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()

    def forward(self, batch_dict):
        # the point head can run the backbone's last FP module inside its own launch (pdm_fp_head_fused)
        head = getattr(self, 'point_head', None)
        if head is not None and hasattr(head, 'wants_deferred_fp') and head.wants_deferred_fp():
            batch_dict['defer_last_fp'] = True
        from .. import fused_bn
        with fused_bn.counter_scope():      # the BatchNorm step counters of every stack: one multi-tensor add
            for cur_module in self.module_list:
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
            loss_hm, tb_dict = self.dense_head.get_loss(tb_dict)
            loss = loss + loss_hm
        return loss, tb_dict, disp_dict
