"""Detector assembly: the spconv-free equivalent of the reference's Detector3DTemplate for the point path.

Contract restated from /root/reference/pcdet/models/detectors/detector3d_template.py:14-139 (module topology, one
`build_<slot>` per slot, modules chosen by `NAME` from registry dicts, `model_info_dict` threading channel counts),
:178-263 (post_processing: per-sample class-agnostic NMS over the head's boxes) and :330-359 (shape-filtered
checkpoint loading).  The reference's template cannot be imported on this platform (it pulls in spconv,
pcdet/utils/spconv_utils.py:3); the slots that only exist for voxel models (vfe, pfe, backbone_2d, roi_head) are kept
as names so reference configs read naturally, and build nothing here.
"""
import torch
import torch.nn as nn

from .. import dense_heads
from ..iou3d_nms import iou3d_nms_utils
from ..pdm_neck import PDMNeck
from ..pointnet2_backbone import PointNet2MSG

# registries keyed by NAME, as pcdet/models/backbones_3d/__init__.py:10-22 and map_to_bev/__init__.py:5-10
BACKBONES_3D = {'PointNet2MSG': PointNet2MSG}
MAP_TO_BEV = {'PDMNeck': PDMNeck}


def _get(cfg, key, default=None):
    return cfg.get(key, default) if isinstance(cfg, dict) else getattr(cfg, key, default)


class Detector3DTemplate(nn.Module):
    def __init__(self, model_cfg, num_class, dataset):
        """dataset: an object with class_names, point_feature_encoder.num_point_features, grid_size,
        point_cloud_range, voxel_size (the attributes build_networks reads, ref :36-43)."""
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.dataset = dataset
        self.class_names = dataset.class_names
        self.register_buffer('global_step', torch.LongTensor(1).zero_())
        self.module_topology = ['vfe', 'backbone_3d', 'map_to_bev_module', 'pfe', 'backbone_2d', 'dense_head',
                                'point_head', 'roi_head']

    @property
    def mode(self):
        return 'TRAIN' if self.training else 'TEST'

    def update_global_step(self):
        self.global_step += 1

    def build_networks(self):
        model_info_dict = {
            'module_list': [],
            'num_rawpoint_features': self.dataset.point_feature_encoder.num_point_features,
            'num_point_features': self.dataset.point_feature_encoder.num_point_features,
            'grid_size': self.dataset.grid_size,
            'point_cloud_range': self.dataset.point_cloud_range,
            'voxel_size': self.dataset.voxel_size,
        }
        for module_name in self.module_topology:
            module, model_info_dict = getattr(self, 'build_%s' % module_name)(model_info_dict=model_info_dict)
            self.add_module(module_name, module)
        return model_info_dict['module_list']

    def _unsupported(self, key, model_info_dict):
        assert _get(self.model_cfg, key, None) is None, f'{key} needs spconv / voxel modules: not part of the point path'
        return None, model_info_dict

    def build_vfe(self, model_info_dict):
        return self._unsupported('VFE', model_info_dict)

    def build_pfe(self, model_info_dict):
        return self._unsupported('PFE', model_info_dict)

    def build_backbone_2d(self, model_info_dict):
        return self._unsupported('BACKBONE_2D', model_info_dict)

    def build_roi_head(self, model_info_dict):
        return self._unsupported('ROI_HEAD', model_info_dict)

    def build_backbone_3d(self, model_info_dict):
        cfg = _get(self.model_cfg, 'BACKBONE_3D', None)
        if cfg is None:
            return None, model_info_dict
        module = BACKBONES_3D[_get(cfg, 'NAME')](
            model_cfg=cfg, input_channels=model_info_dict['num_point_features'], grid_size=model_info_dict['grid_size'],
            voxel_size=model_info_dict['voxel_size'], point_cloud_range=model_info_dict['point_cloud_range'])
        model_info_dict['module_list'].append(module)
        model_info_dict['num_point_features'] = module.num_point_features
        model_info_dict['backbone_channels'] = getattr(module, 'backbone_channels', None)
        return module, model_info_dict

    def build_map_to_bev_module(self, model_info_dict):
        cfg = _get(self.model_cfg, 'MAP_TO_BEV', None)
        if cfg is None:
            return None, model_info_dict
        # (the reference passes model_cfg and grid_size only, ref :89-92; the PDM neck also needs the metric grid)
        module = MAP_TO_BEV[_get(cfg, 'NAME')](
            model_cfg=cfg, grid_size=model_info_dict['grid_size'], voxel_size=model_info_dict['voxel_size'],
            point_cloud_range=model_info_dict['point_cloud_range'])
        model_info_dict['module_list'].append(module)
        model_info_dict['num_bev_features'] = module.num_bev_features
        return module, model_info_dict

    def build_dense_head(self, model_info_dict):
        cfg = _get(self.model_cfg, 'DENSE_HEAD', None)
        if cfg is None:
            return None, model_info_dict
        module = dense_heads.__all__[_get(cfg, 'NAME')](
            model_cfg=cfg, input_channels=model_info_dict.get('num_bev_features', None),
            num_class=self.num_class if not _get(cfg, 'CLASS_AGNOSTIC', False) else 1, class_names=self.class_names,
            grid_size=model_info_dict['grid_size'], point_cloud_range=model_info_dict['point_cloud_range'],
            predict_boxes_when_training=_get(self.model_cfg, 'ROI_HEAD', False),
            voxel_size=model_info_dict.get('voxel_size', False))
        model_info_dict['module_list'].append(module)
        return module, model_info_dict

    def build_point_head(self, model_info_dict):
        cfg = _get(self.model_cfg, 'POINT_HEAD', None)
        if cfg is None:
            return None, model_info_dict
        module = dense_heads.__all__[_get(cfg, 'NAME')](
            model_cfg=cfg, input_channels=model_info_dict['num_point_features'],
            num_class=self.num_class if not _get(cfg, 'CLASS_AGNOSTIC', False) else 1,
            predict_boxes_when_training=_get(self.model_cfg, 'ROI_HEAD', False))
        model_info_dict['module_list'].append(module)
        return module, model_info_dict

    def forward(self, **kwargs):
        raise NotImplementedError

    def post_processing(self, batch_dict):
        """batch_cls_preds (N1 + N2 + ..., num_class | 1) logits, batch_box_preds (.., 7), batch_index (..) ->
        [{'pred_boxes', 'pred_scores', 'pred_labels'}] per sample: sigmoid, arg-max class, score threshold, top
        NMS_PRE_MAXSIZE, rotated NMS (HIP kernels of iou3d_nms.hip), first NMS_POST_MAXSIZE (ref :178-263, the
        single-class-NMS branch the point heads use)."""
        cfg = _get(self.model_cfg, 'POST_PROCESSING')
        nms_cfg = _get(cfg, 'NMS_CONFIG')
        assert not _get(nms_cfg, 'MULTI_CLASSES_NMS', False), 'multi-class NMS is not part of the point path'
        pred_dicts = []
        for index in range(batch_dict['batch_size']):
            if batch_dict.get('batch_index', None) is not None:
                assert batch_dict['batch_box_preds'].dim() == 2
                batch_mask = batch_dict['batch_index'] == index
            else:
                assert batch_dict['batch_box_preds'].dim() == 3
                batch_mask = index
            box_preds = batch_dict['batch_box_preds'][batch_mask]
            cls_preds = batch_dict['batch_cls_preds'][batch_mask]
            src_cls_preds = cls_preds
            assert cls_preds.shape[1] in [1, self.num_class]
            if not batch_dict['cls_preds_normalized']:
                cls_preds = torch.sigmoid(cls_preds)
            cls_preds, label_preds = torch.max(cls_preds, dim=-1)
            label_preds = label_preds + 1
            selected, selected_scores = iou3d_nms_utils.class_agnostic_nms(
                box_scores=cls_preds, box_preds=box_preds, nms_config=nms_cfg, score_thresh=_get(cfg, 'SCORE_THRESH'))
            if _get(cfg, 'OUTPUT_RAW_SCORE', False):
                selected_scores = torch.max(src_cls_preds, dim=-1)[0][selected]
            pred_dicts.append({'pred_boxes': box_preds[selected], 'pred_scores': selected_scores,
                               'pred_labels': label_preds[selected]})
        return pred_dicts, {}

    def load_params_from_state_dict(self, model_state_disk, strict=True):
        """Copy every entry whose key and shape match (ref :330-359, without the spconv weight re-layout).  As in the
        reference (:354-358): strict=True loads ONLY the matching entries, strictly — a checkpoint that lacks a
        parameter of this model (or holds it with another shape) raises; strict=False keeps this model's own values
        for those.  Returns the keys that were not taken from the checkpoint."""
        state = self.state_dict()
        update = {k: v for k, v in model_state_disk.items() if k in state and state[k].shape == v.shape}
        if strict:
            self.load_state_dict(update)
        else:
            state.update(update)
            self.load_state_dict(state)
        return [k for k in state if k not in update]
