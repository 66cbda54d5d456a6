"""Detector registry and the model entry points of /root/reference/pcdet/models/__init__.py:17-53 and
detectors/__init__.py:19-46: build_detector / build_network by NAME, model_fn_decorator for the training loop."""
from collections import namedtuple

from .detector3d_template import BACKBONES_3D, MAP_TO_BEV, Detector3DTemplate
from .pdm_ssd import PDMSSD

__all__ = {
    'Detector3DTemplate': Detector3DTemplate,
    'PDMSSD': PDMSSD,
}


def build_detector(model_cfg, num_class, dataset):
    name = model_cfg['NAME'] if isinstance(model_cfg, dict) else model_cfg.NAME
    return __all__[name](model_cfg=model_cfg, num_class=num_class, dataset=dataset)


def build_network(model_cfg, num_class, dataset):
    return build_detector(model_cfg=model_cfg, num_class=num_class, dataset=dataset)


def model_fn_decorator():
    """model_func(model, batch_dict) -> ModelReturn(loss, tb_dict, disp_dict); batch_dict already holds device tensors
    (the reference's load_data_to_gpu, pcdet/models/__init__.py:24-38, is the input path's job here)."""
    ModelReturn = namedtuple('ModelReturn', ['loss', 'tb_dict', 'disp_dict'])

    def model_func(model, batch_dict):
        ret_dict, tb_dict, disp_dict = model(batch_dict)
        loss = ret_dict['loss'].mean()
        (model if hasattr(model, 'update_global_step') else model.module).update_global_step()
        return ModelReturn(loss, tb_dict, disp_dict)

    return model_func
