"""Heads of the detector, registered by NAME as /root/reference/pcdet/models/dense_heads/__init__.py:11-21 does."""
from .pdm_heatmap_head import PDMHeatmapHead
from .point_head_box import PointHeadBox
from .point_head_template import PointHeadTemplate

__all__ = {
    'PointHeadTemplate': PointHeadTemplate,
    'PointHeadBox': PointHeadBox,
    'PDMHeatmapHead': PDMHeatmapHead,
}
