"""Selection-based sentence generation: region features for USER-SUPPLIED boxes,
i.e. the path the reference's bbox-variation study drives by reaching into the detector
(``evaluate_bbox_variations.py:92-109`` of ttanida/rgrg: backbone -> box_roi_pool on the given boxes -> 8x8
average -> squeeze -> dim_reduction), followed by ``model.language_model.generate``.  The reference's own
function works unchanged on ``rgrg_amd.ReportGenerationModel`` because every attribute it touches is callable
and HIP-engine backed; this module offers the same result through one call that skips the NCHW round trip."""
from typing import List

import torch


def get_bbox_features(model, images: torch.Tensor, bbox_coordinates: List[torch.Tensor]) -> torch.Tensor:
    """[sum_i n_i, 1024] region features of the given xyxy boxes (one [n_i,4] tensor per image)."""
    detector = model.object_detector
    heads = detector.roi_heads
    feature_map = detector.backbone(images)                      # NCHW view of the NHWC trunk output
    wrapped_images, feature_dict = detector._transform_inputs_for_rpn_and_roi(images, feature_map)
    roi_maps = heads.box_roi_pool(feature_dict, bbox_coordinates, wrapped_images.image_sizes)
    pooled = torch.squeeze(heads.avg_pool(roi_maps))             # the fused kernel already produced the 8x8 mean
    return heads.dim_reduction(pooled)
