"""Selection-based sentence generation: the ``get_bbox_features`` path of
``src/full_model/evaluate_full_model/../evaluate_bbox_variations/evaluate_bbox_variations.py:92-109`` -
user-supplied boxes -> RoIAlign -> 8x8 average -> dim_reduction -> ``language_model.generate``.
Same function body as the reference: every attribute it reaches into is HIP-engine backed."""
import torch


def get_bbox_features(model, images, bbox_coordinates):
    features = model.object_detector.backbone(images)
    images, features = model.object_detector._transform_inputs_for_rpn_and_roi(images, features)
    image_shapes = images.image_sizes
    bbox_roi_pool_feature_maps = model.object_detector.roi_heads.box_roi_pool(features, bbox_coordinates, image_shapes)
    bbox_features = model.object_detector.roi_heads.avg_pool(bbox_roi_pool_feature_maps)
    bbox_features = torch.squeeze(bbox_features)
    bbox_features = model.object_detector.roi_heads.dim_reduction(bbox_features)
    return bbox_features
