"""Constants of the RGRG inference path (names follow the reference)."""

# src/dataset/constants.py:1-31 - region order; detector class id = index + 1 (0 = background)
ANATOMICAL_REGIONS = {
    name: i for i, name in enumerate([
        "right lung", "right upper lung zone", "right mid lung zone", "right lower lung zone",
        "right hilar structures", "right apical zone", "right costophrenic angle", "right hemidiaphragm",
        "left lung", "left upper lung zone", "left mid lung zone", "left lower lung zone", "left hilar structures",
        "left apical zone", "left costophrenic angle", "left hemidiaphragm", "trachea", "spine", "right clavicle",
        "left clavicle", "aortic arch", "mediastinum", "upper mediastinum", "svc", "cardiac silhouette",
        "cavoatrial junction", "right atrium", "carina", "abdomen"])
}
NUM_REGIONS = 29

# src/full_model/generate_reports_for_images.py:25-30
BERTSCORE_SIMILARITY_THRESHOLD = 0.9
IMAGE_INPUT_SIZE = 512
MAX_NUM_TOKENS_GENERATE = 300
NUM_BEAMS = 4
mean = 0.471
std = 0.302

# src/object_detector/object_detector.py:78-97
ANCHOR_SIZES = (20, 40, 60, 80, 100, 120, 140, 160, 180, 300)
ANCHOR_RATIOS = (0.2, 0.25, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0, 1.3, 1.5, 2.1, 2.6, 3.0, 5.0, 8.0)
RPN_PRE_NMS_TOP_N = 1000   # "testing"
RPN_POST_NMS_TOP_N = 1000  # "testing"
RPN_NMS_THRESH = 0.7
RESNET50_LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))  # planes, blocks, stride

# src/binary_classifier/binary_classifier_region_selection.py:53
SELECTION_LOGIT_THRESHOLD = -1.0

# src/language_model/language_model.py:200-202
BOS_TOKEN_ID = EOS_TOKEN_ID = PAD_TOKEN_ID = 50256
