"""rgrg_amd - MI355X (gfx950) native implementation of the RGRG inference hot path
(``ReportGenerationModel.generate`` of ttanida/rgrg): Python host mirroring the
reference's module API over hand-written HIP kernels in ``librgrg_hip.so``."""
from .binary_classifier import BinaryClassifierRegionAbnormal, BinaryClassifierRegionSelection  # noqa: F401
from .language_model import LanguageModel  # noqa: F401
from .object_detector import ObjectDetector  # noqa: F401
from .report_generation_model import ReportGenerationModel  # noqa: F401

__version__ = "0.1.0"
