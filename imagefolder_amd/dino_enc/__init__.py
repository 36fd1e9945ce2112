from .dinov2 import DINOv2Encoder, DINOv2Decoder  # noqa: F401
from .vision_transformer import Attention, VisionTransformer  # noqa: F401
