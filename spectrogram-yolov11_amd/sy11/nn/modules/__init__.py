"""sy11.nn.modules — same public names as ultralytics.nn.modules for the hot-path subset."""
from .block import C2PSA, C3, DFL, SPPF, Attention, Bottleneck, C2f, C3k, C3k2, PSABlock
from .conv import GCT, Concat, Conv, DDWConv, DWConv, Fusion, WeightedSpatialAttention, autopad
from .head import Detect

__all__ = ("Conv", "DWConv", "DDWConv", "Concat", "autopad", "DFL", "SPPF", "C2f", "C3", "C3k", "C3k2", "Bottleneck", "Attention",
           "PSABlock", "C2PSA", "Detect", "GCT", "WeightedSpatialAttention", "Fusion")
