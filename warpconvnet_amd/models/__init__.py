"""Model definitions the hot path's configurations exercise (reference `warpconvnet/models/`: only the MinkUNet family)."""
from .mink_unet import (BasicBlock, BottleneckBlock, ConvBlock, ConvTrBlock, MinkUNet14, MinkUNet18, MinkUNet34, MinkUNet50,
                        MinkUNet101, MinkUNetBase)

__all__ = ["BasicBlock", "BottleneckBlock", "ConvBlock", "ConvTrBlock", "MinkUNet14", "MinkUNet18", "MinkUNet34", "MinkUNet50",
           "MinkUNet101", "MinkUNetBase"]
