"""Backbone constructors with the reference's names (backbone/__init__.py:2-3).

The reference returns pytorchcv nn.Modules; here a backbone is a *specification* that
PoseModuleKD turns into kd6d engine layers (HIP implicit-GEMM convs).  `pretrained=True`
(libs/train_libs.py:82-87) needs the pytorchcv model zoo over the network; offline it falls
back to random initialisation with a warning, or loads `root/<model_name>.pth` if present.
"""
import os
import warnings


class BackboneSpec:
    def __init__(self, arch, pretrained=False, root=os.path.join("~", ".torch", "models")):
        self.arch = arch
        self.pretrained_file = None
        if pretrained:
            cand = os.path.join(os.path.expanduser(root), arch + ".pth")
            if os.path.exists(cand):
                self.pretrained_file = cand
            else:
                warnings.warn("pretrained weights for %s not found at %s (no network): random init" % (arch, cand))


def darknet53(pretrained=False, **kw):
    return BackboneSpec("darknet53", pretrained, **kw)


def darknet_tiny(pretrained=False, **kw):
    return BackboneSpec("darknet_tiny", pretrained, **kw)


def darknet_tiny_h(pretrained=False, **kw):
    return BackboneSpec("darknet_tiny_h", pretrained, **kw)
