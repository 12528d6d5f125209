"""Backbone constructors with the reference's names (backbone/__init__.py:2-3).

The reference returns pytorchcv nn.Modules; here a backbone is a *specification* that PoseModuleKD turns into kd6d
engine layers (HIP implicit-GEMM convs).  `pretrained=True` (libs/train_libs.py:82-87) resolves the pytorchcv
model-zoo file the reference would download -- `<root>/<name>-<error>-<sha1[:8]>.pth`, root `~/.torch/models`
(backbone/model_store.py:131,133,540-592) -- from the LOCAL directory only (there is no network on the GPU box):
the file is used if its SHA-1 matches the zoo's, otherwise (or when it is missing) the backbone starts from random
initialisation with a warning.  Its keys (`features.*`, `output.*`) are the `backbone.*` entries of the pose
module's state_dict; PoseModuleKD loads them by name and shape.
"""
import hashlib
import os
import warnings

# name -> (top-1 error tag, sha1) of the imgclsmob release the reference pins (backbone/model_store.py:131,133)
MODEL_STORE = {
    "darknet_tiny": ("1784", "4561e1ada619e33520d1f765b3321f7f8ea6196b"),
    "darknet53": ("0564", "b36bef6b297055dda3d17a3f79596511730e1963"),
}


def _sha1(path):
    h = hashlib.sha1()
    with open(path, "rb") as f:
        for block in iter(lambda: f.read(1 << 20), b""):
            h.update(block)
    return h.hexdigest()


def get_model_file(model_name, local_model_store_dir_path=os.path.join("~", ".torch", "models")):
    """Path of the pretrained file in the local model store, or None (model_store.py:540-592 without the download)."""
    if model_name not in MODEL_STORE:
        return None
    error, sha1 = MODEL_STORE[model_name]
    root = os.path.expanduser(local_model_store_dir_path)
    path = os.path.join(root, "%s-%s-%s.pth" % (model_name, error, sha1[:8]))
    if not os.path.exists(path):
        return None
    if _sha1(path) != sha1:
        warnings.warn("Mismatch in the content of model file %s detected (no network to download it again)" % path)
        return None
    return path


class BackboneSpec:
    def __init__(self, arch, pretrained=False, root=os.path.join("~", ".torch", "models")):
        self.arch = arch
        self.pretrained_file = None
        if pretrained:
            self.pretrained_file = get_model_file(arch, root)
            if self.pretrained_file is None:
                warnings.warn("pretrained weights for %s not found in %s (no network): random init" % (arch, root))


def darknet53(pretrained=False, **kw):
    return BackboneSpec("darknet53", pretrained, **kw)


def darknet_tiny(pretrained=False, **kw):
    return BackboneSpec("darknet_tiny", pretrained, **kw)


def darknet_tiny_h(pretrained=False, **kw):
    return BackboneSpec("darknet_tiny_h", pretrained, **kw)
