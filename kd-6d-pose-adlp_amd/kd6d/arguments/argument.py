"""Derived configuration keys (the hard-coded layer of arguments/argument.py:51-104)."""

_BACKBONES = {
    # name: (FEAT_CHANNELS, OUT_CHANNEL, VAL_FREQ)
    "darknet_tiny": ([0, 0, 128, 128], 256, 500),
    "darknet_tiny_h": ([0, 0, 64, 64], 128, 500),       # half the channels of darknet_tiny
    "darknet53": ([0, 0, 256, 512, 1024], 256, 2000),
}

_SOLVER_DEFAULTS = {
    "GRAD_CLIP": 1.0, "VAL_FREQ": 5000, "AUGMENTATION_OCCLUSION": 0, "AUGMENTATION_Grayscalize": False,
    "AUGMENTATION_Smooth": 0, "AUGMENTATION_Sharpen": 0, "AUGMENTATION_BACKGROUND_DIR": None,
}


def custom_cfg(cfg):
    name = cfg["MODEL"]["BACKBONE"]
    if name not in _BACKBONES:
        raise AssertionError("Unsupported backbone %r (the HIP path implements %s)" % (name, sorted(_BACKBONES)))
    feat, out_c, val_freq = _BACKBONES[name]
    cfg["MODEL"]["OUT_CHANNEL"] = out_c
    cfg["MODEL"]["FEAT_CHANNELS"] = list(feat)
    cfg["SOLVER"]["VAL_FREQ"] = val_freq
    cfg["MODEL"]["N_CONV"] = 4
    cfg["MODEL"]["PRIOR"] = 0.01
    cfg["MODEL"].setdefault("USE_HIGHER_LEVELS", True)
    cfg["SOLVER"].update(FOCAL_GAMMA=2.0, FOCAL_ALPHA=0.25, TOP_K=9, POSITIVE_NUM=10)
    cfg["INPUT"].update(PIXEL_MEAN=[0.485, 0.456, 0.406], PIXEL_STD=[0.229, 0.224, 0.225], SIZE_DIVISIBLE=32)
    for k, v in _SOLVER_DEFAULTS.items():
        cfg["SOLVER"].setdefault(k, v)
    cfg["DATASETS"].setdefault("SYMMETRY_TYPES", {})
    return cfg
