"""CPU: host-side logic of the product path that needs no device (parameter store / state_dict
surface, CLI+config surface, synthetic batches, sampler, lr schedule vs the reference capture)."""
import os

import numpy as np
import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _cfg(arch):
    from kd6d.arguments.argument import custom_cfg
    with open(os.path.join(ROOT, "configs", "ape.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg["RUNTIME"] = {"PRECISION": "bf16"}
    cfg["MODEL"]["BACKBONE"] = arch
    cfg = custom_cfg(cfg)
    cfg["KD"] = dict(LOSS_WEIGHT_KD=5.0, LEVEL="pred", GLEVEL="point", GTYPE="sinkhorn", GP=2.0, GBLUR=0.001, GnD=2,
                     WEIGHTED_OT=True, DETACH=False, SCALING=0.5, REACH=0.5)
    return cfg


@pytest.mark.parametrize("arch,n_keys,n_params", [("darknet_tiny_h", 150, 2304468), ("darknet_tiny", 150, 8486076),
                                                   ("darknet53", 376, 52101836)])
def test_state_dict_surface_matches_reference(arch, n_keys, n_params):
    """key names, shapes and parameter counts of SURVEY App. C.3 / train_kd.py:76-78."""
    from kd6d import backbone as BB
    from kd6d.models.model_kd import PoseModuleKD
    from oracle import kd_step_ref as O
    m = PoseModuleKD(_cfg(arch), getattr(BB, arch)())
    sd = m.state_dict()
    ref = O.PoseNetRef(arch)
    rsd = ref.state_dict()
    assert len(sd) == n_keys and set(sd) == set(rsd)
    assert all(sd[k].shape == rsd[k].shape for k in rsd)
    assert sum(p.numel() for p in m.parameters()) == n_params
    seeded = O.seeded_state_dict(ref, 5)
    m.load_state_dict(seeded)
    back = m.state_dict()
    assert all(torch.equal(back[k].float(), seeded[k].float()) for k in seeded)
    # storage really is the flat KRSC buffer: padded input channels of the first conv stay zero
    e = m.net.store.entries["backbone.features.stage1.unit1.conv.weight" if arch != "darknet53"
                            else "backbone.features.init_block.conv.weight"]
    raw = m.net.store.storage(e).view(e.store_shape)
    assert float(raw[..., 3:].abs().max()) == 0.0 and float(raw[..., :3].abs().max()) > 0.0


def test_unused_parameters_are_outside_the_optimised_region():
    from kd6d import backbone as BB
    from kd6d.models.model_kd import PoseModuleKD
    m = PoseModuleKD(_cfg("darknet_tiny_h"), BB.darknet_tiny_h())
    st = m.net.store
    frozen = [e.name for e in st.order if e.region == "frozen"]
    assert sorted(frozen) == ["backbone.output.final_conv.bias", "backbone.output.final_conv.weight",
                              "head.scales.4.scale"]


def test_cli_surface_and_cfg():
    from kd6d.arguments.argument_kd import get_argparser, get_args
    flags = {a.option_strings[0] for a in get_argparser()._actions if a.option_strings}
    for f in ["--local_rank", "--config_file", "--num_workers", "--working_dir", "--test_file", "--weight_file",
              "--running_device", "--backbone", "--max_iters", "--base_lr", "--config_file_t", "--backbone_t",
              "--weight_file_t", "--kd_weight", "--kd_level", "--gtype", "--glevel", "--p", "--blur", "--gnD",
              "--weightedOT", "--wot_detach", "--scaling", "--reach"]:
        assert f in flags, f
    y = os.path.join(ROOT, "configs", "ape.yaml")
    cfg, cfg_t = get_args(["--config_file", y, "--config_file_t", y, "--kd_weight", "5.", "--max_iters", "10000",
                           "--wot_detach", "true"])
    assert cfg["KD"] == dict(LOSS_WEIGHT_KD=5.0, LEVEL="pred", GLEVEL="point", GTYPE="sinkhorn", GP=2.0, GBLUR=0.001,
                             GnD=2, WEIGHTED_OT=True, DETACH=True, SCALING=0.5, REACH=0.5)
    assert cfg["MODEL"]["FEAT_CHANNELS"] == [0, 0, 64, 64] and cfg["MODEL"]["OUT_CHANNEL"] == 128
    assert cfg_t["MODEL"]["BACKBONE"] == "darknet53" and cfg_t["MODEL"]["FEAT_CHANNELS"] == [0, 0, 256, 512, 1024]
    assert cfg["SOLVER"]["GRAD_CLIP"] == 1.0 and cfg["SOLVER"]["POSITIVE_NUM"] == 10 and cfg["SOLVER"]["MAX_ITER"] == 10000


def test_unsupported_options_fail_loudly():
    from kd6d.kd_losses import KDLoss
    from kd6d.synthetic import INTERNAL_K, MESH_DIAMETERS
    with pytest.raises(NotImplementedError):
        KDLoss(INTERNAL_K, MESH_DIAMETERS, kd_cfg=dict(GTYPE="energy"))
    with pytest.raises(NotImplementedError):
        KDLoss(INTERNAL_K, MESH_DIAMETERS, kd_cfg=dict(GTYPE="sinkhorn", WEIGHTED_OT=False))


def test_synthetic_batches_are_seeded_and_linemod_shaped():
    from kd6d.synthetic import make_batch
    a_img, a_t = make_batch(3, 7)
    b_img, b_t = make_batch(3, 7)
    assert torch.equal(a_img.tensors, b_img.tensors) and a_img.tensors.shape == (3, 3, 256, 256)
    t = a_t[0]
    assert t.keypoints_3d.shape == (15, 8, 3) and t.mask.shape == (256, 256) and t.bbox_trans.shape == (2, 3)
    assert torch.equal(t.bbox_trans, b_t[0].bbox_trans)
    R = t.rotations[0]
    assert torch.allclose(R @ R.T, torch.eye(3), atol=1e-5) and float(torch.det(R)) > 0
    # projected box is centred in the crop with extent 1/1.5 of it
    kp = t.keypoints_3d[int(t.class_ids[0])]
    uv = t.K @ (R @ kp.T + t.translations[0])
    xy = t.bbox_trans[:, :2] @ (uv[:2] / uv[2]) + t.bbox_trans[:, 2:3]
    ext = max(float(xy[0].max() - xy[0].min()), float(xy[1].max() - xy[1].min()))
    assert ext == pytest.approx(256 / 1.5, rel=1e-3)
    m_img, m_t = make_batch(14, 1, mixed_classes=True)
    assert len({int(x.class_ids[0]) for x in m_t}) == 13


def test_distributed_sampler_semantics():
    from kd6d.libs.distributed import DistributedSampler
    data = list(range(10))
    parts = [list(DistributedSampler(data, num_replicas=4, rank=r, shuffle=True)) for r in range(4)]
    assert all(len(p) == 3 for p in parts)
    flat = sum(parts, [])
    g = torch.Generator(); g.manual_seed(0)
    perm = torch.randperm(10, generator=g).tolist()
    assert flat == perm + perm[:2]          # wrap-around padding, contiguous rank slices


def test_onecycle_lr_trajectory_matches_reference_capture():
    z = np.load(os.path.join(G, "optim.npz"))
    p = torch.nn.Parameter(torch.zeros(4))
    opt = torch.optim.SGD([p], lr=1e-3)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, 1e-3, 10100, pct_start=0.05, cycle_momentum=False,
                                              anneal_strategy="linear")
    lrs = []
    for _ in range(3):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step(); sch.step()
    np.testing.assert_allclose(lrs, z["lrs"], rtol=1e-12)


def test_model_zoo_file_and_backbone_only_state_dict(tmp_path, monkeypatch):
    """backbone/model_store.py:540-592 + libs/train_libs.py:82-87: `pretrained=True` resolves
    `<root>/<name>-<error>-<sha1[:8]>.pth` in the local model store (no download), checks its SHA-1, and the
    backbone-only state_dict (`features.*`, `output.*`) lands under `backbone.*` of the pose module, leaving FPN and
    head at their initialisation."""
    import hashlib
    import warnings
    import torch
    from kd6d import backbone as BB
    from kd6d.models.model_kd import PoseModuleKD
    from oracle import kd_step_ref as O
    assert BB.MODEL_STORE["darknet53"] == ("0564", "b36bef6b297055dda3d17a3f79596511730e1963")
    assert BB.MODEL_STORE["darknet_tiny"][0] == "1784"
    root = str(tmp_path)
    zoo = O.DarkNetTinyRef("darknet_tiny")
    sd = O.seeded_state_dict(zoo, 9)
    tmp = os.path.join(root, "blob.pth")
    torch.save(sd, tmp)
    sha = hashlib.sha1(open(tmp, "rb").read()).hexdigest()
    monkeypatch.setitem(BB.MODEL_STORE, "darknet_tiny", ("1784", sha))
    os.rename(tmp, os.path.join(root, "darknet_tiny-1784-%s.pth" % sha[:8]))
    spec = BB.darknet_tiny(pretrained=True, root=root)
    assert spec.pretrained_file and spec.pretrained_file.endswith("darknet_tiny-1784-%s.pth" % sha[:8])
    torch.manual_seed(0)
    m = PoseModuleKD(_cfg("darknet_tiny"), spec)
    got = m.state_dict()
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        assert torch.equal(got["backbone." + k], v), k
    assert float(got["head.cls_logits.bias"][0]) == pytest.approx(-4.59512, abs=1e-4)      # untouched: prior init
    # wrong content -> ignored with a warning; missing -> random init with a warning
    monkeypatch.setitem(BB.MODEL_STORE, "darknet_tiny", ("1784", "0" * 40))
    os.rename(os.path.join(root, "darknet_tiny-1784-%s.pth" % sha[:8]), os.path.join(root, "darknet_tiny-1784-00000000.pth"))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert BB.darknet_tiny(pretrained=True, root=root).pretrained_file is None
        assert BB.darknet53(pretrained=True, root=root).pretrained_file is None
    assert any("Mismatch" in str(x.message) for x in w) and any("random init" in str(x.message) for x in w)
    assert BB.darknet_tiny_h(pretrained=False).pretrained_file is None
