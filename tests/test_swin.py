"""Swin backbone (PyTorch, config 4 of BASELINE): parameter names / shapes of the Swin-L configuration and the outputs
of a small seeded configuration against the reference's own module (tests/golden/swin.npz, tests/golden/make_golden_swin.py)."""
import os
import sys

import numpy as np
import pytest
import torch

from util import GOLDEN, ROOT
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden_swin import LARGE, SMALL, seeded_state   # noqa: E402  (seed / config shared with the generator)

from slotvps_amd.config import Config
from slotvps_amd.registry import BACKBONES, build_detector
from slotvps_amd.swin import SwinTransformer


def test_swin_large_parameter_names_match_reference():
    z = np.load(os.path.join(GOLDEN, "swin.npz"))
    sd = SwinTransformer(**LARGE).state_dict()
    assert list(sd) == z["large_keys"].tolist()
    assert [",".join(map(str, v.shape)) for v in sd.values()] == z["large_shapes"].tolist()
    assert sum(p.numel() for p in SwinTransformer(**LARGE).parameters()) > 190e6          # Swin-L trunk


def test_swin_small_outputs_match_reference():
    z = np.load(os.path.join(GOLDEN, "swin.npz"))
    m = SwinTransformer(**SMALL).eval()
    m.load_state_dict(seeded_state(m, 7))
    x = torch.randn(2, 3, 70, 91, generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        outs = m(x)
    assert [tuple(o.shape) for o in outs] == [(2, 32, 18, 23), (2, 64, 9, 12), (2, 128, 5, 6), (2, 256, 3, 3)]
    for i, o in enumerate(outs):
        ref = z[f"out{i}"]
        err = np.abs(o.numpy() - ref).max()
        assert err <= 2e-4 * max(1.0, np.abs(ref).max()), (i, err)       # fused attention vs explicit softmax: fp32 reassociation


def test_swin_config_resolves():
    ref_cfg = "/root/reference/configs/cityscapes/swinL_fpn_slotvps.py"
    if not os.path.exists(ref_cfg):
        pytest.skip("reference tree not present (GPU box)")
    cfg = Config.fromfile(ref_cfg)
    det = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    assert type(det.image_model.backbone).__name__ == "SwinTransformer" and BACKBONES.get("SwinTransformer") is not None
    head = det.image_model.dynamic_mask_head
    assert sum(p.numel() for p in det.parameters()) > 200e6
    assert head.head_series_0[0].activation.__name__ in ("relu",)       # swinL config: FFN ReLU, temporal GELU


@pytest.mark.gpu
def test_swin_small_outputs_match_reference_on_gpu():
    """Same fixture through the ROCm attention kernels (bias + shifted-window mask as one additive attention bias)."""
    z = np.load(os.path.join(GOLDEN, "swin.npz"))
    dev = torch.device("cuda:0")
    m = SwinTransformer(**SMALL).eval()
    m.load_state_dict(seeded_state(m, 7))
    m.to(dev)
    x = torch.randn(2, 3, 70, 91, generator=torch.Generator().manual_seed(8)).to(dev)
    with torch.no_grad():
        outs = m(x)
    for i, o in enumerate(outs):
        ref = z[f"out{i}"]
        assert np.abs(o.cpu().numpy() - ref).max() <= 5e-4 * max(1.0, np.abs(ref).max())
