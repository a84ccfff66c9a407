"""PyTorch backbone / neck mirrors against the reference's own ResNet-50 and FPN (tests/golden/backbone.npz, made by
tests/golden/make_golden_backbone.py): state-dict key / shape lists (the checkpoint contract) and outputs on a seeded input."""
import os
import sys

import numpy as np
import torch

from util import GOLDEN, ROOT
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden_backbone import FPN as FPN_CFG, R50, UPS, seeded_state   # noqa: E402

from slotvps_amd.backbones import FPN, ResNet, UPSNetFPN


def test_resnet50_and_fpn_match_reference():
    z = np.load(os.path.join(GOLDEN, "backbone.npz"))
    bb, neck = ResNet(**R50).eval(), FPN(**FPN_CFG).eval()
    for m, tag in ((bb, "resnet"), (neck, "fpn")):
        sd = m.state_dict()
        assert list(sd) == z[f"{tag}_keys"].tolist()
        assert [",".join(map(str, v.shape)) for v in sd.values()] == z[f"{tag}_shapes"].tolist()
    bb.load_state_dict(seeded_state(bb, 3))
    neck.load_state_dict(seeded_state(neck, 4))
    x = torch.randn(2, 3, 64, 96, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        c = bb(x)
        p = neck(c)
    assert len(c) == 4 and len(p) == 5
    for i, o in enumerate(c):
        ref = z[f"c{i}"]
        assert o.shape == ref.shape and np.abs(o.numpy() - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    for i, o in enumerate(p):
        ref = z[f"p{i}"]
        assert o.shape == ref.shape and np.abs(o.numpy() - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())


def test_semantic_tower_checkpoint_contract():
    """UPSNetFPN's parameter names / shapes equal the reference module's (tests/golden/semantic_tower.npz)."""
    z = np.load(os.path.join(GOLDEN, "semantic_tower.npz"))
    sd = UPSNetFPN(**UPS).state_dict()
    assert list(sd) == z["keys"].tolist()
    assert [",".join(map(str, v.shape)) for v in sd.values()] == z["shapes"].tolist()
