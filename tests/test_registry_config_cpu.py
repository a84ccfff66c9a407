"""Host-side mirror of the reference's registry / builder / config semantics
(mmdet/utils/registry.py:6-76, mmdet/models/builder.py:7-33)."""
import os

import pytest

from util import ROOT


def test_registry_semantics():
    from slotvps_amd.registry import Registry, build_from_cfg
    R = Registry("thing")

    @R.register_module                      # bare decorator, no call parens
    class A:
        def __init__(self, x, y=2, train_cfg=None):
            self.x, self.y, self.train_cfg = x, y, train_cfg
    assert R.get("A") is A and "A" in R.module_dict
    with pytest.raises(KeyError):
        R.register_module(A)                # duplicate class name
    with pytest.raises(TypeError):
        R.register_module(lambda: 0)
    a = build_from_cfg(dict(type="A", x=1), R, dict(train_cfg="T", y=5))
    assert (a.x, a.y, a.train_cfg) == (1, 5, "T")
    b = build_from_cfg(dict(type=A, x=3, y=4), R)          # a class instead of a name
    assert (b.x, b.y) == (3, 4)
    with pytest.raises(KeyError):
        build_from_cfg(dict(type="Missing"), R)


def test_head_is_registered_and_builds_from_own_config():
    from slotvps_amd.config import Config
    from slotvps_amd.registry import HEADS, build_head
    import slotvps_amd.slot_head  # noqa: F401  (registers the heads)
    assert HEADS.get("MultiScaleDynamicMaskHead") is not None and HEADS.get("TemporalSlotsHead") is not None
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "r50_fpn_slotvps_mi355x.py"))
    assert cfg.model.other_config.pos_config.hidden_dim == 256          # attribute access on nested dicts
    head = build_head(dict(type="MultiScaleDynamicMaskHead", **cfg.model.dynamic_mask_head))
    assert sum(p.numel() for p in head.parameters()) == 15495052       # measured on the reference
    assert len(head.state_dict()) == 392
    # temporal sub-head only where the level's first stage index is a temporal stage (:83-106)
    assert head.head_series_0[0].temporal_query_head is None and head.head_series_1[0].temporal_query_head is None
    assert head.head_series_2[1].temporal_query_head is not None and head.head_series_3[0].temporal_query_head is not None
    # focal prior on every parameter whose last dim is num_classes (:127-136)
    assert abs(float(head.head_series_0[0].class_logits.bias.detach()[0]) + 4.59512) < 1e-4


@pytest.mark.skipif(not os.path.exists("/root/reference/configs/cityscapes/r50_fpn_slotvps.py"),
                    reason="reference tree not present (GPU box)")
def test_reference_config_file_loads_unchanged():
    from slotvps_amd.config import Config
    from slotvps_amd.registry import build_head
    import slotvps_amd.slot_head  # noqa: F401
    cfg = Config.fromfile("/root/reference/configs/cityscapes/r50_fpn_slotvps.py")
    assert cfg.model.type == "VPS_Temporal_Slots"
    dmh = dict(cfg.model.dynamic_mask_head)
    assert "type" not in dmh                                            # SURVEY: instantiated by class reference
    head = build_head(dict(type="MultiScaleDynamicMaskHead", other_config=cfg.model.other_config, **dmh))
    assert len(head.state_dict()) == 392
    swin = Config.fromfile("/root/reference/configs/cityscapes/swinL_fpn_slotvps.py")
    head2 = build_head(dict(type="MultiScaleDynamicMaskHead", **dict(swin.model.dynamic_mask_head)))
    assert head2.head_series_0[0].activation.__name__ == "relu"
