"""Golden fixtures for the panoptic post-process by RUNNING the reference's
PostProcessPanopticInstances (mmdet/models/detectors/vps_temporal_slots.py:528-807) in the build container.

Stand-ins (none of them arithmetic of the path, except the one flagged):
  mmdet.core.auto_fp16 (identity decorator), ..registry.DETECTORS (identity), mmcv.parallel.DataContainer,
  .vps_capsule.VPS_Capsule / .simple_track_head.SimpleTrackHead / ..utils.conv_module.init_weights (unused by
  the post-process class), mmdet.core.utils.misc.{NestedTensor, nested_tensor_from_tensor_list, interpolate}
  (interpolate = torch.nn.functional.interpolate, what the real wrapper calls for non-empty inputs),
  Tensor.cuda() -> no-op (the class moves numpy results back with .cuda(), :655-656; no GPU here).
  RESTATED: panopticapi.utils.id2rgb / rgb2id (un-vendored dependency) - published base-256 pack/unpack.
The reference's own Instances container (mmdet/models/structures/instances.py) is imported as is.
Outputs only are stored (tests regenerate the inputs from slotvps_amd.synth.make_post_case).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from slotvps_amd import synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def load_reference():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Reg:
        def register_module(self, cls):
            return cls

    def auto_fp16(apply_to=None):
        return lambda f: f

    class NestedTensor:
        def __init__(self, tensors, mask):
            self.tensors, self.mask = tensors, mask

    def id2rgb(id_map):
        id_map = np.asarray(id_map)
        rgb = np.zeros(id_map.shape + (3,), dtype=np.uint8)
        tmp = id_map.copy()
        for i in range(3):
            rgb[..., i] = tmp % 256
            tmp = tmp // 256
        return rgb

    def rgb2id(color):
        color = np.asarray(color).astype(np.int32)
        return color[..., 0] + 256 * color[..., 1] + 256 * 256 * color[..., 2]

    mod("mmdet"); mod("mmdet.core", auto_fp16=auto_fp16); mod("mmdet.core.utils")
    mod("mmdet.core.utils.misc", NestedTensor=NestedTensor, nested_tensor_from_tensor_list=None,
        interpolate=torch.nn.functional.interpolate)
    mod("mmcv"); mod("mmcv.parallel", DataContainer=object)
    mod("panopticapi"); mod("panopticapi.utils", rgb2id=rgb2id, id2rgb=id2rgb)
    mod("refpkg2"); mod("refpkg2.models"); mod("refpkg2.models.detectors")
    mod("refpkg2.models.registry", DETECTORS=_Reg())
    mod("refpkg2.models.utils"); mod("refpkg2.models.utils.conv_module", init_weights=None)
    mod("refpkg2.models.detectors.vps_capsule", VPS_Capsule=object)
    mod("refpkg2.models.detectors.simple_track_head", SimpleTrackHead=object)
    spec = importlib.util.spec_from_file_location("refpkg2.models.structures.instances",
                                                  os.path.join(REF, "mmdet/models/structures/instances.py"))
    inst = importlib.util.module_from_spec(spec); sys.modules[spec.name] = inst; spec.loader.exec_module(inst)
    mod("refpkg2.models.structures", Instances=inst.Instances)
    spec = importlib.util.spec_from_file_location("refpkg2.models.detectors.vps_temporal_slots",
                                                  os.path.join(REF, "mmdet/models/detectors/vps_temporal_slots.py"))
    m = importlib.util.module_from_spec(spec); sys.modules[spec.name] = m; spec.loader.exec_module(m)
    return m, inst.Instances


def main():
    torch.set_grad_enabled(False)
    torch.Tensor.cuda = lambda self, *a, **k: self
    vts, Instances = load_reference()
    pp = vts.PostProcessPanopticInstances(is_thing_map={i: i > 10 for i in range(20)}, threshold=0.85,
                                          fraction_threshold=0.03, pixel_threshold=0.4, apply_mask_removal=True,
                                          apply_mask_removal_only_ins=True, use_mask_low_constant=False)
    out = {}
    for tag, (seed, L, h, w, nk) in {"a": (11, 100, 16, 32, 18), "b": (12, 100, 24, 40, 26), "c": (13, 60, 12, 20, 9)}.items():
        logits, masks = synth.make_post_case(seed, L, h, w, 20, nk)
        inst = Instances((1, 1))
        inst.pred_logits = torch.from_numpy(logits)
        inst.pred_masks = torch.from_numpy(masks)
        inst.slot_index = torch.arange(L)
        res = pp(inst, [(4 * h, 4 * w)], id=0)
        out[f"{tag}_slot_index"] = res.slot_index.numpy().astype(np.int64)
        out[f"{tag}_labels"] = res.labels.numpy().astype(np.int64)
        out[f"{tag}_probs"] = res.probs.numpy().astype(np.float32)
        out[f"{tag}_masks"] = res.masks.numpy().astype(np.float32)
        out[f"{tag}_meta"] = np.array([seed, L, h, w, nk], dtype=np.int64)
        print(tag, "kept", len(res.labels), "labels", res.labels.tolist())
    np.savez_compressed(os.path.join(GOLDEN, "postprocess.npz"), **out)
    print("postprocess.npz", os.path.getsize(os.path.join(GOLDEN, "postprocess.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
