"""Golden fixture for the test-time flow AFTER the head - post-process, panoptic relabel, tracker assignment, result dict -
by RUNNING the reference's own VPS_Temporal_Slots.simple_test (mmdet/models/detectors/vps_temporal_slots.py:207-469) on a
four-frame synthetic video in the build container.

The detector object is created without its constructor (which needs mmcv / the CUDA-only ops) and given stand-ins for
everything UPSTREAM of the head outputs, none of them arithmetic of the pinned part: backbone / neck / semantic head / slot
head return the canned tensors of slotvps_amd.synth.make_simple_test_case; generate_final_outputs returns the canned mask
logits (that function is pinned by tests/golden/head_small.npz on its own). Everything downstream is the reference's
code as it stands: PostProcessPanopticInstances, the Instances container, SimpleTrackHead (mmdet/models/detectors/
simple_track_head.py, seeded weights), the greedy assignment (:328-409), the relabel (:411-435), the result dict.
Other stand-ins as in tests/golden/make_golden_post.py (registry decorators, auto_fp16, DataContainer, Tensor.cuda() -> no-op,
torch.cuda.current_device() -> "cpu",
panopticapi's id2rgb / rgb2id restated - un-vendored dependency, unused on this path).
Stored: the result dict of every frame and the tracker memory after the last frame (inputs are regenerated from the seed).
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
SEED, N_FRAMES, L, LH, LW = 31, 4, 100, 16, 32
POST = dict(is_thing_map={i: i > 10 for i in range(20)}, threshold=0.85, fraction_threshold=0.03, pixel_threshold=0.4,
            apply_mask_removal=True, apply_mask_removal_only_ins=True, use_mask_low_constant=False)


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

    class DataContainer:
        pass

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

    mod("mmdet"); mod("mmdet.core", auto_fp16=lambda apply_to=None: (lambda f: f)); mod("mmdet.core.utils")
    mod("mmdet.core.utils.misc", NestedTensor=NestedTensor, nested_tensor_from_tensor_list=None,
        interpolate=torch.nn.functional.interpolate)
    mod("mmcv"); mod("mmcv.parallel", DataContainer=DataContainer)
    mod("panopticapi"); mod("panopticapi.utils", rgb2id=rgb2id, id2rgb=id2rgb)
    mod("refpkg3"); mod("refpkg3.models"); mod("refpkg3.models.detectors")
    mod("refpkg3.models.registry", DETECTORS=_Reg(), HEADS=_Reg())
    mod("refpkg3.models.utils"); mod("refpkg3.models.utils.conv_module", init_weights=None)
    mod("refpkg3.models.detectors.vps_capsule", VPS_Capsule=object)

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    inst = load("refpkg3.models.structures.instances", "mmdet/models/structures/instances.py")
    mod("refpkg3.models.structures", Instances=inst.Instances)
    load("refpkg3.models.detectors.simple_track_head", "mmdet/models/detectors/simple_track_head.py")
    return load("refpkg3.models.detectors.vps_temporal_slots", "mmdet/models/detectors/vps_temporal_slots.py")


def main():
    torch.set_grad_enabled(False)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.current_device = lambda: "cpu"        # SimpleTrackHead.forward allocates its zero column on "the current GPU" (:90)
    vts = load_reference()
    frames, fc_w, fc_b = synth.make_simple_test_case(SEED, N_FRAMES, L, LH, LW)
    H, W = 4 * LH, 4 * LW

    det = vts.VPS_Temporal_Slots.__new__(vts.VPS_Temporal_Slots)
    torch.nn.Module.__init__(det)
    det.num_classes, det.stuff_num = 20, 11
    det.other_config = {"test_forward_ref_img": True}
    det.postprocess_panoptic = vts.PostProcessPanopticInstances(**POST)
    det.temporal_track_head = vts.SimpleTrackHead(num_fcs_query=2, in_channels_query=256)
    for fc, w_, b_ in zip(det.temporal_track_head.fcs_query, fc_w, fc_b):
        fc.weight.copy_(torch.from_numpy(w_))
        fc.bias.copy_(torch.from_numpy(b_))
    cur = {}
    t = lambda a: torch.from_numpy(a)[None]
    det.image_model = types.SimpleNamespace(
        backbone=lambda img: [img], with_neck=False, panopticFPN=types.SimpleNamespace(num_levels=4),
        init_mask_query=torch.nn.Embedding(L, 256),
        dynamic_mask_head=lambda **kw: ([[t(cur["f"]["logits"])], [t(cur["f"]["logits"])]],
                                        [[t(cur["f"]["embed"])], [t(cur["f"]["embed"])]],
                                        [[torch.zeros(1, 256, LH, LW)], [torch.zeros(1, 256, LH, LW)]]))
    det.extract_semantic_feats = lambda x: (t(cur["f"]["fcn"]), None, [torch.zeros(1, 128, LH, LW)])
    det.semantic_trans_ins = lambda feats: feats
    det.generate_position_embedding = lambda feats: None
    det.generate_final_outputs = lambda feats, outputs_masks, generate_aux_output=True: (feats, t(cur["f"]["masks"]), [])

    out = {}
    img = torch.zeros(1, 3, H, W)
    for f, fr in enumerate(frames):
        cur["f"] = fr
        meta = dict(iid=3 * 10000 + f + 1, filename=f"v3_f{f + 1}.png", ori_shape=(H, W, 3), img_shape=(H, W, 3))
        r = det.simple_test(img, [meta], rescale=True, ref_img=[img])
        out[f"f{f}_fcn_outputs"] = r["fcn_outputs"].numpy().astype(np.uint8)
        out[f"f{f}_panoptic_outputs"] = r["panoptic_outputs"].numpy().astype(np.uint8)
        out[f"f{f}_panoptic_cls_inds"] = r["panoptic_cls_inds"].numpy().astype(np.int64)
        out[f"f{f}_panoptic_cls_prob"] = r["panoptic_cls_prob"].numpy().astype(np.float32)
        out[f"f{f}_panoptic_det_obj_ids"] = r["panoptic_det_obj_ids"].numpy().astype(np.int64)
        print(f, "cls_inds", out[f"f{f}_panoptic_cls_inds"].tolist(), "obj_ids", out[f"f{f}_panoptic_det_obj_ids"].tolist(),
              "ids", np.unique(out[f"f{f}_panoptic_outputs"]).tolist())
    out["memory"] = det.prev_instances.output_embedding.numpy().astype(np.float32)

    # ---- a8 by the reference's own generate_final_outputs (:144-160) ------------------------------------------------
    # (tests/golden/make_golden.py could only execute the torch ops of those lines: that module loader has no vps_temporal_slots)
    case = synth.make_decode_case(SEED + 1)
    feat_bn, fg_bn = torch.nn.BatchNorm2d(256).eval(), torch.nn.BatchNorm2d(1).eval()
    for bn, (w_, b_, mu, var) in ((feat_bn, case["feat_bn"]), (fg_bn, case["fg_bn"])):
        bn.weight.copy_(torch.from_numpy(w_)); bn.bias.copy_(torch.from_numpy(b_))
        bn.running_mean.copy_(torch.from_numpy(mu)); bn.running_var.copy_(torch.from_numpy(var))
    fake = types.SimpleNamespace(image_model=types.SimpleNamespace(feat_bn=feat_bn, fg_bn=fg_bn), other_config={})
    _, mask, _ = vts.VPS_Temporal_Slots.generate_final_outputs(
        fake, [torch.from_numpy(case["feat"])[None].clone()], [torch.from_numpy(case["embed"])[None]], generate_aux_output=False)
    out["decode_mask"] = mask[0].numpy().astype(np.float32)
    out["decode_seed"] = np.array([SEED + 1], dtype=np.int64)
    print("decode", out["decode_mask"].shape, float(np.abs(out["decode_mask"]).max()))
    out["meta"] = np.array([SEED, N_FRAMES, L, LH, LW], dtype=np.int64)
    np.savez_compressed(os.path.join(GOLDEN, "simple_test.npz"), **out)
    print("simple_test.npz", os.path.getsize(os.path.join(GOLDEN, "simple_test.npz")) // 1024, "KiB; memory", out["memory"].shape)


if __name__ == "__main__":
    main()
