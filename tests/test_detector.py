"""Detector-level mirror: config resolution, parameter names, the Instances container, the
tracker's greedy assignment against the oracle restatement, and (GPU) the whole test-time flow of
VPS_Temporal_Slots against the CPU oracle applied to the same head outputs."""
import os

import numpy as np
import pytest
import torch

from oracle import postprocess_oracle as porc
from slotvps_amd.config import Config
from slotvps_amd.detector import SimpleTrackHead, greedy_track_assign
from slotvps_amd.instances import Instances
from slotvps_amd.registry import BACKBONES, DETECTORS, HEADS, NECKS, PANOPTIC, build_detector

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "configs", "r50_fpn_slotvps_mi355x.py")
REF_CFG = "/root/reference/configs/cityscapes/r50_fpn_slotvps.py"


def build(cfg_file=CFG):
    cfg = Config.fromfile(cfg_file)
    return build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)


def test_registry_resolves_detector_types():
    det = build()
    for reg, name in [(BACKBONES, "ResNet"), (NECKS, "FPN"), (PANOPTIC, "UPSNetFPN"), (HEADS, "SimpleTrackHead"),
                      (DETECTORS, "VPS_Temporal_Slots"), (DETECTORS, "VPS_Capsule")]:
        assert reg.get(name) is not None
    sd = det.state_dict()
    assert len(sd) == 760
    # key layout of a reference checkpoint (module attribute names of vps_temporal_slots.py / vps_capsule.py /
    # resnet.py / fpn.py / upsnetFPN.py / simple_track_head.py)
    for k in ["image_model.backbone.conv1.weight", "image_model.backbone.layer3.5.bn3.running_var",
              "image_model.backbone.layer2.0.downsample.0.weight", "image_model.neck.lateral_convs.3.conv.bias",
              "image_model.neck.fpn_convs.0.conv.weight", "image_model.panopticFPN.deform_convs.0.0.conv_offset.weight",
              "image_model.panopticFPN.deform_convs.0.6.conv.weight", "image_model.panopticFPN.deform_convs.0.7.bias",
              "image_model.panopticFPN.conv_pred.conv.weight", "image_model.init_mask_query.weight",
              "image_model.conv_trans.conv.weight", "image_model.dynamic_mask_head.conv_trans.conv.weight",
              "image_model.dynamic_mask_head.head_series_3.1.temporal_query_head.inst_interact.to_q.weight",
              "image_model.fg_bn.weight", "image_model.feat_bn.running_mean", "temporal_track_head.fcs_query.1.bias"]:
        assert k in sd, k
    assert tuple(sd["image_model.init_mask_query.weight"].shape) == (100, 256)
    assert tuple(sd["image_model.panopticFPN.conv_pred.conv.weight"].shape) == (19, 512, 1, 1)
    assert float(sd["image_model.fg_bn.weight"]) == pytest.approx(0.1)             # vps_capsule.py:129
    assert not sd["image_model.panopticFPN.deform_convs.0.0.conv_offset.weight"].any()   # zero-initialised offsets
    assert sum(p.numel() for p in det.image_model.backbone.parameters()) == 23508032      # torchvision R50 trunk


@pytest.mark.skipif(not os.path.exists(REF_CFG), reason="reference tree not present (GPU box)")
def test_reference_config_builds_same_model():
    with pytest.warns(UserWarning, match="model zoo"):
        ref = build(REF_CFG)
    ours = build()
    a, b = ref.state_dict(), ours.state_dict()
    assert list(a) == list(b)
    assert all(a[k].shape == b[k].shape for k in a)
    assert ref.postprocess_panoptic.threshold == 0.85 and ref.postprocess_panoptic.pixel_threshold == 0.4
    assert ref.stuff_num == 11 and ref.num_classes == 20


def test_train_path_is_not_released():
    det = build()
    with pytest.raises(AssertionError, match="NOT RELEASED TRAIN CODE"):
        det(img=None, img_meta=None, return_loss=True)
    with pytest.raises(TypeError):
        det(img=torch.zeros(1, 3, 8, 8), img_meta=[[{}]], return_loss=False)


def test_instances_container():
    a = Instances((1, 1), emb=torch.arange(12.).view(4, 3), ids=torch.tensor([3, 1, 2, 0]))
    assert len(a) == 4 and a.has("emb") and not a.has("x")
    b = a[torch.tensor([True, False, True, False])]
    assert b.ids.tolist() == [3, 2] and b.emb.shape == (2, 3)
    c = a[1]
    assert len(c) == 1 and c.ids.tolist() == [1]
    d = Instances.cat([a[:1], c, a[2:]])
    assert d.ids.tolist() == [3, 1, 2, 0]
    with pytest.raises(AssertionError):
        a.bad = torch.zeros(3)
    with pytest.raises(AttributeError):
        a.missing
    with pytest.raises(IndexError):
        a[4]
    assert a.to(torch.device("cpu")).ids.tolist() == [3, 1, 2, 0]


@pytest.mark.parametrize("seed", range(6))
def test_tracker_matches_oracle(seed):
    rng = np.random.default_rng(seed)
    K, P, D = int(rng.integers(1, 14)), int(rng.integers(1, 14)), 32
    head = SimpleTrackHead(num_fcs_query=2, in_channels_query=D)
    with torch.no_grad():
        for fc in head.fcs_query:
            fc.weight.normal_(0, 0.4, generator=torch.Generator().manual_seed(seed))
            fc.bias.normal_(0, 0.1, generator=torch.Generator().manual_seed(seed + 9))
    cur = rng.standard_normal((K, D)).astype(np.float32)
    prev = rng.standard_normal((P, D)).astype(np.float32)
    if K > 2 and P > 1:
        cur[1] = prev[0] * 2.0            # force a contested match: two segments both closest to tracked object 0
        cur[2] = prev[0] * 2.1
    with torch.no_grad():
        score = head(torch.from_numpy(cur), torch.from_numpy(prev))[0]
        logprob = torch.log_softmax(score, 1).numpy()
    det, updates = greedy_track_assign(logprob, P)
    mem = [p for p in prev]
    for p, c in updates:
        if p == len(mem):
            mem.append(cur[c])
        else:
            mem[p] = cur[c]
    fw = [fc.weight.detach().numpy() for fc in head.fcs_query]
    fb = [fc.bias.detach().numpy() for fc in head.fcs_query]
    want_det, want_mem = porc.track_assign(cur, prev, fw, fb)
    assert det.tolist() == want_det.tolist()
    np.testing.assert_array_equal(np.stack(mem), want_mem)
    assert len(set(det.tolist())) == K                     # ids are unique within a frame
    assert score.shape == (K, P + 1) and not score[:, 0].any()


# ------------------------------------------------ GPU: the whole test-time flow --------------------------------
def _make_detector(dev):
    torch.manual_seed(0)
    det = build().to(dev).eval()
    with torch.no_grad():
        det.image_model.fg_bn.weight.fill_(40.0)            # spread the mask logits so that things claim pixels
        for fc in det.temporal_track_head.fcs_query:
            fc.weight.normal_(0, 0.2)
        cls = det.image_model.dynamic_mask_head.head_series_3[-1].class_logits
        cls.weight.normal_(0, 0.5)                          # confident, varied classes from random weights
    det._fold = None
    det.postprocess_panoptic.threshold = 0.3
    return det


@pytest.mark.gpu
def test_clip_flow_matches_oracle_pipeline():
    dev = torch.device("cuda:0")
    det = _make_detector(dev)
    T, H, W = 3, 128, 256
    imgs = torch.randn(T, 3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    metas = [dict(iid=10001 + t, ori_shape=(H, W, 3), img_shape=(H, W, 3), filename=f"f{t}.png") for t in range(T)]
    logits, embeds, masks, fcn = det.slot_path(imgs)
    assert logits.shape == (T, 100, 20) and embeds.shape == (T, 100, 256) and masks.shape == (T, 100, H // 4, W // 4)
    assert fcn.shape == (T, 19, H, W)
    # the decode runs on demand for the slots the post-process keeps; the kept rows equal those of the all-slot decode bit for bit
    pick = torch.tensor([7, 3, 99, 0, 41], device=dev)
    assert torch.equal(masks[1].decode_slots(pick), masks.dense()[1][pick])
    # random-init slots all predict one class: add a fixed per-slot class preference so that stuff, things,
    # duplicates of a stuff class and "no object" all occur. The frozen outputs are handed to both sides (the
    # PyTorch backbone is not run-to-run deterministic - MIOpen picks algorithms at first use).
    logits = logits + 4.0 * torch.randn(100, 20, device=dev, generator=torch.Generator(device=dev).manual_seed(11))
    det.slot_path = lambda _imgs: (logits, embeds, masks, fcn)
    results = det.clip_test(imgs, metas)
    fw = [fc.weight.detach().cpu().numpy() for fc in det.temporal_track_head.fcs_query]
    fb = [fc.bias.detach().cpu().numpy() for fc in det.temporal_track_head.fcs_query]
    memory, n_things = None, 0
    for t in range(T):
        want = porc.postprocess(logits[t].cpu().numpy(), masks[t].cpu().numpy(), (H, W), threshold=0.3)
        pan, cls_inds, _ = porc.panoptic_relabel(want["masks"], want["labels"])
        emb = embeds[t].cpu().numpy()[want["slot_index"]]
        if memory is None:
            det_ids, memory = np.arange(len(emb)), emb.copy()
        else:
            det_ids, memory = porc.track_assign(emb, memory, fw, fb)
        ins = want["labels"] > 10
        got = results[t]
        np.testing.assert_array_equal(got["panoptic_outputs"][0].cpu().numpy(), pan)
        assert got["panoptic_cls_inds"].tolist() == cls_inds.tolist()
        assert got["panoptic_det_obj_ids"].tolist() == det_ids[ins].tolist()
        np.testing.assert_allclose(got["panoptic_cls_prob"].numpy(), want["probs"][ins], rtol=1e-6)
        np.testing.assert_array_equal(got["fcn_outputs"][0].cpu().numpy(), fcn[t].argmax(0).cpu().numpy())
        assert got["panoptic_outputs"].shape == (1, H, W) and got["fcn_outputs"].shape == (1, H, W)
        n_things += int(ins.sum())
    assert n_things > 0, "the synthetic case must exercise the instance / tracker branch"
    np.testing.assert_allclose(det.prev_embedding.cpu().numpy(), memory, rtol=0, atol=0)
    # harness loop on top (tools/test_vpq.py:23-59 + get_unified_pan_result): uint8 maps, 3-channel encoding
    from slotvps_amd import harness
    r = harness.clip_gpu_test(det, [(imgs, metas)])
    assert r["all_names"] == [f"f{t}.png" for t in range(T)] and r["all_panos"][0].dtype == np.uint8
    enc = harness.get_unified_pan_result(r["all_ssegs"], r["all_panos"], r["all_pano_cls_inds"], r["all_pano_obj_ids"],
                                         stuff_area_limit=64, names=r["all_names"])
    assert enc["f0.png"].shape == (H, W, 3) and enc["f0.png"].dtype == np.uint8


@pytest.mark.gpu
def test_simple_test_reference_call_convention():
    dev = torch.device("cuda:0")
    det = _make_detector(dev)
    H, W = 128, 256
    g = torch.Generator(device=dev).manual_seed(5)
    frames = [torch.randn(1, 3, H, W, device=dev, generator=g) for _ in range(3)]
    outs = []
    for t in (1, 2):                                         # frame t with frame t-1 as its reference image
        meta = dict(iid=20000 + t, ori_shape=(H, W, 3), img_shape=(H, W, 3), filename=f"f{t}.png")
        outs.append(det(img=[frames[t]], img_meta=[[meta]], return_loss=False, rescale=True, ref_img=[frames[t - 1]]))
    assert det.ref_reuse_hits == 1           # frame 2's reference frame is frame 1: its level maps were kept
    for r in outs:
        assert set(r) == {"fcn_outputs", "panoptic_cls_inds", "panoptic_cls_prob", "panoptic_det_obj_ids", "panoptic_outputs"}
        ids = torch.unique(r["panoptic_outputs"])
        assert (ids[ids > 10]).numel() == len(r["panoptic_cls_inds"])          # the reference's MISMATCH check (:453-458)
        assert len(r["panoptic_det_obj_ids"]) == len(r["panoptic_cls_inds"]) == len(r["panoptic_cls_prob"])
    # same frames with the reference's recompute-everything behaviour: same result structure, tracker ids of frame 1
    det2 = _make_detector(dev)
    det2.reuse_ref_features = False
    meta = dict(iid=20001, ori_shape=(H, W, 3), img_shape=(H, W, 3), filename="f1.png")
    r2 = det2(img=[frames[1]], img_meta=[[meta]], return_loss=False, rescale=True, ref_img=[frames[0]])
    assert det2.ref_reuse_hits == 0 and set(r2) == set(outs[0])
    assert r2["panoptic_outputs"].shape == outs[0]["panoptic_outputs"].shape


@pytest.mark.gpu
def test_hip_slot_path_is_deterministic():
    """Everything after the PyTorch trunk (K4, K3, K1, K5, K2) must be bit-reproducible run to run."""
    dev = torch.device("cuda:0")
    det = _make_detector(dev)
    im = det.image_model
    g = torch.Generator(device=dev).manual_seed(2)
    sizes = [(4, 8), (8, 16), (16, 32), (32, 64)]
    feats = [torch.randn(2, 128, h, w, device=dev, generator=g) for h, w in sizes]
    from slotvps_amd import ops
    tabs = [ops.pos_embed_sine_tables(h, w, 256, dev) for h, w in sizes]
    a = im.dynamic_mask_head.forward_clip(feats, im.init_mask_query.weight, tabs)
    b = im.dynamic_mask_head.forward_clip(feats, im.init_mask_query.weight, tabs)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert all(torch.equal(x, y) for x, y in zip(a[2], b[2]))


@pytest.mark.gpu
def test_head_graph_follows_weight_and_mode_changes():
    """detector._head_clip with use_graph=True: the captured hipGraph bakes in weight-derived tensors (packed K8 weights, QR
    factors) and the kernel choices of the current modes. After an in-place weight edit, a load_state_dict and a mode switch the
    graph path must return what the eager path returns for the NEW state (the cache is keyed on parameter versions + modes)."""
    from slotvps_amd import ops
    dev = torch.device("cuda:0")
    det = _make_detector(dev)
    im = det.image_model
    g = torch.Generator(device=dev).manual_seed(3)
    sizes = [(4, 8), (8, 16), (16, 32), (32, 64)]
    feats = [torch.randn(2, 128, h, w, device=dev, generator=g) for h, w in sizes]

    def both():
        det.use_graph = True
        a = det._head_clip(feats)
        a = [a[0].clone(), a[1].clone()] + [f.clone() for f in a[2]]
        det.use_graph = False
        b = det._head_clip(feats)
        b = [b[0], b[1]] + list(b[2])
        torch.cuda.synchronize()
        return a, b

    with torch.no_grad():
        a, b = both()
        assert all(torch.equal(x, y) for x, y in zip(a, b))
        first = a[1].clone()
        lin = im.dynamic_mask_head.head_series_1[0].inst_interact.to_k          # feeds the QR factor of the statistics kernel
        lin.weight.mul_(1.5)                                                    # in-place edit: _version changes, data_ptr does not
        a, b = both()
        assert all(torch.equal(x, y) for x, y in zip(a, b)) and not torch.equal(a[1], first)
        sd = {k: v.clone() for k, v in im.dynamic_mask_head.state_dict().items()}
        sd["head_series_2.0.linear1.weight"] = sd["head_series_2.0.linear1.weight"] * 0.5     # packed K8 weight
        im.dynamic_mask_head.load_state_dict(sd)
        second = a[1].clone()
        a, b = both()
        assert all(torch.equal(x, y) for x, y in zip(a, b)) and not torch.equal(a[1], second)
        saved = im.dynamic_mask_head.stats_form
        try:
            im.dynamic_mask_head.stats_form = "stage"                           # a switch of the statistics form re-captures as well
            a, b = both()
            assert all(torch.equal(x, y) for x, y in zip(a, b))
        finally:
            im.dynamic_mask_head.stats_form = saved


@pytest.mark.gpu
def test_baseline_config0_512x1024_first_frame():
    """BASELINE config 0: the reference's own r50 config, one 512x1024 frame paired with itself (first frame of a
    video, tools/dataset/cityscapes_vps.py:262), random init. The head sees two identical frames; with random weights no
    slot reaches the 0.85 score threshold and the reference's mask_removal fails on the empty selection
    (vps_temporal_slots.py:578, :652) - same failure here. A fixed slot -> class preference makes slots
    survive; the two identical frames must then give identical head outputs and a well-formed result."""
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    det = build(REF_CFG if os.path.exists(REF_CFG) else CFG).to(dev).eval()
    H, W = 512, 1024
    img = torch.randn(1, 3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
    meta = dict(iid=30001, ori_shape=(H, W, 3), img_shape=(H, W, 3), filename="frankfurt_000000_000001.png")
    with pytest.raises(ValueError, match="no slot passes the score threshold"):
        det(img=[img], img_meta=[[meta]], return_loss=False, rescale=True, ref_img=[img])
    feats, fcn = det.trunk(img)
    assert feats.hws == [(16, 32), (32, 64), (64, 128), (128, 256)] and fcn.shape == (1, 19, H, W)
    # the default head mode (fp16x2): the tower's own rows as two fp16 planes hi + lo, conv_trans folded into K4-HL's composed weights
    assert det.image_model.dynamic_mask_head.mode == "fp16x2"
    assert feats.folded and all(f.shape == (2, 1, h * w, 128) and f.dtype == torch.float16 for f, (h, w) in zip(feats, feats.hws))
    from slotvps_amd.detector import LevelMaps
    logits, embeds, masks = det.head_path(LevelMaps.cat(feats, feats), dense=True)             # the reference's all-slot form
    assert logits.shape == (2, 100, 20) and masks.shape == (2, 100, 128, 256) and torch.isfinite(masks).all()
    assert torch.equal(logits[0], logits[1]) and torch.equal(masks[0], masks[1])      # identical frames, per-frame kernels
    with torch.no_grad():                                      # slot l prefers class l % 19, strongly
        cls = det.image_model.dynamic_mask_head.head_series_3[-1].class_logits
        table = torch.zeros(100, 20, device=dev)
        table[torch.arange(100), torch.arange(100) % 19] = 12.0
        det.image_model.fg_bn.weight.fill_(40.0)
        det._fold = None
    base_forward = det.head_path
    det.head_path = lambda f: (lambda lg, em, mk: (lg + table, em, mk))(*base_forward(f))
    r = det(img=[img], img_meta=[[meta]], return_loss=False, rescale=True, ref_img=[img])
    assert r["panoptic_outputs"].shape == (1, H, W) and r["fcn_outputs"].shape == (1, H, W)
    ids = torch.unique(r["panoptic_outputs"])
    assert (ids[ids > 10]).numel() == len(r["panoptic_cls_inds"]) == len(r["panoptic_det_obj_ids"])
    assert det.ref_reuse_hits == 2                            # ref frame == current frame: its level maps were reused


@pytest.mark.gpu
def test_trunk_bf16_autocast_option():
    """Optional bf16 autocast of the PyTorch trunk: same shapes / dtypes, level maps close to the fp32 trunk."""
    dev = torch.device("cuda:0")
    det = _make_detector(dev)
    det.fold_trans = False                                     # the reference's tensors behind conv_trans in both runs
    imgs = torch.randn(2, 3, 128, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    f32, s32 = det.trunk(imgs)
    det.trunk_bf16 = True
    f16, s16 = det.trunk(imgs)
    assert s16.dtype == torch.float32 and all(f.dtype == torch.float32 and f.is_contiguous() for f in f16)
    for a, b in zip(f32, f16):
        assert a.shape == b.shape
        d, scale = (a - b).abs(), a.abs().max().item()          # ~50 bf16 convolution layers in a row, random weights
        assert d.max().item() <= 0.12 * scale + 1e-3 and d.mean().item() <= 0.02 * scale


# ---- the two other BASELINE configurations resolve from the repository's own config files (no reference tree needed) ----
def test_swinL_and_viper_configs_build_on_any_box():
    from slotvps_amd.config import Config
    from slotvps_amd.registry import build_detector
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "viper_r50_slotvps_mi355x.py"))
    det = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    im = det.image_model
    assert det.stuff_num == 13 and det.num_classes == 24                    # vps_temporal_slots.py:68-70
    assert im.init_mask_query.weight.shape == (200, 256)                    # proposal_num = 200
    assert im.dynamic_mask_head.head_series_3[1].class_logits.weight.shape == (24, 256)
    assert cfg.clip == dict(frames=10, height=1088, width=1920)
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "swinL_fpn_slotvps_mi355x.py"))
    det = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    bb = det.image_model.backbone
    assert type(bb).__name__ == "SwinTransformer" and sum(p.numel() for p in bb.parameters()) > 190e6      # Swin-L
    stage = det.image_model.dynamic_mask_head.head_series_0[0]
    assert stage.activation is torch.nn.functional.relu and stage.temporal_query_head is None
    assert det.image_model.dynamic_mask_head.head_series_2[0].temporal_query_head.activation is torch.nn.functional.gelu
    swin_ref = "/root/reference/configs/cityscapes/swinL_fpn_slotvps.py"
    if os.path.exists(swin_ref):                                            # same model as the reference's own file
        ref = build_detector(Config.fromfile(swin_ref).model, train_cfg=None, test_cfg=cfg.test_cfg)
        assert {k: tuple(v.shape) for k, v in ref.state_dict().items()} == {k: tuple(v.shape) for k, v in det.state_dict().items()}


@pytest.mark.gpu
def test_viper_config_clip_runs_through_the_whole_detector():
    """VIPER geometry at reduced size (level sizes 17x30 ... no multiple of the tile), 200 slots, 24 classes, T = 3:
    trunk + fused retriever for more than 128 slots + selected-slot decode + post-process + tracker."""
    from slotvps_amd.config import Config
    from slotvps_amd.registry import build_detector
    dev = torch.device("cuda:0")
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "viper_r50_slotvps_mi355x.py"))
    torch.manual_seed(1)
    det = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg).to(dev).eval()
    T, H, W = 3, 544, 960
    imgs = torch.randn(T, 3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    table = torch.zeros(200, 24, device=dev)
    table[torch.arange(200), torch.arange(200) % 23] = 12.0
    with torch.no_grad():
        det.image_model.fg_bn.weight.fill_(40.0)
    base = det.head_path
    det.head_path = lambda f: (lambda lg, em, mk: (lg + table, em, mk))(*base(f))
    metas = [dict(iid=100001 + t, ori_shape=(H, W, 3), img_shape=(H, W, 3), filename=f"f{t}.png") for t in range(T)]   # vid * 100000 + fid (:220-222)
    res = det.clip_test(imgs, metas)
    assert len(res) == T
    for r in res:
        assert r["panoptic_outputs"].shape == (1, H, W) and r["fcn_outputs"].shape == (1, H, W)
        assert int(r["fcn_outputs"].max()) <= 22
        ids = torch.unique(r["panoptic_outputs"])
        assert (ids[ids > 12]).numel() == len(r["panoptic_cls_inds"]) == len(r["panoptic_det_obj_ids"])


@pytest.mark.gpu
def test_fp16_level_maps_through_the_whole_detector():
    """`other_config=dict(mode="fp16")` on the head (or head.set_mode): trunk + fp16 level maps + selected-slot decode +
    post-process + tracker run as with bf16 maps. The two storages agree on most pixels of the panoptic output (measured 75 %: a
    random-initialised head is chaotic - its free-running slot argmax agrees with the reference's on 87 - 91 % of the pixels with bf16
    maps and on 97 - 98 % with fp16 maps, tests/test_head_gpu.py - and one segment more or less renumbers the instance ids)."""
    from slotvps_amd.config import Config
    from slotvps_amd.registry import build_detector
    dev = torch.device("cuda:0")
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "r50_fpn_slotvps_mi355x.py"))
    outs = {}
    for md in ("bf16", "fp16"):
        torch.manual_seed(1)
        det = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg).to(dev).eval()
        det.image_model.dynamic_mask_head.set_mode(md)
        T, H, W = 2, 256, 512
        imgs = torch.randn(T, 3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
        ncls = det.image_model.dynamic_mask_head.num_classes
        table = torch.zeros(100, ncls, device=dev)
        table[torch.arange(100), torch.arange(100) % (ncls - 1)] = 12.0
        with torch.no_grad():
            det.image_model.fg_bn.weight.fill_(40.0)
        base = det.head_path
        det.head_path = lambda f, base=base, table=table: (lambda lg, em, mk: (lg + table, em, mk))(*base(f))
        metas = [dict(iid=100001 + t, ori_shape=(H, W, 3), img_shape=(H, W, 3), filename=f"f{t}.png") for t in range(T)]
        outs[md] = det.clip_test(imgs, metas)
        assert len(outs[md]) == T and all(r["panoptic_outputs"].shape == (1, H, W) for r in outs[md])
    same = np.mean([(a["panoptic_outputs"] == b["panoptic_outputs"]).float().mean().item() for a, b in zip(outs["bf16"], outs["fp16"])])
    print(f"\npanoptic ids equal between bf16 and fp16 level maps on {100 * same:.2f} % of the pixels")
    assert same >= 0.6


@pytest.mark.gpu
@pytest.mark.parametrize("map_dtype", ["fp16x2", "bf16", "fp16"])
def test_conv_trans_folded_into_level_fusion(map_dtype):
    """conv_trans (a linear 1x1 conv, vps_capsule.py:76-79) folded into K4's weights - the tower's last GroupNorm +
    ReLU hands K4 its own output as 16-bit pixel-major rows. Against the reference's order of operations (conv_trans in fp32 by the
    framework, then K4 on the fp32 NCHW map) the fused maps agree to the rounding of the 16-bit operands, and the semantic logits are
    the same tensor. Mode fp16x2: the rows are two fp16 planes hi + lo (22 bits), conv_trans composed into K4-HL's
    weights in float64 - the maps agree to fp32 rounding (the framework's conv_trans is itself an fp32 sum in another order)."""
    dev = torch.device("cuda:0")
    det = _make_detector(dev)
    head = det.image_model.dynamic_mask_head
    head.set_mode(map_dtype)
    with torch.no_grad():
        ct = det.image_model.conv_trans.conv
        ct.weight.normal_(0, 0.12)
        ct.bias.normal_(0, 0.3)
    imgs = torch.randn(2, 3, 128, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    det.fold_trans = True
    fa, sa = det.trunk(imgs)
    f32 = lambda f: (f[0].float() + f[1].float()) if f.dim() == 4 else f.float()      # fp16x2: the map is the sum of its planes
    la, ea, fused_a = det._head_clip(fa)
    fused_a = [f32(f).clone() for f in fused_a]
    det.fold_trans = False
    fb, sb = det.trunk(imgs)
    assert not fb.folded and fb[0].dtype == torch.float32
    if map_dtype == "fp16x2":
        # the framework's convolutions are not run-to-run identical (the two trunk runs differ by ~1e-6 of the tower's output - as much as
        # this mode resolves): the unfolded side gets x = conv_trans(y) computed by the framework from the SAME rows y the folded side read
        from slotvps_amd.detector import LevelMaps
        ct = det.image_model.conv_trans.conv
        xs = []
        for y, (h_, w_) in zip(fa, fa.hws):
            y32 = (y[0].float() + y[1].float()).transpose(1, 2).reshape(y.shape[1], 128, h_, w_)
            xs.append(torch.nn.functional.conv2d(y32, ct.weight, ct.bias).contiguous())
        fb2 = LevelMaps(xs)
        fb2.hws = fb.hws
        fb = fb2
    lb, eb, fused_b = det._head_clip(fb)
    assert fa.folded and not fb.folded and fa.hws == fb.hws
    assert fa[0].dtype == (torch.bfloat16 if map_dtype == "bf16" else torch.float16) and fb[0].dtype == torch.float32
    assert (fa[0].dim() == 4 and fa[0].shape[0] == 2) == (map_dtype == "fp16x2")
    assert (sa - sb).abs().max().item() <= 1e-4 * sb.abs().max().item()      # (the framework's convolutions are not run-to-run identical)
    ulp = {"bf16": 2.0 ** -7, "fp16": 2.0 ** -10, "fp16x2": 2.0 ** -20}[map_dtype]     # (fp16x2: fp32-class; the framework's conv_trans is an fp32 sum itself)
    for a, b in zip(fused_a, fused_b):
        b = f32(b)
        scale = b.abs().max().item()
        # different rounding points (y rounded instead of W_t y + b_t, composed weights rounded once): a few operand ulps of the map's scale
        assert (a - b).abs().max().item() <= 3 * ulp * scale, ((a - b).abs().max().item(), scale)
        assert (a - b).abs().mean().item() <= 0.3 * ulp * scale
