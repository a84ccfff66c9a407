"""Detector-level mirror of the reference: `VPS_Capsule` (image model,
mmdet/models/detectors/vps_capsule.py:26-133), `SimpleTrackHead` (simple_track_head.py:21-92) and
`VPS_Temporal_Slots` with its test-time flow (vps_temporal_slots.py:40-499), under the reference's class and
parameter names so `Config.fromfile(<reference config>)` + `build_detector` resolve and a reference
checkpoint's keys load unchanged.

What runs where:
  backbone / FPN / GroupNorm / 1x1 convs       PyTorch-ROCm (MIOpen), by design (north star)
  deformable conv of UPSNetFPN                 K7
  level fusion, k/v projection, slot attention, slot-side LayerNorms, mask decode   K4 / K3 / K1 / K5 / K2
  panoptic post-process, argmax, relabel       K6 (+ K x K host tables)
  tracker                                      two 256x256 Linear layers + one [K, P] product (PyTorch) and
                                               the reference's greedy sequential matching on the host

Besides the reference's frame-by-frame `simple_test` (current + reference image per call) there is
`clip_test`: T frames of a clip through the backbone in one batch and through the slot head as one
T-frame clip (the reference's own temporal attention is defined over the frames handed in, :284-291).
"""
import warnings

import numpy as np
import sys
import torch
import torch.nn.functional as F
from torch import nn

from . import ops, registry
from .instances import Instances
from .position_encoding import build_position_encoding
from .postprocess import PostProcessPanopticInstances
from .registry import DETECTORS, HEADS
from .slot_head import ConvModule, MultiScaleDynamicMaskHead, fold_bn_eval


class _AttrDict(dict):
    __getattr__ = dict.get


class LevelMaps(list):
    """The level maps the trunk hands to the slot head, coarse -> fine. `hws` their sizes; `folded`: the maps are the semantic tower's OWN
    output as 16-bit pixel-major rows [T, Hi*Wi, 128] (head mode fp16x2: two fp16 planes hi + lo, [2, T, Hi*Wi, 128]) and conv_trans
    (vps_capsule.py:76-79, a linear 1x1 conv) is still to be applied - the head folds it into K4's weights; otherwise [T, 128, Hi, Wi]
    fp32 maps behind conv_trans, the reference's tensors."""
    hws = None
    folded = False

    @staticmethod
    def cat(a, b):
        """Frames of `a` followed by the frames of `b` (same geometry, same form)."""
        planes = bool(getattr(a, "folded", False)) and a[0].dim() == 4         # [2, T, HW, 128]: the frame axis is dim 1
        out = LevelMaps(torch.cat([x, y], 1 if planes else 0) for x, y in zip(a, b))
        out.hws = getattr(a, "hws", None) or [tuple(x.shape[-2:]) for x in a]
        out.folded = bool(getattr(a, "folded", False))
        assert out.folded == bool(getattr(b, "folded", False))
        return out


class SlotMasks:
    """Mask logits of a clip, decoded on demand (K2). The reference materialises `pred_masks` for all L slots
    ([L, h, w] fp32 per frame, vps_temporal_slots.py:297-308) and its post-process then keeps the few slots whose class
    score passes the threshold (:685-691, computed from the [L, nc] class logits alone). Here the decode of a frame runs
    AFTER that selection, on the kept slots only, in the order the post-process wants them: K2 then writes 4 K HW bytes
    per frame instead of 4 L HW (per-frame path), and nobody gathers rows out of a [L, HW] tensor. The clip path decides on the device
    without waiting for the host, so it decodes the first `PostProcessPanopticInstances.clip_decode_cap` slots of every frame's score
    order (64; all L again if a frame keeps more). `dense()` is the reference's all-slot form (same kernel, bit-identical rows)."""

    def __init__(self, fused, embeds, fold, hw):
        self.fused, self.embeds, self.fold, self.hw = fused, embeds, fold, hw     # [T, HW, 256], [T, L, 256]
        self._dense = None

    @property
    def shape(self):
        return (self.embeds.shape[0], self.embeds.shape[1]) + tuple(self.hw)

    @property
    def is_cuda(self):
        return self.fused.is_cuda

    @property
    def device(self):
        return self.fused.device

    def _decode(self, t0, t1, embed):
        scale, shift, fs, fb = self.fold
        if self.fused.dim() == 4:                        # precision "fp16x2": the map as fp16 hi + lo planes [2, T, HW, 256]
            m = ops.mask_decode_hl(self.fused[:, t0:t1].contiguous() if (t0, t1) != (0, self.fused.shape[1]) else self.fused,
                                   embed.contiguous(), scale, shift, fs, fb)
            return m.view(m.shape[0], m.shape[1], *self.hw)
        decode = ops.mask_decode_f32 if self.fused.dtype == torch.float32 else ops.mask_decode   # exact mode: fp32 map
        m = decode(self.fused[t0:t1], embed.contiguous(), scale, shift, fs, fb)
        return m.view(m.shape[0], m.shape[1], *self.hw)

    def dense(self):
        if self._dense is None:
            self._dense = self._decode(0, self.embeds.shape[0], self.embeds)
        return self._dense

    def decode_clip(self, index):
        """index [T, K] int64 (device): slot ids per frame (padded rows repeat any valid id) -> mask logits [T, K, h, w] of exactly those
        slots, in that order, in ONE launch (the clip path of the post-process: the kept slots of every frame in score order)."""
        emb = torch.gather(self.embeds, 1, index[:, :, None].expand(-1, -1, self.embeds.shape[2]))
        return self._decode(0, self.embeds.shape[0], emb)

    def __getitem__(self, t):
        return FrameSlotMasks(self, int(t))


class FrameSlotMasks:
    """One frame of `SlotMasks`: `decode_slots(idx)` decodes the listed slots ([K, h, w], in that order)."""

    def __init__(self, clip, t):
        self.clip, self.t = clip, t

    shape = property(lambda self: self.clip.shape[1:])
    is_cuda = property(lambda self: self.clip.is_cuda)
    device = property(lambda self: self.clip.device)

    def decode_slots(self, idx):
        c = self.clip
        return c._decode(self.t, self.t + 1, c.embeds[self.t][idx][None])[0]

    def dense(self):
        return self.clip.dense()[self.t]

    def cpu(self):
        return self.dense().cpu()


def _stuff_num(num_classes):
    """vps_capsule.py:47-57 / vps_temporal_slots.py:62-72."""
    if num_classes <= 20:
        return 11            # Cityscapes(-VPS)
    if num_classes in (46, 47):
        return 34            # Mapillary Vistas
    if num_classes in (23, 24):
        return 13            # VIPER
    raise AssertionError(f"self.num_classes: {num_classes}")


@HEADS.register_module
class SimpleTrackHead(nn.Module):
    """Match scores between the current segments' slot embeddings and the tracked ones: shared fc stack on
    both, dot products, and a leading all-zero "new object" column (simple_track_head.py:58-92)."""

    def __init__(self, num_fcs_query=0, in_channels_query=0, loss_match=None, query_matched_weight=1.0):
        super().__init__()
        self.query_matched_weight = query_matched_weight
        self.num_fcs_query = num_fcs_query
        if num_fcs_query > 0:
            self.fcs_query = nn.ModuleList([nn.Linear(in_channels_query, in_channels_query) for _ in range(num_fcs_query)])
            self.relu = nn.ReLU(inplace=True)
        self.init_weights()

    def init_weights(self):
        if self.num_fcs_query > 0:
            for fc in self.fcs_query:
                nn.init.normal_(fc.weight, 0, 0.01)
                nn.init.constant_(fc.bias, 0)

    def _embed(self, x):
        for idx in range(self.num_fcs_query):
            x = self.fcs_query[idx](x)
            if idx < self.num_fcs_query - 1:
                x = self.relu(x)
        return x

    def forward(self, x_query=None, ref_x_query=None):
        x = self._embed(x_query)
        refs = ref_x_query if isinstance(ref_x_query, list) else [ref_x_query]
        scores = []
        for r in refs:
            prod = x @ self._embed(r).t()
            scores.append(torch.cat([prod.new_zeros(prod.size(0), 1), prod], dim=1))
        return scores


def greedy_track_assign(match_logprob, n_prev):
    """The sequential id assignment of simple_test (vps_temporal_slots.py:350-408) on a [K, 1 + P] matrix of
    log-probabilities (column 0 = "new object").
    Returns (det_obj_ids [K] int32, updates): `updates` is the ordered list of (prev_index, cur_index) writes the
    reference applies to its memory of tracked segments - prev_index == current length appends."""
    likelihood = match_logprob.max(axis=1)
    match_ids = match_logprob.argmax(axis=1).astype(np.int32)
    K = match_ids.shape[0]
    det = np.full(K, -1, dtype=np.int32)
    best_score = np.full(n_prev, -100.0)
    best_idx = np.full(n_prev, -1, dtype=np.int32)
    n = n_prev
    updates = []
    for idx in range(K):
        m = int(match_ids[idx])
        if m == 0:                                           # new object
            det[idx] = n
            updates.append((n, idx))
            n += 1
            continue
        obj = m - 1
        if likelihood[idx] > best_score[obj]:                # several candidates may pick one tracked object
            det[idx] = obj
            if best_idx[obj] >= 0:
                det[best_idx[obj]] = -1                      # undo the earlier, weaker match
            best_score[obj] = likelihood[idx]
            best_idx[obj] = idx
            updates.append((obj, idx))
    for idx in range(K):                                     # losers of a contested match become new objects
        if det[idx] < 0:
            det[idx] = n
            updates.append((n, idx))
            n += 1
    return det, updates


@DETECTORS.register_module
class VPS_Capsule(nn.Module):
    """Image model: backbone, neck, semantic tower, slot initialisation, 1x1 transfer conv, slot head, decode BNs."""

    def __init__(self, backbone, train_cfg, test_cfg, neck=None, panoptic=None, dynamic_mask_head=None,
                 pretrained=None, other_config=None):
        super().__init__()
        self.fp16_enabled = False
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.other_config = other_config if other_config is not None else {}
        self.cfg = train_cfg if train_cfg is not None else test_cfg
        self.has_no_obj = self.other_config.get("has_no_obj", True)
        self.num_classes = dynamic_mask_head["num_classes"]
        self.stuff_num = _stuff_num(self.num_classes)
        if self.has_no_obj:
            assert dynamic_mask_head["num_classes"] in [9, 20, 47, 24]
        self.backbone = registry.build_backbone(backbone)
        if neck is not None:
            self.neck = registry.build_neck(neck)
        if panoptic is not None:
            panoptic = dict(panoptic)
            if "feat_num_levels" in dynamic_mask_head:
                panoptic["return_feat_levels"] = dynamic_mask_head["feat_num_levels"]
            self.panopticFPN = registry.build_panoptic(panoptic)
        if dynamic_mask_head is not None:
            self.init_mask_query = nn.Embedding(self.other_config.get("proposal_num", 100), dynamic_mask_head["dh_dim"])
            nn.init.xavier_uniform_(self.init_mask_query.weight)
            out_ch = panoptic["out_channels"]
            self.conv_trans = ConvModule(out_ch, self.other_config.get("main_trans_out_dim", out_ch), 1, padding=0,
                                         activation=None)
            head_cfg = dict(dynamic_mask_head)
            head_cfg["other_config"] = other_config
            self.dynamic_mask_head = MultiScaleDynamicMaskHead(**head_cfg)
            self.query_feat_num_levels = dynamic_mask_head.get("feat_num_levels", 4)
            self.multi_scale_heads_num = dynamic_mask_head.get("per_dh_num_heads", [1, 2, 2, 2])
        pos_cfg = self.other_config.get("pos_config", None)
        self.position_embedding = build_position_encoding(_AttrDict(pos_cfg) if isinstance(pos_cfg, dict) else pos_cfg)
        self.fg_bn = nn.BatchNorm2d(1)
        self.feat_bn = nn.BatchNorm2d(dynamic_mask_head["dh_dim"])
        self.upsample = nn.Upsample(scale_factor=4, mode="bilinear", align_corners=True)
        self.init_weights(pretrained=pretrained)

    @property
    def with_neck(self):
        return hasattr(self, "neck") and self.neck is not None

    @property
    def with_panoptic(self):
        return hasattr(self, "panopticFPN") and self.panopticFPN is not None

    def init_weights(self, pretrained=None):
        if isinstance(pretrained, str):          # 'modelzoo://resnet50' in the reference config: no network here
            warnings.warn(f"pretrained={pretrained!r} ignored: no model zoo access, load weights with load_state_dict")
            pretrained = None
        self.backbone.init_weights(pretrained=pretrained)
        if self.with_neck:
            self.neck.init_weights()
        if self.with_panoptic:
            self.panopticFPN.init_weights()
        with torch.no_grad():                    # vps_capsule.py:129-133
            self.fg_bn.weight.fill_(0.1)
            self.fg_bn.bias.zero_()
            self.feat_bn.weight.fill_(1)
            self.feat_bn.bias.zero_()


@DETECTORS.register_module
class VPS_Temporal_Slots(nn.Module):
    def __init__(self, backbone, train_cfg, test_cfg, neck=None, panoptic=None, dynamic_mask_head=None, pretrained=None,
                 postprocess_panoptic=None, simple_track_head=None, other_config=None):
        super().__init__()
        self.fp16_enabled = False
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.other_config = other_config if other_config is not None else {}
        self.cfg = train_cfg if train_cfg is not None else test_cfg
        self.has_no_obj = self.other_config.get("has_no_obj", True)
        self.num_classes = dynamic_mask_head["num_classes"]
        self.stuff_num = _stuff_num(self.num_classes)
        self.image_model = VPS_Capsule(backbone, train_cfg, test_cfg, neck=neck, panoptic=panoptic,
                                       dynamic_mask_head=dynamic_mask_head, pretrained=pretrained, other_config=other_config)
        self.track_head_config = simple_track_head
        if simple_track_head is not None:
            self.temporal_track_head = SimpleTrackHead(**simple_track_head)
        if postprocess_panoptic is not None:
            pp = dict(postprocess_panoptic)
            pp.setdefault("num_stuff", self.stuff_num)
            self.postprocess_panoptic = PostProcessPanopticInstances(**pp)
        self.prev_embedding = None
        self.test_track_instances = None         # segment table of the last frame (the reference's attribute, :303-346)
        self._fold = None
        self.reuse_ref_features = True           # keep the previous frame's level maps
        self.decode_selected = True              # K2 decodes only the slots the post-process keeps (SlotMasks)
        self.clip_postprocess = True             # clip_test: post-process + tracker of all frames in lock-step (_clip_results)
        self.use_graph = False                   # replay the slot head as one hipGraph per input geometry (_head_clip)
        # conv_trans folded into K4's weights (round 4): the semantic tower's last GroupNorm + ReLU writes its output as 16-bit
        # pixel-major rows and K4 reads THOSE (256 instead of 512 B per pixel, no framework conv, no layout copy in between). Applies in
        # the 16-bit modes of the head and - as two fp16 planes hi + lo, round 6 - in mode fp16x2, with the pixel-major tower (fp32
        # trunk); otherwise (exact mode, bf16 trunk) the reference's tensors as before
        self.fold_trans = True
        self._head_cache = {}
        self._trunk_bf16 = False
        self._ref_cache = None
        self.ref_reuse_hits = 0

    # ---- pieces of simple_test ---------------------------------------------------------------------
    def extract_semantic_feats(self, x):
        n = self.image_model.panopticFPN.num_levels
        fcn_output, fcn_score, fcn_feature = self.image_model.panopticFPN(x[0:n])
        return fcn_output.float(), fcn_score, fcn_feature

    def semantic_trans_ins(self, fcn_feature):
        assert len(fcn_feature) == self.image_model.query_feat_num_levels
        return [self.image_model.conv_trans(f) for f in fcn_feature]

    @property
    def trunk_bf16(self):
        """bf16 for the PyTorch trunk: autocast around backbone / FPN / semantic tower AND bf16 matrix-core operands in the
        deformable convolutions (K7). Off by default: the reference's trunk is fp32 (fp16_enabled = False, :55)."""
        return self._trunk_bf16

    @trunk_bf16.setter
    def trunk_bf16(self, on):
        self._trunk_bf16 = bool(on)
        for m in self.modules():
            if hasattr(m, "bf16_operands"):
                m.bf16_operands = bool(on)

    def _decode_fold(self):
        """Eval-mode BatchNorms as the scalars / vectors K2 takes. Keyed on the identity and version of every tensor that
        enters the fold, so in-place edits, submodule / mmcv-style checkpoint loads and .to(device) all invalidate it."""
        bns = (self.image_model.feat_bn, self.image_model.fg_bn)
        key = tuple((t.data_ptr(), t._version, str(t.device))
                    for bn in bns for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))
        if self._fold is None or self._fold[0] != key:
            scale, shift = fold_bn_eval(bns[0])
            fs, fb = fold_bn_eval(bns[1])
            self._fold = (key, (scale, shift, float(fs.item()), float(fb.item())))
        return self._fold[1]

    @torch.no_grad()
    def trunk(self, imgs):
        """PyTorch part: imgs [T, 3, H, W] -> (per level [T, 128, Hi, Wi] fp32 maps coarse -> fine, the slot head's
        input; semantic logits [T, nc_sem, H, W])."""
        im = self.image_model
        # trunk_bf16: PyTorch autocast (bf16 convolutions / GEMMs, fp32 normalisation layers) around the PyTorch part;
        # off by default: the reference's trunk is fp32
        head = im.dynamic_mask_head
        pf = im.panopticFPN
        want16 = None
        if self.fold_trans and head.precision == "fp16x2" and hasattr(pf, "emit_pm16"):
            want16 = "hl"                                    # two fp16 planes hi + lo: K4-HL's operand tile is the tower's rows (round 6)
        elif self.fold_trans and head.precision == "bf16" and hasattr(pf, "emit_pm16"):
            want16 = torch.float16 if head._map_form() == "fp16" else torch.bfloat16
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=self.trunk_bf16):
            x = im.backbone(imgs)
            if im.with_neck:
                x = im.neck(x)
            if hasattr(pf, "emit_pm16"):
                pf.emit_pm16 = want16
            fcn_output, _, fcn_feature = self.extract_semantic_feats(x)
            pm16 = pf.last_pm16 if want16 is not None else None
            if pm16 is None:
                feats = LevelMaps(f.float().contiguous() for f in self.semantic_trans_ins(fcn_feature))
            else:
                feats = LevelMaps(pm16)
                feats.folded = True
            feats.hws = [tuple(f.shape[-2:]) for f in fcn_feature]
        return feats, fcn_output.float()

    def _head_clip(self, feats):
        """The slot head on one clip's level maps; with `use_graph` the whole head (about 200 launches) is captured once per
        input geometry into a hipGraph and replayed on static buffers (the returned tensors are then valid until the next call)."""
        im = self.image_model
        head = im.dynamic_mask_head
        # The captured graph bakes in pointers to weight-DERIVED tensors (packed K8 weights, QR factors, position tables) and the
        # kernel choices of the current modes: key it on the identity + version of every head parameter and on the mode switches,
        # so load_state_dict / in-place edits / set_mode / set_slot_gemm / the statistics form re-capture.
        hws = getattr(feats, "hws", None) or [tuple(f.shape[-2:]) for f in feats]
        folded = bool(getattr(feats, "folded", False))
        pre = (im.conv_trans.conv.weight, im.conv_trans.conv.bias) if folded else None
        wkey = tuple((p_.data_ptr(), p_._version) for p_ in head.parameters()) + (im.init_mask_query.weight.data_ptr(), im.init_mask_query.weight._version) \
            + (tuple((p_.data_ptr(), p_._version) for p_ in pre if p_ is not None) if folded else ())
        modes = (head.mode, head.stats_form, head.map_encoding) + tuple(sorted({(type(m).__name__, getattr(m, "use_slot_gemm", None), getattr(m, "query_side", None), getattr(m, "range_check", None))
                                                                                 for m in head.modules() if hasattr(m, "precision")}))
        key = tuple(tuple(f.shape) for f in feats) + (str(feats[0].device), str(feats[0].dtype), folded, tuple(hws), hash(wkey), modes)
        ent = self._head_cache.get(key)
        if ent is None:
            D = im.init_mask_query.weight.shape[1]
            ent = {"pos": [ops.pos_embed_sine_tables(h_, w_, D, feats[0].device) for (h_, w_) in hws], "graph": None}
            self._head_cache = {key: ent}                    # one geometry at a time
        run = lambda fs: im.dynamic_mask_head.forward_clip(fs, im.init_mask_query.weight, ent["pos"], hws=hws, pre_linear=pre)
        if not self.use_graph:
            return run(feats)
        if ent["graph"] is None:
            dev = feats[0].device
            ent["static"] = [f.clone() for f in feats]
            # a launch the runtime refuses during capture is simply missing from the graph: validate the FIRST replay against the
            # eager step on the same inputs (every kernel is deterministic), as SlotClipRunner.run does: one repeated capture after a
            # mismatch (reported on stderr), a second mismatch raises
            for attempt in range(2):
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    for _ in range(2):                       # lazy initialisations, weight-derived constants, allocator
                        eager = run(ent["static"])
                    eager = [eager[0].clone(), eager[1].clone()] + [f.clone() for f in eager[2]]
                torch.cuda.current_stream(dev).wait_stream(side)
                torch.cuda.synchronize(dev)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    ent["out"] = run(ent["static"])
                g.replay()
                torch.cuda.synchronize(dev)
                got = [ent["out"][0], ent["out"][1]] + list(ent["out"][2])
                bad = [(i, float((a_.float() - b_.float()).abs().max())) for i, (a_, b_) in enumerate(zip(got, eager)) if not torch.equal(a_, b_)]
                if not bad:
                    ent["graph"] = g
                    break
                msg = (f"hipGraph replay of the slot head differs from the eager step (outputs, max abs: {bad}; attempt {attempt + 1}): "
                       "a launch was refused or reordered during capture, or a kernel is not deterministic")
                print("[slotvps_amd.detector] " + msg, file=sys.stderr, flush=True)
                del g
                if attempt == 1:
                    raise RuntimeError(msg)
        for dst, src in zip(ent["static"], feats):
            dst.copy_(src)
        ent["graph"].replay()
        return ent["out"]

    @torch.no_grad()
    def head_path(self, feats, dense=None):
        """HIP part: level maps -> (class logits [T, L, nc] of the last stage, slot embeddings [T, L, 256], mask logits).
        Mask logits: `SlotMasks` (decoded per frame for the slots the post-process keeps) or, with dense=True /
        self.decode_selected = False, the reference's [T, L, H/4, W/4] tensor of all slots."""
        logits, embeds, fused = self._head_clip(feats)
        hw = feats.hws[-1] if getattr(feats, "hws", None) else tuple(feats[-1].shape[-2:])
        masks = SlotMasks(fused[-1], embeds[-1].contiguous(), self._decode_fold(), tuple(hw))
        if dense is None:
            dense = not self.decode_selected
        return logits[-1], embeds[-1], masks.dense() if dense else masks

    @torch.no_grad()
    def slot_path(self, imgs):
        """imgs [T, 3, H, W] -> (class logits, slot embeddings, mask logits, semantic logits [T, nc_sem, H, W])."""
        feats, fcn_output = self.trunk(imgs)
        return self.head_path(feats) + (fcn_output,)

    def _track(self, embedding, first):
        """Tracker step over ALL surviving segments of the frame (stuff and things, :345-409)."""
        K = embedding.shape[0]
        if first or self.prev_embedding is None:
            self.prev_embedding = embedding.clone()
            return np.arange(K, dtype=np.int64)
        assert K > 0 and self.prev_embedding.shape[0] > 0
        score = self.temporal_track_head(embedding, self.prev_embedding)[0]
        logprob = F.log_softmax(score, dim=1).cpu().numpy()
        det, updates = greedy_track_assign(logprob, self.prev_embedding.shape[0])
        n_new = max([p for p, _ in updates] + [self.prev_embedding.shape[0] - 1]) + 1
        mem = torch.cat([self.prev_embedding, embedding.new_zeros(n_new - self.prev_embedding.shape[0], embedding.shape[1])])
        for p, c in updates:                      # in order: a later, stronger match overwrites an earlier one
            mem[p] = embedding[c]
        self.prev_embedding = mem
        return det.astype(np.int64)

    def _frame_result(self, pred_logits, pred_masks, embedding, fcn_output, ori_shape, first):
        H, W = int(ori_shape[0]), int(ori_shape[1])
        res = self.postprocess_panoptic.forward_tensors(pred_logits, pred_masks, (H, W))
        det = self._track(embedding[res.slot_index], first)
        labels = res.labels.cpu()
        self.test_track_instances = Instances((H, W), slot_index=res.slot_index, labels=res.labels,
                                              output_embedding=embedding[res.slot_index],
                                              obj_ids=torch.from_numpy(det))
        ins = labels > self.stuff_num - 1
        pan, cls_inds, cls_prob = self.postprocess_panoptic.panoptic_ids(res, self.stuff_num)
        if fcn_output.shape[-2] != H or fcn_output.shape[-1] != W:
            fcn_output = F.interpolate(fcn_output, size=(H, W), mode="bilinear", align_corners=False)
        fcn = fcn_output.argmax(dim=1)            # argmax of the softmax (:447)
        return {"fcn_outputs": fcn[:, :H, :W], "panoptic_cls_inds": cls_inds, "panoptic_cls_prob": cls_prob.cpu(),
                "panoptic_det_obj_ids": torch.from_numpy(det)[ins], "panoptic_outputs": pan[None, :H, :W].long()}

    # ---- reference entry points ------------------------------------------------------------------------
    @torch.no_grad()
    def simple_test(self, img, img_meta, rescale=False, ref_img=None):
        """One call per frame like the reference (:208-469): `img` [1, 3, H, W], `ref_img` a list holding the
        reference frame; the slot head sees the clip [ref, cur] and the current frame's outputs are used."""
        if ref_img is not None and isinstance(ref_img, (list, tuple)):
            ref_img = ref_img[0]
        meta = img_meta[0] if isinstance(img_meta, (list, tuple)) else img_meta
        iid = meta["iid"]
        div_mod = 100000 if self.num_classes in (23, 24) else 10000
        self.vid, self.fid = iid // div_mod, iid % div_mod
        assert self.other_config.get("test_forward_ref_img", False) is True and ref_img is not None
        if self.num_classes in (19, 20):
            assert meta["ori_shape"][0] == img.shape[2] and meta["ori_shape"][1] == img.shape[3]
        if not self.reuse_ref_features:
            logits, embeds, masks, fcn = self.slot_path(torch.cat([ref_img, img], 0))
            return self._frame_result(logits[1], masks[1], embeds[1], fcn[1:2], meta["ori_shape"], self.fid == 1)
        # The reference runs backbone + neck + semantic tower on the reference frame again at every step (:245-260),
        # although it is the frame it processed in the previous call (or the current frame itself for the first frame
        # of a video, tools/dataset/cityscapes_vps.py:262). Its level maps are kept instead; identity is checked on
        # the pixels (a copy: loaders may recycle their buffers).
        cur_feats, fcn = self.trunk(img)
        if ref_img.shape == img.shape and torch.equal(ref_img, img):
            ref_feats = cur_feats
            self.ref_reuse_hits += 1
        elif self._ref_cache is not None and self._ref_cache[0].shape == ref_img.shape and torch.equal(self._ref_cache[0], ref_img):
            ref_feats = self._ref_cache[1]
            self.ref_reuse_hits += 1
        else:
            ref_feats, _ = self.trunk(ref_img)
        self._ref_cache = (img.clone(), cur_feats)
        logits, embeds, masks = self.head_path(LevelMaps.cat(ref_feats, cur_feats))
        return self._frame_result(logits[1], masks[1], embeds[1], fcn, meta["ori_shape"], self.fid == 1)

    @torch.no_grad()
    def clip_test(self, imgs, img_metas):
        """T frames of one video at once: imgs [T, 3, H, W], img_metas list of T dicts (iid, ori_shape).
        Returns one result dict per frame; the tracker runs over the frames in order."""
        logits, embeds, masks, fcn = self.slot_path(imgs)
        div_mod = 100000 if self.num_classes in (23, 24) else 10000
        firsts = [meta["iid"] % div_mod == 1 for meta in img_metas]
        shapes = {(int(m["ori_shape"][0]), int(m["ori_shape"][1])) for m in img_metas}
        if self.clip_postprocess and len(shapes) == 1:
            return self._clip_results(logits, masks, embeds, fcn, shapes.pop(), firsts)
        out = []
        for t, meta in enumerate(img_metas):
            out.append(self._frame_result(logits[t], masks[t], embeds[t], fcn[t:t + 1], meta["ori_shape"], firsts[t]))
        return out

    def _clip_results(self, logits, masks, embeds, fcn, size, firsts):
        """_frame_result for all frames of a clip in lock-step (same decisions, same results): the post-process phases of all frames run
        together (postprocess.forward_clip / panoptic_ids_clip: one device -> host copy per phase), the tracker's match scores come from
        ONE copy of the embedded segment vectors of all frames (the fc stack is per row, so a memory row's embedding is the embedding of
        the segment it was copied from) and its sequential id assignment runs on the host."""
        H, W = size
        T = logits.shape[0]
        pp = self.postprocess_panoptic
        # ---- tracker (:345-409): embedded vectors of the kept slots of every frame (+ of the memory a previous clip left), enqueued by the
        # post-process BEHIND its own kernels so that they come back with its one wait
        have_mem = self.prev_embedding is not None and not firsts[0]
        o0 = self.prev_embedding.shape[0] if have_mem else 0
        box = {}

        def side(index_d):                       # index_d [T, Kmax]: each frame's kept slots in score order (padded rows: slot 0)
            g = torch.gather(embeds, 1, index_d[:, :, None].expand(-1, -1, embeds.shape[2])).reshape(-1, embeds.shape[2])
            box["raw"] = torch.cat([self.prev_embedding, g]) if have_mem else g
            box["kmax"] = index_d.shape[1]
            return self.temporal_track_head._embed(box["raw"])
        results = pp.forward_clip(logits, masks, (H, W), stuff_num=self.stuff_num, side=side)
        emb_h, raw, Kmax = results.side_host, box["raw"], box["kmax"]
        assert Kmax == results[0]._row_stride
        rows = [o0 + t * Kmax + np.asarray(r._sorted_pos, dtype=np.int64) for t, r in enumerate(results)]   # rows of `raw` = segments
        mem_src = list(range(o0))                                    # memory row -> row of `raw` it currently holds
        dets = []
        for t in range(T):
            K = len(rows[t])
            if firsts[t] or (t == 0 and not have_mem):
                mem_src = rows[t].tolist()
                det = np.arange(K, dtype=np.int64)
            else:
                assert K > 0 and len(mem_src) > 0
                prod = emb_h[rows[t]] @ emb_h[mem_src].T
                score = torch.from_numpy(np.concatenate([np.zeros((K, 1), dtype=prod.dtype), prod], axis=1))
                logprob = F.log_softmax(score, dim=1).numpy()
                det, updates = greedy_track_assign(logprob, len(mem_src))
                for p_, c_ in updates:                               # in order: a later, stronger match overwrites an earlier one
                    if p_ == len(mem_src):
                        mem_src.append(int(rows[t][c_]))
                    else:
                        mem_src[p_] = int(rows[t][c_])
                det = det.astype(np.int64)
            dets.append(det)
        self.prev_embedding = raw[torch.as_tensor(mem_src, dtype=torch.long, device=raw.device)].clone()
        last = results[-1]
        self.test_track_instances = Instances((H, W), slot_index=last.slot_index, labels=torch.from_numpy(last.labels_host).to(raw.device),
                                              output_embedding=embeds[T - 1][last.slot_index], obj_ids=torch.from_numpy(dets[-1]))
        pans = pp.panoptic_ids_clip(results, self.stuff_num)
        if fcn.shape[-2] != H or fcn.shape[-1] != W:
            fcn = F.interpolate(fcn, size=(H, W), mode="bilinear", align_corners=False)
        fcn_ids = fcn.argmax(dim=1)                                  # argmax of the softmax (:447)
        out = []
        for t, (r, (pan, cls_inds, cls_prob)) in enumerate(zip(results, pans)):
            ins = r.labels_host > self.stuff_num - 1
            out.append({"fcn_outputs": fcn_ids[t:t + 1, :H, :W], "panoptic_cls_inds": cls_inds, "panoptic_cls_prob": cls_prob,
                        "panoptic_det_obj_ids": torch.from_numpy(dets[t])[torch.from_numpy(ins)], "panoptic_outputs": pan[None, :H, :W].long()})
        return out

    def forward_test(self, imgs, img_metas, rescale=False, ref_img=None):
        for var, name in [(imgs, "imgs"), (img_metas, "img_metas")]:
            if not isinstance(var, list):
                raise TypeError(f"{name} must be a list, but got {type(var)}")
        if len(imgs) != len(img_metas):
            raise ValueError(f"num of augmentations ({len(imgs)}) != num of image meta ({len(img_metas)})")
        assert imgs[0].size(0) == 1
        if len(imgs) != 1:
            raise NotImplementedError("test-time augmentation is not part of the released code path")
        return self.simple_test(imgs[0], img_metas[0], rescale, ref_img)

    def forward(self, img, img_meta, return_loss=True, rescale=None, ref_img=None):
        if return_loss:
            raise AssertionError("NOT RELEASED TRAIN CODE YET !!!!!!")          # vps_temporal_slots.py:497
        return self.forward_test(img, img_meta, rescale, ref_img)
