"""Clip driver of the hot path: T frames of FPN feature maps -> slot head (K1 per stage) -> mask
decode (K2), with the launch-bound slot-side operators captured in a hipGraph.

The reference's detector runs [ref_frame, cur_frame] pairs (T = 2, vps_temporal_slots.py:283-291) and
decodes the current frame only (:297-299); the head itself is generic in T (dynamic_mask_head.py:143-164).
This driver feeds a whole T-frame clip through the head at once and decodes every frame.
"""
import os
import sys
import torch
from torch import nn

from . import ops, synth
from .slot_head import MultiScaleDynamicMaskHead, fold_bn_eval


def build_r50_head(cfg=None):
    cfg = cfg or synth.R50_HEAD_CFG
    return MultiScaleDynamicMaskHead(
        dh_dim=cfg["dh_dim"], num_classes=cfg["num_classes"], dim_feedforward=cfg["dim_feedforward"],
        nhead=cfg["nhead"], dropout=0.0, activation=cfg["activation"], dh_num_heads=sum(cfg["per_dh_num_heads"]),
        per_dh_num_heads=list(cfg["per_dh_num_heads"]), feat_num_levels=cfg["feat_num_levels"],
        merge_operation="concat", trans_in_dim=cfg["trans_in_dim"], num_cls=cfg["num_cls"], num_reg=cfg["num_reg"],
        temporal_query_attention_config=dict(d_model=cfg["dh_dim"], dim_feedforward=cfg["temporal_dim_feedforward"],
                                             dropout=0.0, activation=cfg["temporal_activation"],
                                             softmax_dim="slots", drop_path=0.),
        apply_temporal_query_atten_stages=list(cfg["apply_temporal_query_atten_stages"]))


class SlotClipRunner:
    """Owns a head replica, the slot initialisation, the decode BatchNorms and (optionally) a captured
    hipGraph of one clip step. All tensors live on `device`."""

    def __init__(self, device, T, H, W, L=100, param_seed=0, cfg=None, split_p=True, use_graph=True, n_slots=1,
                 clips_per_launch=1, decode_logits=True, input_form="nchw_f32"):
        if torch.device(device).type != "cuda":
            raise RuntimeError("SlotClipRunner runs on the GPU only; there is no CPU fallback")
        self.device = torch.device(device)
        # decode_logits=False: K2 in argmax-only mode - the step returns the per-pixel slot assignment (uint8) and the class logits,
        # not the [T, L, HW] fp32 mask logits (a clip driver that post-processes decodes the kept slots only, detector.clip_test)
        self.decode_logits = decode_logits
        self.clip_frames, self.clips_per_launch = T, clips_per_launch
        T = T * clips_per_launch                  # clips stacked along the frame axis: one launch covers them all
        self.T, self.H, self.W, self.L = T, H, W, L
        self.cfg = cfg or synth.R50_HEAD_CFG
        self.sizes = synth.level_sizes(H, W)
        params = synth.make_params(synth.head_shapes(self.cfg), param_seed)
        self.head = build_r50_head(self.cfg)
        sd = self.head.state_dict()
        self.head.load_state_dict({k: torch.from_numpy(v).reshape(sd[k].shape) for k, v in params.items()}, strict=True)
        self.head.to(self.device).eval()
        for m in self.head.modules():
            if hasattr(m, "split_p"):
                m.split_p = split_p
        # the slot initialisation is a model parameter in the detector (VPS_Capsule.init_mask_query.weight): as a Parameter here too, so
        # that the head caches its broadcast over the frames (forward_clip) instead of launching a copy kernel inside every step
        self.init_slots = nn.Parameter(torch.from_numpy(synth.make_slots(param_seed + 1, L)).to(self.device), requires_grad=False)
        # decode BatchNorms at the reference's initial values (vps_capsule.py:129-133): fg_bn weight 0.1
        self.feat_bn = nn.BatchNorm2d(self.cfg["dh_dim"]).to(self.device).eval()
        self.fg_bn = nn.BatchNorm2d(1).to(self.device).eval()
        with torch.no_grad():
            self.fg_bn.weight.fill_(0.1)
            self.fg_bn.bias.zero_()
        self.refold()
        self.pos_tabs = [ops.pos_embed_sine_tables(h, w, self.cfg["dh_dim"], self.device) for (h, w) in self.sizes]
        # n_slots independent input buffer sets, each with its own captured graph: the producer of the
        # FPN maps writes clip i straight into slot i % n_slots, so no staging copy sits in the step
        self.n_slots = n_slots
        # input_form: "nchw_f32" - the reference's tensors, [T, 128, Hi, Wi] fp32 behind conv_trans (vps_capsule.py:76-79); "tower16" -
        # the semantic tower's OWN output as 16-bit pixel-major rows [T, Hi*Wi, 128] (what its last GroupNorm + ReLU kernel writes,
        # csrc/gn_relu.hip) with conv_trans, a linear 1x1 conv, folded into K4's weights: K4 reads 256 instead of 512 B per pixel
        if input_form not in ("nchw_f32", "tower16"):
            raise ValueError(f"input_form {input_form!r}")
        self.input_form = input_form
        self.pre_linear = None
        if input_form == "tower16":
            g = torch.Generator().manual_seed(param_seed + 2)
            c = self.cfg["trans_in_dim"] - self.cfg["dh_dim"]
            self.pre_linear = (nn.Parameter((torch.randn(c, c, 1, 1, generator=g) * (2.0 / c) ** 0.5).to(self.device), requires_grad=False),
                               nn.Parameter((torch.randn(c, generator=g) * 0.1 - 0.45).to(self.device), requires_grad=False))
        self._alloc_inputs()
        self.use_graph = use_graph
        # graph mode, fp16x2: K4 / K3 as a parallel branch of the slot chain (see _step). OFF by default: built, bitwise the single-stream result,
        # the branches do overlap in the kernel trace (rocprofv3: 4 queues), and the step measures the SAME (61.9 - 62.4 against 61.75 ms, same
        # box, profiles/r06/README.md) - the pixel-side kernels hold every CU for their whole launch (one workgroup per CU, most of its LDS and
        # registers), so a small slot-side kernel of the other branch waits for a CU instead of filling idle ones
        self.overlap_pixel_side = os.environ.get("SVPS_OVERLAP_PIXEL_SIDE", "0") == "1"
        self.graphs = [None] * n_slots
        self.validation_reports = []                      # graph validations that failed and were repeated (run())
        self.outs = [None] * n_slots
        self.out = None

    def _planes_in(self):
        """input_form "tower16" in head mode fp16x2: the tower's rows as two fp16 planes hi + lo [2, T, HW, 128] (gn_relu.hip, round 6)."""
        return self.input_form == "tower16" and self.head.precision == "fp16x2"

    def _input_dtype(self):
        if self.input_form == "nchw_f32":
            return torch.float32
        if self._planes_in():
            return torch.float16
        return torch.float16 if self.head._map_form() == "fp16" else torch.bfloat16

    def _alloc_inputs(self):
        dt = self._input_dtype()
        shape = (lambda h, w: (self.T, 128, h, w)) if self.input_form == "nchw_f32" else (lambda h, w: (self.T, h * w, 128))
        if self._planes_in():
            shape = lambda h, w: (2, self.T, h * w, 128)
        self._alloc_key = (dt, self._planes_in())
        self.slots_feats = [[torch.zeros(shape(h, w), dtype=dt, device=self.device) for (h, w) in self.sizes] for _ in range(self.n_slots)]
        self.static_feats = self.slots_feats[0]
        self.graphs = [None] * self.n_slots

    def refold(self):
        """Fold the eval BatchNorms into the scalars / vectors K2 takes (host floats: no sync per step)."""
        with torch.no_grad():
            self.bn_scale, self.bn_shift = fold_bn_eval(self.feat_bn)
            fs, fb = fold_bn_eval(self.fg_bn)
            self.fg_scale, self.fg_shift = float(fs.item()), float(fb.item())

    def _step(self, slot=0):
        # graph mode: the pixel side (K4, K3) as a parallel branch of the slot chain (MultiScaleDynamicMaskHead.forward_clip, pixel_stream)
        pix = None
        if self.use_graph and self.overlap_pixel_side and getattr(self.head, "precision", "") == "fp16x2":
            if getattr(self, "_pix_stream", None) is None:
                self._pix_stream = torch.cuda.Stream(device=self.device)
            pix = self._pix_stream
        logits, embeds, fused = self.head.forward_clip(self.slots_feats[slot], self.init_slots, self.pos_tabs, hws=self.sizes,
                                                       clip_frames=self.clip_frames, pre_linear=self.pre_linear, pixel_stream=pix)
        if fused[-1].dim() == 4:                             # precision "fp16x2": the map as fp16 hi + lo planes; the logits are always written
            masks, amax = ops.mask_decode_hl(fused[-1], embeds[-1].contiguous(), self.bn_scale, self.bn_shift, self.fg_scale, self.fg_shift,
                                             want_argmax=True)
            if not self.decode_logits:
                masks = None
        elif fused[-1].dtype == torch.float32:               # exact mode (head.set_mode("fp32")): fp32 map, fp32 decode kernel
            masks = ops.mask_decode_f32(fused[-1], embeds[-1].contiguous(), self.bn_scale, self.bn_shift, self.fg_scale, self.fg_shift)
            amax = masks.argmax(dim=1).to(torch.uint8)
            if not self.decode_logits:
                masks = None
        else:
            masks, amax = ops.mask_decode(fused[-1], embeds[-1].contiguous(), self.bn_scale, self.bn_shift,
                                          self.fg_scale, self.fg_shift, want_argmax=True, want_logits=self.decode_logits)
        out = dict(class_logits=logits, slot_embeds=embeds, slot_argmax=amax)
        if masks is not None:
            out["mask_logits"] = masks
        return out

    def load_clip(self, feats, slot=0):
        if getattr(self, "_alloc_key", None) != (self._input_dtype(), self._planes_in()):   # the head's mode was switched: the rows follow it
            self._alloc_inputs()
        for dst, src in zip(self.slots_feats[slot], feats):
            dst.copy_(src)

    @torch.no_grad()
    def run(self, slot=0):
        """One clip step on the clip currently in input slot `slot`."""
        if not self.use_graph:
            self.out = self._step(slot)
            return self.out
        if self.graphs[slot] is None:
            # A launch the runtime refuses during capture does not fail there, it is simply missing from the graph: the FIRST replay
            # is validated against the eager step on the same inputs (every kernel is deterministic). A mismatch is reported with
            # what it looked like (_mismatch_report: is the replay reproducible, is the eager step, where they differ) and the capture
            # is repeated ONCE - a second mismatch raises. `validation_reports` keeps the reports (bench.py prints them in its line).
            for attempt in range(2):
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    for _ in range(2):                      # warm-up: lazy inits, kernel attributes, allocator
                        eager = self._step(slot)
                    eager = {k: v.clone() for k, v in eager.items()}
                torch.cuda.current_stream(self.device).wait_stream(side)
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self.outs[slot] = self._step(slot)
                g.replay()
                torch.cuda.synchronize(self.device)
                bad = [(k, float((self.outs[slot][k].float() - ref.float()).abs().max())) for k, ref in eager.items()
                       if not torch.equal(self.outs[slot][k], ref)]
                if not bad:
                    self.graphs[slot] = g
                    break
                report = (f"hipGraph replay of the clip step differs from the eager step in {bad} (attempt {attempt + 1}): a launch was refused "
                          "or reordered during capture, or a kernel is not deterministic. " + self._mismatch_report(slot, eager, g))
                self.validation_reports.append(report)
                print("[slotvps_amd.clip] " + report, file=sys.stderr, flush=True)
                del g
                if attempt == 1:
                    raise RuntimeError(report)
        self.graphs[slot].replay()
        self.out = self.outs[slot]
        return self.out

    def _mismatch_report(self, slot, eager, g):
        """What a failed graph validation looked like: is the replay reproducible, is the eager step, where do they differ."""
        first = {k: v.clone() for k, v in self.outs[slot].items()}
        g.replay()
        torch.cuda.synchronize(self.device)
        again = {k: v.clone() for k, v in self.outs[slot].items()}
        e2 = {k: v.clone() for k, v in self._step(slot).items()}
        torch.cuda.synchronize(self.device)
        parts = []
        for k in eager:
            d = (first[k].float() - eager[k].float()).abs()
            where = ""
            if d.numel() and float(d.max()) > 0:
                idx = torch.nonzero(d.reshape(d.shape[0], -1).amax(dim=1) > 0).flatten().tolist()
                where = f", differing leading indices {idx[:8]}{'...' if len(idx) > 8 else ''} of {d.shape[0]}, {int((d > 0).sum())} of {d.numel()} elements"
            parts.append(f"{k}: replay1 == eager1 {torch.equal(first[k], eager[k])}, replay2 == replay1 {torch.equal(again[k], first[k])}, "
                         f"replay2 == eager1 {torch.equal(again[k], eager[k])}, eager2 == eager1 {torch.equal(e2[k], eager[k])}{where}")
        return " | ".join(parts)

    def random_clip(self, seed):
        g = torch.Generator(device=self.device).manual_seed(seed)
        if self._planes_in():
            return [ops.split_hl(torch.randn((self.T, h * w, 128), generator=g, device=self.device).relu_()) for (h, w) in self.sizes]
        if self.input_form == "tower16":                                   # behind a ReLU: non-negative rows
            return [torch.randn((self.T, h * w, 128), generator=g, device=self.device).relu_().to(self._input_dtype()) for (h, w) in self.sizes]
        return [torch.randn((self.T, 128, h, w), generator=g, device=self.device) for (h, w) in self.sizes]

    # ---- algorithmic accounting ------------------------------------------------
    def k1_launch_shapes(self):
        """[(HW, launches per step)] of the retriever: one launch per stage covering all T frames."""
        return [(h * w, n) for (h, w), n in zip(self.sizes, self.cfg["per_dh_num_heads"])]

    @property
    def retriever_form(self):
        for m in self.head.modules():
            if hasattr(m, "retriever"):
                return m.retriever
        return "kv"

    def k1_algorithmic_bytes_per_step(self):
        """kv form (K1): k and v read exactly once + q in + out: 2*HW*D*2 + L*D*(2+4) per frame-stage."""
        D = self.cfg["dh_dim"]
        return sum(n * self.T * (2 * hw * D * 2 + self.L * D * (2 + 4)) for hw, n in self.k1_launch_shapes())

    def algorithmic_per_step(self):
        """Per library kernel: algorithmic HBM bytes and matrix flops of one step (all frames of the launch), the figures the
        roofline fractions are computed from. Per pixel and stage (D = 256, L slots):
          retr_stats (K3')  bytes: 512 (map) in + 16 (the aux row: both statistics) out
                            flops: the two triangular products |R x|^2, 36 of 64 blocks each: 2 * (36/64) * 2 * D^2
          retr_attn  (K1')  bytes: 512 (map) + 16 (aux row) in, + per frame-stage L*D*(2+2) (Q'' hi / lo) + tables (H+W)*128*4 + L*260*4 out
                            flops: 4 * L * D (logits + attn.v; the hi / lo splits are not algorithmic)
          kv_project (K3)   bytes: 512 in + 1024 out; flops 4 * D^2
          slot_attn  (K1)   bytes: 1024 in (+ q, out per frame-stage); flops 4 * L * D
          level_fuse (K4)   bytes: 512 (fp32 NCHW map; 256 for the tower's 16-bit rows, input_form "tower16") in + 512 out (+ 128 of the 4x
                            smaller previous level); flops 2 * 384 * D
          mask_decode (K2)  bytes: 512 in + 4 L + 1 out (finest level only); flops 2 * L * D"""
        D, L, T = self.cfg["dh_dim"], self.L, self.T
        px = [h * w for (h, w) in self.sizes]
        ps = sum(n * hw for hw, n in self.k1_launch_shapes())              # pixel-stages per frame
        stages = sum(n for _, n in self.k1_launch_shapes())
        if getattr(self.head, "precision", "bf16") == "fp16x2":
            # reference precision on the matrix cores: the maps are 1 KiB per pixel (fp16 hi + lo planes). Algorithmic flops = one product
            # per multiply; executed = the MFMAs issued (three per product in K4 / K3-HL / K2; K1'-HL32, round 6: 4 x 48 producer + 4 x 52
            # consumer per 32-pixel tile - 4 x 32 + 4 x 26 per SIXTEEN pixels before; more than 128 slots: two passes + 8 x 48 of the logit statistics).
            # K3-HL reads both planes once per stage; K4-HL: K = 128 products only (composed weights).
            tabs = sum(n * (h + w) * 128 * 4 for (h, w), n in zip(self.sizes, self.cfg["per_dh_num_heads"]))
            return {
                # K4-HL: per pixel 512 B of the incoming fp32 map in, the two planes (1 KiB) out, the 4x smaller level below (its fp32 G, 1 KiB
                # per coarse pixel) in - the function's own I/O in this storage; the G^(m) outputs a level writes for its finer levels (fp32,
                # n - 1 - i launches of the same K = 128 kernel, csrc/level_fuse_hl.hip) are executed work, not algorithmic bytes
                "level_fuse": {"bytes": T * sum(hw * (512 + 1024 + (256 if i else 0)) for i, hw in enumerate(px)),
                               "flops": T * sum(px) * 2 * 384 * D,
                               "executed_flops": T * sum(hw * (len(px) - i) for i, hw in enumerate(px)) * 3 * 2 * 128 * D},
                "mask_decode": {"bytes": T * px[-1] * (1024 + 4 * L + 1), "flops": T * px[-1] * 2 * L * D,
                                "executed_flops": T * px[-1] * ((4 if L <= 128 else 8) * 48 * 32768 // 32)},
                "retr_stats": {"bytes": T * ps * (1024 + 16), "flops": T * ps * int(2 * 36 / 64 * 2 * D * D),
                               "executed_flops": T * ps * 3 * int(2 * 36 / 64 * 2 * D * D)},
                "retr_attn": {"bytes": T * (ps * (1024 + 16) + stages * (L * D * 4 + L * 260 * 4) + tabs), "flops": T * ps * 4 * L * D,
                              "executed_flops": T * ps * ((400 if L <= 128 else 2 * 400 + 8 * 48) * 32768 // 32)},
            }
        out = {
            "level_fuse": {"bytes": T * sum(hw * ((768 if self.input_form == "tower16" else 1024) + (128 if i else 0)) for i, hw in enumerate(px)),
                           "flops": T * sum(px) * 2 * 384 * D},
            # executed: 32 MFMA 32x32x16 per 32-pixel tile and wave (e as bf16 hi + lo), 4 waves (8 for more than 128 slots)
            "mask_decode": {"bytes": T * px[-1] * (512 + (4 * L if self.decode_logits else 0) + 1), "flops": T * px[-1] * 2 * L * D,
                            "executed_flops": T * px[-1] * ((4 if L <= 128 else 8) * 32 * 32768 // 32)},
        }
        if self.retriever_form == "fused":
            tabs = sum(n * (h + w) * 128 * 4 for (h, w), n in zip(self.sizes, self.cfg["per_dh_num_heads"]))
            # K3' reads the map (512 B / pixel) and writes one 16-byte aux row (both statistics) per stage; K3'' (head.stats_form
            # "level", the default) reads the map once per LEVEL and writes the rows of all its stages
            if self.head.stats_form == "level":
                sbytes = T * sum(hw * (512 + 16 * n) if n == 2 else n * hw * (512 + 16) for hw, n in self.k1_launch_shapes())
            else:
                sbytes = T * ps * (512 + 16)
            out["retr_stats"] = {"bytes": sbytes, "flops": T * ps * int(2 * 36 / 64 * 2 * D * D),
                                 "executed_flops": T * ps * int(2 * 36 / 64 * 2 * D * D)}     # the triangular products are all it executes
            # executed matrix work of K1' per pixel (informational): (4 x 32 producer + 4 x 18 consumer) MFMA 32x32x16 per 32-pixel
            # tile - Q'' is carried as fp16 hi + lo. K1' stages the 16-byte aux row with every pixel. More than 128 slots (two
            # passes, the probabilities of 256 slot rows through HBM): pass 1 reads map + aux and writes 512 B of P per pixel,
            # pass 2 reads map + aux + P; 256 + 144 MFMAs per tile. The ALGORITHMIC bytes stay those of one read of the map.
            mf = 200 if L <= 128 else 400
            out["retr_attn"] = {"bytes": T * (ps * (512 + 16) + stages * (L * D * 4 + L * 260 * 4) + tabs), "flops": T * ps * 4 * L * D,
                                "executed_flops": T * ps * (mf * 32768 // 32)}
        else:
            out["kv_project"] = {"bytes": T * ps * 1536, "flops": T * ps * 4 * D * D}
            out["slot_attn"] = {"bytes": self.k1_algorithmic_bytes_per_step(), "flops": T * ps * 4 * L * D}
        return out
