"""PyTorch-CPU restatement of the slot head + mask decode.  TEST INFRASTRUCTURE ONLY (see slotvps_oracle.py).

Same algorithm, same parameter dict and same call structure as `slotvps_oracle.head_forward` (which is pinned
against the reference's own modules), written with the torch ops the reference itself uses - F.linear,
F.layer_norm, nn.functional.multi_head_attention_forward, F.interpolate, einsum, F.softmax - and, like the
reference (mmdet/models/detectors/dynamic_mask_head.py:302-338), looping over the frames of a clip in Python.
It exists for ONE purpose: `bench.py`'s `cpu_baseline` leg times it on the GPU box's host cores as "the
reference's CPU PyTorch path" (fp32, torch.set_num_threads(N), T = 5), since the
reference's files do not travel. `tests/test_torch_cpu_head.py` checks it against the NumPy oracle.
"""
import math

import torch
import torch.nn.functional as F


def _t(params, name):
    v = params[name]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


class TorchCpuHead:
    def __init__(self, params, cfg=None):
        from .slotvps_oracle import DEFAULT_CFG
        self.cfg = dict(DEFAULT_CFG, **(cfg or {}))
        self.p = {k: _t(params, k).float().contiguous() for k in params}

    def g(self, prefix, name):
        return self.p[prefix + name]

    # dynamic_mask_head.py:423-461
    def retriever(self, slots, feat, pos, prefix):
        g = lambda n: self.g(prefix, n)
        q = F.layer_norm(F.linear(slots, g("to_q.weight"), g("to_q.bias")), (256,), g("norm_q.weight"), g("norm_q.bias"))
        k = F.layer_norm(F.linear(feat + pos, g("to_k.weight"), g("to_k.bias")), (256,), g("norm_k.weight"), g("norm_k.bias"))
        v = F.layer_norm(F.linear(feat, g("to_v.weight"), g("to_v.bias")), (256,), g("norm_v.weight"), g("norm_v.bias"))
        attn = F.softmax(q @ k.t(), dim=0)                                   # over slots (:446)
        return F.relu(F.layer_norm(attn @ v, (256,), g("norm1.weight"), g("norm1.bias")))

    # :550-572, :494-527
    def temporal(self, S, prefix):
        g = lambda n: self.g(prefix, n)
        gi = lambda n: self.g(prefix + "inst_interact.", n)
        q = F.layer_norm(F.linear(S, gi("to_q.weight"), gi("to_q.bias")), (256,), gi("norm_q.weight"), gi("norm_q.bias"))
        k = F.layer_norm(F.linear(S, gi("to_k.weight"), gi("to_k.bias")), (256,), gi("norm_k.weight"), gi("norm_k.bias"))
        v = F.layer_norm(F.linear(S, gi("to_v.weight"), gi("to_v.bias")), (256,), gi("norm_v.weight"), gi("norm_v.bias"))
        r = F.relu(F.layer_norm(F.softmax(q @ k.t(), dim=0) @ v, (256,), gi("norm1.weight"), gi("norm1.bias")))
        x = F.layer_norm(S + r, (256,), g("norm2.weight"), g("norm2.bias"))
        act = F.relu if self.cfg["temporal_activation"] == "relu" else F.gelu
        y = F.linear(act(F.linear(x, g("linear1.weight"), g("linear1.bias"))), g("linear2.weight"), g("linear2.bias"))
        return F.layer_norm(x + y, (256,), g("norm3.weight"), g("norm3.bias"))

    # :342-388
    def till_ffn(self, s, feat, pos, prefix):
        g = lambda n: self.g(prefix, n)
        x = s[:, None, :]                                                    # [L, 1, C] sequence-first (:346)
        a, _ = F.multi_head_attention_forward(
            x, x, x, 256, self.cfg["nhead"], g("self_attn.in_proj_weight"), g("self_attn.in_proj_bias"), None, None,
            False, 0.0, g("self_attn.out_proj.weight"), g("self_attn.out_proj.bias"), training=False, need_weights=False)
        s1 = F.layer_norm(s + a[:, 0], (256,), g("norm1.weight"), g("norm1.bias"))
        r = self.retriever(s1, feat, pos, prefix + "inst_interact.")
        s2 = F.layer_norm(s1 + r, (256,), g("norm2.weight"), g("norm2.bias"))
        act = F.gelu if self.cfg["activation"] == "gelu" else F.relu
        y = F.linear(act(F.linear(s2, g("linear1.weight"), g("linear1.bias"))), g("linear2.weight"), g("linear2.bias"))
        return F.layer_norm(s2 + y, (256,), g("norm3.weight"), g("norm3.bias"))

    # :390-400
    def after_ffn(self, obj, prefix):
        g = lambda n: self.g(prefix, n)
        c = e = obj
        for i in range(self.cfg["num_cls"]):
            c = F.relu(F.layer_norm(F.linear(c, g(f"cls_module.{3 * i}.weight")), (256,), g(f"cls_module.{3 * i + 1}.weight"),
                                    g(f"cls_module.{3 * i + 1}.bias")))
        for i in range(self.cfg["num_reg"]):
            e = F.relu(F.layer_norm(F.linear(e, g(f"reg_module.{3 * i}.weight")), (256,), g(f"reg_module.{3 * i + 1}.weight"),
                                    g(f"reg_module.{3 * i + 1}.bias")))
        return F.linear(c, g("class_logits.weight"), g("class_logits.bias")), e

    @torch.no_grad()
    def forward(self, features, init_slots, pos):
        """features[t][i] [128, Hi, Wi] tensors, init_slots [L, 256], pos[i] [Hi*Wi, 256] -> (logits [T][S], embeds [T][S],
        fused [T][4] pixel-major) - the signature of slotvps_oracle.head_forward."""
        T, nlev = len(features), len(features[0])
        wc = self.p["conv_trans.conv.weight"].reshape(256, -1, 1, 1)
        bc = self.p["conv_trans.conv.bias"]
        slots = [init_slots.float() for _ in range(T)]
        logits, embeds = [[] for _ in range(T)], [[] for _ in range(T)]
        fused = [[None] * nlev for _ in range(T)]
        prev = None
        stage_idx = 0
        for i in range(nlev):
            x = torch.stack([features[t][i] for t in range(T)]).float()                  # :159-164
            if prev is None:
                cat = torch.cat([x, x, x], dim=1)                                        # :183
            else:
                cat = torch.cat([F.interpolate(prev, None, 2, mode="bilinear", align_corners=False), x], dim=1)   # :178-179
            cur = F.conv2d(cat, wc, bc)                                                  # :181/:185
            prev = cur
            cur_pm = [cur[t].permute(1, 2, 0).reshape(-1, 256) for t in range(T)]        # 'b c h w -> b h w c' (:428)
            for t in range(T):
                fused[t][i] = cur_pm[t]
            for j in range(self.cfg["per_level_stages"][i]):
                prefix = f"head_series_{i}.{j}."
                objs = [self.till_ffn(slots[t], cur_pm[t], pos[i], prefix) for t in range(T)]     # :302
                if stage_idx in self.cfg["temporal_stages"]:
                    S = torch.cat(objs, 0)
                    S = S + self.temporal(S, prefix + "temporal_query_head.")            # :313-317
                    objs = list(S.split(S.shape[0] // T, 0))
                outs = [self.after_ffn(o, prefix) for o in objs]
                for t in range(T):
                    logits[t].append(outs[t][0])
                    embeds[t].append(outs[t][1])
                slots = [o[1] for o in outs]                                             # :210-211
                stage_idx += 1
        return logits, embeds, fused

    @staticmethod
    @torch.no_grad()
    def mask_decode(feat_pm, embed, bn_scale, bn_shift, fg_scale, fg_shift):
        """vps_temporal_slots.py:144-160 on a pixel-major map: [HW, 256], [L, 256] -> [L, HW]."""
        gmap = F.normalize(feat_pm * bn_scale + bn_shift, p=2, dim=1)
        return (embed @ gmap.t()) * fg_scale + fg_shift
