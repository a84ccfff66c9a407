"""Teacher-forced localisation of head differences (GPU vs oracle), stage 0 of level 0/3."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from util import orc
from test_head_gpu import build_head
from slotvps_amd import ops

dev = torch.device("cuda:0")
T, H, W, L, seed = 2, 64, 128, 100, 301
params = synth.make_params(synth.head_shapes(), seed)
feats = synth.make_clip_features(seed + 1, T, H, W)
slots = synth.make_slots(seed + 2, L)
sizes = synth.level_sizes(H, W)
head = build_head(dev, params)
st = orc.Storage.bf16_policy(torch_gemm=True)
with torch.no_grad():
    tf = [torch.from_numpy(np.stack([feats[t][i] for t in range(T)])).to(dev) for i in range(4)]
    pos_pm = [ops.pos_embed_sine(h, w, 256, dev) for (h, w) in sizes]
    y0 = head.fuse_level(0, tf[0], None)
    f_pm = y0.permute(0, 2, 3, 1).reshape(T, -1, 256).contiguous()
    o_logits, o_embeds, o_fused = orc.head_forward(feats, slots, [p.cpu().numpy() for p in pos_pm], params, st=st)
    d = np.abs(f_pm.float().cpu().numpy()[0] - o_fused[0][0])
    print("fused lvl0: max", d.max(), "n diff", (d > 0).sum(), "of", d.size, "| values max", np.abs(o_fused[0][0]).max())
    stage = head.head_series_0[0]
    s = torch.from_numpy(slots).to(dev).unsqueeze(0).expand(T, -1, -1).contiguous()
    x = s.transpose(0, 1)
    x = stage.norm1(x + stage.self_attn(x, x, value=x, need_weights=False)[0]).transpose(0, 1).contiguous()
    pfx = "head_series_0.0."
    g = lambda n: params[pfx + n]
    s1_o = orc.layer_norm(slots + orc.multihead_self_attention(slots, params, pfx + "self_attn.", 8), g("norm1.weight"), g("norm1.bias"))
    print("s1 (self-attn + norm1) diff", np.abs(x[0].cpu().numpy() - s1_o).max())
    ic = stage.inst_interact
    q = ic.norm_q(ic.to_q(x)).to(torch.bfloat16)
    k, v = ic.project_kv(f_pm, pos_pm[0])
    # oracle projections from the GPU's own s1 / f_pm
    qo, ko, vo = orc.retriever_project(x[0].cpu().numpy(), f_pm[0].float().cpu().numpy(), pos_pm[0].cpu().numpy(),
                                       params, pfx + "inst_interact.", st, np.float32)
    for nm, a, b in (("q", q[0], qo), ("k", k[0], ko), ("v", v[0], vo)):
        dd = np.abs(a.float().cpu().numpy() - b)
        print(f"{nm}: max diff {dd.max():.3e}  n diff {(dd > 0).sum()} of {dd.size}")
    r = ops.slot_attn(q.contiguous(), k.contiguous(), v.contiguous(), ic.norm1.weight, ic.norm1.bias)
    ro = orc.retriever_core(q[0].float().cpu().numpy().astype(np.float64), k[0].float().cpu().numpy().astype(np.float64),
                            v[0].float().cpu().numpy().astype(np.float64), g("inst_interact.norm1.weight").astype(np.float64),
                            g("inst_interact.norm1.bias").astype(np.float64))
    print("K1 on GPU q/k/v vs oracle core:", np.abs(r[0].cpu().numpy() - ro).max())
    ro2 = orc.retriever_core(qo.astype(np.float64), ko.astype(np.float64), vo.astype(np.float64),
                             g("inst_interact.norm1.weight").astype(np.float64), g("inst_interact.norm1.bias").astype(np.float64))
    print("oracle core on oracle q/k/v vs on GPU q/k/v (sensitivity to the flips):", np.abs(ro - ro2).max(), "| HW =", k.shape[1])
    lg = q[0].float() @ k[0].float().T
    print("logit range", lg.min().item(), lg.max().item(), "std", lg.std().item())
