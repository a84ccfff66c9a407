"""Where does a stage's deviation from the float64 oracle come from? Teacher-forced (the HIP path's own incoming slots and fused map),
sub-block by sub-block: self-attention + norm1 | retriever | norm2 | feed-forward + norm3 | temporal head | towers."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import slotvps_oracle as orc
from slotvps_amd import ops, synth
from test_head_gpu import build_head
cuda = torch.device("cuda:0")
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
z = np.load(os.path.join(GOLDEN, "head_small.npz"))
tag = sys.argv[1] if len(sys.argv) > 1 else "T2_64x128"
mode = sys.argv[2] if len(sys.argv) > 2 else "balanced"
T, H, W, L, seed = (int(x) for x in z[f"{tag}_meta"])
params = synth.make_params(synth.head_shapes(), seed)
feats = synth.make_clip_features(seed + 1, T, H, W)
slots = synth.make_slots(seed + 2, L)
sizes = synth.level_sizes(H, W)
head = build_head(cuda, params).set_retriever("fused").set_statistics(mode)
with torch.no_grad():
    tf = [torch.from_numpy(np.stack([feats[t][i] for t in range(T)])).to(cuda) for i in range(4)]
    pos_tabs = [ops.pos_embed_sine_tables(h, w, 256, cuda) for (h, w) in sizes]
    logits, embeds, fused = head.forward_clip(tf, torch.from_numpy(slots).to(cuda), pos_tabs)
embeds = embeds.cpu().numpy()
pos = [orc.pos_embed_sine(h, w) for (h, w) in sizes]
st = orc.Storage.fused_policy()
cfg = dict(orc.DEFAULT_CFG)
dt = np.float64
sidx = 0
d = lambda a, b: float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())
for lvl, n in enumerate(cfg["per_level_stages"]):
    h, w = sizes[lvl]
    for j in range(n):
        prefix = f"head_series_{lvl}.{j}."
        stage = getattr(head, f"head_series_{lvl}")[j]
        s_in = np.stack([slots.astype(np.float32) if sidx == 0 else embeds[sidx - 1, t] for t in range(T)])
        fm = [fused[lvl][t].float().cpu().numpy().astype(np.float64) for t in range(T)]
        g = lambda nme: orc._p(params, prefix, nme, dt)
        with torch.no_grad():
            x0 = torch.from_numpy(s_in).to(cuda)
            x1 = stage._self_attention(x0, residual_norm=True)
            r = stage.inst_interact.forward_pm(x1, fused[lvl], (h, w), pos_tabs[lvl])
            x2 = ops.row_ln(r, stage.norm2.weight, stage.norm2.bias, stage.norm2.eps, pre=x1)
            obj = stage.forward_till_ffn_pm(x0, fused[lvl], (h, w), pos_tabs[lvl])
        e = {"self-attn+norm1": 0.0, "retriever (own x1)": 0.0, "norm2 (own r)": 0.0, "ffn+norm3 (own x2)": 0.0, "till_ffn (chained)": 0.0}
        for t in range(T):
            s = s_in[t].astype(dt)
            o1 = orc.layer_norm(s + orc.multihead_self_attention(s, params, prefix + "self_attn.", cfg["nhead"], dt), g("norm1.weight"), g("norm1.bias"))
            e["self-attn+norm1"] = max(e["self-attn+norm1"], d(x1[t].cpu().numpy(), o1))
            own1 = x1[t].cpu().numpy().astype(dt)
            orr = orc.retriever(own1, fm[t], pos[lvl], params, prefix + "inst_interact.", st, dt)
            e["retriever (own x1)"] = max(e["retriever (own x1)"], d(r[t].cpu().numpy(), orr))
            ownr = r[t].cpu().numpy().astype(dt)
            o2 = orc.layer_norm(own1 + ownr, g("norm2.weight"), g("norm2.bias"))
            e["norm2 (own r)"] = max(e["norm2 (own r)"], d(x2[t].cpu().numpy(), o2))
            own2 = x2[t].cpu().numpy().astype(dt)
            y = orc.linear(orc._act(cfg["activation"])(orc.linear(own2, g("linear1.weight"), g("linear1.bias"))), g("linear2.weight"), g("linear2.bias"))
            o3 = orc.layer_norm(own2 + y, g("norm3.weight"), g("norm3.bias"))
            e["ffn+norm3 (own x2)"] = max(e["ffn+norm3 (own x2)"], d(obj[t].cpu().numpy(), o3))
            oc = orc.stage_till_ffn(s_in[t], fm[t], pos[lvl], params, prefix, cfg["nhead"], cfg["activation"], st, dt)
            e["till_ffn (chained)"] = max(e["till_ffn (chained)"], d(obj[t].cpu().numpy(), oc))
        print(f"stage {sidx} (level {lvl}, {h}x{w}): " + ", ".join(f"{k} {v:.1e}" for k, v in e.items()), flush=True)
        sidx += 1

# ---- inside the retriever of the last stage: [A | s1 | s0] from K1', the value fold, norm1 ----------------------------------------
print("\nlast stage, retriever internals (float64 evaluation of every step from the kernels' own intermediate):")
lvl, j, sidx = 3, 1, 6
h, w = sizes[lvl]
prefix = f"head_series_{lvl}.{j}.inst_interact."
stage = getattr(head, f"head_series_{lvl}")[j]
mdc = stage.inst_interact
s_in = np.stack([embeds[sidx - 1, t] for t in range(T)])
keep = {}
real_attn = ops.retr_attn
def spy(*a, **k):
    out = real_attn(*a, **k)
    keep["ext"] = out.clone()
    return out
ops.retr_attn = spy
with torch.no_grad():
    x1 = stage._self_attention(torch.from_numpy(s_in).to(cuda), residual_norm=True)
    r = mdc.forward_pm(x1, fused[lvl], (h, w), pos_tabs[lvl])
ops.retr_attn = real_attn
ext = keep["ext"].cpu().numpy().astype(np.float64)            # [T, L, 272]: A (256) | s1 | s0 | pad
g = lambda nme: orc._p(params, prefix, nme, np.float64)
Wv, bv = g("to_v.weight"), g("to_v.bias")
Wc = Wv - Wv.mean(axis=0, keepdims=True); bc = bv - bv.mean()
gv, betav = g("norm_v.weight"), g("norm_v.bias")
for t in range(T):
    own1 = x1[t].cpu().numpy().astype(np.float64)
    fm = fused[lvl][t].float().cpu().numpy().astype(np.float64)
    q, k, v = orc.retriever_project(own1, fm, pos[lvl].astype(np.float64), params, prefix, orc.Storage.exact(), np.float64)
    out64, pre64 = orc.retriever_core(q, k, v, g("norm1.weight"), g("norm1.bias"), return_pre=True)
    logits = q @ k.T
    P = orc.softmax(logits, axis=0)
    xv = fm @ Wv.T + bv
    rstd = 1.0 / np.sqrt(xv.var(axis=1) + 1e-5)
    A64 = (P * rstd[None, :]) @ fm; s1_64 = (P * rstd[None, :]).sum(axis=1); s0_64 = P.sum(axis=1)
    A, s1, s0 = ext[t, :, :256], ext[t, :, 256], ext[t, :, 257]
    pre_from_ext = gv[None, :] * (A @ Wc.T + s1[:, None] * bc[None, :]) + betav[None, :] * s0[:, None]
    pre_from_64 = gv[None, :] * (A64 @ Wc.T + s1_64[:, None] * bc[None, :]) + betav[None, :] * s0_64[:, None]
    out_from_ext = orc.relu(orc.layer_norm(pre_from_ext, g("norm1.weight"), g("norm1.bias")))
    rowstd = pre64.std(axis=1)
    print(f" frame {t}: logits |max| {np.abs(logits).max():.1f} std {logits.std():.1f}; A rel err {np.abs(A - A64).max() / np.abs(A64).max():.1e} (|A| max {np.abs(A64).max():.1f}), "
          f"s1 rel {np.abs(s1 - s1_64).max() / np.abs(s1_64).max():.1e}, s0 rel {np.abs(s0 - s0_64).max() / np.abs(s0_64).max():.1e}; "
          f"algebra check pre(A64) vs pre64 {np.abs(pre_from_64 - pre64).max():.1e}; pre(ext) vs pre64 {np.abs(pre_from_ext - pre64).max():.1e} with row std min {rowstd.min():.2e} max {rowstd.max():.2e} |pre| max {np.abs(pre64).max():.1f}; "
          f"out(pre(ext)) vs out64 {np.abs(out_from_ext - out64).max():.1e}; kernel out vs out64 {np.abs(r[t].cpu().numpy() - out64).max():.1e}", flush=True)
    worst = np.unravel_index(np.argmax(np.abs(out_from_ext - out64)), out64.shape)
    l = worst[0]
    print(f"   worst slot {l}: s0 {s0_64[l]:.3e}, row std of pre {rowstd[l]:.2e}, |pre err| max in that row {np.abs(pre_from_ext[l] - pre64[l]).max():.1e}, |A64 row| max {np.abs(A64[l]).max():.2e}, A err in that row {np.abs(A[l] - A64[l]).max():.1e}")
