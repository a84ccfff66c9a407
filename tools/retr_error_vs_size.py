import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import slotvps_oracle as orc
from slotvps_amd import ops
from slotvps_amd.slot_head import MaskDynamicConv
cuda = torch.device("cuda:0")
rng = np.random.default_rng(7)
m = MaskDynamicConv(256).to(cuda).eval()
P = {}
with torch.no_grad():
    for n in ("to_q", "to_k", "to_v"):
        lim = float(np.sqrt(6.0 / 512))
        P[f"{n}.weight"] = rng.uniform(-lim, lim, (256, 256)).astype(np.float32)
        P[f"{n}.bias"] = (0.1 * rng.standard_normal(256)).astype(np.float32)
        getattr(m, n).weight.copy_(torch.from_numpy(P[f"{n}.weight"])); getattr(m, n).bias.copy_(torch.from_numpy(P[f"{n}.bias"]))
    for n in ("norm_q", "norm_k", "norm_v", "norm1"):
        P[f"{n}.weight"] = rng.uniform(0.5, 1.5, 256).astype(np.float32)
        P[f"{n}.bias"] = (0.1 * rng.standard_normal(256)).astype(np.float32)
        getattr(m, n).weight.copy_(torch.from_numpy(P[f"{n}.weight"])); getattr(m, n).bias.copy_(torch.from_numpy(P[f"{n}.bias"]))
for (H, W, L) in ((16, 32, 100), (32, 64, 100), (64, 128, 100), (128, 256, 100)):
    feat = orc.round_bf16(rng.standard_normal((1, H * W, 256)).astype(np.float32))
    slots = rng.standard_normal((1, L, 256)).astype(np.float32)
    tabs = ops.pos_embed_sine_tables(H, W, 256, cuda)
    ft = torch.from_numpy(feat).to(cuda).to(torch.bfloat16)
    ref, pre = None, None
    ref = orc.retriever(slots[0], feat[0], orc.pos_embed_sine(H, W), P, "", st=orc.Storage.exact(), dt=np.float64)
    ref32 = orc.retriever(slots[0], feat[0], orc.pos_embed_sine(H, W), P, "", st=orc.Storage.exact(), dt=np.float32)
    line = f"{H}x{W}: fp32 oracle vs fp64 {np.abs(ref32 - ref).max():.1e};"
    with torch.no_grad():
        for mode in ("fast", "balanced", "tight"):
            m.tight_stats = mode == "tight"; m.precise_query_p = mode == "balanced"
            for qs in ("fp16", "bf16", "fp32"):                     # operands of the query-side products
                m.query_side = qs
                got = m.forward_fused(torch.from_numpy(slots).to(cuda), ft, (H, W), tabs).cpu().numpy()
                line += f" {mode}/{qs} {np.abs(got[0] - ref).max():.2e}"
        m.query_side = "fp16"
    print(line, flush=True)
