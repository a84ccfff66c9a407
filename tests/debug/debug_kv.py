import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from util import orc, to_bf16_t, bf16_t_to_np
from slotvps_amd import ops
dev = torch.device("cuda:0")
for (T, H, W) in [(3, 64, 128)]:
    seed = H * 1000 + W + T
    params = synth.make_params(synth.retriever_shapes(""), seed)
    rng = np.random.default_rng(seed)
    feat = np.stack([synth.smooth_features(rng, 256, H, W).reshape(256, H * W).T for _ in range(T)])
    tf = to_bf16_t(feat, dev)
    g = lambda n: torch.from_numpy(params[n]).to(dev)
    tabs = ops.pos_embed_sine_tables(H, W, 256, dev)
    k, v = ops.kv_project(tf, H, W, tabs, g("to_k.weight").to(torch.bfloat16).contiguous(), g("to_k.bias"),
                          g("norm_k.weight"), g("norm_k.bias"), 1e-5, g("to_v.weight").to(torch.bfloat16).contiguous(),
                          g("to_v.bias"), g("norm_v.weight"), g("norm_v.bias"), 1e-5)
    torch.cuda.synchronize()
    k, v, fb = bf16_t_to_np(k), bf16_t_to_np(v), bf16_t_to_np(tf)
    yt, xt = tabs[0].cpu().numpy(), tabs[1].cpu().numpy()
    pos = np.concatenate([np.repeat(yt[:, None, :], W, 1), np.repeat(xt[None, :, :], H, 0)], -1).reshape(H * W, 256)
    st = orc.Storage.bf16_policy()
    for t in range(T):
        _, ko, vo = orc.retriever_project(np.zeros((1, 256), np.float32), fb[t], pos, params, "", st, np.float32)
        for nm, got, ref in (("k", k[t], ko), ("v", v[t], vo)):
            d = np.abs(got - ref)
            rel = d / np.maximum(np.abs(ref), 1e-3)
            bad = np.argwhere(rel > 0.006)
            print(f"{T}x{H}x{W} t={t} {nm}: max abs {d.max():.4f} max rel {rel.max():.4f} n(rel>0.6%) {len(bad)} of {d.size}; n diff {(d>0).sum()}")
            if len(bad):
                print("   bad pixels (px%32):", sorted(set((bad[:, 0] % 32).tolist()))[:32], " channels:", sorted(set(bad[:, 1].tolist()))[:20], " tiles:", sorted(set((bad[:, 0] // 32).tolist()))[:40])
                big = np.argwhere(d > 0.1)
                print("   n(abs>0.1)", len(big), "first:", [tuple(x) for x in big[:24].tolist()])
                if len(big):
                    import collections
                    print("   by channel%8:", collections.Counter((big[:, 1] % 8).tolist()), " by px%32:", sorted(collections.Counter((big[:, 0] % 32).tolist()).items())[:32])
                i, j = bad[0]
                print("   e.g.", got[i, j], ref[i, j], "pixel", i, "ch", j)
        # also check k computed WITHOUT pos to see if pos was ignored
        _, k_nopos, _ = orc.retriever_project(np.zeros((1, 256), np.float32), fb[t], None, params, "", st, np.float32)
        print("   |k - k_nopos| max", np.abs(k[t] - k_nopos).max(), " |ko - k_nopos| max", np.abs(ko - k_nopos).max())
