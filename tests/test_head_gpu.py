"""End-to-end parity of the HIP slot head (host mirror + kernels) on a small clip, in both forms of the bf16 mode:
"fused" (statistics-fused retriever K3' + K1', the default) and "kv" (K3 + K1 through bf16 k / v tensors).

Two yardsticks:
  * the CPU oracle run under the SAME storage policy: what the HIP path is supposed to compute; the residual is
    accumulation order, the kernels' own hi / lo splits and (kv form) bf16 rounding flips of single k / v elements;
  * the golden fixture captured from the reference's fp32 modules (`head_small.npz`): asserted with measured bounds -
    teacher-forced per stage (each stage fed the REFERENCE's own incoming slots), free-running, final mask logits and the
    per-pixel slot argmax. The bf16 storage of the fused maps is the floor of these numbers (exact mode removes it:
    tests/test_exact_mode_gpu.py).
Tolerances are written next to each assertion."""
import os

import numpy as np
import pytest

import synth
from util import orc, GOLDEN

pytestmark = pytest.mark.gpu


# per-pixel slot argmax of the FREE-running head (its own embeddings through all seven stages) against the argmax of the reference's
# fp32 mask logits, last frame: measured minimum over the fixture cases, asserted with a margin
SAME_FREE_MIN = {"fused": 0.85, "kv": 0.80}      # measured 87.1 / 91.0 % (fused), 85.0 / 83.2 % (kv)


def build_head(cuda, params, mode=None):
    """mode None: the product's default (MultiScaleDynamicMaskHead DEFAULT_MODE = "fp16x2")."""
    import torch
    from slotvps_amd.slot_head import MultiScaleDynamicMaskHead
    cfg = synth.R50_HEAD_CFG
    head = MultiScaleDynamicMaskHead(
        dh_dim=256, num_classes=cfg["num_classes"], dim_feedforward=cfg["dim_feedforward"], nhead=cfg["nhead"],
        dropout=0.0, activation=cfg["activation"], dh_num_heads=7, per_dh_num_heads=list(cfg["per_dh_num_heads"]),
        feat_num_levels=4, merge_operation="concat", trans_in_dim=cfg["trans_in_dim"], num_cls=cfg["num_cls"],
        num_reg=cfg["num_reg"],
        temporal_query_attention_config=dict(d_model=256, dim_feedforward=cfg["temporal_dim_feedforward"], dropout=0.0,
                                             activation=cfg["temporal_activation"], softmax_dim="slots", drop_path=0.),
        apply_temporal_query_atten_stages=list(cfg["apply_temporal_query_atten_stages"]))
    sd = {k: torch.from_numpy(v).reshape(head.state_dict()[k].shape) for k, v in params.items()}
    head.load_state_dict(sd, strict=True)
    head = head.to(cuda).eval()
    return head if mode is None else head.set_mode(mode)


def frames(fmap, sl):
    """Frames `sl` of a fused level map: [T, HW, 256] (16-bit / fp32 modes) or [2 (hi, lo), T, HW, 256] (fp16x2)."""
    return fmap[:, sl] if fmap.dim() == 4 else fmap[sl]


@pytest.mark.parametrize("form", ["fused", "kv"])
@pytest.mark.parametrize("tag", ["T2_64x128", "T3_64x64"])
def test_head_matches_oracle_and_reference(cuda, tag, form):
    import torch
    from slotvps_amd import ops
    from slotvps_amd.slot_head import generate_final_outputs
    z = np.load(os.path.join(GOLDEN, "head_small.npz"))
    T, H, W, L, seed = (int(x) for x in z[f"{tag}_meta"])
    params = synth.make_params(synth.head_shapes(), seed)
    feats = synth.make_clip_features(seed + 1, T, H, W)
    slots = synth.make_slots(seed + 2, L)
    sizes = synth.level_sizes(H, W)
    head = build_head(cuda, params).set_mode("bf16" if form == "fused" else "bf16_kv")
    if os.environ.get("SVPS_TEST_SLOT_GEMM") == "0":          # experiment switches: the slot side's dense layers in library fp32,
        head.set_slot_gemm(False)
    if os.environ.get("SVPS_TEST_BGEMM") == "torch":          # K9 replaced by float64 matmuls (which layer limits a stage's parity?)
        def _bg(a_, b_, bias=None, alpha=1.0, out=None):
            a3 = a_ if a_.dim() == 3 else a_.unsqueeze(0)
            b3 = b_ if b_.dim() == 3 else b_.unsqueeze(0)
            r = alpha * torch.matmul(a3.double(), b3.double().transpose(1, 2))
            if bias is not None:
                r = r + (bias if bias.dim() == 2 else bias.unsqueeze(0)).double()[:, None, :]
            r = r.float()
            if out is not None:
                out.copy_(r)
                return out
            return r
        ops.bgemm = _bg
    with torch.no_grad():
        tf = [torch.from_numpy(np.stack([feats[t][i] for t in range(T)])).to(cuda) for i in range(4)]
        pos_tabs = [ops.pos_embed_sine_tables(h, w, 256, cuda) for (h, w) in sizes]
        logits, embeds, fused = head.forward_clip(tf, torch.from_numpy(slots).to(cuda), pos_tabs)
        torch.cuda.synchronize()
    logits, embeds = logits.cpu().numpy(), embeds.cpu().numpy()          # [7, T, L, *]

    # ---- yardstick 1: oracle under the same storage policy -------------------------------------
    pos = [orc.pos_embed_sine(h, w) for (h, w) in sizes]
    st = orc.Storage.fused_policy() if form == "fused" else orc.Storage.bf16_policy()
    _, _, o_fused = orc.head_forward(feats, slots, pos, params, st=st)
    g_fused = [[fused[i][t].float().cpu().numpy() for i in range(4)] for t in range(T)]
    f_err = max(np.abs(g_fused[t][i] - o_fused[t][i]).max() for t in range(T) for i in range(4))
    f_frac = np.mean([(g_fused[t][3] != o_fused[t][3]).mean() for t in range(T)])
    # (a) level fusion (K4), free-running over the four levels: bf16 maps of magnitude < 8 -> one bf16
    #     ulp is 2^-5; a one-ulp flip at level i is blended into level i+1, so flips accumulate slowly.
    assert f_err <= 6.3e-2 and f_frac < 0.02, (f_err, f_frac)
    # (b) every stage on IDENTICAL inputs (teacher forcing: the oracle stage is fed the HIP path's own
    #     incoming slots and fused map). This is the per-function parity of a3/a4/a5 + K1. The chain
    #     itself is chaotic with these random weights - logits have sigma ~ 19, so the softmax over
    #     slots is close to a hard assignment and perturbations grow ~5x per stage (the reference's own
    #     fp32 result moves 2e-5 -> 6e-4 over the 7 stages under a mere change of summation order) -
    #     so free-running end-to-end distances are reported, not asserted tightly.
    cfg = dict(orc.DEFAULT_CFG)
    stage_err, logit_err = [], []
    sidx = 0
    for lvl, n in enumerate(cfg["per_level_stages"]):
        for j in range(n):
            s_in = [slots.astype(np.float32)] * T if sidx == 0 else [embeds[sidx - 1, t] for t in range(T)]
            # float64: a float32 evaluation of the same stage is itself 1e-4 ... 1e-3 away at the fine levels (sharp softmax over 8192 pixels)
            lg, em = orc.stage(s_in, [g_fused[t][lvl] for t in range(T)], [pos[lvl]] * T, params,
                               f"head_series_{lvl}.{j}.", sidx in cfg["temporal_stages"], cfg, st, dt=np.float64)
            stage_err.append(max(np.abs(embeds[sidx, t] - em[t]).max() for t in range(T)))
            logit_err.append(max(np.abs(logits[sidx, t] - lg[t]).max() for t in range(T)))
            sidx += 1
    o_logits, o_embeds, _ = orc.head_forward(feats, slots, pos, params, st=st, fused_override=g_fused)
    free = [max(np.abs(embeds[s_, t] - o_embeds[t][s_]).max() for t in range(T)) for s_ in range(7)]
    free_mean = float(np.mean([np.abs(embeds[6, t] - o_embeds[t][6]).mean() for t in range(T)]))
    print(f"\n[{tag}] fused maps vs oracle: max {f_err:.2e} ({100 * f_frac:.0f}% of elements one ulp off)")
    print(f"[{tag}] per-stage (identical inputs) embed err " + " ".join(f"{x:.1e}" for x in stage_err)
          + " | logits " + " ".join(f"{x:.1e}" for x in logit_err))
    print(f"[{tag}] free-running embed err " + " ".join(f"{x:.1e}" for x in free) + f" | stage-6 mean abs {free_mean:.2e}")
    # Per-stage bound on identical inputs, slot embeddings are O(1). kv form: bf16 rounding flips of single q / k / v
    # elements seen through one sharp softmax (measured <= 9e-3). fused form: nothing is rounded as a tensor; what is left
    # are the 16-bit splits of Q'' and P and the fp16 statistics (measured <= 2e-3).
    # Since the probabilities travel as 2^7 * P * rstd_v (csrc/common.h: slots that own almost no pixel no longer fall into fp16's
    # subnormal range) the whole stage measures 2.2e-4 ... 9.6e-4 in the default form (1.2e-3 before, growing with the level).
    bound = 2e-3 if form == "fused" else 2e-2
    assert max(stage_err) <= bound and max(logit_err) <= bound, (stage_err, logit_err)
    assert free[0] <= 5e-3 and free_mean <= 5e-2

    # ---- yardstick 2: the reference's own fp32 outputs ------------------------------------------------
    # (a) teacher-forced: stage s is fed the REFERENCE's stage s-1 embeddings, the kernels' own bf16 fused map of the level
    tf_err = []
    sidx = 0
    with torch.no_grad():
        for lvl, n in enumerate(cfg["per_level_stages"]):
            h, w = sizes[lvl]
            for j in range(n):
                s_in = np.stack([slots.astype(np.float32) if sidx == 0 else z[f"{tag}_embeds_{t}"][sidx - 1] for t in range(T)])
                stage = getattr(head, f"head_series_{lvl}")[j]
                _, em = stage.forward_pm(torch.from_numpy(s_in).to(cuda), fused[lvl], (h, w), pos_tabs[lvl],
                                         sidx in cfg["temporal_stages"], 1)
                tf_err.append(max(np.abs(em[t].cpu().numpy() - z[f"{tag}_embeds_{t}"][sidx]).max() for t in range(T)))
                sidx += 1
    r_free = [max(np.abs(embeds[s_, t] - z[f"{tag}_embeds_{t}"][s_]).max() for t in range(T)) for s_ in range(7)]
    r_f3 = max(np.abs(fused[3][t].float().cpu().numpy() - z[f"{tag}_fused3_{t}"]).max() for t in range(T))
    print(f"[{tag}/{form}] vs the reference's fp32 outputs: fused map (finest) {r_f3:.2e}")
    print(f"[{tag}/{form}]   teacher-forced per stage " + " ".join(f"{x:.1e}" for x in tf_err))
    print(f"[{tag}/{form}]   free-running per stage   " + " ".join(f"{x:.1e}" for x in r_free))
    # The fused maps are STORED as bf16 (|f| < 8: half an ulp = 1.6e-2): that rounding, seen through one sharp softmax,
    # is the floor of the teacher-forced distance in both forms; the kv form adds the bf16 rounding of every k / v element.
    assert r_f3 <= 3.2e-2
    assert max(tf_err) <= (6e-2 if form == "fused" else 2e-1), tf_err
    assert r_free[0] <= (6e-2 if form == "fused" else 2e-1)

    # ---- K2 on the head's own outputs vs oracle decode of the same tensors -----------------------------
    w, b, mu, var = z[f"{tag}_bn"]
    fg = z[f"{tag}_fg"]
    feat_bn = torch.nn.BatchNorm2d(256).to(cuda).eval()
    fg_bn = torch.nn.BatchNorm2d(1).to(cuda).eval()
    with torch.no_grad():
        feat_bn.weight.copy_(torch.from_numpy(w)); feat_bn.bias.copy_(torch.from_numpy(b))
        feat_bn.running_mean.copy_(torch.from_numpy(mu)); feat_bn.running_var.copy_(torch.from_numpy(var))
        fg_bn.weight.fill_(float(fg[0])); fg_bn.bias.fill_(float(fg[1]))
        fg_bn.running_mean.fill_(float(fg[2])); fg_bn.running_var.fill_(float(fg[3]))
        emb_last = torch.from_numpy(embeds[6]).to(cuda)
        masks, amax = generate_final_outputs(fused[3], emb_last, feat_bn, fg_bn, want_argmax=True)
        torch.cuda.synchronize()
    scale, shift = orc.bn_eval_affine(w.astype(np.float64), b.astype(np.float64), mu.astype(np.float64), var.astype(np.float64))
    fgs, fgb = orc.bn_eval_affine(np.float64(fg[0]), np.float64(fg[1]), np.float64(fg[2]), np.float64(fg[3]))
    worst = 0.0
    for t in range(T):
        ref = orc.mask_decode(fused[3][t].float().cpu().numpy().astype(np.float64), embeds[6, t].astype(np.float64),
                              scale, shift, fgs, fgb)
        worst = max(worst, np.abs(masks[t].cpu().numpy() - ref).max())
        srt = np.sort(ref, axis=0)
        decided = (srt[-1] - srt[-2]) > 2e-4
        np.testing.assert_array_equal(amax[t].cpu().numpy()[decided], orc.slot_argmax(ref)[decided])
    print(f"[{tag}] mask logits vs oracle decode of the same tensors: {worst:.2e}")
    assert worst <= 1e-4          # north star: 1e-4 on the float mask logits

    # ---- final mask logits and per-pixel slot argmax vs the reference fixture (last frame), teacher-forced decode: the
    #      reference's own last-stage embeddings, the kernels' bf16 fused map
    m_ref = z[f"{tag}_mask"]
    with torch.no_grad():
        emb_ref = torch.from_numpy(np.stack([z[f"{tag}_embeds_{t}"][6] for t in range(T)])).to(cuda)
        m_tf, a_tf = generate_final_outputs(fused[3], emb_ref, feat_bn, fg_bn, want_argmax=True)
    m_err = float(np.abs(m_tf[T - 1].cpu().numpy() - m_ref).max())
    same = (a_tf[T - 1].cpu().numpy() == np.argmax(m_ref, axis=0)).mean()
    same_free = (amax[T - 1].cpu().numpy() == np.argmax(m_ref, axis=0)).mean()
    print(f"[{tag}/{form}] mask logits vs the reference (its embeddings, our bf16 map): {m_err:.2e}; slot argmax equal on "
          f"{100 * same:.2f} % of the pixels (free-running head: {100 * same_free:.2f} %)")
    # mask logits = fg_scale * e . normalize(bn(f)): the bf16 map moves them by ~1e-3 of their O(0.1) range
    assert m_err <= 2e-3 and same >= 0.97
    assert same_free >= SAME_FREE_MIN[form], same_free          # free-running: seven stages of bf16-map rounding through sharp softmaxes


def test_default_mode_is_the_contract_mode(cuda):
    """a head built without any mode key runs the mode that meets the north star's tolerance (fp16x2), so do its
    stand-alone sub-modules; the 16-bit storage policies are opt-in; the switches of rounds 1 - 4 in other_config are refused."""
    from slotvps_amd.slot_head import MultiScaleDynamicMaskHead, MaskDynamicConv, DEFAULT_MODE
    params = synth.make_params(synth.head_shapes(), 7)
    head = build_head(cuda, params)
    assert DEFAULT_MODE == "fp16x2" and head.mode == "fp16x2" and head.precision == "fp16x2"
    assert all(m.precision == "fp16x2" for m in head.modules() if hasattr(m, "precision"))
    assert MaskDynamicConv(256).precision == "fp16x2"
    assert MultiScaleDynamicMaskHead(num_classes=20, dh_num_heads=8, merge_operation="concat", trans_in_dim=384,
                                     apply_temporal_query_atten_stages=[], other_config=dict(mode="bf16")).mode == "bf16"
    for legacy in ("precision", "map_dtype"):
        with pytest.raises(ValueError):
            MultiScaleDynamicMaskHead(num_classes=20, dh_num_heads=8, merge_operation="concat", trans_in_dim=384,
                                      apply_temporal_query_atten_stages=[], other_config={legacy: "bf16"})


def test_pixel_side_on_its_own_stream_is_bitwise_the_same(cuda):
    """forward_clip(pixel_stream=...) (round 6): level fusion and the LayerNorm statistics of every stage on a second stream, each stage's
    retriever behind its own event - same kernels, same order per tensor, so the results are bitwise those of the single-stream call
    (and the captured form of it is what clip.SlotClipRunner validates against the eager step when SVPS_OVERLAP_PIXEL_SIDE=1)."""
    import torch
    from slotvps_amd import ops
    params = synth.make_params(synth.head_shapes(), 3)
    head = build_head(cuda, params)
    T, H, W, L = 3, 64, 96, 37
    sizes = synth.level_sizes(H, W)
    g = torch.Generator(device=cuda).manual_seed(5)
    feats = [torch.randn((T, 128, h, w), generator=g, device=cuda) for (h, w) in sizes]
    slots = torch.randn((L, 256), generator=g, device=cuda)
    tabs = [ops.pos_embed_sine_tables(h, w, 256, cuda) for (h, w) in sizes]
    with torch.no_grad():
        a = head.forward_clip(feats, slots, tabs)
        side = torch.cuda.Stream(device=cuda)
        b = head.forward_clip(feats, slots, tabs, pixel_stream=side)
        torch.cuda.synchronize()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert all(torch.equal(x, y) for x, y in zip(a[2], b[2]))


@pytest.mark.parametrize("map_dtype", ["fp16x2", "bf16", "fp16"])
def test_reference_signature_roundtrip(cuda, map_dtype):
    """The reference-style list-of-frames call returns the reference's structure (with every storage of the level maps), and the maps it
    returns are the ones forward_clip computes (in fp16x2 a map is two planes - the frame is their sum, not a plane)."""
    import torch
    from slotvps_amd.position_encoding import PositionEmbeddingSine, nested_tensor_from_tensor_list
    params = synth.make_params(synth.head_shapes(), 7)
    head = build_head(cuda, params).set_mode(map_dtype)
    T, H, W, L = 2, 64, 64, 100
    feats = synth.make_clip_features(8, T, H, W)
    pe = PositionEmbeddingSine(128, normalize=True)
    features = [[torch.from_numpy(f[None]).to(cuda) for f in feats[t]] for t in range(T)]
    pos = [[pe(nested_tensor_from_tensor_list(f)) for f in features[t]] for t in range(T)]
    init = [torch.from_numpy(synth.make_slots(9, L)).to(cuda) for _ in range(T)]
    with torch.no_grad():
        logits, embeds, fused = head(features=features, init_masks=init, pad_mask=None, pos=pos, query_pos=None)
    assert len(logits) == T and tuple(logits[0].shape) == (7, 1, L, 20)
    assert tuple(embeds[1].shape) == (7, 1, L, 256)
    assert tuple(fused[0][3].shape) == (1, 256, 16, 16) and tuple(init[0].shape) == (1, L, 256)
    assert torch.isfinite(embeds[0]).all()
    from slotvps_amd.slot_head import pos_tables_from_map
    with torch.no_grad():
        tf = [torch.cat([features[t][i] for t in range(T)], 0) for i in range(4)]
        lg_c, em_c, fu_c = head.forward_clip(tf, torch.from_numpy(synth.make_slots(9, L)).to(cuda), [pos_tables_from_map(pos[0][i]) for i in range(4)])
    for t in range(T):
        assert torch.equal(embeds[t][:, 0], em_c[:, t]) and torch.equal(logits[t][:, 0], lg_c[:, t])
        for i, (h, w) in enumerate(synth.level_sizes(H, W)):
            f = fu_c[i]
            want = (f[0, t].float() + f[1, t].float()) if f.dim() == 4 else f[t]
            got = fused[t][i]
            assert tuple(got.shape) == (1, 256, h, w) and torch.equal(got[0].permute(1, 2, 0).reshape(h * w, 256).to(want.dtype), want)
    # a stage through its own reference-signature entry point (MaskRCNNHead.forward): the map is stored the way the head stores it
    stage = head.head_series_3[0]
    with torch.no_grad():
        lg, em, _, _ = stage(features=[fused[t][3] for t in range(T)], mask_query=[embeds[t][4] for t in range(T)], pad_mask=None,
                             pos=[pos[t][3] for t in range(T)])
    assert len(lg) == T and tuple(em[0].shape) == (1, L, 256) and torch.isfinite(em[0]).all()


def test_head_rejects_cpu():
    import torch
    params = synth.make_params(synth.head_shapes(), 7)
    from slotvps_amd.slot_head import MultiScaleDynamicMaskHead  # noqa: F401
    head = build_head(torch.device("cpu"), params)
    feats = [torch.zeros(1, 128, 2 * 2 ** i, 4 * 2 ** i) for i in range(4)]
    with pytest.raises(RuntimeError):
        head.forward_clip(feats, torch.zeros(100, 256), [None] * 4)


# BASELINE configs 4 and 5 as parity cases: the Swin-L head (FFN act ReLU, temporal act
# GELU - swinL_fpn_slotvps.py:41) and the VIPER geometry (24 classes, 200 slots, level sizes that are not
# multiples of the 32-pixel tile: 1088x1920 gives 34x60 ... 272x480; here the same shape family scaled down).
VARIANTS = {
    "swinL_head": dict(cfg=dict(activation="relu", temporal_activation="gelu"), T=2, L=100, sizes=[(2, 4), (4, 8), (8, 16), (16, 32)]),
    "viper_geometry": dict(cfg=dict(num_classes=24), T=3, L=200, sizes=[(3, 5), (6, 10), (12, 20), (24, 40)]),
}


@pytest.mark.parametrize("name", list(VARIANTS))
def test_head_variants_per_stage_parity(cuda, name):
    import torch
    from slotvps_amd import ops
    from slotvps_amd.slot_head import MultiScaleDynamicMaskHead
    v = VARIANTS[name]
    cfg = dict(synth.R50_HEAD_CFG, **v["cfg"])
    T, L, sizes = v["T"], v["L"], v["sizes"]
    params = synth.make_params(synth.head_shapes(cfg), 21)
    head = MultiScaleDynamicMaskHead(
        dh_dim=256, num_classes=cfg["num_classes"], dim_feedforward=cfg["dim_feedforward"], nhead=8, dropout=0.0,
        activation=cfg["activation"], dh_num_heads=7, per_dh_num_heads=[1, 2, 2, 2], feat_num_levels=4,
        merge_operation="concat", trans_in_dim=384, num_cls=2, num_reg=2,
        temporal_query_attention_config=dict(d_model=256, dim_feedforward=1024, dropout=0.0,
                                             activation=cfg["temporal_activation"], softmax_dim="slots", drop_path=0.),
        apply_temporal_query_atten_stages=[3, 4, 5, 6])
    head.load_state_dict({k: torch.from_numpy(p).reshape(head.state_dict()[k].shape) for k, p in params.items()}, strict=True)
    head.to(cuda).eval().set_mode("bf16")                          # the oracle below follows the 16-bit storage policy
    rng = np.random.default_rng(5)
    feats = [[synth.smooth_features(rng, 128, h, w) for (h, w) in sizes] for _ in range(T)]
    slots = synth.make_slots(6, L)
    with torch.no_grad():
        tf = [torch.from_numpy(np.stack([feats[t][i] for t in range(T)])).to(cuda) for i in range(4)]
        tabs = [ops.pos_embed_sine_tables(h, w, 256, cuda) for (h, w) in sizes]
        logits, embeds, fused = head.forward_clip(tf, torch.from_numpy(slots).to(cuda), tabs)
    logits, embeds = logits.cpu().numpy(), embeds.cpu().numpy()
    assert logits.shape == (7, T, L, cfg["num_classes"]) and embeds.shape == (7, T, L, 256)
    pos = [orc.pos_embed_sine(h, w) for (h, w) in sizes]
    form = "fused"                                                 # 200 slots: statistics kernel + two retriever launches
    st = orc.Storage.fused_policy() if form == "fused" else orc.Storage.bf16_policy()
    _, _, o_fused = orc.head_forward(feats, slots, pos, params, st=st)
    g_fused = [[fused[i][t].float().cpu().numpy() for i in range(4)] for t in range(T)]
    f_err = max(np.abs(g_fused[t][i] - o_fused[t][i]).max() for t in range(T) for i in range(4))
    assert f_err <= 6.3e-2, f_err                                  # one bf16 ulp at magnitude < 8 (K4)
    ocfg = dict(orc.DEFAULT_CFG, activation=cfg["activation"], temporal_activation=cfg["temporal_activation"])
    errs, sidx = [], 0
    for lvl, n in enumerate(ocfg["per_level_stages"]):
        for j in range(n):
            s_in = [slots.astype(np.float32)] * T if sidx == 0 else [embeds[sidx - 1, t] for t in range(T)]
            lg, em = orc.stage(s_in, [g_fused[t][lvl] for t in range(T)], [pos[lvl]] * T, params,
                               f"head_series_{lvl}.{j}.", sidx in ocfg["temporal_stages"], ocfg, st)
            errs.append(max(max(np.abs(embeds[sidx, t] - em[t]).max(), np.abs(logits[sidx, t] - lg[t]).max()) for t in range(T)))
            sidx += 1
    print(f"\n[{name}] fused {f_err:.2e} | per-stage (identical inputs) " + " ".join(f"{e:.1e}" for e in errs))
    assert max(errs) <= 4e-3, errs                                 # fused form (default): measured <= 9.6e-4 (Swin-L head), <= 4.8e-4 (VIPER geometry)


@pytest.mark.parametrize("mode", ["fp16x2", "bf16"])
def test_stacked_clips_equal_separate_clips(cuda, mode):
    """Several clips stacked along the frame axis of one launch (clip_frames) give each clip what it gets alone:
    all kernels are per frame, the temporal slot attention is blocked per clip."""
    import torch
    from slotvps_amd import ops
    params = synth.make_params(synth.head_shapes(), 3)
    head = build_head(cuda, params, mode)
    Tc, H, W, L = 2, 64, 128, 100
    sizes = synth.level_sizes(H, W)
    clips = [synth.make_clip_features(40 + c, Tc, H, W) for c in range(3)]
    slots = torch.from_numpy(synth.make_slots(4, L)).to(cuda)
    tabs = [ops.pos_embed_sine_tables(h, w, 256, cuda) for (h, w) in sizes]
    with torch.no_grad():
        alone = []
        for c in clips:
            tf = [torch.from_numpy(np.stack([c[t][i] for t in range(Tc)])).to(cuda) for i in range(4)]
            alone.append(head.forward_clip(tf, slots, tabs))
        tf = [torch.from_numpy(np.stack([c[t][i] for c in clips for t in range(Tc)])).to(cuda) for i in range(4)]
        lg, em, fu = head.forward_clip(tf, slots, tabs, clip_frames=Tc)
        with pytest.raises(ValueError):
            head.forward_clip(tf, slots, tabs, clip_frames=4)
    for ci, (lg1, em1, fu1) in enumerate(alone):
        sl = slice(ci * Tc, (ci + 1) * Tc)
        for i in range(4):
            assert torch.equal(frames(fu[i], sl), fu1[i])                  # K4 is per frame: bit-identical
        # K1's pixel chunking depends on the number of frames per launch (summation order of the partials), so the
        # slot side is equal up to fp32 reassociation seen through the chaotic chain: tight on the first stage
        assert (em[0, sl] - em1[0]).abs().max().item() <= 5e-4
        assert (lg[0, sl] - lg1[0]).abs().max().item() <= 5e-4
        # a clip must not see the other clips: feeding different neighbours leaves it unchanged
    with torch.no_grad():
        tf2 = [t.clone() for t in tf]
        for t in tf2:
            t[Tc:] = t[Tc:].flip(0)                                        # permute the frames of the OTHER clips
        lg2, em2, _ = head.forward_clip(tf2, slots, tabs, clip_frames=Tc)
    assert torch.equal(em2[:, :Tc], em[:, :Tc]) and torch.equal(lg2[:, :Tc], lg[:, :Tc])


def test_init_slots_follow_their_source(cuda):
    """the expanded init slots may only be cached for a model PARAMETER, keyed on its version. A temporary that
    reuses the address of the previous one, and a parameter updated in place, must both be seen."""
    import torch
    from slotvps_amd import ops
    params = synth.make_params(synth.head_shapes(), 5)
    head = build_head(cuda, params)
    T, H, W, L = 1, 32, 64, 100
    sizes = synth.level_sizes(H, W)
    c = synth.make_clip_features(50, T, H, W)
    tf = [torch.from_numpy(np.stack([c[t][i] for t in range(T)])).to(cuda) for i in range(4)]
    tabs = [ops.pos_embed_sine_tables(h, w, 256, cuda) for (h, w) in sizes]
    a = synth.make_slots(6, L)
    b = synth.make_slots(7, L)
    with torch.no_grad():
        ref_a = head.forward_clip(tf, torch.from_numpy(a).to(cuda), tabs)[1][0].clone()
        ref_b = head.forward_clip(tf, torch.from_numpy(b).to(cuda), tabs)[1][0].clone()
        assert not torch.equal(ref_a, ref_b)
        tmp = torch.from_numpy(a).to(cuda)
        got_a = head.forward_clip(tf, tmp, tabs)[1][0].clone()
        tmp.copy_(torch.from_numpy(b).to(cuda))                            # same address, new contents
        got_b = head.forward_clip(tf, tmp, tabs)[1][0].clone()
        assert torch.equal(got_a, ref_a) and torch.equal(got_b, ref_b)
        par = torch.nn.Parameter(torch.from_numpy(a).to(cuda), requires_grad=False)
        got_pa = head.forward_clip(tf, par, tabs)[1][0].clone()
        got_pa2 = head.forward_clip(tf, par, tabs)[1][0].clone()           # second call: served from the cache
        par.copy_(torch.from_numpy(b).to(cuda))                            # an optimiser step: the version moves
        got_pb = head.forward_clip(tf, par, tabs)[1][0].clone()
    assert torch.equal(got_pa, ref_a) and torch.equal(got_pa2, ref_a) and torch.equal(got_pb, ref_b)


@pytest.mark.parametrize("map_dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("tag", ["T2_64x128", "T3_64x64"])
def test_free_running_bf16_head_to_panoptic_ids(cuda, tag, map_dtype):
    """End-to-end INTEGER parity of the fast (bf16, fused-retriever) path: free-running head -> K2 -> K6 post-process -> relabel
    (vps_temporal_slots.py:284-299 -> 411-435), against the panoptic ids the ORACLE pipeline produces from the REFERENCE's own fp32
    head outputs (class logits + mask logits of the fixture, tests/golden/head_small.npz). Both sides get the same fixed
    slot -> class bias (random-init slots all predict "no object") and the same scalar gain on the mask logits (the
    fixture's fg_bn weight 0.1 leaves them in a +-0.1 range where no slot reaches the 0.4 pixel threshold).
    The bf16 storage of the fused maps moves slot embeddings by up to 6e-2 per stage (test above), so the ids cannot be bit-equal:
    the measured pixel agreement is asserted."""
    import sys
    import torch
    from util import ROOT
    sys.path.insert(0, ROOT)
    from oracle import postprocess_oracle as po
    from slotvps_amd import ops
    from slotvps_amd.slot_head import generate_final_outputs
    from slotvps_amd.postprocess import PostProcessPanopticInstances
    z = np.load(os.path.join(GOLDEN, "head_small.npz"))
    T, H, W, L, seed = (int(x) for x in z[f"{tag}_meta"])
    params = synth.make_params(synth.head_shapes(), seed)
    feats = synth.make_clip_features(seed + 1, T, H, W)
    slots = synth.make_slots(seed + 2, L)
    sizes = synth.level_sizes(H, W)
    h, w = sizes[-1]
    head = build_head(cuda, params).set_mode(map_dtype)       # 16-bit storage of the level maps: bf16 (default) or fp16
    wb, bb, mu, var = z[f"{tag}_bn"]
    fg = z[f"{tag}_fg"]
    feat_bn = torch.nn.BatchNorm2d(256).to(cuda).eval()
    fg_bn = torch.nn.BatchNorm2d(1).to(cuda).eval()
    gain = 400.0
    bias = np.zeros((L, 20), dtype=np.float32)
    bias[np.arange(L), np.arange(L) % 19] = 12.0
    cfg = dict(is_thing_map={i: i > 10 for i in range(20)}, threshold=0.85, fraction_threshold=0.03, pixel_threshold=0.4,
               apply_mask_removal=True, apply_mask_removal_only_ins=True, use_mask_low_constant=False)
    with torch.no_grad():
        feat_bn.weight.copy_(torch.from_numpy(wb)); feat_bn.bias.copy_(torch.from_numpy(bb))
        feat_bn.running_mean.copy_(torch.from_numpy(mu)); feat_bn.running_var.copy_(torch.from_numpy(var))
        fg_bn.weight.fill_(float(fg[0])); fg_bn.bias.fill_(float(fg[1]))
        fg_bn.running_mean.fill_(float(fg[2])); fg_bn.running_var.fill_(float(fg[3]))
        tf = [torch.from_numpy(np.stack([feats[t][i] for t in range(T)])).to(cuda) for i in range(4)]
        pos_tabs = [ops.pos_embed_sine_tables(hh, ww, 256, cuda) for (hh, ww) in sizes]
        logits, embeds, fused = head.forward_clip(tf, torch.from_numpy(slots).to(cuda), pos_tabs)
        masks = generate_final_outputs(fused[3], embeds[6].contiguous(), feat_bn, fg_bn)
        pp = PostProcessPanopticInstances(**cfg)
        t = T - 1
        res = pp.forward_tensors(logits[6, t] + torch.from_numpy(bias).to(cuda), (gain * masks[t]).view(L, h, w).contiguous(), (4 * h, 4 * w))
        ids, cls_inds, _ = pp.panoptic_ids(res)
        torch.cuda.synchronize()
    ids = ids.cpu().numpy().astype(np.int64).reshape(4 * h, 4 * w)
    # the reference's fp32 outputs through the oracle pipeline
    o = po.postprocess(z[f"{tag}_logits_{t}"][6] + bias, (gain * z[f"{tag}_mask"]).reshape(L, h, w), (4 * h, 4 * w))
    want_ids, want_cls, _ = po.panoptic_relabel(o["masks"], o["labels"])
    want_ids = np.asarray(want_ids).reshape(4 * h, 4 * w)
    agree = float((ids == want_ids).mean())
    # semantic agreement (class of the pixel's segment; instance numbering can permute when two scores are within rounding)
    def sem(x):
        return np.where(x >= 1000, x // 1000, x)
    agree_sem = float((sem(ids) == sem(want_ids)).mean())
    print(f"\n[{tag}] free-running head ({map_dtype} level maps) -> K2 -> K6 -> relabel vs oracle pipeline on the reference's fp32 outputs: panoptic ids equal on "
          f"{100 * agree:.2f} % of the pixels (semantic class: {100 * agree_sem:.2f} %); segments {len(cls_inds)} vs {len(want_cls)}")
    assert len(np.unique(want_ids)) > 3, "degenerate case: the reference side kept (almost) nothing"
    assert len(cls_inds) == len(want_cls)                     # the same segments survive
    # measured on MI355X: 86.2 % (T2_64x128: seven kept segments, the last stages' embeddings are 0.4 - 1.9 apart) and 98.4 % (T3_64x64)
    # fp16 level maps: 98.24 % and 99.63 % (the storage rounding of the maps is what separates the free-running head from the reference)
    bound = {"bf16": {"T2_64x128": 0.84, "T3_64x64": 0.97}, "fp16": {"T2_64x128": 0.97, "T3_64x64": 0.99}}[map_dtype][tag]
    assert agree >= bound, agree


@pytest.mark.parametrize("tag", ["T2_64x128", "T3_64x64"])
def test_fp16_level_maps_against_the_reference(cuda, tag):
    """head.set_mode("fp16"): the fused level maps (and the operands of the level-fusion conv) as fp16 instead of bf16 - the same
    bytes, three more mantissa bits. What limits the distance of the bf16 path from the REFERENCE's own fp32 outputs is the rounding
    of those maps (one bf16 ulp at magnitude 8 is 3e-2); measured here for both storages on the reference fixture: the finest fused
    map, every stage teacher-forced on the reference's embeddings, the mask logits decoded from the reference's embeddings, and the
    per-pixel slot argmax of the free-running head."""
    import torch
    from slotvps_amd import ops
    from slotvps_amd.slot_head import generate_final_outputs
    z = np.load(os.path.join(GOLDEN, "head_small.npz"))
    T, H, W, L, seed = (int(x) for x in z[f"{tag}_meta"])
    params = synth.make_params(synth.head_shapes(), seed)
    feats = synth.make_clip_features(seed + 1, T, H, W)
    slots = synth.make_slots(seed + 2, L)
    sizes = synth.level_sizes(H, W)
    cfg = dict(orc.DEFAULT_CFG)
    w, b, mu, var = z[f"{tag}_bn"]
    fg = z[f"{tag}_fg"]
    feat_bn = torch.nn.BatchNorm2d(256).to(cuda).eval()
    fg_bn = torch.nn.BatchNorm2d(1).to(cuda).eval()
    with torch.no_grad():
        feat_bn.weight.copy_(torch.from_numpy(w)); feat_bn.bias.copy_(torch.from_numpy(b))
        feat_bn.running_mean.copy_(torch.from_numpy(mu)); feat_bn.running_var.copy_(torch.from_numpy(var))
        fg_bn.weight.fill_(float(fg[0])); fg_bn.bias.fill_(float(fg[1]))
        fg_bn.running_mean.fill_(float(fg[2])); fg_bn.running_var.fill_(float(fg[3]))
    res = {}
    for md in ("bf16", "fp16"):
        head = build_head(cuda, params).set_mode(md)
        with torch.no_grad():
            tf = [torch.from_numpy(np.stack([feats[t][i] for t in range(T)])).to(cuda) for i in range(4)]
            pos_tabs = [ops.pos_embed_sine_tables(h, w_, 256, cuda) for (h, w_) in sizes]
            logits, embeds, fused = head.forward_clip(tf, torch.from_numpy(slots).to(cuda), pos_tabs)
            assert fused[3].dtype == torch.float16          # fp16 values, or (bf16 policy) bf16 values in the fp16 encoding
            if md == "bf16":
                assert torch.equal(fused[3].float(), fused[3].to(torch.bfloat16).float())
            r_f3 = max(np.abs(fused[3][t].float().cpu().numpy() - z[f"{tag}_fused3_{t}"]).max() for t in range(T))
            tf_err, sidx = [], 0
            for lvl, n in enumerate(cfg["per_level_stages"]):
                h, w_ = sizes[lvl]
                for j in range(n):
                    s_in = np.stack([slots.astype(np.float32) if sidx == 0 else z[f"{tag}_embeds_{t}"][sidx - 1] for t in range(T)])
                    stage = getattr(head, f"head_series_{lvl}")[j]
                    _, em = stage.forward_pm(torch.from_numpy(s_in).to(cuda), fused[lvl], (h, w_), pos_tabs[lvl], sidx in cfg["temporal_stages"], 1)
                    tf_err.append(max(np.abs(em[t].cpu().numpy() - z[f"{tag}_embeds_{t}"][sidx]).max() for t in range(T)))
                    sidx += 1
            emb_ref = torch.from_numpy(np.stack([z[f"{tag}_embeds_{t}"][6] for t in range(T)])).to(cuda)
            m_tf, a_tf = generate_final_outputs(fused[3], emb_ref, feat_bn, fg_bn, want_argmax=True)
            _, a_free = generate_final_outputs(fused[3], embeds[6].contiguous(), feat_bn, fg_bn, want_argmax=True)
        m_ref = z[f"{tag}_mask"]
        res[md] = dict(map=r_f3, tf=max(tf_err), mask=float(np.abs(m_tf[T - 1].cpu().numpy() - m_ref).max()),
                       same=float((a_tf[T - 1].cpu().numpy() == np.argmax(m_ref, axis=0)).mean()),
                       free=float((a_free[T - 1].cpu().numpy() == np.argmax(m_ref, axis=0)).mean()))
        print(f"\n[{tag}/{md} maps] vs the reference's fp32 outputs: finest fused map {r_f3:.2e}; teacher-forced stages " + " ".join(f"{e:.1e}" for e in tf_err)
              + f"; mask logits {res[md]['mask']:.2e}, slot argmax equal {100 * res[md]['same']:.2f} % (free-running head {100 * res[md]['free']:.2f} %)")
    b16, f16 = res["bf16"], res["fp16"]
    assert f16["map"] <= 0.25 * b16["map"] and f16["tf"] <= 0.4 * b16["tf"] and f16["mask"] <= 0.3 * b16["mask"], (b16, f16)
    # ABSOLUTE bounds (measured on MI355X: T2_64x128 1.03e-4 / 100 % / 96.9 %, T3_64x64 7.7e-5 / 100 % / 98.4 %): with fp16 level maps the mask logits decoded from the reference's embeddings sit AT the north star's
    # 1e-4 (one fp16 rounding of the map, 3.4e-3 at |f| < 8, through the normalised dot product) and their slot argmax is the reference's on
    # every pixel; the free-running head stays at 97 - 98 %: 16-bit maps cannot do better, the form that meets 1e-4 / 100 % free-running is
    # precision "fp16x2" (tests/test_refprec_gpu.py)
    for md in ("fp16",):
        assert res[md]["mask"] <= 1.2e-4 and res[md]["same"] == 1.0 and res[md]["free"] >= 0.96, (md, res[md])
    assert res["fp16"]["map"] <= 4e-3 and res["fp16"]["tf"] <= 1.2e-2


def test_map_dtype_switch_is_checked(cuda):
    """set_mode accepts the names of MultiScaleDynamicMaskHead.MODES only; the bf16 policy's two encodings of the level maps agree."""
    import torch
    from slotvps_amd import ops
    params = synth.make_params(synth.head_shapes(), 7)
    head = build_head(cuda, params)
    assert head.mode == "fp16x2"                                   # the product's default
    with pytest.raises(ValueError):
        head.set_mode("fp8")
    head.set_mode("fp16")
    assert head.mode == "fp16" and all(m.map_dtype == "fp16" for m in head.modules() if hasattr(m, "precision"))
    T, H, W, L = 1, 64, 64, 100
    feats = synth.make_clip_features(8, T, H, W)
    tf = [torch.from_numpy(np.stack([feats[t][i] for t in range(T)])).to(cuda) for i in range(4)]
    tabs = [ops.pos_embed_sine_tables(h, w, 256, cuda) for (h, w) in synth.level_sizes(H, W)]
    with torch.no_grad():
        logits, embeds, fused = head.forward_clip(tf, torch.from_numpy(synth.make_slots(9, L)).to(cuda), tabs)
    assert all(f.dtype == torch.float16 for f in fused) and torch.isfinite(embeds).all()
    # the bf16 policy: bf16 VALUES; in the fp16 encoding by default (no conversion pass in the consumers), as bf16 tensors on request -
    # the same values and the same result
    head.set_mode("bf16")
    with torch.no_grad():
        lg_c, em_c, fused_c = head.forward_clip(tf, torch.from_numpy(synth.make_slots(9, L)).to(cuda), tabs)
        head.map_encoding = "bf16"
        lg_b, em_b, fused_b = head.forward_clip(tf, torch.from_numpy(synth.make_slots(9, L)).to(cuda), tabs)
        head.map_encoding = "auto"
    assert all(f.dtype == torch.float16 for f in fused_c) and all(f.dtype == torch.bfloat16 for f in fused_b)
    for fc, fb in zip(fused_c, fused_b):
        assert torch.equal(fc.float(), fc.to(torch.bfloat16).float())                 # bf16 values
        big = fb.float().abs() >= 2.0 ** -13                                              # (below that fp16 is subnormal)
        assert torch.equal(fc.float()[big], fb.float()[big])
    print(f"\nbf16 policy, fp16 encoding against bf16 tensors: slot embeddings differ by {float((em_c - em_b).abs().max()):.2e}")
    assert float((em_c - em_b).abs().max()) <= 1e-3


@pytest.mark.parametrize("map_dtype", ["bf16", "fp16"])
def test_clip_runner_from_the_tower_rows(cuda, map_dtype):
    """SlotClipRunner(input_form="tower16") - the bench's default step since round 4: the semantic tower's 16-bit pixel-major rows in,
    conv_trans folded into K4's weights - against the same runner fed the reference's tensors x = conv_trans(y) (fp32 NCHW, computed by
    the framework): graph replay == eager (run() validates), fused maps equal to a few operand ulps, slot argmax equal almost everywhere."""
    import torch
    import torch.nn.functional as F
    from slotvps_amd.clip import SlotClipRunner
    kw = dict(T=2, H=64, W=128, L=100, param_seed=3, use_graph=True, clips_per_launch=2)
    ra = SlotClipRunner(cuda, input_form="tower16", **kw)
    rb = SlotClipRunner(cuda, input_form="nchw_f32", **kw)
    for r in (ra, rb):
        r.head.set_mode(map_dtype)
    rows = ra.random_clip(5)
    ra.load_clip(rows)
    wt, bt = ra.pre_linear
    with torch.no_grad():
        rb.load_clip([F.conv2d(y.float().transpose(1, 2).reshape(y.shape[0], 128, h, w), wt, bt) for y, (h, w) in zip(ra.slots_feats[0], ra.sizes)])
    assert ra.slots_feats[0][0].dtype == (torch.float16 if map_dtype == "fp16" else torch.bfloat16) and ra.slots_feats[0][0].shape == (4, 2 * 4, 128)
    oa, ob = ra.run(), rb.run()
    torch.cuda.synchronize()
    fa = ra.head.forward_clip(ra.slots_feats[0], ra.init_slots, ra.pos_tabs, hws=ra.sizes, clip_frames=2, pre_linear=ra.pre_linear)[2]
    fb = rb.head.forward_clip(rb.slots_feats[0], rb.init_slots, rb.pos_tabs, hws=rb.sizes, clip_frames=2)[2]
    ulp = 2.0 ** -7 if map_dtype == "bf16" else 2.0 ** -10
    for a, b in zip(fa, fb):
        scale = b.float().abs().max().item()
        d = (a.float() - b.float()).abs()
        print(f"[{map_dtype}] fused map {tuple(a.shape)}: max {d.max().item() / (ulp * scale):.2f}, mean {d.mean().item() / (ulp * scale):.3f} operand ulps of the map's scale")
        assert d.max().item() <= 2 * ulp * scale and d.mean().item() <= 0.2 * ulp * scale      # measured 0.81 / 0.09
    assert torch.isfinite(oa["mask_logits"]).all() and oa["mask_logits"].shape == ob["mask_logits"].shape
    same = (oa["slot_argmax"] == ob["slot_argmax"]).float().mean().item()
    print(f"[{map_dtype}] slot argmax equal on {100 * same:.2f} % of the pixels")
    # (the free-running head amplifies one-ulp differences of the maps: bf16 maps 89.8 %, fp16 maps 98.1 % - the class of
    # SAME_FREE_MIN above, which compares with the reference's own outputs)
    assert same >= (0.85 if map_dtype == "bf16" else 0.96)
