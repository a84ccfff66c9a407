"""Deterministic synthetic parameters and inputs for the slot head (shared by tests/golden/make_golden.py,
the tests, bench.py and smoke()). Everything derives from NumPy Generators with fixed seeds, so a
golden fixture only has to carry the seeds and the reference's outputs.

Parameter names and shapes follow the reference's state_dict for MultiScaleDynamicMaskHead
(SURVEY.md 8b 'Checkpoint key contract'); values are deliberately non-trivial (LayerNorm affines
away from (1, 0), non-zero biases) so that every term of every formula is exercised.
"""
import numpy as np

D = 256
R50_HEAD_CFG = dict(dh_dim=256, num_classes=20, dim_feedforward=2048, nhead=8, activation="gelu",
                    per_dh_num_heads=[1, 2, 2, 2], feat_num_levels=4, trans_in_dim=384, num_cls=2, num_reg=2,
                    temporal_dim_feedforward=1024, temporal_activation="relu",
                    apply_temporal_query_atten_stages=[3, 4, 5, 6])


def retriever_shapes(prefix):
    s = {}
    for n in ("to_q", "to_k", "to_v"):
        s[f"{prefix}{n}.weight"] = (D, D)
        s[f"{prefix}{n}.bias"] = (D,)
    for n in ("norm_q", "norm_k", "norm_v", "norm1"):
        s[f"{prefix}{n}.weight"] = (D,)
        s[f"{prefix}{n}.bias"] = (D,)
    return s


def temporal_shapes(prefix, ff):
    s = retriever_shapes(prefix + "inst_interact.")
    s[prefix + "linear1.weight"] = (ff, D)
    s[prefix + "linear1.bias"] = (ff,)
    s[prefix + "linear2.weight"] = (D, ff)
    s[prefix + "linear2.bias"] = (D,)
    for n in ("norm1", "norm2", "norm3"):
        s[f"{prefix}{n}.weight"] = (D,)
        s[f"{prefix}{n}.bias"] = (D,)
    return s


def stage_shapes(prefix, cfg, temporal):
    ff, nc = cfg["dim_feedforward"], cfg["num_classes"]
    s = {
        prefix + "self_attn.in_proj_weight": (3 * D, D),
        prefix + "self_attn.in_proj_bias": (3 * D,),
        prefix + "self_attn.out_proj.weight": (D, D),
        prefix + "self_attn.out_proj.bias": (D,),
        prefix + "linear1.weight": (ff, D),
        prefix + "linear1.bias": (ff,),
        prefix + "linear2.weight": (D, ff),
        prefix + "linear2.bias": (D,),
        prefix + "class_logits.weight": (nc, D),
        prefix + "class_logits.bias": (nc,),
    }
    s.update(retriever_shapes(prefix + "inst_interact."))
    for n in ("norm1", "norm2", "norm3"):
        s[f"{prefix}{n}.weight"] = (D,)
        s[f"{prefix}{n}.bias"] = (D,)
    for tower, cnt in (("cls_module", cfg["num_cls"]), ("reg_module", cfg["num_reg"])):
        for i in range(cnt):
            s[f"{prefix}{tower}.{3 * i}.weight"] = (D, D)
            s[f"{prefix}{tower}.{3 * i + 1}.weight"] = (D,)
            s[f"{prefix}{tower}.{3 * i + 1}.bias"] = (D,)
    if temporal:
        s.update(temporal_shapes(prefix + "temporal_query_head.", cfg["temporal_dim_feedforward"]))
    return s


def head_shapes(cfg=None):
    cfg = cfg or R50_HEAD_CFG
    s = {"conv_trans.conv.weight": (D, cfg["trans_in_dim"], 1, 1), "conv_trans.conv.bias": (D,)}
    idx = 0
    for lvl, n in enumerate(cfg["per_dh_num_heads"]):
        # the reference attaches the temporal sub-head to every stage of a level whose FIRST stage
        # index is a temporal stage (dynamic_mask_head.py:83-106)
        temporal = idx in cfg["apply_temporal_query_atten_stages"]
        for j in range(n):
            s.update(stage_shapes(f"head_series_{lvl}.{j}.", cfg, temporal))
        idx += n
    return s


def make_params(shapes, seed):
    """name -> float32 array. Matrices xavier-uniform, LayerNorm weights U(0.5, 1.5), biases N(0, 0.1)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name in sorted(shapes):
        shp = shapes[name]
        if len(shp) >= 2:
            fan_out, fan_in = shp[0], int(np.prod(shp[1:]))
            a = np.sqrt(6.0 / (fan_in + fan_out))
            out[name] = rng.uniform(-a, a, shp).astype(np.float32)
        elif name.endswith("weight"):          # LayerNorm scale
            out[name] = rng.uniform(0.5, 1.5, shp).astype(np.float32)
        else:
            out[name] = (0.1 * rng.standard_normal(shp)).astype(np.float32)
    return out


def level_sizes(H, W, nlev=4):
    """Coarse -> fine feature sizes for an image of H x W padded to /32 (strides 32, 16, 8, 4)."""
    assert H % 32 == 0 and W % 32 == 0
    return [(H // s, W // s) for s in (32, 16, 8, 4)][:nlev]


def smooth_features(rng, C, h, w):
    """[C, h, w] float32 with image-like spatial correlation: bilinear-ish blend of coarse noise + detail."""
    ch, cw = max(h // 4, 1), max(w // 4, 1)
    coarse = rng.standard_normal((C, ch, cw)).astype(np.float32)
    up = np.repeat(np.repeat(coarse, -(-h // ch), axis=1), -(-w // cw), axis=2)[:, :h, :w]
    return (0.8 * up + 0.6 * rng.standard_normal((C, h, w)).astype(np.float32)).astype(np.float32)


def make_clip_features(seed, T, H, W, C=128):
    """features[t][lvl] : [C, Hi, Wi] float32, coarse -> fine (the conv_trans'ed UPSNetFPN maps)."""
    rng = np.random.default_rng(seed)
    return [[smooth_features(rng, C, h, w) for (h, w) in level_sizes(H, W)] for _ in range(T)]


def make_slots(seed, L):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((L, D)).astype(np.float32)


def make_post_case(seed, L=100, h=16, w=32, num_classes=20, n_keep=18):
    """Synthetic head outputs for the panoptic post-process: class_logits [L, nc] with `n_keep` confident
    slots (stuff classes incl. duplicates, thing classes incl. heavily overlapping same-class pairs and
    tiny blobs) and smooth blob-shaped mask logits [L, h, w]."""
    rng = np.random.default_rng(seed)
    logits = rng.standard_normal((L, num_classes)).astype(np.float32)
    logits[:, num_classes - 1] += 6.0                                   # default: "no object"
    chosen = rng.choice(L, size=n_keep, replace=False)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    masks = (0.3 * rng.standard_normal((L, h, w)) - 1.0).astype(np.float32)
    centers = rng.uniform([0, 0], [h, w], size=(n_keep, 2))
    for n, s in enumerate(chosen):
        if n < 6:
            c = [0, 1, 2, 2, 8, 10][n]                                  # stuff, one duplicated class
            sig = rng.uniform(0.35, 0.6) * max(h, w)
        else:
            c = int(rng.integers(11, num_classes - 1))                  # things
            sig = rng.uniform(0.05, 0.2) * max(h, w)
            if n % 4 == 0:                                              # same class + nearly same place as the previous thing
                c = prev_c
                centers[n] = centers[n - 1] + rng.normal(0, 0.5, 2)
            if n % 7 == 6:
                sig = 0.35                                              # sub-pixel blob: dies in the small-area filter
        prev_c = c
        logits[s] = rng.standard_normal(num_classes).astype(np.float32) * 0.3
        logits[s, c] += rng.uniform(6.0, 9.0)
        d2 = (yy - centers[n, 0]) ** 2 + (xx - centers[n, 1]) ** 2
        amp = rng.uniform(1.5, 3.0) if n < 6 else rng.uniform(5.0, 9.0)     # things stand out of the stuff
        masks[s] = (amp * np.exp(-d2 / (2 * sig * sig)) - rng.uniform(0.3, 1.2)
                    + 0.15 * rng.standard_normal((h, w))).astype(np.float32)
    return logits, masks


def make_harness_case(seed, n_frames=3, H=96, W=160):
    """Synthetic frames exercising every branch: agreeing / out-voted / disagreeing instances, a frame without
    instances, duplicate tracked ids, small stuff areas, an unlabeled (255) region."""
    rng = np.random.default_rng(seed)
    segs, pans, cls_inds, obj_ids, names = [], [], [], [], []
    for f in range(n_frames):
        seg = np.zeros((H, W), np.uint8)
        pan = np.zeros((H, W), np.uint8)
        for _ in range(6):                                   # stuff layout shared by both branches
            c = int(rng.integers(0, 11))
            y, x = int(rng.integers(0, H - 8)), int(rng.integers(0, W - 8))
            h, w = int(rng.integers(4, H // 2)), int(rng.integers(4, W // 2))
            seg[y:y + h, x:x + w] = c
            pan[y:y + h, x:x + w] = c
        n_ins = 0 if f == 1 else int(rng.integers(3, 7))
        cls = rng.integers(1, 9, size=n_ins).astype(np.int64)
        for k in range(n_ins):
            y, x = int(rng.integers(0, H - 12)), int(rng.integers(0, W - 12))
            h, w = int(rng.integers(6, 30)), int(rng.integers(6, 40))
            pan[y:y + h, x:x + w] = 11 + k
            mode = k % 3
            if mode == 0:
                seg[y:y + h, x:x + w] = 10 + cls[k]           # semantic branch agrees
            elif mode == 1:
                seg[y:y + h, x:x + w] = int(rng.integers(0, 11))   # a stuff class out-votes the instance
            else:
                seg[y:y + h, x:x + w // 2] = 10 + (cls[k] % 8) + 1  # another thing class on part of it
        present = [k for k in range(n_ins) if (pan == 11 + k).any()]
        cls = cls[present]
        remap = {11 + k: 11 + j for j, k in enumerate(present)}
        pan2 = pan.copy()
        for a, b in remap.items():
            pan2[pan == a] = b
        if f == 1:
            pan2[5:15, 5:25] = 13                              # instance ids without any cls_ind -> 255
        oid = rng.integers(0, 6, size=len(cls)).astype(np.int32)   # duplicates likely
        segs.append(seg); pans.append(pan2); cls_inds.append(cls); obj_ids.append(oid); names.append(f"frame_{seed}_{f}.png")
    return segs, pans, cls_inds, obj_ids, names


def make_simple_test_case(seed, n_frames=4, L=100, h=16, w=32, num_classes=20, embed_dim=256):
    """Head outputs of an `n_frames` video for the test-time flow after the head (post-process, relabel, tracker):
    per frame class logits [L, nc] and mask logits [L, h, w] (make_post_case: the surviving slots change from frame to
    frame), slot embeddings [L, D] that drift slowly (slot l of frame f resembles slot l of frame f-1, so the tracker
    re-identifies most instances, meets new ones and has contested matches), semantic logits [19, 4h, 4w]; and the
    tracker's two fully connected layers."""
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((L, embed_dim)).astype(np.float32)
    frames = []
    for f in range(n_frames):
        logits, masks = make_post_case(seed * 10 + f // 2, L, h, w, num_classes, 16 + 2 * (f % 2))
        if f % 2 == 1:                                       # odd frames: same scene as the frame before, perturbed
            logits = (logits + 0.3 * rng.standard_normal(logits.shape)).astype(np.float32)
            masks = (masks + 0.2 * rng.standard_normal(masks.shape)).astype(np.float32)
        embed = (base + 0.15 * (f + 1) * rng.standard_normal(base.shape)).astype(np.float32)
        if f >= 1:                                           # two slots collapse onto one older identity: contested match
            embed[(7 * f) % L] = frames[-1]["embed"][(7 * f + 1) % L] * 1.05
        fcn = rng.standard_normal((num_classes - 1, 4 * h, 4 * w)).astype(np.float32)
        frames.append(dict(logits=logits, masks=masks, embed=embed, fcn=fcn))
    fc_w = [(0.2 * rng.standard_normal((embed_dim, embed_dim))).astype(np.float32) for _ in range(2)]
    fc_b = [(0.1 * rng.standard_normal(embed_dim)).astype(np.float32) for _ in range(2)]
    return frames, fc_w, fc_b


def make_decode_case(seed, L=100, h=12, w=20, D=256):
    """Inputs of generate_final_outputs: finest fused map [D, h, w], slot embeddings [L, D], eval-mode BatchNorm
    parameters (weight, bias, running_mean, running_var) of feat_bn [D] and fg_bn [1]."""
    rng = np.random.default_rng(seed)
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    return dict(feat=f32(rng.standard_normal((D, h, w))), embed=f32(rng.standard_normal((L, D))),
                feat_bn=(f32(rng.uniform(0.5, 1.5, D)), f32(0.1 * rng.standard_normal(D)), f32(0.2 * rng.standard_normal(D)),
                         f32(rng.uniform(0.5, 2.0, D))),
                fg_bn=(f32([0.1]), f32([0.03]), f32([0.2]), f32([1.7])))


def temper_queries(params, tau):
    """Scale the query LayerNorm (inst_interact.norm_q.{weight,bias}) of every slot<->pixel retriever by `tau` (in place).
    make_params gives LayerNorm gains U(0.5, 1.5) on 256-wide rows, i.e. UNSCALED logits q.k (dynamic_mask_head.py:435) with
    sigma ~ 16: each of the seven stages then amplifies a perturbation of its incoming slots 2 - 4 x, and the reference's OWN
    fp32 result sits 5e-4 ... 1.5e-3 (mask logits) from its float64 evaluation - no implementation can be held to 1e-4 free-running
    on such weights, the reference under another summation order included. tau = 0.25 (logit sigma ~ 4) keeps the chain near
    contractive (reference fp32 vs float64: ~2e-5): the regime in which the north star's free-running tolerance is decidable.
    The full-size fixture (tests/golden/make_golden_full.py) holds both regimes and records the reference's own floor for each."""
    for k in params:
        if "inst_interact.norm_q." in k and "temporal_query_head" not in k:
            params[k] = (params[k] * np.float32(tau)).astype(np.float32)
    return params


def make_feat_bn(seed, D=256):
    """Eval-mode BatchNorm parameters of the decode (feat_bn [4, D]: weight, bias, running_mean, running_var; fg_bn [4])."""
    rng = np.random.default_rng(seed)
    bn = np.stack([rng.uniform(0.5, 1.5, D), 0.1 * rng.standard_normal(D), 0.2 * rng.standard_normal(D),
                   rng.uniform(0.5, 2.0, D)]).astype(np.float32)
    return bn, np.array([0.1, 0.03, 0.2, 1.7], dtype=np.float32)


FULL_SIZE_MASK_GAIN = 400.0        # the decode's fg_bn keeps the synthetic mask logits within +-0.33: scaled so that the post-process sees decisive masks


def full_size_class_bias(L, nc):
    """[L, nc] bias added to the last stage's class logits in the full-size integer-target test: every third slot becomes a confident
    segment of class (slot % (nc - 1)) - stuff and things, duplicated stuff classes included -, every other slot a confident "no object"."""
    b = np.zeros((L, nc), dtype=np.float32)
    b[:, nc - 1] = 12.0
    for l in range(0, L, 3):
        b[l, nc - 1] = 0.0
        b[l, l % (nc - 1)] = 12.0
    return b
