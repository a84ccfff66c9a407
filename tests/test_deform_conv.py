"""K7 (deformable conv forward): oracle properties on CPU, HIP-vs-oracle parity on the GPU."""
import sys

import numpy as np
import pytest

from util import orc


def _torch_conv(x, w, pad):
    import torch
    return torch.nn.functional.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), padding=pad)[0].numpy()


def test_oracle_zero_offset_is_plain_convolution():
    rng = np.random.default_rng(0)
    x, w = rng.standard_normal((6, 7, 9)), rng.standard_normal((4, 6, 3, 3))
    assert np.abs(orc.deform_conv(x, np.zeros((18, 7, 9)), w, 1, 1, 1) - _torch_conv(x, w, 1)).max() < 1e-12


def test_oracle_integer_offset_is_shifted_convolution():
    rng = np.random.default_rng(1)
    x, w = rng.standard_normal((4, 8, 8)), rng.standard_normal((3, 4, 3, 3))
    off = np.zeros((18, 8, 8))
    off[0::2] = 1.0                                  # dy = +1 on every tap: rows y .. y+2 instead of y-1 .. y+1
    got = orc.deform_conv(x, off, w, 1, 1, 1)
    xs = np.zeros_like(x)
    xs[:, :-1] = x[:, 1:]
    want = _torch_conv(xs, w, 1)
    assert np.abs(got - want)[:, 1:].max() < 1e-12   # row 0 differs by construction (zero padding vs a real row)


def test_oracle_half_pixel_offset_is_average_of_neighbours():
    x = np.arange(5 * 6, dtype=np.float64).reshape(1, 5, 6)
    w = np.zeros((1, 1, 1, 1)); w[0, 0, 0, 0] = 1.0
    off = np.zeros((2, 5, 6)); off[1] = 0.5          # dx = 0.5, 1x1 kernel
    got = orc.deform_conv(x, off, w, 1, 0, 1)[0]
    want = 0.5 * x[0] + 0.5 * np.concatenate([x[0][:, 1:], np.zeros((5, 1))], axis=1)
    assert np.abs(got - want).max() < 1e-12


# ---- hand-computed vectors for the border rules of deform_conv_cuda_kernel.cu:82-114 (bilinear taps, corners outside the image
# read as zero) and :224 (a sample is taken only for h in the OPEN interval (-1, H), w in (-1, W)). Image I[h][w] = 1 .. 9 on a
# 3 x 3 grid, a single centre tap of weight 1, so every output pixel is the sample at (y + dy, x + dx) of its own offset.
# Worked by hand from the published formula, not produced by any implementation:
#   pixel (0,0) -> (-0.5,  0.0): rows -1 | 0, lh = 0.5; row -1 is outside -> 0.5 * I[0][0]                     = 0.5
#   pixel (0,1) -> (-1.0,  1.0): h = -1 is not > -1 -> no sample                                                = 0
#   pixel (0,2) -> ( 2.5,  2.5): rows 2 | 3, cols 2 | 3; only (2,2) inside -> 0.5 * 0.5 * I[2][2]               = 2.25
#   pixel (1,0) -> ( 3.0,  1.0): h = 3 is not < 3 -> no sample                                                  = 0
#   pixel (1,1) -> (-0.25,-0.75): rows -1 | 0 (lh = .75), cols -1 | 0 (lw = .25); only (0,0) -> .75 * .25 * 1    = 0.1875
#   pixel (1,2) -> ( 0.5,  1.25): .5*.75*I[0][1] + .5*.25*I[0][2] + .5*.75*I[1][1] + .5*.25*I[1][2]             = 3.75
#   pixel (2,0) -> ( 2.0, -0.5): row 2 exactly (lh = 0), cols -1 | 0 (lw = .5) -> 1 * .5 * I[2][0]               = 3.5
#   pixel (2,1) -> ( 1.0,  2.999..): not used (no exact binary value); instead (1.0, 2.75): cols 2 | 3, lw = .75 -> .25 * I[1][2] = 1.5
#   pixel (2,2) -> ( 2.0,  2.0): the pixel itself                                                               = 9
_HAND_TARGETS = [[(-0.5, 0.0), (-1.0, 1.0), (2.5, 2.5)], [(3.0, 1.0), (-0.25, -0.75), (0.5, 1.25)], [(2.0, -0.5), (1.0, 2.75), (2.0, 2.0)]]
_HAND_EXPECT = np.array([[0.5, 0.0, 2.25], [0.0, 0.1875, 3.75], [3.5, 1.5, 9.0]])


def _hand_case(C, O, kernel=3):
    x = np.zeros((C, 3, 3), dtype=np.float64)
    x[0] = np.arange(1, 10, dtype=np.float64).reshape(3, 3)
    w = np.zeros((O, C, kernel, kernel), dtype=np.float64)
    w[0, 0, kernel // 2, kernel // 2] = 1.0
    off = np.zeros((2 * kernel * kernel, 3, 3), dtype=np.float64)
    t = (kernel // 2) * kernel + kernel // 2                      # centre tap: samples at (y + dy, x + dx) with padding = kernel // 2
    for y in range(3):
        for xx in range(3):
            off[2 * t, y, xx] = _HAND_TARGETS[y][xx][0] - y
            off[2 * t + 1, y, xx] = _HAND_TARGETS[y][xx][1] - xx
    return x, off, w


def test_oracle_border_rules_match_hand_computed_vectors():
    for kernel in (1, 3):
        x, off, w = _hand_case(2, 2, kernel)
        got = orc.deform_conv(x, off, w, 1, kernel // 2, 1)
        assert np.array_equal(got[0], _HAND_EXPECT), got[0]
        assert not got[1].any()


@pytest.mark.gpu
def test_hip_deform_conv_border_rules_match_hand_computed_vectors(cuda):
    """Both HIP forms (K7 column buffer + GEMM, K7' fused) on the hand-computed border vectors: every value is a short dyadic
    sum, so the fp32 results are exact."""
    import torch
    from slotvps_amd.dcn import DeformConv, deform_conv
    x, off, w = _hand_case(64, 128)
    tx, to, tw = (torch.from_numpy(a.astype(np.float32)).to(cuda) for a in (x[None], off[None], w))
    out = deform_conv(tx, to, tw, 1, 1, 1, 1, 1)
    m = DeformConv(64, 128, 3, padding=1).to(cuda)
    with torch.no_grad():
        m.weight.copy_(tw)
        assert m.fused
        out_f = m(tx, to)
    torch.cuda.synchronize()
    for o in (out, out_f):
        got = o[0].cpu().numpy()
        assert np.array_equal(got[0], _HAND_EXPECT.astype(np.float32)), got[0]
        assert not got[1:].any()


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,O,H,W,dg", [(1, 16, 8, 9, 11, 1), (2, 256, 128, 16, 32, 1), (1, 32, 16, 12, 10, 2)])
def test_hip_deform_conv_matches_oracle(cuda, N, C, O, H, W, dg):
    import torch
    from slotvps_amd.dcn import deform_conv
    rng = np.random.default_rng(C + H)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((O, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
    off = (2.5 * rng.standard_normal((N, dg * 18, H, W))).astype(np.float32)     # reaches outside the image
    out = deform_conv(torch.from_numpy(x).to(cuda), torch.from_numpy(off).to(cuda), torch.from_numpy(w).to(cuda),
                      1, 1, 1, 1, dg)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    for n in range(N):
        ref = orc.deform_conv(x[n].astype(np.float64), off[n].astype(np.float64), w.astype(np.float64), 1, 1, 1, dg)
        assert np.abs(out[n] - ref).max() < 2e-5     # fp32 sampling + fp32 GEMM vs float64


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,O,H,W,scale", [(2, 256, 256, 16, 32, 2.5), (1, 256, 128, 33, 20, 2.5), (2, 128, 128, 9, 7, 6.0),
                                             (1, 64, 128, 40, 40, 0.0),
                                             # K7'' (the staged form): offsets inside the staged region (|offset| <= 2) on most / some / all tiles,
                                             # sizes that are no multiple of the 8 x 16 tile
                                             (2, 256, 128, 24, 40, 0.5), (1, 128, 128, 33, 20, 0.7), (1, 64, 128, 17, 50, 1.0), (1, 128, 256, 8, 16, 0.3)])
def test_fused_deform_conv_without_column_buffer_matches_oracle(cuda, N, C, O, H, W, scale):
    """K7' (csrc/deform_conv_fused.hip): one kernel, no column buffer, split-bf16 products with fp32 accumulation, against
    the float64 oracle on the same fp32 input: fp32-class tolerance (the im2col + fp32 GEMM path holds 2e-5). Offsets reach
    far outside the image (scale 6), ragged tiles (pixel counts no multiple of 128), and zero offsets == conv2d."""
    import torch
    from slotvps_amd.dcn import DeformConv
    rng = np.random.default_rng(C + O + H)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    m = DeformConv(C, O, 3, padding=1).to(cuda)
    w = (rng.standard_normal((O, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy(w))
    off = (scale * rng.standard_normal((N, 18, H, W))).astype(np.float32)
    with torch.no_grad():
        assert m.fused and not m.bf16_operands
        out = m(torch.from_numpy(x).to(cuda), torch.from_numpy(off).to(cuda))
        for _ in range(3):                                                      # bitwise reproducible from run to run
            assert torch.equal(out, m(torch.from_numpy(x).to(cuda), torch.from_numpy(off).to(cuda)))
        m.fused = False
        out_im2col = m(torch.from_numpy(x).to(cuda), torch.from_numpy(off).to(cuda))
    torch.cuda.synchronize()
    assert out.shape == (N, O, H, W)
    out, out_im2col = out.cpu().numpy(), out_im2col.cpu().numpy()
    worst = 0.0
    for n in range(N):
        ref = orc.deform_conv(x[n].astype(np.float64), off[n].astype(np.float64), w.astype(np.float64), 1, 1, 1, 1)
        worst = max(worst, float(np.abs(out[n] - ref).max()))
    print(f"\nK7' C={C} O={O} {H}x{W}: max abs err vs float64 oracle {worst:.2e} (outputs of order 1); "
          f"vs the im2col + fp32 GEMM path {np.abs(out - out_im2col).max():.2e}")
    assert worst < 3e-5
    if scale == 0.0:
        want = torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(w), padding=1).numpy()
        assert np.abs(out - want).max() < 3e-5


@pytest.mark.gpu
def test_fused_deform_conv_full_size_properties(cuda):
    """At the P2 size of BASELINE config 1 (256 x 512, 256 -> 256 channels): zero offsets == conv2d (the reference's
    initialisation of the offset conv), random offsets: equal to the column-buffer path and bitwise reproducible."""
    import torch
    from slotvps_amd.dcn import DeformConv
    g = torch.Generator(device=cuda).manual_seed(4)
    m = DeformConv(256, 256, 3, padding=1).to(cuda)
    x = torch.randn((1, 256, 256, 512), generator=g, device=cuda)
    with torch.no_grad():
        zero = torch.zeros((1, 18, 256, 512), device=cuda)
        got = m(x, zero)
        want = torch.nn.functional.conv2d(x, m.weight, padding=1)
        assert (got - want).abs().max().item() < 1e-4
        off = 3.0 * torch.randn((1, 18, 256, 512), generator=g, device=cuda)
        a = m(x, off)
        assert torch.equal(a, m(x, off)) and torch.equal(a, m(x, off))
        m.fused = False
        b = m(x, off)
        assert (a - b).abs().max().item() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,O,H,W,dg", [(2, 256, 128, 16, 32, 1), (1, 32, 16, 12, 10, 2), (1, 128, 128, 33, 20, 1)])
def test_hip_deform_conv_bf16_operands_match_oracle(cuda, N, C, O, H, W, dg):
    """bf16 operand path: the oracle (float64) on the same bf16-rounded input and weight; what is left is the bf16
    rounding of the blended samples: 9*C terms of size ~ |x| |w| 2^-9 ~ 0.8 * (9C)^-1/2 * 2e-3 with random signs, i.e. a
    standard deviation of ~1.6e-3 on outputs of order 1 (max over 1e5 outputs ~ 5 sigma), plus fp32 accumulation."""
    import torch
    from slotvps_amd.dcn import deform_conv
    rng = np.random.default_rng(C + H + 1)
    x = orc.round_bf16(rng.standard_normal((N, C, H, W)).astype(np.float32))
    w = orc.round_bf16((rng.standard_normal((O, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32))
    off = (2.5 * rng.standard_normal((N, dg * 18, H, W))).astype(np.float32)
    out = deform_conv(torch.from_numpy(x).to(cuda), torch.from_numpy(off).to(cuda), torch.from_numpy(w).to(cuda),
                      1, 1, 1, 1, dg, bf16_operands=True)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    for n in range(N):
        ref = orc.deform_conv(x[n].astype(np.float64), off[n].astype(np.float64), w.astype(np.float64), 1, 1, 1, dg)
        err = np.abs(out[n] - ref)
        assert err.max() < 1.2e-2 and err.mean() < 2e-3, (err.max(), err.mean())


@pytest.mark.gpu
def test_hip_deform_conv_with_offset_module_zero_init_is_conv(cuda):
    """The offset conv is zero-initialised (deform_conv_with_offset.py:25-26): a fresh module == conv2d."""
    import torch
    from slotvps_amd.dcn import DeformConvWithOffset
    m = DeformConvWithOffset(32, 16, kernel_size=3, padding=1).to(cuda)
    x = torch.randn(2, 32, 10, 14, device=cuda)
    with torch.no_grad():
        m.conv.bf16_operands = False
        got = m(x)
        want = torch.nn.functional.conv2d(x, m.conv.weight, padding=1)
        assert (got - want).abs().max().item() < 1e-5
        m.conv.bf16_operands = True           # default: bf16 operands == conv2d of the bf16-rounded input and weight
        got = m(x)
        want = torch.nn.functional.conv2d(x.bfloat16().float(), m.conv.weight.bfloat16().float(), padding=1)
        assert (got - want).abs().max().item() < 1e-4


@pytest.mark.gpu
def test_semantic_tower_matches_reference_structure(cuda):
    """UPSNetFPN (three deformable convolutions + GroupNorm + ReLU per level, x2/x4/x8 upsampling, prediction conv) against
    tests/golden/semantic_tower.npz: the REFERENCE's module run with the oracle's deformable convolution in place of its
    CUDA-only op (tests/golden/make_golden_backbone.py), offsets non-zero. fp32 operands: 1e-3; bf16 operands: storage tolerance."""
    import os
    import torch
    from util import GOLDEN, ROOT
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden_backbone import UPS, seeded_state
    from slotvps_amd.backbones import UPSNetFPN
    from slotvps_amd.dcn import DeformConv

    z = np.load(os.path.join(GOLDEN, "semantic_tower.npz"))
    tower = UPSNetFPN(**UPS).eval()
    st = seeded_state(tower, 6)
    for k in st:
        if "conv_offset" in k:
            st[k] = st[k] * 0.3
    tower.load_state_dict(st)
    tower = tower.cuda()
    g = torch.Generator().manual_seed(7)
    lv = [torch.randn(1, 32, 16 >> i, 24 >> i, generator=g).cuda() for i in range(4)]
    for bf16, tol_max, tol_mean in ((False, 1e-3, 1e-4), (True, 0.15, 0.02)):
        for m in tower.modules():
            if isinstance(m, DeformConv):
                m.bf16_operands = bf16
        with torch.no_grad():
            up, score, feats = tower(lv)
        torch.cuda.synchronize()
        pairs = [(up, z["up"]), (score, z["score"])] + [(f, z[f"feat{i}"]) for i, f in enumerate(feats)]
        for got, ref in pairs:
            d = np.abs(got.float().cpu().numpy() - ref)
            scale = max(1.0, np.abs(ref).max())
            assert got.shape == ref.shape and d.max() <= tol_max * scale and d.mean() <= tol_mean * scale, (bf16, d.max(), d.mean())


@pytest.mark.gpu
@pytest.mark.parametrize("N,HW,C,groups", [(2, 1000, 256, 32), (1, 4096, 128, 32), (3, 77, 32, 32), (1, 131, 64, 8)])
def test_group_norm_relu_pixel_major(cuda, N, HW, C, groups):
    """csrc/gn_relu.hip against torch's GroupNorm + ReLU (the modules of the reference's tower, upsnetFPN.py:36-49) on the same data:
    pixel-major result and its NCHW copy, ragged pixel counts."""
    import torch
    from slotvps_amd import ops
    g = torch.Generator(device=cuda).manual_seed(N + HW + C)
    x = 2.0 * torch.randn((N, HW, C), generator=g, device=cuda) + 0.7
    gamma = torch.rand(C, generator=g, device=cuda) + 0.5
    beta = 0.3 * torch.randn(C, generator=g, device=cuda)
    ref = torch.relu(torch.nn.functional.group_norm(x.double().transpose(1, 2), groups, gamma.double(), beta.double(), 1e-5))   # [N, C, HW]
    y, yn = ops.group_norm_relu_pm(x, gamma, beta, groups, 1e-5, want_nchw=True)
    y2, none = ops.group_norm_relu_pm(x, gamma, beta, groups, 1e-5)
    torch.cuda.synchronize()
    assert none is None and torch.equal(y, y2)
    assert torch.equal(yn, y.transpose(1, 2).contiguous())
    assert float((yn.double() - ref).abs().max()) < 2e-5


@pytest.mark.gpu
def test_pixel_major_tower_equals_module_sequence(cuda):
    """UPSNetFPN._tower with the pixel-major path (K7' -> GroupNorm + ReLU kernel, no layout copies) against the module sequence
    (K7' per layer, torch GroupNorm / ReLU) on a 256 -> 256 -> 128 -> 128 tower, non-zero offsets."""
    import torch
    from slotvps_amd.backbones import UPSNetFPN
    torch.manual_seed(0)
    tower = UPSNetFPN(in_channels=256, out_channels=128, num_levels=4, num_things_classes=8, num_classes=19, ignore_label=255,
                      loss_weight=1.0).cuda().eval()
    with torch.no_grad():
        for m in tower.modules():
            if hasattr(m, "conv_offset"):
                m.conv_offset.weight.normal_(0, 0.02)
                m.conv_offset.bias.normal_(0, 0.5)
            if isinstance(m, torch.nn.GroupNorm):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
    x = torch.randn(2, 256, 24, 40, device=cuda)
    with torch.no_grad():
        tower.fuse_norm = True
        a, none16 = tower._tower(x)
        tower.fuse_norm = False
        b, _ = tower._tower(x)
        tower.fuse_norm = True
        outs = {}
        for dt in (torch.bfloat16, torch.float16):                 # the last layer's 16-bit pixel-major rows (what K4 reads when
            tower.emit_pm16 = dt                                   # conv_trans is folded): the same values, rounded once
            outs[dt] = tower._tower(x)
        tower.emit_pm16 = None
    torch.cuda.synchronize()
    assert none16 is None
    assert a.shape == b.shape == (2, 128, 24, 40)
    assert float((a - b).abs().max()) < 2e-4 * max(1.0, float(b.abs().max()))
    for dt, (nchw, y16) in outs.items():
        assert float((nchw - a).abs().max()) < 1e-4 * max(1.0, float(a.abs().max()))   # (the offset convs are not run-to-run identical)
        assert y16.dtype == dt and y16.shape == (2, 24 * 40, 128)
        assert torch.equal(y16, nchw.permute(0, 2, 3, 1).reshape(2, -1, 128).to(dt))   # the same values of the same run, rounded once


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,H,W,O", [(2, 256, 24, 40, 256), (1, 128, 17, 23, 128), (3, 64, 8, 16, 128)])
def test_fused_kernel_group_norm_statistics(cuda, N, C, H, W, O):
    """K7' with the GroupNorm statistics in its epilogue (svps_deform_conv_fused_stats_fwd): the same output bits, the per-channel
    partial sums add up to the sums of the result (ragged last tile: HW not a multiple of 128), and GroupNorm + ReLU from those
    statistics equals GroupNorm + ReLU with its own moments pass to fp32 rounding; nchw_to_pixel_major == permute + contiguous."""
    import torch
    from slotvps_amd import ops
    from slotvps_amd.dcn import deform_conv_fused_pm, pack_weight_fragments
    g = torch.Generator(device=cuda).manual_seed(N + C + H)
    x = torch.randn((N, C, H, W), generator=g, device=cuda)
    xp = ops.nchw_to_pixel_major(x)
    assert torch.equal(xp, x.permute(0, 2, 3, 1).contiguous())
    off = 0.7 * torch.randn((N, 18, H, W), generator=g, device=cuda)
    wgt = torch.randn((O, C, 3, 3), generator=g, device=cuda) / (3 * C ** 0.5)
    wp = pack_weight_fragments(wgt)
    a = deform_conv_fused_pm(xp, off, wp, O, 1, 1, 1)
    b, (part, chunks) = deform_conv_fused_pm(xp, off, wp, O, 1, 1, 1, gn_stats=True)
    assert torch.equal(a, b) and part.shape == (N, chunks, 2, O)
    s1, s2 = part[:, :, 0].double().sum(1), part[:, :, 1].double().sum(1)
    assert (s1 - a.double().sum(1)).abs().max().item() <= 1e-4 * a.abs().sum(1).max().item()
    assert (s2 - (a.double() ** 2).sum(1)).abs().max().item() <= 1e-5 * (a.double() ** 2).sum(1).max().item()
    gamma = torch.rand((O,), generator=g, device=cuda) + 0.5
    beta = 0.2 * torch.randn((O,), generator=g, device=cuda)
    y0, n0 = ops.group_norm_relu_pm(a, gamma, beta, 32, 1e-5, want_nchw=True)
    y1, n1 = ops.group_norm_relu_pm(a, gamma, beta, 32, 1e-5, want_nchw=True, stats=(part, chunks))
    assert (y0 - y1).abs().max().item() <= 2e-6 * max(1.0, y0.abs().max().item()) and (n0 - n1).abs().max().item() <= 2e-6 * max(1.0, y0.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,K,H,W", [(2, 128, 19, 32, 64), (1, 64, 24, 16, 40), (1, 128, 30, 24, 40), (1, 128, 19, 256, 512)])
def test_semantic_prediction_layer_in_one_kernel(cuda, N, C, K, H, W):
    """csrc/semantic_pred.hip against the framework's three bilinear upsamplings + concatenation + 1x1 convolution
    (upsnetFPN.py forward) on the same inputs: the class scores to fp32 rounding of a 4 C-term sum, their argmax on all but a handful
    of pixels (near-ties; the framework's convolution has its own summation order too)."""
    import torch
    import torch.nn.functional as F
    from slotvps_amd import ops
    g = torch.Generator(device=cuda).manual_seed(N + C + H)
    px = [torch.randn((N, C, H >> i, W >> i), generator=g, device=cuda) for i in range(4)]
    wgt = torch.randn((K, 4 * C, 1, 1), generator=g, device=cuda) / (4 * C) ** 0.5
    bias = 0.1 * torch.randn((K,), generator=g, device=cuda)
    got = ops.semantic_pred(px, wgt, bias)
    ups = [px[0]] + [F.interpolate(px[i], None, 2 ** i, mode="bilinear", align_corners=False) for i in (1, 2, 3)]
    cat = torch.cat(ups, dim=1)
    want = F.conv2d(cat, wgt, bias)
    ref64 = F.conv2d(cat.double(), wgt.double(), bias.double())
    e_got, e_fw = (got.double() - ref64).abs().max().item(), (want.double() - ref64).abs().max().item()
    same = (got.argmax(1) == want.argmax(1)).double().mean().item()
    print(f"\nsemantic_pred N={N} C={C} K={K} {H}x{W}: {e_got:.2e} against float64 (the framework's convolution: {e_fw:.2e}), argmax equal on {100 * same:.4f} %")
    assert e_got <= 5e-6 and e_got <= 4 * e_fw + 1e-6 and same >= 0.9999
    assert torch.equal(got, ops.semantic_pred(px, wgt, bias))


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,H,W", [(2, 256, 24, 40), (1, 128, 17, 23), (1, 64, 5, 7), (1, 256, 64, 128)])
def test_offset_convolution_on_pixel_major_rows(cuda, N, C, H, W):
    """csrc/offset_conv.hip (the conv_offset of DeformConvWithOffset on the tower's pixel-major activations, split-bf16 on the matrix
    cores) against a float64 convolution: fp32-class, like the framework's fp32 convolution; ragged sizes and the zero padding."""
    import torch
    import torch.nn.functional as F
    from slotvps_amd import ops
    g = torch.Generator(device=cuda).manual_seed(N + C + H)
    x = torch.randn((N, C, H, W), generator=g, device=cuda)
    wgt = torch.randn((18, C, 3, 3), generator=g, device=cuda) / (3 * C ** 0.5)
    bias = 0.5 * torch.randn((18,), generator=g, device=cuda)
    got = ops.conv3x3_pm_small(x.permute(0, 2, 3, 1).contiguous(), ops.pack_conv3x3_small(wgt), bias, 18)
    ref = F.conv2d(x.double(), wgt.double(), bias.double(), padding=1)
    fw = F.conv2d(x, wgt, bias, padding=1)
    e_got, e_fw = (got.double() - ref).abs().max().item(), (fw.double() - ref).abs().max().item()
    print(f"\noffset conv N={N} C={C} {H}x{W}: {e_got:.2e} against float64 (the framework's fp32 convolution {e_fw:.2e})")
    assert got.shape == (N, 18, H, W) and e_got <= 4e-5          # split-bf16: 16 bits of mantissa over 9 C terms (K7' itself: 2.2e-5)
    assert torch.equal(got, ops.conv3x3_pm_small(x.permute(0, 2, 3, 1).contiguous(), ops.pack_conv3x3_small(wgt), bias, 18))
