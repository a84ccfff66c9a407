"""The torch-CPU restatement that bench.py times as `cpu_baseline` equals the (reference-pinned) NumPy oracle."""
import numpy as np
import torch

from util import orc
from oracle.torch_cpu_head import TorchCpuHead
from slotvps_amd import synth


def test_torch_cpu_head_equals_numpy_oracle():
    params = synth.make_params(synth.head_shapes(), 5)
    sizes = synth.level_sizes(64, 128)
    pos = [orc.pos_embed_sine(h, w) for (h, w) in sizes]
    rng = np.random.default_rng(2)
    T = 2
    feats = [[rng.standard_normal((128, h, w)).astype(np.float32) for (h, w) in sizes] for _ in range(T)]
    slots = synth.make_slots(3, 100)
    lg0, em0, fu0 = orc.head_forward(feats, slots, pos, params, dt=np.float64)
    head = TorchCpuHead(params)
    lg1, em1, fu1 = head.forward([[torch.from_numpy(f) for f in fr] for fr in feats], torch.from_numpy(slots),
                                 [torch.from_numpy(p) for p in pos])
    for t in range(T):
        assert np.abs(fu1[t][3].numpy() - fu0[t][3]).max() < 1e-4
        assert np.abs(em1[t][0].numpy() - em0[t][0]).max() < 1e-3          # stage 0
        assert np.abs(em1[t][-1].numpy() - em0[t][-1]).max() < 5e-2        # stage 6: fp32 noise grows ~5x per stage
        assert np.abs(lg1[t][0].numpy() - lg0[t][0]).max() < 1e-3
    scale, shift = orc.bn_eval_affine(np.ones(256, np.float32), np.zeros(256, np.float32), np.zeros(256, np.float32),
                                      np.ones(256, np.float32))
    m0 = orc.mask_decode(fu0[0][3], em0[0][-1], scale, shift, 0.1, 0.0)
    m1 = TorchCpuHead.mask_decode(torch.from_numpy(fu0[0][3]).float(), torch.from_numpy(em0[0][-1]).float(),
                                  torch.from_numpy(scale), torch.from_numpy(shift), 0.1, 0.0)
    assert np.abs(m1.numpy() - m0).max() < 1e-5
