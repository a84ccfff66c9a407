import numpy as np
import pytest

from util import orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,W", [(16, 32), (33, 65), (34, 60), (256, 512)])
def test_pos_embed_matches_oracle(cuda, H, W):
    import torch
    from slotvps_amd import ops
    got = ops.pos_embed_sine(H, W, 256, cuda)
    torch.cuda.synchronize()
    ref = orc.pos_embed_sine(H, W, 256)
    err = np.abs(got.cpu().numpy() - ref).max()
    assert err < 5e-6, f"max abs err {err:.3e}"   # device powf/sinf vs libm: a few ulp on args <= 2*pi
