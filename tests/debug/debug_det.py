import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_detector import _make_detector
from oracle import postprocess_oracle as porc
dev = torch.device("cuda:0")
det = _make_detector(dev)
T, H, W = 3, 128, 256
imgs = torch.randn(T, 3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
logits, embeds, masks, fcn = det.slot_path(imgs)
for t in range(T):
    want = porc.postprocess(logits[t].cpu().numpy(), masks[t].cpu().numpy(), (H, W), threshold=0.3)
    res = det.postprocess_panoptic.forward_tensors(logits[t], masks[t], (H, W), materialize_masks=True)
    print(t, "slot_index", res.slot_index.tolist() == want["slot_index"].tolist(), res.slot_index.tolist(), want["slot_index"].tolist())
    print("labels", res.labels.tolist(), want["labels"].tolist())
    print("area", res.area, want["area"])
    if res.masks.shape == want["masks"].shape:
        print("masks maxdiff", np.abs(res.masks.cpu().numpy() - want["masks"]).max())
    pan, cls_inds, sem = porc.panoptic_relabel(want["masks"], want["labels"])
    g, ci, _ = det.postprocess_panoptic.panoptic_ids(res, 11)
    g = g.cpu().numpy()
    d = g != pan
    print("pan mismatch", d.sum(), np.unique(g[d]), np.unique(pan[d]), np.unique(g), np.unique(pan))
