"""CPU oracle for the consumers of the mask decode: the panoptic post-process and the
per-pixel id map.  TEST INFRASTRUCTURE ONLY - same rules as slotvps_oracle.py.

NumPy restatement of
  * PostProcessPanopticInstances.forward / mask_removal / get_ids_area
    (mmdet/models/detectors/vps_temporal_slots.py:564-657, :659-807), configuration of
    configs/cityscapes/r50_fpn_slotvps.py:68-76 (threshold 0.85, pixel_threshold 0.4,
    fraction_threshold 0.03, apply_mask_removal + only_ins, filter_small_option '4')
  * the stuff-first reorder, per-pixel argmax and id relabel of simple_test (:411-435)

Parity status: PINNED for the class above - tests/golden/make_golden_post.py imports the reference's
vps_temporal_slots.py in the build container (non-arithmetic stand-ins for mmcv / registry / sibling
modules, Tensor.cuda() made a no-op) and records its outputs in tests/golden/postprocess.npz.
One dependency is RESTATED, not imported: panopticapi.utils.id2rgb / rgb2id (un-vendored, unpinned git
install, README.md:13 of the reference) - the published base-256 pack / unpack; together with the
same-size PIL NEAREST resize at :745-751 it is the identity for ids < 2^24, which is how it is
modelled here. The relabel (:411-435) and the tracker assignment (:328-409) of simple_test are PINNED too:
tests/golden/make_golden_flow.py runs the reference's own simple_test on a four-frame synthetic video (detector
object without its constructor, canned tensors upstream of the head outputs) and tests/test_simple_test_golden.py
checks the pipeline postprocess -> panoptic_relabel -> track_assign against every frame's result dict, bit for bit.
"""
import numpy as np

STUFF_NUM_CITYSCAPES = 11


def softmax(x, axis):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def bilinear_resize(m, size):
    """F.interpolate(m[:, None], size, mode='bilinear') (align_corners=False) for m [K, h, w] -> [K, H, W],
    float32 arithmetic in torch's operation order (vps_temporal_slots.py:697-698 via misc.interpolate)."""
    m = np.asarray(m, dtype=np.float32)
    K, h, w = m.shape
    H, W = size

    def taps(n_out, n_in):
        scale = np.float32(n_in) / np.float32(n_out)
        src = np.maximum(scale * (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) - np.float32(0.5), np.float32(0))
        i0 = np.minimum(src.astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l1 = (src - i0.astype(np.float32)).astype(np.float32)
        return i0, i1, (np.float32(1) - l1).astype(np.float32), l1

    y0, y1, hy0, hy1 = taps(H, h)
    x0, x1, wx0, wx1 = taps(W, w)
    top = m[:, y0][:, :, x0] * wx0 + m[:, y0][:, :, x1] * wx1
    bot = m[:, y1][:, :, x0] * wx0 + m[:, y1][:, :, x1] * wx1
    return (hy0[None, :, None] * top + hy1[None, :, None] * bot).astype(np.float32)


def select_slots(class_logits, threshold=0.85, num_classes=20):
    """:684-691 -> (scores [L], classes [L], keep [L] bool)."""
    p = softmax(np.asarray(class_logits, dtype=np.float32), axis=-1)
    scores, classes = p.max(-1), p.argmax(-1)
    if class_logits.shape[-1] == num_classes - 1:
        keep = scores > threshold
    else:
        keep = (classes != class_logits.shape[-1] - 1) & (scores > threshold)
    return scores, classes, keep


def mask_removal(cls_prob, mask_logits, cls_idx, pixel_threshold=0.4, fraction_threshold=0.03,
                 num_stuff=STUFF_NUM_CITYSCAPES, only_ins=True, low_constant=False):
    """:564-657. cls_prob [K], mask_logits [K, H, W], cls_idx [K] -> (probs, masks, classes, keep_inds).
    Overlap filter among instances ("things"), processed by descending class score; every kept thing keeps
    its logits only on the pixels it is the first to claim."""
    mask_prob = softmax(mask_logits.astype(np.float32), axis=0)
    K = len(cls_prob)
    im_shape = mask_logits.shape[1:]
    mask_image = np.zeros((int(np.max(cls_idx)) + 1,) + im_shape, dtype=np.float32)
    panoptic_image = np.zeros(im_shape, dtype=np.float32)
    # the reference: np.argsort(cls_prob)[::-1] (:580) - an unstable sort, whose order among EQUAL scores differs between numpy's AVX-512
    # and scalar builds; pinned here (and in the product) to the scalar path's: ties in descending slot order
    order = np.argsort(cls_prob, kind="stable")[::-1]
    cls_prob, cls_idx = cls_prob[order], cls_idx[order]
    mask_prob, mask_copy = mask_prob[order], mask_logits[order]
    keep_prob, keep_idx, keep_mask, keep_inds = [], [], [], []
    stuff = []
    if only_ins:
        for i in range(K):
            if cls_idx[i] <= num_stuff - 1:
                stuff.append(i)
                keep_prob.append(cls_prob[i]); keep_idx.append(cls_idx[i]); keep_mask.append(mask_copy[i]); keep_inds.append(order[i])
    for i in range(K):
        if only_ins and i in stuff:
            continue
        logit = (mask_prob[i] >= pixel_threshold).astype(np.float32)
        mask_sum = logit.sum()
        cur = mask_image[cls_idx[i]]
        if logit.max() == logit.min() or mask_sum == 0 or \
                (np.logical_and(cur >= 1, logit == 1).sum() / mask_sum > fraction_threshold):
            continue
        assign = np.logical_and(panoptic_image == 0, logit == 1)
        keep_prob.append(cls_prob[i]); keep_idx.append(cls_idx[i])
        empty = np.full(im_shape, -99999.0 if low_constant else 0.0, dtype=np.float32)
        empty[assign] = mask_copy[i][assign]
        keep_mask.append(empty)
        panoptic_image[assign] = 1
        mask_image[cls_idx[i]] += assign.astype(np.float32)
        keep_inds.append(order[i])
    if not keep_prob:
        raise ValueError("mask_removal: nothing kept (the reference's np.stack fails here too, :652)")
    return np.stack(keep_prob), np.stack(keep_mask), np.stack(keep_idx), [int(i) for i in keep_inds]


def ids_area(masks, n, classes, is_thing, dedup):
    """get_ids_area :724-757 -> (area list, m_id [H, W]). id2rgb/rgb2id + same-size NEAREST resize = identity."""
    K, H, W = masks.shape
    if K == 0:
        m_id = np.zeros((H, W), dtype=np.int64)
    else:
        m_id = np.argmax(masks, axis=0)                 # argmax of softmax == argmax of logits, first max wins
    if dedup:
        equiv = {}
        for k, lab in enumerate(classes):
            if not is_thing(int(lab)):
                equiv.setdefault(int(lab), []).append(k)
        for ids in equiv.values():
            if len(ids) > 1:
                for e in ids:
                    m_id[m_id == e] = ids[0]
    return [int((m_id == i).sum()) for i in range(n)], m_id


def postprocess(class_logits, mask_logits_lowres, size, threshold=0.85, pixel_threshold=0.4,
                fraction_threshold=0.03, num_classes=20, num_stuff=STUFF_NUM_CITYSCAPES,
                apply_mask_removal=True, only_ins=True, filter_small_option="4"):
    """PostProcessPanopticInstances.forward :659-807 for one frame.
    class_logits [L, nc], mask_logits_lowres [L, h, w], size (H, W).
    Returns dict(slot_index [K''] into the L slots, probs, labels, masks [K'', H, W])."""
    is_thing = lambda c: c > num_stuff - 1                            # is_thing_map = {i: i > 10}
    scores, classes, keep = select_slots(class_logits, threshold, num_classes)
    slot_index = np.nonzero(keep)[0]
    cur_scores, cur_classes = scores[keep], classes[keep]
    cur_masks = bilinear_resize(mask_logits_lowres[keep], size)       # :697-698
    if apply_mask_removal:
        cur_scores, cur_masks, cur_classes, keep_inds = mask_removal(cur_scores, cur_masks, cur_classes,
                                                                     pixel_threshold, fraction_threshold, num_stuff, only_ins)
        slot_index = slot_index[keep_inds]
    area, _ = ids_area(cur_masks, len(cur_scores), cur_classes, is_thing, dedup=True)
    if len(cur_classes) > 0:
        while True:
            if filter_small_option == "4":
                small = np.array([a <= 4 for a in area], dtype=bool)
            elif filter_small_option == "4_256":
                small = np.array([a < 256 if is_thing(int(c)) else a < 4 for a, c in zip(area, cur_classes)], dtype=bool)
            elif filter_small_option == "4096_256":
                small = np.array([a < 4096 if not is_thing(int(c)) else a < 256 for a, c in zip(area, cur_classes)], dtype=bool)
            else:
                raise AssertionError("filter_small_option is not valid")
            if small.any():
                cur_scores, cur_classes, cur_masks = cur_scores[~small], cur_classes[~small], cur_masks[~small]
                slot_index = slot_index[~small]
                area, _ = ids_area(cur_masks, len(cur_scores), cur_classes, is_thing, dedup=False)
            else:
                break
    return dict(slot_index=slot_index, probs=cur_scores, labels=cur_classes, masks=cur_masks, area=area)


def panoptic_relabel(masks, labels, stuff_num=STUFF_NUM_CITYSCAPES):
    """simple_test :411-435: stuff-first reorder, per-pixel argmax, id relabel. masks [K, H, W], labels [K].
    Returns (panoptic_output [H, W] int64, cls_inds, semantic_labels after reorder)."""
    labels = np.asarray(labels)
    panoptic_num = len(labels)
    ins = labels > stuff_num - 1
    cls_inds = labels[ins] - (stuff_num - 1)
    instance_num = len(cls_inds)
    masks = np.concatenate([masks[~ins], masks[ins]], axis=0)
    sem = np.concatenate([labels[~ins], labels[ins]], axis=0)
    pan = np.argmax(masks, axis=0)
    ids = np.unique(pan)
    out = np.zeros(pan.shape, dtype=np.int64)                          # zeros * (-1) == 0  (:422-423)
    count = instance_num
    for i in range(len(ids) - 1, -1, -1):
        oid = ids[i]
        region = pan == oid
        if oid >= panoptic_num - instance_num:
            out[region] = stuff_num + count - 1                        # instance ids counted downward (:429-430)
            count -= 1
        else:
            out[region] = sem[i]                                       # position in unique(), not the id (:433)
    return out, cls_inds, sem


def track_assign(cur_embed, prev_embed, fc_w, fc_b):
    """Tracker step of simple_test (vps_temporal_slots.py:345-409) with SimpleTrackHead.forward
    (simple_track_head.py:58-92) for num_fcs_query = len(fc_w) layers, test_only_save_main_results=True (the
    memory holds output embeddings only, :32-37). Pinned by tests/golden/simple_test.npz (the reference's own
    simple_test, see the module header). cur_embed [K, D], prev_embed [P, D] ->
    (det_obj_ids [K] over ALL segments, updated memory [P', D])."""
    def fcs(x):
        for i, (w, b) in enumerate(zip(fc_w, fc_b)):
            x = x @ w.T + b
            if i < len(fc_w) - 1:
                x = np.maximum(x, 0)
        return x
    prod = fcs(cur_embed.astype(np.float32)) @ fcs(prev_embed.astype(np.float32)).T
    score = np.concatenate([np.zeros((prod.shape[0], 1), np.float32), prod], axis=1)
    z = score - score.max(axis=1, keepdims=True)
    logprob = z - np.log(np.exp(z).sum(axis=1, keepdims=True))
    likelihood, match_ids = logprob.max(axis=1), logprob.argmax(axis=1)
    memory = [e for e in prev_embed]
    det = -np.ones(len(match_ids), dtype=np.int32)
    best_scores = -100.0 * np.ones(len(prev_embed))
    best_ids = -np.ones(len(prev_embed), dtype=np.int32)
    for idx, mid in enumerate(match_ids):
        if mid == 0:
            det[idx] = len(memory)
            memory.append(cur_embed[idx])
        else:
            obj = mid - 1
            if likelihood[idx] > best_scores[obj]:
                det[idx] = obj
                if best_ids[obj] >= 0:
                    det[best_ids[obj]] = -1
                best_scores[obj] = likelihood[idx]
                best_ids[obj] = idx
                memory[obj] = cur_embed[idx]
    for idx in range(len(det)):
        if det[idx] < 0:
            det[idx] = len(memory)
            memory.append(cur_embed[idx])
    return det, np.stack(memory)
