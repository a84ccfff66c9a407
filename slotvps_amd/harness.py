"""Evaluation-loop glue of the reference: `single_gpu_test` (tools/test_vpq.py:23-59), its
clip-batched variant, and the semantic/panoptic reconciliation `get_unified_pan_result`
(tools/dataset/cityscapes_vps.py:215-303) that turns the detector's result dict into the 3-channel
(semantic class, per-frame instance index, tracked object id + 1) map the VPQ evaluation reads.
Integer / byte work on the host; inputs are the uint8 maps the detector's GPU post-process (K6) produced."""
import numpy as np
import torch

from .parallel import host_pools


def _append(pano_results, result, filename):
    pano_results["all_ssegs"].append(result["fcn_outputs"].data.cpu().numpy()[0].astype(np.uint8))
    pano_results["all_panos"].append(result["panoptic_outputs"].data.cpu().numpy()[0].astype(np.uint8))
    pano_results["all_pano_cls_inds"].append(result["panoptic_cls_inds"].data.cpu().numpy())
    pano_results["all_names"].append(filename)
    if "panoptic_det_obj_ids" in result and pano_results["all_pano_obj_ids"] is not None:
        pano_results["all_pano_obj_ids"].append(result["panoptic_det_obj_ids"].data.cpu().numpy())
    else:
        pano_results["all_pano_obj_ids"] = None


def _empty():
    return {"all_names": [], "all_ssegs": [], "all_panos": [], "all_pano_cls_inds": [], "all_pano_obj_ids": []}


def _meta_of(entry):
    meta = entry[0]
    if hasattr(meta, "data"):                    # mmcv DataContainer
        meta = meta.data[0][0]
    elif isinstance(meta, (list, tuple)):
        meta = meta[0]
    return meta


def single_gpu_test(model, data_loader, show=False, size_pools=True):
    """One detector call per frame, like the reference: `data` = dict(img=[Tensor], img_meta=[...], ref_img=[Tensor]).
    size_pools: host thread pools sized to the cgroup's CPU share for the duration of the loop (parallel.host_pools: why), the caller's
    settings restored afterwards."""
    import contextlib
    model.eval()
    pano_results = _empty()
    with (host_pools() if size_pools else contextlib.nullcontext()):
        for data in data_loader:
            filename = _meta_of(data["img_meta"])["filename"].split("/")[-1]
            with torch.no_grad():
                result = model(return_loss=False, rescale=not show, **data)
            _append(pano_results, result, filename)
    return pano_results


def clip_gpu_test(model, clips, size_pools=True):
    """Clip-batched loop: `clips` yields (imgs [T, 3, H, W], [T metas]) of consecutive frames of one video; the
    backbone runs once per frame (the reference recomputes the reference frame at every step, :245-252)."""
    import contextlib
    model.eval()
    pano_results = _empty()
    with (host_pools() if size_pools else contextlib.nullcontext()):
        for imgs, metas in clips:
            for result, meta in zip(model.clip_test(imgs, metas), metas):
                _append(pano_results, result, meta["filename"].split("/")[-1])
    return pano_results


def get_unified_pan_result(segs, pans, cls_inds, obj_ids=None, stuff_area_limit=4 * 64 * 64, names=None,
                           num_seg_classes=19, num_classes=9):
    """segs / pans: per-frame uint8 maps [H, W] (semantic argmax, panoptic ids: stuff class ids <= id_last_stuff,
    instances id_last_stuff + 1 + k); cls_inds: per frame the 1-based thing class of instance k; obj_ids: per frame
    the tracked id of instance k. Returns {name: uint8 [H, W, 3]}. `max_oid` is shared by all frames (:220)."""
    if obj_ids is None:
        obj_ids = [None] * len(cls_inds)
    id_last_stuff = num_seg_classes - num_classes            # 10 for Cityscapes-VPS
    out = {}
    max_oid = 100
    for seg, pan, cls_ind, obj_id, name in zip(segs, pans, cls_inds, obj_ids, names):
        if obj_id is not None:
            # an id that occurs several times in a frame stays on its LAST occurrence; the earlier ones get fresh ids
            # >= 100, handed out back to front (:233-243: the reference edits a reversed copy and reverses it back)
            uniq, cnt = np.unique(obj_id, return_counts=True)
            rev = obj_id[::-1].copy()
            if np.any(cnt > 1):
                for red in uniq[cnt > 1]:
                    part = obj_id[obj_id == red]
                    for i in range(1, len(part)):
                        part[i] = max_oid
                        max_oid += 1
                    rev[rev == red] = part
                obj_id = rev[::-1]
        pan_seg = pan.copy()
        if len(cls_ind) == 0:
            pan[pan > id_last_stuff] = 255                  # in place, like the reference (:251-252)
        pan_ins = pan.copy()
        pan_obj = pan.copy()
        ids = np.unique(pan)
        ids_ins = ids[ids > id_last_stuff]
        pan_ins[pan_ins <= id_last_stuff] = 0
        for idx, sid in enumerate(ids_ins):
            region = pan_ins == sid
            if sid == 255:
                pan_seg[region] = 255
                pan_ins[region] = 0
                continue
            cls, cnt = np.unique(seg[region], return_counts=True)
            thing_cls = cls_ind[sid - id_last_stuff - 1] + id_last_stuff
            major = cls[np.argmax(cnt)]
            if major != thing_cls and np.max(cnt) / np.sum(cnt) >= 0.5 and major <= id_last_stuff:
                pan_seg[region] = major                     # the semantic branch out-votes the instance with a stuff class
                pan_ins[region] = 0
                pan_obj[region] = 0
            else:
                pan_seg[region] = thing_cls
                pan_ins[region] = idx + 1
                if obj_id is not None:
                    pan_obj[region] = obj_id[idx] + 1
        for c in np.unique(pan_seg):
            if c <= id_last_stuff:
                area = pan_seg == c
                if area.sum() < stuff_area_limit:
                    pan_seg[area] = 255
        pan_2ch = np.zeros(pan.shape + (3,), dtype=np.uint8)
        pan_2ch[:, :, 0] = pan_seg
        pan_2ch[:, :, 1] = pan_ins
        pan_2ch[:, :, 2] = pan_obj
        out[name] = pan_2ch
    return out
