"""GPU panoptic post-process: host-side mirror of the reference's
PostProcessPanopticInstances (mmdet/models/detectors/vps_temporal_slots.py:528-807) and of the stuff-first
reorder / argmax / relabel of simple_test (:411-435) on top of the K6 kernels.

The pixel work (upsampling, softmax, candidate detection, argmax, areas) runs in two HIP kernels; the
order-dependent decisions of mask_removal and the small-area loop run here on K x K integer tables (K <= 100
kept slots), exactly as the reference's loops would decide them. Full-resolution masks are never
materialised unless asked for (`materialize_masks`), the reference's K x 8 MB device->host copy disappears.
"""
import ctypes
from types import SimpleNamespace

import torch
from torch import nn

from . import _lib, ops

_SOFTMAX_MASKING_CONSTANT = -99999.0

# per-frame state block of svps_panoptic_clip (include/slotvps_hip.h: SVPS_PPC_*)
PPC_K, PPC_N, PPC_PHASE, PPC_ROUNDS, PPC_LUT_IDENT = 0, 1, 2, 3, 4
PPC_THING = 16
PPC_CL, PPC_COUNTS, PPC_KEPT, PPC_CUR, PPC_LUT, PPC_HIST, PPC_AREA, PPC_LUT2, PPC_SLOT, PPC_SCORE = (PPC_THING + 256 * i for i in range(1, 11))
PPC_STATE_INTS = PPC_SCORE + 256
_SMALL_OPTION = {"4": 0, "4_256": 1, "4096_256": 2}


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


from .ops import _on


class _ClipResults(list):
    """Per-frame results of forward_clip; `side_host` carries the host copy of the caller's side work."""
    side_host = None


class PostProcessPanopticInstances(nn.Module):
    """Same constructor arguments and defaults as the reference class (:532-562)."""

    def __init__(self, is_thing_map=None, threshold=0.85, output_dir="", debug=False, fraction_threshold=0.03,
                 pixel_threshold=0.4, apply_mask_removal=False, apply_mask_removal_only_ins=False,
                 use_mask_low_constant=False, catgories_color=None, filter_small_option="4", num_classes=20,
                 num_stuff=11):
        super().__init__()
        self.threshold = threshold
        self.is_thing_map = is_thing_map if is_thing_map is not None else {i: i > 10 for i in range(21)}
        self.fraction_threshold = fraction_threshold
        self.pixel_threshold = pixel_threshold
        self.apply_mask_removal = apply_mask_removal
        self.apply_mask_removal_only_ins = apply_mask_removal_only_ins
        if use_mask_low_constant:
            raise NotImplementedError("use_mask_low_constant=True is not used by the released configs")
        if not (apply_mask_removal and apply_mask_removal_only_ins):
            raise NotImplementedError("the GPU path implements the released configuration: mask removal among instances")
        self.filter_small_option = filter_small_option
        self.num_classes = num_classes
        self.num_stuff = num_stuff

    # ---- kernels -------------------------------------------------------------------------------
    def _candidates(self, m_sorted, thing_u8, size):
        lib = _lib.load()
        K, h, w = m_sorted.shape
        H, W = size
        dev = m_sorted.device
        cand = torch.empty((H * W, 2), dtype=torch.uint8, device=dev)
        counts = torch.zeros(K, dtype=torch.int32, device=dev)
        pairs = torch.zeros((K, K), dtype=torch.int32, device=dev)
        with _on(m_sorted, thing_u8, cand, counts, pairs) as ctx:
            _lib.check(lib.svps_panoptic_candidates(_p(m_sorted), _p(thing_u8), K, h, w, H, W, float(self.pixel_threshold),
                                                    _p(cand), _p(counts), _p(pairs), ctx.stream), "svps_panoptic_candidates")
        return cand, counts, pairs

    @staticmethod
    def _argmax(m_sorted, sel, sel_thing, kept_u8, cand, lut, size, want_ids=False, want_hist=True, want_masks=False, tables=None,
                hist=None):
        """tables: (sel, sel_thing, lut) as uint8 DEVICE tensors already uploaded (the clip path batches its uploads); hist: a zeroed
        [256] int32 device buffer to accumulate into (the clip path keeps one row per frame and copies them back together)."""
        lib = _lib.load()
        K, h, w = m_sorted.shape
        H, W = size
        dev = m_sorted.device
        n = len(sel)
        if tables is None:
            t_sel = torch.tensor(sel if n else [0], dtype=torch.uint8, device=dev)
            t_thing = torch.tensor(sel_thing if n else [0], dtype=torch.uint8, device=dev)
            t_lut = torch.tensor(lut if n else [0], dtype=torch.uint8, device=dev)
        else:
            t_sel, t_thing, t_lut = tables
        ids = torch.empty(H * W, dtype=torch.uint8, device=dev) if want_ids else None
        if hist is None:
            hist = torch.zeros(256, dtype=torch.int32, device=dev) if want_hist else None
        masks = torch.empty((n, H, W), dtype=torch.float32, device=dev) if want_masks else None
        with _on(m_sorted, t_sel, t_thing, kept_u8, cand, t_lut, ids, hist, masks) as ctx:
            _lib.check(lib.svps_panoptic_argmax(_p(m_sorted), _p(t_sel), _p(t_thing), n, _p(kept_u8), _p(cand), _p(t_lut),
                                                h, w, H, W, _p(ids), _p(hist), _p(masks), ctx.stream), "svps_panoptic_argmax")
        return ids, hist, masks

    # ---- the reference's forward, on tensors ---------------------------------------------------------
    @torch.no_grad()
    def forward_tensors(self, pred_logits, pred_masks, size, materialize_masks=False):
        """pred_logits [L, nc], pred_masks [L, h, w] (GPU, fp32) or a lazy frame of mask logits with
        `.decode_slots(slot_indices) -> [K, h, w]` (detector.FrameSlotMasks: only the kept slots are ever decoded), size (H, W).
        Returns a namespace: slot_index [K''] (into the L slots, the reference's filtered Instances order),
        probs, labels, (masks [K'', H, W] if materialize_masks), plus the state `panoptic_ids` needs."""
        if not pred_masks.is_cuda:
            raise RuntimeError("the panoptic post-process runs on the GPU only; there is no CPU fallback")
        dev = pred_masks.device
        nc = pred_logits.shape[-1]
        scores, classes = pred_logits.float().softmax(-1).max(-1)                                   # :684
        if nc == self.num_classes - 1:
            keep = scores > self.threshold
        else:
            keep = classes.ne(nc - 1) & (scores > self.threshold)                                    # :688-691
        idx = torch.nonzero(keep).flatten()
        if idx.numel() == 0:
            raise ValueError("no slot passes the score threshold (the reference's mask_removal fails here too, :652)")
        if idx.numel() > 255:
            raise NotImplementedError("more than 255 kept slots")
        sc = scores[idx].cpu().numpy()
        cl = classes[idx].cpu().numpy()
        order = sc.argsort(kind="stable")[::-1]               # :580; equal scores: see svps_panoptic_clip_select (include/slotvps_hip.h)
        sorted_idx = idx[torch.from_numpy(order.copy()).to(dev)]
        sc, cl = sc[order], cl[order]
        K = len(sc)
        thing = [bool(c > self.num_stuff - 1) for c in cl]                                           # :594
        if hasattr(pred_masks, "decode_slots"):                            # decode the kept slots only, already in score order
            m_sorted = pred_masks.decode_slots(sorted_idx).float().contiguous()
        else:
            m_sorted = pred_masks[sorted_idx].float().contiguous()
        thing_u8 = torch.tensor(thing, dtype=torch.uint8, device=dev)
        cand, counts, pairs = self._candidates(m_sorted, thing_u8, size)
        n_px = size[0] * size[1]
        counts_h, pairs_h = counts.cpu().numpy(), pairs.cpu().numpy()

        # ---- mask_removal :601-640 on the tables: stuff kept first, then things by descending score --------
        kept = [not t for t in thing]
        for i in range(K):
            if not thing[i]:
                continue
            n_i = int(counts_h[i])
            if n_i == 0 or n_i == n_px:                               # logit.max() == logit.min() / mask_sum == 0
                continue
            overlap = sum(int(pairs_h[j, i]) for j in range(i) if thing[j] and kept[j] and cl[j] == cl[i])
            if overlap / float(n_i) > self.fraction_threshold:
                continue
            kept[i] = True
        kept_u8 = torch.tensor(kept, dtype=torch.uint8, device=dev)
        cur = [i for i in range(K) if not thing[i]] + [i for i in range(K) if thing[i] and kept[i]]   # keep_inds order

        # ---- get_ids_area(dedup=True) :759, then the small-area loop :760-790 ------------------------------
        first_of_class = {}
        lut = []
        for j, i in enumerate(cur):
            if not thing[i]:
                first_of_class.setdefault(int(cl[i]), j)
                lut.append(first_of_class[int(cl[i])])
            else:
                lut.append(j)
        _, hist, _ = self._argmax(m_sorted, cur, [thing[i] for i in cur], kept_u8, cand, lut, size)
        area = hist.cpu().numpy()[:len(cur)].tolist()
        while len(cur) > 0:
            if self.filter_small_option == "4":
                small = [a <= 4 for a in area]
            elif self.filter_small_option == "4_256":
                small = [a < 256 if thing[i] else a < 4 for a, i in zip(area, cur)]
            elif self.filter_small_option == "4096_256":
                small = [a < 4096 if not thing[i] else a < 256 for a, i in zip(area, cur)]
            else:
                raise AssertionError("filter_small_option is not valid !!!!!!")
            if not any(small):
                break
            cur = [i for i, s in zip(cur, small) if not s]
            _, hist, _ = self._argmax(m_sorted, cur, [thing[i] for i in cur], kept_u8, cand, list(range(len(cur))), size)
            area = hist.cpu().numpy()[:len(cur)].tolist()
        res = SimpleNamespace(slot_index=sorted_idx[torch.tensor(cur, dtype=torch.long, device=dev)] if cur else sorted_idx[:0],
                              probs=torch.from_numpy(sc[cur].copy()).to(dev), labels=torch.from_numpy(cl[cur].copy()).to(dev),
                              area=area, size=size, _m_sorted=m_sorted, _cur=cur, _thing=thing, _kept_u8=kept_u8, _cand=cand)
        if materialize_masks:
            _, _, res.masks = self._argmax(m_sorted, cur, [thing[i] for i in cur], kept_u8, cand, list(range(len(cur))),
                                           size, want_hist=False, want_masks=True)
        return res

    # ---- the same for ALL frames of a clip, in lock-step: one device -> host copy per PHASE instead of seven per frame ---------------
    @staticmethod
    def _upload(dev, arrays):
        """Several small uint8 host arrays -> device views of ONE uploaded buffer (one H2D copy)."""
        import numpy as np
        lens = [max(1, len(a)) for a in arrays]
        flat = np.zeros(sum(lens), dtype=np.uint8)
        o = 0
        for a, n in zip(arrays, lens):
            flat[o:o + len(a)] = a
            o += n
        t = torch.from_numpy(flat).to(dev)                  # (a few hundred bytes: a pageable copy; pinning a fresh buffer costs more)
        out, o = [], 0
        for n in lens:
            out.append(t[o:o + n])
            o += n
        return out

    def _small(self, area, cur, thing):
        if self.filter_small_option == "4":
            return [a <= 4 for a in area]
        if self.filter_small_option == "4_256":
            return [a < 256 if thing[i] else a < 4 for a, i in zip(area, cur)]
        if self.filter_small_option == "4096_256":
            return [a < 4096 if not thing[i] else a < 256 for a, i in zip(area, cur)]
        raise AssertionError("filter_small_option is not valid !!!!!!")

    # ---- K6c: the decisions on the device, ONE copy back per clip (csrc/panoptic_clip.hip) ---------------------------------------
    device_decisions = True        # False: the lock-step host path below (also taken when the output is not exactly 4x the logits)
    clip_rounds = 4                # (area pass, step) pairs enqueued speculatively; unfinished frames get more (never a wrong result)

    # rows of every frame that are decoded (and handed to `side`) BEFORE the host knows how many slots pass the score filter: the first
    # `clip_decode_cap` slots of the score order. A frame that keeps more (the released configs keep ~30 of 100 - 200) makes the clip run
    # again with all L rows - never a wrong result - and the cap is then RAISED for the clips that follow (to the next power of two above
    # the largest count seen, at most L; logged once per raise), so that a model that keeps many slots pays the second pass once, not per
    # clip. None: always all L rows (K2 then writes 4 L h w bytes per frame: ~1 GB per VIPER clip)
    clip_decode_cap = 64

    def _clip_on_device(self, scores, classes, nc, pred_masks, size, stuff_num, side, cap=None):
        """The whole post-process of a clip enqueued without waiting for anything: score filter + order (svps_panoptic_clip_select),
        decode of the first `cap` slots in that order (K2), candidates, mask_removal, de-duplication, small-area loop, relabel table and the
        id maps (svps_panoptic_clip); `side(index_d)` may enqueue more work whose result comes back with the same wait. Returns the
        per-frame results (a list carrying `side_host`)."""
        import numpy as np
        lib = _lib.load()
        assert lib.svps_panoptic_clip_state_ints() == PPC_STATE_INTS
        dev = scores.device
        T, L = scores.shape
        H, W = size
        h, w = H // 4, W // 4
        if cap is None:
            cap = L if self.clip_decode_cap is None else min(L, int(self.clip_decode_cap))
        scores, classes = scores.contiguous(), classes.contiguous()
        state = torch.zeros((T, PPC_STATE_INTS), dtype=torch.int32, device=dev)
        index_full = torch.empty((T, L), dtype=torch.int64, device=dev)
        with _on(scores, classes, index_full, state) as ctx:
            _lib.check(lib.svps_panoptic_clip_select(_p(scores), _p(classes), T, L, nc, self.num_classes, self.num_stuff,
                                                     float(self.threshold), _p(index_full), _p(state), ctx.stream), "svps_panoptic_clip_select")
        index_d = index_full[:, :cap].contiguous() if cap < L else index_full
        if hasattr(pred_masks, "decode_clip"):                             # every frame's first `cap` slots in score order (rows past K: slot 0)
            m_clip = pred_masks.decode_clip(index_d)
        else:
            m_clip = torch.gather(pred_masks.float(), 1, index_d[:, :, None, None].expand(-1, -1, *pred_masks.shape[2:]))
        m_clip = m_clip.float().contiguous()
        pairs = torch.zeros((T, L * L), dtype=torch.int32, device=dev)
        cand = torch.empty((T, H * W, 2), dtype=torch.uint8, device=dev)
        ids = torch.empty((T, H, W), dtype=torch.uint8, device=dev)

        def enqueue(rounds, stages):
            with _on(m_clip, state, pairs, cand, ids) as ctx:
                _lib.check(lib.svps_panoptic_clip(_p(m_clip), cap * h * w, T, h, w, H, W, _p(state), _p(pairs), L * L, _p(cand),
                                                  _p(ids), float(self.pixel_threshold), float(self.fraction_threshold),
                                                  _SMALL_OPTION[self.filter_small_option], int(stuff_num), rounds, stages, ctx.stream),
                           "svps_panoptic_clip")
        enqueue(self.clip_rounds, 7)
        side_d = side(index_d) if side is not None else None
        st = state.cpu().numpy()                                                                     # THE wait of the clip
        if (st[:, PPC_K] == 0).any():
            raise ValueError("no slot passes the score threshold (the reference's mask_removal fails here too, :652)")
        if (st[:, PPC_K] > cap).any():
            # a frame keeps more slots than were decoded: the kernels skipped it (svps_panoptic_clip: Args::rows); everything again with all
            # rows - and the following clips start with a cap that covers what this one kept
            kmax = int(st[:, PPC_K].max())
            new_cap = min(L, 1 << max(0, kmax - 1).bit_length())
            if self.clip_decode_cap is not None and new_cap > int(self.clip_decode_cap):
                import sys
                print(f"[slotvps_amd.postprocess] a frame kept {kmax} slots, more than clip_decode_cap = {self.clip_decode_cap}: this clip was "
                      f"post-processed twice; clip_decode_cap is now {new_cap}", file=sys.stderr, flush=True)
                self.clip_decode_cap = new_cap
            return self._clip_on_device(scores, classes, nc, pred_masks, size, stuff_num, side, cap=L)
        while (st[:, PPC_PHASE] != 2).any():                                # a frame with more small-area rounds than were enqueued
            enqueue(2, 6)
            st = state.cpu().numpy()
        side_h = side_d.cpu().numpy() if side_d is not None else None
        curs = [st[t, PPC_CUR:PPC_CUR + int(st[t, PPC_N])].tolist() for t in range(T)]
        flat = np.concatenate([np.asarray(c, dtype=np.int64) + t * L for t, c in enumerate(curs)]) if any(curs) else np.zeros(0, np.int64)
        sel_all = index_full.reshape(-1)[torch.from_numpy(flat).to(dev)]       # the surviving slot ids of all frames: one upload, one gather
        offs = np.cumsum([0] + [len(c) for c in curs])
        out = []
        for t in range(T):
            K, cur = int(st[t, PPC_K]), curs[t]
            sorted_idx = st[t, PPC_SLOT:PPC_SLOT + K].astype(np.int64)
            sc = st[t, PPC_SCORE:PPC_SCORE + K].copy().view(np.float32)
            cl = st[t, PPC_CL:PPC_CL + K].astype(np.int64)
            thing = [bool(x) for x in st[t, PPC_THING:PPC_THING + K]]
            out.append(SimpleNamespace(slot_index=sel_all[offs[t]:offs[t + 1]], slot_index_host=sorted_idx[cur] if cur else sorted_idx[:0],
                                       probs_host=sc[cur].copy(), labels_host=cl[cur].copy(), area=st[t, PPC_AREA:PPC_AREA + len(cur)].tolist(),
                                       size=size, rounds=int(st[t, PPC_ROUNDS]), _m_sorted=m_clip[t, :K], _cur=cur, _thing=thing,
                                       _kept=st[t, PPC_KEPT:PPC_KEPT + K].astype(np.uint8), _cand=cand[t], _ids=ids[t],
                                       _stuff_num=int(stuff_num), _sorted_pos=cur, _row_stride=cap))
        out = _ClipResults(out)
        out.side_host = side_h
        return out

    @torch.no_grad()
    def forward_clip(self, pred_logits, pred_masks, size, stuff_num=None, side=None):
        """forward_tensors for the T frames of a clip at once (same decisions, same results frame by frame): pred_logits
        [T, L, nc]; pred_masks a clip of lazy mask logits (`decode_clip(index [T, Kmax]) -> [T, Kmax, h, w]`, detector.SlotMasks) or a
        dense [T, L, h, w] tensor; size (H, W). The frames move through the phases together - score filter, decode of the kept slots
        (ONE K2 launch), candidates, mask_removal tables, areas, small-area loop. With an output of exactly 4x the logits (every
        configuration of the repository) the decisions run on the device too (`_clip_on_device`): the host waits twice per clip - for
        the class scores and for the finished state - instead of seven times per frame; otherwise once per phase. `side(index_d)` (index
        of the kept slots [T, Kmax]) may enqueue work of the caller whose host copy is wanted with the same wait (`.side_host` of the
        returned list). Returns the per-frame result namespaces of forward_tensors (as a list subclass carrying `side_host`)."""
        import numpy as np
        if not pred_masks.is_cuda:
            raise RuntimeError("the panoptic post-process runs on the GPU only; there is no CPU fallback")
        dev = pred_masks.device
        T, L, nc = pred_logits.shape
        H, W = size
        n_px = H * W
        scores, classes = pred_logits.float().softmax(-1).max(-1)                                   # :684
        hw = tuple(pred_masks.shape[-2:])
        if self.device_decisions and H == 4 * hw[0] and W == 4 * hw[1] and L <= 255:
            return self._clip_on_device(scores, classes, nc, pred_masks, size, self.num_stuff if stuff_num is None else stuff_num, side)
        both = torch.stack([scores, classes.float()])
        host = both.cpu().numpy()                                                                    # copy 1: [2, T, L]
        fr = []
        for t in range(T):
            sc_l, cl_l = host[0, t], host[1, t].astype(np.int64)
            keep = sc_l > self.threshold if nc == self.num_classes - 1 else (cl_l != nc - 1) & (sc_l > self.threshold)   # :688-691
            idx = np.nonzero(keep)[0]
            if idx.size == 0:
                raise ValueError("no slot passes the score threshold (the reference's mask_removal fails here too, :652)")
            if idx.size > 255:
                raise NotImplementedError("more than 255 kept slots")
            sc, cl = sc_l[idx], cl_l[idx]
            order = sc.argsort(kind="stable")[::-1]                                                  # :580 (ties: as above)
            fr.append(SimpleNamespace(sorted_idx=idx[order], sc=sc[order], cl=cl[order], K=int(idx.size),
                                      thing=[bool(c > self.num_stuff - 1) for c in cl[order]]))      # :594
        Kmax = max(f.K for f in fr)
        index = np.zeros((T, Kmax), dtype=np.int64)
        for t, f in enumerate(fr):
            index[t, :f.K] = f.sorted_idx
        index_d = torch.from_numpy(index).to(dev)
        if hasattr(pred_masks, "decode_clip"):                             # the kept slots of every frame, score order, one launch
            m_clip = pred_masks.decode_clip(index_d)
        else:
            m_clip = torch.gather(pred_masks.float(), 1, index_d[:, :, None, None].expand(-1, -1, *pred_masks.shape[2:]))
        m_clip = m_clip.float().contiguous()
        h, w = m_clip.shape[2:]
        side_d = side(index_d) if side is not None else None
        things = self._upload(dev, [np.asarray(f.thing, dtype=np.uint8) for f in fr])
        lib = _lib.load()
        cand = torch.empty((T, n_px, 2), dtype=torch.uint8, device=dev)
        tabs = torch.zeros((T, Kmax + Kmax * Kmax), dtype=torch.int32, device=dev)                  # per frame: counts [K], pairs [K, K]
        for t, f in enumerate(fr):
            f.m_sorted = m_clip[t, :f.K]
            f.cand = cand[t]
            with _on(f.m_sorted, things[t], cand, tabs) as ctx:
                _lib.check(lib.svps_panoptic_candidates(_p(f.m_sorted), _p(things[t]), f.K, h, w, H, W, float(self.pixel_threshold),
                                                        _p(f.cand), _p(tabs[t]), _p(tabs[t, Kmax:]), ctx.stream), "svps_panoptic_candidates")
        tabs_h = tabs.cpu().numpy()                                                                  # copy 2
        # ---- mask_removal :601-640 on the tables: stuff kept first, then things by descending score --------
        for t, f in enumerate(fr):
            K, thing, cl = f.K, f.thing, f.cl
            counts_h, pairs_h = tabs_h[t, :K], tabs_h[t, Kmax:Kmax + K * K].reshape(K, K)
            kept = [not x for x in thing]
            for i in range(K):
                if not thing[i]:
                    continue
                n_i = int(counts_h[i])
                if n_i == 0 or n_i == n_px:
                    continue
                overlap = sum(int(pairs_h[j, i]) for j in range(i) if thing[j] and kept[j] and cl[j] == cl[i])
                if overlap / float(n_i) > self.fraction_threshold:
                    continue
                kept[i] = True
            f.kept = kept
            f.cur = [i for i in range(K) if not thing[i]] + [i for i in range(K) if thing[i] and kept[i]]
            first_of_class, lut = {}, []
            for j, i in enumerate(f.cur):                                 # get_ids_area(dedup=True) :759
                if not thing[i]:
                    first_of_class.setdefault(int(cl[i]), j)
                    lut.append(first_of_class[int(cl[i])])
                else:
                    lut.append(j)
            f.lut = lut
        up = self._upload(dev, [np.asarray(x, dtype=np.uint8) for f in fr for x in (f.kept, f.cur, [f.thing[i] for i in f.cur], f.lut)])
        for t, f in enumerate(fr):
            f.kept_u8 = up[4 * t]
            f.tables = (up[4 * t + 1], up[4 * t + 2], up[4 * t + 3])
        # ---- areas, then the small-area loop :760-790: every frame that still has a small segment goes round again (together)
        todo = list(range(T))
        while todo:
            hist = torch.zeros((len(todo), 256), dtype=torch.int32, device=dev)
            for n, t in enumerate(todo):
                f = fr[t]
                self._argmax(f.m_sorted, f.cur, None, f.kept_u8, f.cand, None, size, tables=f.tables, hist=hist[n])
            hist_h = hist.cpu().numpy()                                                              # copy 3 (+ one per extra round)
            again = []
            for n, t in enumerate(todo):
                f = fr[t]
                f.area = hist_h[n, :len(f.cur)].tolist()
                f.area_lut_identity = f.lut == list(range(len(f.cur)))
                if not f.cur:
                    continue
                small = self._small(f.area, f.cur, f.thing)
                if any(small):
                    f.cur = [i for i, s_ in zip(f.cur, small) if not s_]
                    f.lut = list(range(len(f.cur)))
                    again.append(t)
            if again:
                up = self._upload(dev, [np.asarray(x, dtype=np.uint8) for t in again
                                        for x in (fr[t].cur, [fr[t].thing[i] for i in fr[t].cur], fr[t].lut)])
                for n, t in enumerate(again):
                    fr[t].tables = (up[3 * n], up[3 * n + 1], up[3 * n + 2])
            todo = again
        out = []
        for t, f in enumerate(fr):
            cur = f.cur
            sel_d = index_d[t, :f.K][torch.as_tensor(cur, dtype=torch.long, device=dev)] if cur else index_d[t, :0]
            out.append(SimpleNamespace(slot_index=sel_d, slot_index_host=f.sorted_idx[cur] if cur else f.sorted_idx[:0],
                                       probs_host=f.sc[cur].copy(), labels_host=f.cl[cur].copy(), area=f.area, size=size,
                                       _m_sorted=f.m_sorted, _cur=cur, _thing=f.thing, _kept_u8=f.kept_u8, _cand=f.cand,
                                       _tables=f.tables, _hist_identity=f.area if f.area_lut_identity else None, _sorted_pos=cur,
                                       _row_stride=Kmax))
        out = _ClipResults(out)
        out.side_host = side_d.cpu().numpy() if side_d is not None else None
        return out

    @torch.no_grad()
    def panoptic_ids_clip(self, results, stuff_num=None):
        """panoptic_ids for the frames of a clip in lock-step (results of forward_clip): one copy back for the "which positions own
        pixels" histograms of all frames, one upload of all relabel tables. Returns per frame (panoptic_output [H, W] uint8 on the
        device, cls_inds, instance probabilities) - the last two as host tensors."""
        import numpy as np
        stuff_num = self.num_stuff if stuff_num is None else stuff_num
        dev = results[0]._m_sorted.device
        if all(getattr(r, "_ids", None) is not None and r._stuff_num == stuff_num for r in results):      # K6c wrote them already
            out = []
            for res in results:
                ins = [j for j in range(len(res._cur)) if res._thing[res._cur[j]]]
                cls_inds = torch.tensor([int(res.labels_host[j]) - (stuff_num - 1) for j in ins], dtype=torch.long)
                probs = torch.from_numpy(res.probs_host[ins].copy()) if ins else torch.zeros(0)
                out.append((res._ids, cls_inds, probs))
            return out
        if any(getattr(r, "_ids", None) is not None for r in results):
            raise RuntimeError("panoptic_ids_clip: stuff_num differs from the one forward_clip relabelled with; pass it to forward_clip")
        plan = []
        for res in results:
            cur, thing = res._cur, res._thing
            order = [j for j, i in enumerate(cur) if not thing[i]] + [j for j, i in enumerate(cur) if thing[i]]
            plan.append(SimpleNamespace(order=order, sel=[cur[j] for j in order], sem=[int(res.labels_host[j]) for j in order],
                                        identity=order == list(range(len(cur)))))
        # pass 1: which positions own pixels (torch.unique of the argmax map, :420). The area histogram of the last small-area round IS
        # that histogram when it was taken over the same slot list in the same order with the identity table (no stuff de-duplication)
        need = [n for n, (res, pl) in enumerate(zip(results, plan)) if not (pl.identity and res._hist_identity is not None)]
        hist_h = None
        if need:
            up = self._upload(dev, [np.asarray(x, dtype=np.uint8) for n in need
                                    for x in (plan[n].sel, [results[n]._thing[i] for i in plan[n].sel], list(range(len(plan[n].sel))))])
            hist = torch.zeros((len(need), 256), dtype=torch.int32, device=dev)
            for k, n in enumerate(need):
                res = results[n]
                self._argmax(res._m_sorted, plan[n].sel, None, res._kept_u8, res._cand, None, res.size,
                             tables=(up[3 * k], up[3 * k + 1], up[3 * k + 2]), hist=hist[k])
            hist_h = hist.cpu().numpy()
        luts = []
        for n, (res, pl) in enumerate(zip(results, plan)):
            cur, thing = res._cur, res._thing
            areas = hist_h[need.index(n), :len(pl.sel)].tolist() if n in need else res._hist_identity
            present = [j for j, a in enumerate(areas) if a > 0]
            panoptic_num = len(cur)
            instance_num = sum(1 for i in cur if thing[i])
            lut = [0] * max(len(pl.sel), 1)
            count = instance_num
            for pos in range(len(present) - 1, -1, -1):                    # :424-433
                oid = present[pos]
                if oid >= panoptic_num - instance_num:
                    lut[oid] = stuff_num + count - 1
                    count -= 1
                else:
                    lut[oid] = pl.sem[pos]                                 # position in unique(), not the id (:433)
            luts.append(lut)
        up = self._upload(dev, [np.asarray(x, dtype=np.uint8) for res, pl, lut in zip(results, plan, luts)
                                for x in (pl.sel, [res._thing[i] for i in pl.sel], lut)])
        out = []
        for n, (res, pl) in enumerate(zip(results, plan)):
            ids, _, _ = self._argmax(res._m_sorted, pl.sel, None, res._kept_u8, res._cand, None, res.size, want_ids=True, want_hist=False,
                                     tables=(up[3 * n], up[3 * n + 1], up[3 * n + 2]))
            ins = [j for j in range(len(res._cur)) if res._thing[res._cur[j]]]
            cls_inds = torch.tensor([int(res.labels_host[j]) - (stuff_num - 1) for j in ins], dtype=torch.long)
            probs = torch.from_numpy(res.probs_host[ins].copy()) if ins else torch.zeros(0)
            out.append((ids.view(res.size[0], res.size[1]), cls_inds, probs))
        return out

    def forward(self, outputs, processed_sizes, target_sizes=None, id=None):
        """Reference signature (:659): `outputs` carries pred_logits [L, nc] and pred_masks [L, h, w]."""
        if target_sizes is None:
            target_sizes = processed_sizes
        assert len(processed_sizes) == len(target_sizes) == 1
        size = tuple(int(s) for s in processed_sizes[0])
        return self.forward_tensors(outputs.pred_logits, outputs.pred_masks, size, materialize_masks=True)

    # ---- simple_test :411-435 ------------------------------------------------------------------------------
    @torch.no_grad()
    def panoptic_ids(self, res, stuff_num=None):
        """Stuff-first reorder, per-pixel argmax over the surviving masks and the reference's id relabel.
        Returns (panoptic_output [H, W] uint8, cls_inds (1-based thing classes), instance probabilities)."""
        stuff_num = self.num_stuff if stuff_num is None else stuff_num
        cur, thing = res._cur, res._thing
        labels = res.labels.cpu().numpy()
        order = [j for j, i in enumerate(cur) if not thing[i]] + [j for j, i in enumerate(cur) if thing[i]]
        sel = [cur[j] for j in order]
        sem = [int(labels[j]) for j in order]
        panoptic_num = len(cur)
        instance_num = sum(1 for i in cur if thing[i])
        # pass 1: which positions own pixels (torch.unique of the argmax map, :420)
        _, hist, _ = self._argmax(res._m_sorted, sel, [thing[i] for i in sel], res._kept_u8, res._cand,
                                  list(range(len(sel))), res.size)
        present = [j for j, a in enumerate(hist.cpu().numpy()[:len(sel)].tolist()) if a > 0]
        lut = [0] * max(len(sel), 1)
        count = instance_num
        for pos in range(len(present) - 1, -1, -1):                    # :424-433
            oid = present[pos]
            if oid >= panoptic_num - instance_num:
                lut[oid] = stuff_num + count - 1
                count -= 1
            else:
                lut[oid] = sem[pos]                                    # position in unique(), not the id (:433)
        ids, _, _ = self._argmax(res._m_sorted, sel, [thing[i] for i in sel], res._kept_u8, res._cand, lut, res.size,
                                 want_ids=True, want_hist=False)
        ins = [j for j in range(len(cur)) if thing[cur[j]]]
        cls_inds = torch.tensor([int(labels[j]) - (stuff_num - 1) for j in ins], dtype=torch.long)
        return ids.view(res.size[0], res.size[1]), cls_inds, res.probs[torch.tensor(ins, dtype=torch.long, device=res.probs.device)] \
            if ins else res.probs[:0]
