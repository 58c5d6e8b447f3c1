"""Batched per-detection inference: crops + masks -> 2D-3D correspondences, entirely in HBM.

The batched form of the hot section of the reference driver (/root/reference/scripts/infer.py:468-542),
which handles one detection at a time: extractor forward -> mask-filtered grid points -> bilinear feature
sampling -> PCA projection -> establish_correspondences. Detections are independent units, so a node
shards them across its GPUs (one process per GPU) and gathers fixed-size result records at the end.
"""

import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import feature_util, ops
from ._lib import call, ptr, stream
from .bank import DeviceBank
from .dinov2_utils import DinoFeatureExtractor
from .matching import MatchResult, match_batch


class FoundPoseEngine:
    def __init__(self, extractor: DinoFeatureExtractor, bank: DeviceBank, grid_cell_size: float = 14.0,
                 top_n_templates: int = 5, top_k_buddies: int = 300, tie_order: str = "canonical", overlap_matching: bool = False,
                 select_tokens: bool = True, fused_sample: bool = True, prefilter: bool = True) -> None:
        """select_tokens / fused_sample / prefilter: A/B switches of measurements and of the bit-identity tests (the hooked block on the sampled tokens only;
        final norm + sampling in one kernel; the two-stage template retrieval where it applies) -- same outputs bit for bit, attributes of the engine.
        overlap_matching: the matching stage of a batch (projection, retrieval, cyclic buddies: ~1 ms of small, latency-bound
        launches) is enqueued on a second stream behind an event, so the next infer_batch's backbone -- enqueued on the
        caller's stream -- runs beside it: the small launches fill the CUs the big GEMMs leave idle in their tail rounds.
        infer_batch then returns a MatchResult whose tensors are complete when `result.ready` has fired: call
        `result.wait()` (corresp_list does) before touching them on another stream, and run follow-up work that should stay
        overlapped under `with torch.cuda.stream(engine.side_stream)`."""
        self.extractor, self.bank = extractor, bank
        self.overlap_matching = overlap_matching
        self.select_tokens, self.fused_sample, self.prefilter = bool(select_tokens), bool(fused_sample), bool(prefilter)
        self._side = None
        self.cell = grid_cell_size
        self.top_n, self.top_k = top_n_templates, top_k_buddies
        self.tie_order = tie_order
        self._grids = {}
        # per-stage HIP events of the last infer_batch (record_stage_times=True): the reference driver's `times` keys
        # feat_extract / grid_sample / proj / corresp (scripts/infer.py:473-544), read back with stage_times()
        self.record_stage_times = False
        self._stage_events: List[Tuple[str, "torch.cuda.Event"]] = []
        self._sub_events: dict = {}   # name -> event, inside a stage (match_batch's "retrieval_begin" / "retrieval_end")

    def _mark(self, name: str) -> None:
        if self.record_stage_times:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._stage_events.append((name, e))

    def _submark(self, name: str) -> None:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self._sub_events[name] = e

    def retrieval_time(self) -> Optional[float]:
        """Seconds the template retrieval (fp_cosine_topk: descriptor streaming + top-n) of the last infer_batch took INSIDE the step, behind the backbone and the
        word search -- the bank comes from HBM there, not from a warm Infinity Cache (record_stage_times=True; synchronises)."""
        b, e = self._sub_events.get("retrieval_begin"), self._sub_events.get("retrieval_end")
        if b is None or e is None:
            return None
        e.synchronize()
        return 1e-3 * b.elapsed_time(e)

    def stage_times(self) -> dict:
        """Seconds per stage of the last infer_batch (whole batch), keyed like the reference's per-detection `times`
        (scripts/infer.py:464-544).  Synchronises on the last event."""
        ev = self._stage_events
        if len(ev) < 2:
            return {}
        ev[-1][1].synchronize()
        return {ev[i][0]: 1e-3 * ev[i - 1][1].elapsed_time(ev[i][1]) for i in range(1, len(ev))}

    def _grid(self, w: int, h: int, device):
        """Grid points of a w x h crop (generate_grid_points) and their pixels int(point + 0.5) (filter_points_by_mask)."""
        key = (w, h)
        if key not in self._grids:
            pts = feature_util.generate_grid_points((w, h), self.cell).to(device).float().contiguous()
            pix = (pts + 0.5).int()
            self._grids[key] = (pts, pix[:, 0].contiguous(), pix[:, 1].contiguous())
        return self._grids[key]

    def _cells9(self, w: int, h: int, device):
        """For every grid point the 3 x 3 patch cells around the cell its sampling position rounds to: a superset of the four
        bilinear taps whatever way the last ulp of the tap arithmetic falls.  -> [G * 9] long, cells outside the map -> gh * gw."""
        key = ("cells9", w, h)
        if key not in self._grids:
            ps = self.extractor.patch_size
            gh, gw = h // ps, w // ps
            pts = self._grid(w, h, device)[0]
            cx = torch.round(pts[:, 0] / ps - 0.5).long()
            cy = torch.round(pts[:, 1] / ps - 0.5).long()
            d = torch.tensor([-1, 0, 1], device=device)
            xs = (cx[:, None, None] + d[None, None, :]).expand(-1, 3, 3)
            ys = (cy[:, None, None] + d[None, :, None]).expand(-1, 3, 3)
            ok = (xs >= 0) & (xs < gw) & (ys >= 0) & (ys < gh)
            self._grids[key] = torch.where(ok, ys * gw + xs, torch.full_like(xs, gh * gw)).reshape(-1).contiguous()
        return self._grids[key]

    def query_points(self, masks: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, List[int]]:
        """filter_points_by_mask for the whole batch -> (points [sumQ,2], point_img [sumQ] i32, counts)."""
        return self._query_points_end(*self._query_points_begin(masks))

    def _query_points_begin(self, masks: torch.Tensor, select_tokens: bool = False):
        """Enqueues fp_query_select -- the mask test of every grid point, the per-detection point lists and (select_tokens) the
        patch tokens the sampling of those points will read, the only outputs the hooked block has to produce -- and an
        asynchronous copy of the per-detection counts to pinned memory.  The counts are the one thing the host needs from the
        device per batch (buffer sizes, segment tables of the matching stage); the caller enqueues the ViT forward BEFORE
        waiting for them, so the wait ends as soon as the previous batch has drained and the device never idles between
        batches (a blocking .tolist() here left a bubble per step)."""
        B, H, W = masks.shape
        dev = masks.device
        pts, pix_x, pix_y = self._grid(W, H, dev)
        G = pts.shape[0]
        m8 = masks if masks.dtype == torch.uint8 else (masks.view(torch.uint8) if masks.dtype == torch.bool else (masks != 0).to(torch.uint8))
        m8 = m8.contiguous()
        ps = self.extractor.patch_size if select_tokens else 1
        C = (H // ps) * (W // ps) if select_tokens else 0
        n_tok = 1 + self.extractor.arch.registers + C if select_tokens else 0   # cls | registers | patches
        cnt = torch.empty(2 * B if select_tokens else B, dtype=torch.int32, device=dev)
        out_pts = torch.empty(B * G, 2, dtype=torch.float32, device=dev)
        out_img = torch.empty(B * G, dtype=torch.int32, device=dev)
        scratch = torch.empty(B * (G + C), dtype=torch.int32, device=dev)
        sel = None
        if select_tokens:
            sel = (torch.empty(B * C, dtype=torch.int32, device=dev), torch.empty(B + 1, dtype=torch.int32, device=dev),
                   torch.empty(B * C, dtype=torch.int32, device=dev))
        call("fp_query_select", ptr(m8), B, H, W, ptr(pix_x), ptr(pix_y), ptr(pts), G, ptr(self._cells9(W, H, dev)) if select_tokens else None, C, n_tok,
             ptr(scratch), ptr(cnt), ptr(out_pts), ptr(out_img), None, ptr(sel[0]) if sel else None, ptr(sel[1]) if sel else None,
             ptr(sel[2]) if sel else None, stream())
        cnt_host = torch.empty(cnt.shape[0], dtype=torch.int32, pin_memory=True)
        cnt_host.copy_(cnt, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return B, cnt_host, ev, out_pts, out_img, sel

    def _query_points_end(self, B, cnt_host, ev, out_pts, out_img, sel=None):
        ev.synchronize()
        counts = cnt_host[:B].tolist()
        total = sum(counts)
        out = (out_pts[:total], out_img[:total], counts)
        if sel is None:
            return out
        sel_counts = cnt_host[B:].tolist()
        return out + ((sel[0], sel[1], sel[2], sum(sel_counts), max(sel_counts)),)

    def infer_batch(self, images: torch.Tensor, masks: torch.Tensor, det_obj: Optional[Sequence[int]] = None,
                    keep_debug: bool = False) -> MatchResult:
        B, _, H, W = images.shape
        det_obj = [0] * B if det_obj is None else list(det_obj)
        fused = self.extractor.facet == "token" and not self.extractor.use_graph and self.fused_sample
        # The reference samples the hooked block's feature map at the query points and nowhere else (infer.py:452-466), so
        # that block only has to produce the patch tokens under the sampling taps: its attention queries, proj and MLP run on
        # those tokens (keys / values: all tokens).  Same sampled features bit for bit; engine.select_tokens = False is the A/B switch.
        select = fused and self.extractor.supports_token_selection and self.select_tokens
        self._stage_events = []
        self._mark("start")
        # f16x3 / fp8: clamped activations are reported per BATCH -- the sticky device counters are snapshotted before and after the backbone
        # of this batch (two 8-byte device copies, no sync); the result raises / warns for what ITS batch clamped, whatever happened before
        track_sat = self.extractor.precision in ("f16", "f16x3", "f16f8", "fp8")
        sat0 = self.extractor.saturation_snapshot() if track_sat else None
        pending = self._query_points_begin(masks, select_tokens=select)
        if select:
            self.extractor.forward_hidden(images, prefix_only=True)  # ~115 launches enqueued before the host waits for the counts
            q_pts, q_img, counts, (sel_rows, sel_off, row_map, num_sel, max_sel) = self._query_points_end(*pending)
            if num_sel > 0:
                self.extractor.forward_selected_block(sel_rows, sel_off, num_sel, max_sel)
            self._mark("feat_extract")
            raw = self.extractor.sample_patch_features(q_pts, q_img, row_map=row_map)
        elif fused:   # final norm + sampling in one pass over the query points only: the [B, Np, D] map is never written
            self.extractor.forward_hidden(images)                # ~120 launches enqueued before the host waits for the counts
            q_pts, q_img, counts = self._query_points_end(*pending)
            self._mark("feat_extract")
            raw = self.extractor.sample_patch_features(q_pts, q_img)
        else:
            fmap, _ = self.extractor.forward_tokens(images)
            q_pts, q_img, counts = self._query_points_end(*pending)
            self._mark("feat_extract")
            gh, gw = self.extractor.num_patches
            D = fmap.shape[-1]
            raw = ops.sample_bilinear(fmap.reshape(B, gh, gw, D).permute(0, 3, 1, 2), q_pts, q_img, (W, H))
        self._mark("grid_sample")
        sat_delta = None
        if track_sat:
            sat1 = self.extractor.saturation_snapshot()
            # (clamped at 0: a reset_saturation() between the two snapshots must not turn into a "negative count", which would read as a verdict.
            #  The counters belong to the extractor: one extractor must not serve two concurrently running engines in the reporting modes.)
            sat_delta = sat1 if sat0 is None else (sat1 - sat0).clamp_min_(0)
        if not self.overlap_matching:
            feats = self._project(raw, counts, det_obj)
            self._mark("proj")
            res = match_batch(self.bank, feats, q_pts, counts, det_obj, self.top_n, self.top_k, keep_debug, self.tie_order,
                              mark=self._submark if self.record_stage_times else None, prefilter=self.prefilter)
            self._mark("corresp")
            if track_sat:
                res.extractor, res.sat_delta = self.extractor, sat_delta   # corresp_list() reads the verdict of this batch
            return res
        main, side = torch.cuda.current_stream(), self.side_stream
        produced = torch.cuda.Event()
        produced.record(main)
        with torch.cuda.stream(side):
            side.wait_event(produced)
            for t in (raw, q_pts, q_img):       # allocated on the caller's stream, consumed here
                t.record_stream(side)
            feats = self._project(raw, counts, det_obj)
            self._mark("proj")
            res = match_batch(self.bank, feats, q_pts, counts, det_obj, self.top_n, self.top_k, keep_debug, self.tie_order,
                              mark=self._submark if self.record_stage_times else None, prefilter=self.prefilter)
            self._mark("corresp")
            res.ready = torch.cuda.Event()
            res.ready.record(side)
        if track_sat:
            res.extractor, res.sat_delta = self.extractor, sat_delta
        return res

    @property
    def side_stream(self) -> torch.cuda.Stream:
        if self._side is None:
            self._side = torch.cuda.Stream()
        return self._side

    def infer_detections(self, image_hwc: torch.Tensor, masks_modal: torch.Tensor, boxes_amodal, camera_c2w,
                         crop_size, crop_rel_pad: float, det_obj: Optional[Sequence[int]] = None, keep_debug: bool = False):
        """From the uncropped input image: the crop producer (infer.py:411-450, foundpose_amd.crop_util) followed by
        infer_batch, without leaving HBM.  image_hwc float32 [H,W,3] in [0,1], masks_modal uint8 [B,H,W], boxes_amodal
        B x (left, top, right, bottom).  -> (MatchResult, crop cameras: the cameras the PnP tail solves in)."""
        from . import crop_util
        crops, crop_masks, cams = crop_util.crop_detections(image_hwc, masks_modal, boxes_amodal, camera_c2w, crop_size, crop_rel_pad)
        return self.infer_batch(crops, crop_masks, det_obj, keep_debug), cams

    def _project(self, raw: torch.Tensor, counts: Sequence[int], det_obj: Sequence[int]) -> torch.Tensor:
        if raw.shape[1] == self.bank.feat_dim:
            return raw
        if len(set(det_obj)) == 1:  # one object in the batch: its projector chain on all rows, no copy into a joint buffer
            projs = self.bank.objects[det_obj[0]].projectors
            if not projs:
                raise ValueError("feature dims differ from the bank and the object has no projector")
            x = raw
            for p in projs:
                x = p.transform(x)
            return x
        out = torch.empty(raw.shape[0], self.bank.feat_dim, dtype=torch.float32, device=raw.device)
        r0 = 0
        i = 0
        while i < len(counts):
            j = i
            n = 0
            while j < len(counts) and det_obj[j] == det_obj[i]:
                n += counts[j]
                j += 1
            x = raw[r0:r0 + n]
            projs = self.bank.objects[det_obj[i]].projectors
            if not projs:
                raise ValueError("feature dims differ from the bank and the object has no projector")
            for p in projs:
                x = p.transform(x)
            out[r0:r0 + n] = x
            r0 += n
            i = j
        return out


RECORD_FLOATS_PER_CORRESP = 9  # q_id, feat_id, dist, conf, x, y, X, Y, Z


def pack_result(res: MatchResult) -> torch.Tensor:
    """Fixed-size record per detection for the final gather: [B, n*(3 + K*9)] 32-bit words, typed fp32 so the whole
    record travels as one tensor (template id, score, count, then K padded correspondences per template).
    The integer fields (template id, count, q_id = coord_2d_ids, feat_id = nn_vertex_ids, corresp_util.py:135-141 in the
    reference) are BIT-CAST int32 -> fp32, not converted: a bank with more than 2^24 features (BASELINE config 5:
    N_f = 18.7 M) has ids a float conversion would round.  unpack_result is the inverse."""
    B, n, K = res.q_ids.shape
    if res.q_ids.is_cuda:  # one kernel writes the record (no torch cat / stack launches in the step)
        c = lambda t, dt: t if (t.dtype == dt and t.is_contiguous()) else t.to(dt).contiguous()
        i32, f32 = torch.int32, torch.float32
        out = torch.empty(B, n * (3 + K * RECORD_FLOATS_PER_CORRESP), dtype=f32, device=res.q_ids.device)
        call("fp_pack_records", ptr(c(res.template_ids, i32)), ptr(c(res.template_scores, f32)), ptr(c(res.counts, i32)), ptr(c(res.q_ids, i32)),
             ptr(c(res.feat_ids, i32)), ptr(c(res.dists, f32)), ptr(c(res.conf, f32)), ptr(c(res.coord_2d, f32)), ptr(c(res.coord_3d, f32)), B, n, K,
             ptr(out), stream())
        return out
    as_f = lambda t: t.to(torch.int32).contiguous().view(torch.float32)
    head = torch.stack([as_f(res.template_ids), res.template_scores, as_f(res.counts)], -1)  # [B,n,3]
    body = torch.cat([as_f(res.q_ids).unsqueeze(-1), as_f(res.feat_ids).unsqueeze(-1), res.dists.unsqueeze(-1),
                      res.conf.unsqueeze(-1), res.coord_2d, res.coord_3d], -1)  # [B,n,K,9]
    return torch.cat([head, body.reshape(B, n, K * RECORD_FLOATS_PER_CORRESP)], -1).reshape(B, -1).contiguous()


def unpack_result(records: torch.Tensor, n: int, K: int) -> MatchResult:
    """Inverse of pack_result on gathered records [R, n*(3 + K*9)] -> MatchResult of R detections (integer fields bit-exact)."""
    R = records.shape[0]
    per = records.reshape(R, n, 3 + K * RECORD_FLOATS_PER_CORRESP)
    as_i = lambda t: t.contiguous().view(torch.int32)
    body = per[..., 3:].reshape(R, n, K, RECORD_FLOATS_PER_CORRESP)
    return MatchResult(template_ids=as_i(per[..., 0]), template_scores=per[..., 1].contiguous(), counts=as_i(per[..., 2]),
                       q_ids=as_i(body[..., 0]), feat_ids=as_i(body[..., 1]), dists=body[..., 2].contiguous(), conf=body[..., 3].contiguous(),
                       coord_2d=body[..., 4:6].contiguous(), coord_3d=body[..., 6:9].contiguous())


def shard_detections(num_det: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of the detection list for `rank` (detections are pre-sorted by object, so
    contiguous chunks keep each rank's bank accesses on few objects)."""
    base, rem = divmod(num_det, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rows(num_det: int, world_size: int) -> int:
    """Rows every rank contributes to the gather: the largest shard (the tail shards are padded up to it)."""
    return (num_det + world_size - 1) // world_size


def pad_records(local: torch.Tensor, rows: int) -> torch.Tensor:
    """Pads a rank's records to `rows` rows (all_gather wants equal contributions).  Padding rows carry template id -1 (bit-cast,
    like every integer field) and zeros; the receiver drops them by POSITION (gathered_valid_index), never by content."""
    if local.shape[0] > rows:
        raise ValueError(f"{local.shape[0]} records do not fit a shard of {rows}")
    if local.shape[0] == rows:
        return local
    pad = torch.zeros(rows - local.shape[0], local.shape[1], dtype=local.dtype, device=local.device)
    pad.view(torch.int32)[:, 0] = -1
    return torch.cat([local, pad], 0)


def gathered_valid_index(num_det: int, world_size: int) -> torch.Tensor:
    """Positions, in the gathered [world_size * shard_rows, ...] tensor, of detections 0 .. num_det-1 in order."""
    per = shard_rows(num_det, world_size)
    idx = []
    for r in range(world_size):
        lo, hi = shard_detections(num_det, world_size, r)
        idx += [r * per + i for i in range(hi - lo)]
    return torch.tensor(idx, dtype=torch.int64)


def gather_records(local: torch.Tensor, world_size: int) -> torch.Tensor:
    """The one exchange step of the path: all-gather of the per-detection records over RCCL (xGMI).
    Every rank contributes the same number of rows (the driver pads the tail shard)."""
    if world_size == 1:
        return local
    import torch.distributed as dist
    if local.is_cuda and dist.get_backend() == "gloo":  # dry runs of the multi-rank path without RCCL: through host memory
        host = local.cpu()
        out = torch.empty(world_size * host.shape[0], host.shape[1], dtype=host.dtype)
        dist.all_gather_into_tensor(out, host)
        return out.to(local.device)
    out = torch.empty(world_size * local.shape[0], local.shape[1], dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)
    return out
