"""LineEndPipeline: the whole per-frame hot path, batched and device-resident.

Op order = the reference's graph, slam_recognition/recognition_testing.py:136-144 (callback: frame ->
zoom pyramid) and :69-90 (compile: rgc -> rgby -> orientation -> line-end -> clip -> pad -> value ->
keypoint indices), restated for the two workloads of BASELINE.json:

  mode "gray"  (configs 1, 2, 5)  frame[H,W,1] -> pyramid -> CS -> ReLU -> K-orientation end bank -> ReLU -> clip
  mode "rgb"   (config 3)         frame[H,W,3] -> pyramid -> rgc -> rgby -> stripe -> regulate -> end -> clip
                                   -> pad_inwards -> value -> per-region keypoint indices

All buffers are torch GPU tensors allocated once; every launch goes to torch's current stream through the
``*_dev`` C-ABI entry points, so a step is pure kernel launches (no allocation, no synchronisation).
PyTorch is used for device memory and streams only.
"""
import ctypes as C

import numpy as np

from . import _lib, _runtime
from . import constant_convolutions as cc
from .util.normalize import normalize_tensor_positive_negative
from .util.zoom.from_image import classic_levels, reference_levels


def default_constants(mode, n_orient=4):
    """The constant kernels of a pipeline as one dict of float32 HWIO arrays (what gets broadcast)."""
    if mode == "gray":
        cs = normalize_tensor_positive_negative(cc.center_surround_tensor(2, [1], [1], [1], [-1]))
        return {"cs": cs.astype(np.float32), "end": cc.end_bank(n_orient).astype(np.float32)}
    if mode == "rgb":
        return {"rgc": cc.midget_rgc(2).astype(np.float32), "rgby": cc.rgby_3(2).astype(np.float32),
                "stripe": cc.rgb_2d_stripe_tensors().astype(np.float32), "blur": cc.blur_tensor(2, 7).astype(np.float32),
                "end": cc.rgb_2d_end_tensors().astype(np.float32)}
    raise ValueError("mode must be 'gray' or 'rgb'")


def pack_constants(consts):
    """dict of arrays -> (flat float32 blob, layout) for a single broadcast."""
    layout, parts = [], []
    for name in sorted(consts):
        a = np.ascontiguousarray(consts[name], dtype=np.float32)
        layout.append((name, a.shape))
        parts.append(a.reshape(-1))
    return np.concatenate(parts), layout


def unpack_constants(blob, layout):
    out, off = {}, 0
    for name, shape in layout:
        n = int(np.prod(shape))
        out[name] = np.asarray(blob[off:off + n], dtype=np.float32).reshape(shape).copy()
        off += n
    return out


def _spin_ms(torch, dev, streams, microseconds):
    """Wall time of one busy-wait kernel (silent_busy_wait_dev: one wavefront polling the clock) on each of ``streams`` at once."""
    import time
    ctx = _runtime.get_context(dev.index)
    lib = _lib.load()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for st in streams:
        ctx.check(lib.silent_busy_wait_dev(ctx.handle, int(microseconds), C.c_void_p(st.cuda_stream)))
    for st in streams:
        st.synchronize()
    return (time.perf_counter() - t0) * 1e3


def pick_concurrent_stream(torch, dev, beside, priority=0, tries=10, microseconds=400):
    """A new stream whose work really runs CONCURRENTLY with the streams in ``beside``.  HIP multiplexes streams onto a few
    hardware queues, and two streams that share one are served strictly in order -- "overlap" through such a pair is a loss
    (cross-stream events, no concurrency), and which pairs share is a property of the process's stream pool, not of the
    priorities (scripts/ab_overlap_pool.py: pipelines #2, #5 and #7 of a process lost 40 %, the others won 8 %).  So it is
    measured: a spin kernel on the candidate and on every stream of ``beside`` at once must take about as long as one alone
    (about a millisecond each).  A NECESSARY check only: pairs that pass it can still stall each other through the real step's
    cross-stream events -- tune_overlap times the real thing.  Returns (stream, verified); after ``tries`` candidates the last one
    is returned unverified (callers keep the flag: LineEndPipeline.overlap_verified)."""
    one = min(_spin_ms(torch, dev, beside[:1] or [torch.cuda.current_stream(dev)], microseconds) for _ in range(2))
    cand = None
    keep = []                                  # rejected candidates stay alive until the choice is made (the pool hands out new ones)
    for _ in range(tries):
        cand = torch.cuda.Stream(dev, priority=priority)
        together = min(_spin_ms(torch, dev, list(beside) + [cand], microseconds) for _ in range(2))
        if together < 1.5 * one:
            return cand, True
        keep.append(cand)
    return cand, False


_staging_pool = None


def _staging_copy(dst, src, min_bytes=8 << 20, workers=16):
    """Pageable host frames -> the pinned staging buffer.  One memcpy thread moves ~5 GB/s, a tenth of what the link then takes
    away: batches above ``min_bytes`` are split along the batch axis over a small thread pool (NumPy releases the GIL in copyto)."""
    global _staging_pool
    nbytes = src.numel() * src.element_size()
    n = min(workers, int(src.shape[0]))
    if nbytes < min_bytes or n < 2:
        dst.copy_(src)
        return
    if _staging_pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _staging_pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="silent-staging")
    d, s_ = dst.numpy(), src.numpy()
    step = -(-int(src.shape[0]) // n)
    list(_staging_pool.map(lambda i: np.copyto(d[i:i + step], s_[i:i + step]), range(0, int(src.shape[0]), step)))


class _RawBlock(object):
    """A device allocation of the library's own (silent_malloc / silent_free): the placement tuner's spacers and draws."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        ctx.check(_lib.load().silent_malloc(ctx.handle, C.c_size_t(self.nbytes), C.byref(p)))
        self.ptr = int(p.value)

    def __del__(self):
        try:
            if getattr(self, "ptr", 0):
                _lib.load().silent_free(self.ctx.handle, C.c_void_p(self.ptr))
                self.ptr = 0
        except Exception:
            pass


class LineEndPipeline(object):
    def __init__(self, frame_hw, mode="gray", n_levels=5, scale=2.0, n_orient=4, batch=1, device=None,
                 constants=None, center_dimensions=None, clip_hi=255.0, flat_policy="ieee", pad=2,
                 max_keypoints_per_frame=None, selection=False, top_percent=0.1, keep_selection_maps=False, value_map=True,
                 peak_value_map=True, orient_map=True, overlap=False, overlap_priorities=True, placement="auto"):
        import torch
        self.torch = torch
        self.mode = mode
        self.channels = 1 if mode == "gray" else 3
        self.device_index = _runtime.default_device() if device is None else int(device)
        self.tdev = torch.device("cuda", self.device_index)
        self.ctx = _runtime.get_context(self.device_index)
        self.batch = int(batch)
        h, w = int(frame_hw[0]), int(frame_hw[1])
        self.frame_shape = (h, w, self.channels)
        levels = (reference_levels((h, w), center_dimensions, scale) if center_dimensions is not None
                  else classic_levels((h, w), scale, n_levels))
        self.crop_px = None
        if center_dimensions is not None:
            y0 = min(l[0] for l in levels); x0 = min(l[1] for l in levels)
            y1 = max(l[0] + l[2] for l in levels); x1 = max(l[1] + l[3] for l in levels)
            self.crop_px = (y1 - y0) * (x1 - x0)
        self.plan = _runtime.PyramidPlan(h, w, self.channels, levels, self.device_index)
        self.extents = self.plan.extents
        self.frame_px = self.plan.frame_px
        self.n_levels = len(self.extents)
        self.levels_c = (_lib.Extent * self.n_levels)(*[_lib.Extent(eh, ew) for eh, ew in self.extents])
        self.consts = {k: np.ascontiguousarray(v, np.float32) for k, v in
                       (constants or default_constants(mode, n_orient)).items()}
        self.clip_hi, self.flat_policy, self.pad = float(clip_hi), flat_policy, int(pad)
        n = self.batch * self.frame_px
        f32 = dict(dtype=torch.float32, device=self.tdev)
        self.placement_tuning = None
        # overlap: consecutive steps overlap on two internal streams -- rgb: the pyramid of batch n + 1 (latency-bound walk kernel)
        # beside the chain kernel and the small launches of the keypoint tail of batch n; gray: the single-read stream kernel of
        # batch n + 1 beside the filter kernel of batch n's smaller levels.  The pyramid is then double-buffered (pipeline.pyr = the
        # last step's).  step() stays "enqueue the whole path for this batch", but on the
        # pipeline's own streams: wait() orders the caller's stream behind the results, outputs() does so itself.
        # overlap=True and overlap="auto" both MEASURE (tune_overlap): a pair of streams that does not pay -- some pairs of a
        # process's stream pool lose 30 - 40 % -- is never kept on faith.  overlap="force": the first pair, unmeasured (tests of
        # the overlapped path's bit-identity, A/B scripts)
        if overlap not in (False, True, None, "auto", "force"):
            raise ValueError("overlap must be False, True, 'auto' or 'force'")
        self._overlap_auto = overlap in (True, "auto")
        self._chain_priority = -1 if overlap_priorities else 0
        self.overlap = False
        self.overlap_verified = None
        self.overlap_tuning = None
        self._order_caller = True          # (A/B switch of scripts/ab_overlap.py: order the caller's stream behind the frame read)
        self._chain_stream = self._walk_stream = self._copy_stream = None
        if mode == "gray":
            self.n_orient = int(self.consts["end"].shape[3])
        else:
            # orient_map=False: SURVEY.md section 8d config 3 returns line_end + keypoints, the orientation map is optional
            self.orient_map = bool(orient_map)
            # value_map=False: the value map (a-8 of the line-end map) is not kept -- with selection the fused step needs it
            # nowhere (silent_rgb_keypoints), and BASELINE config 3 returns line_end + keypoints (+ orient) only
            self.value_map = bool(value_map) or not selection or bool(keep_selection_maps)
        self._adopt_maps(self._alloc_maps())
        if mode != "gray":
            self.regions = (_lib.Extent * self.n_levels)(*[_lib.Extent(max(eh // 2, 1), max(ew // 2, 1))
                                                           for eh, ew in self.extents])
            # selection=True: SURVEY.md section 8d config 3 -- top-percent threshold (a-10, p = 0.1), 3x3 NMS (a-9),
            # then the per-region keypoint indices (a-11) of what survives; False: the reference graph
            # (recognition_testing.py:90), keypoints straight from the padded line-end map
            self.selection, self.top_percent = bool(selection), float(top_percent)
            self.keep_selection_maps = bool(keep_selection_maps)
            if self.selection:
                if self.keep_selection_maps:
                    self.top = torch.empty(n * 3, **f32)
                    self.peaks = torch.empty(n * 3, **f32)
                # peak_value_map=False (fused step only): nobody wants the selection's value map itself -- the keypoint tail
                # then runs sparse (silent_rgb_keypoints with peak_value_out = NULL, csrc/silent_peaks.h)
                self.peak_value_map = bool(peak_value_map) or self.keep_selection_maps
                self.peak_value = torch.empty(n, **f32) if self.peak_value_map else None
            self.kp_cap = int(max_keypoints_per_frame or self.frame_px)
            self.kp_idx = torch.empty((self.batch, self.kp_cap, 4), dtype=torch.int64, device=self.tdev)
            self.kp_counts = torch.zeros(self.batch, dtype=torch.int64, device=self.tdev)
            fp = C.POINTER(C.c_float)
            self._params = _lib.RgbChainParams(
                *[self.consts[k].ctypes.data_as(fp) for k in ("rgc", "rgby", "stripe", "blur", "end")],
                1.0, 0.1, {"ieee": _lib.FLAT_IEEE, "zero": _lib.FLAT_ZERO}[flat_policy], self.clip_hi, self.pad)
        self._lib = _lib.load()
        # placement="auto" (default): the first batch ``step`` sees tunes the placement of the small maps against the big one AND
        # against that batch's own buffer (tune_placement: <= 1 s, bounded memory, results unaffected); None / False: the maps
        # stay where the constructor's allocations landed -- side by side, which is the slow relation more often than not
        if placement not in (None, False, "auto"):
            raise ValueError("placement must be None or 'auto'")
        self._placement_pending = placement == "auto"
        if overlap == "force":
            self._pyrs.append(torch.empty_like(self._pyrs[0]))
            # the chain + tail of batch n are the critical path, the pyramid of batch n + 1 only has to be ready in time: the chain's
            # stream gets the higher queue priority, the pyramid's stream is checked to run concurrently with it
            self._new_stream_pair()
            self.overlap = True
        elif self._overlap_auto:
            self.tune_overlap()

    # -- the maps a step streams through: pyramid + every dense output ---------------------------------
    def _alloc_maps(self):
        """Fresh device buffers for the pyramid and the dense maps (one allocation each; the keypoint rows and the optional
        selection maps are not part of it)."""
        torch = self.torch
        n = self.batch * self.frame_px
        f32 = dict(dtype=torch.float32, device=self.tdev)
        m = {"pyr": torch.empty(n * self.channels, **f32)}
        if self.mode == "gray":
            m["cs"] = torch.empty(n, **f32)
            m["end"] = torch.empty(n * self.n_orient, **f32)
        else:
            m["orient"] = torch.empty(n * 3, **f32) if self.orient_map else None
            m["line_end"] = torch.empty(n * 3, **f32)
            m["value"] = torch.empty(n, **f32) if self.value_map else None
        return m

    def _adopt_maps(self, m):
        """Make ``m``'s tensors the pipeline's maps ("pyr1": the second pyramid buffer of an overlapped pipeline).  Views an earlier
        ``outputs()`` handed out keep the OLD tensors (alive, no longer written): call ``outputs()`` again after a tuner ran."""
        self._pyrs = [m["pyr"]] + ([m["pyr1"]] if "pyr1" in m else list(getattr(self, "_pyrs", [])[1:]))
        self.pyr = self._pyrs[0]
        for k, v in m.items():
            if k not in ("pyr", "pyr1"):
                setattr(self, k, v)

    def _time_step(self, frames, steps, windows=2):
        import time
        torch = self.torch
        best = None
        for _ in range(windows):
            torch.cuda.synchronize(self.tdev)
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step(frames)
            torch.cuda.synchronize(self.tdev)
            t = (time.perf_counter() - t0) / steps * 1e3
            best = t if best is None else min(best, t)
        return best

    def _raw_tensor(self, like):
        """A device buffer of ``like``'s shape and dtype from the library's allocator (silent_malloc: a plain hipMalloc, given back
        to the driver by silent_free the moment the last reference goes), wrapped as a torch tensor.  The tuner's draws and
        spacers come from here, not from torch's caching allocator: nothing a draw leaves behind stays cached in the process."""
        nbytes = like.numel() * like.element_size()
        owner = _RawBlock(self.ctx, nbytes)
        typestr = {4: "<f4", 1: "|u1", 8: "<i8"}[like.element_size()]
        owner.__cuda_array_interface__ = {"shape": tuple(like.shape), "typestr": typestr, "data": (owner.ptr, False), "version": 2}
        t = self.torch.as_tensor(owner, device=self.tdev)      # (holds a reference to ``owner`` for as long as the tensor lives)
        assert t.data_ptr() == owner.ptr
        return t

    def tune_placement(self, frames=None, tries=10, steps=8, budget_s=1.0, spacer_gib=None, max_held_gib=None):
        """Pick the physical placement of the maps RELATIVE to each other by measurement; the largest map is moved last.

        What is measured (profiles/r06/placement.md): the step's time depends on where the maps lie relative to each other in the
        device's physical memory -- not on where any one of them lies.  The same K-orientation map is fast with one allocation of
        the CS map + pyramid and 25 % slower with another (1.18 vs 1.52 ms for config 5's dominant kernel), and the other way
        round for another K map; a synthetic kernel that only issues the three store streams shows the same times draw by draw;
        maps whose physical chunks are co-located on purpose are reliably SLOW.  The discriminating counter is
        TCC_EA0_WRREQ_STALL (x 4.7 on a slow pair, concentrated in a few L2 channels) at identical request counts: the streams
        collide behind the L2, in the memory's own address mapping, which an unprivileged process can neither see nor choose.
        It can choose AGAIN, cheaply: the big map (most of the bytes) is kept, every other map is allocated anew -- behind a
        spacer, because the classes change every few tens of GiB -- ``steps`` steps are timed on each, the fastest set is kept.
        When three draws in a row have shown no contrast (within 2 %) the big map is drawn again as well, once per three draws.
        ``frames``: the caller's resident batch (its placement is part of the relation; ``placement="auto"`` tunes on the first batch
        ``step`` sees); default: synthetic noise.

        Bounds: ``tries`` draws incl. the first; ``budget_s`` seconds; ``spacer_gib`` per spacer, default AND upper limit = the size
        of the pipeline's maps; ``max_held_gib`` for everything the tuner holds at once (spacers + losing draws), default 16 x the
        maps, and never more than a quarter of the device or than leaves a quarter of it free.  Spacers and draws come from
        silent_malloc and go back to the driver at the end: torch's caching allocator is not involved and
        ``torch.cuda.empty_cache()`` is not called.  Results never depend on the choice.  The pipeline's map tensors are REPLACED:
        ``PackedPyramid`` views an earlier ``outputs()`` handed out keep pointing at the old buffers (alive, but no longer written)
        -- call ``outputs()`` again.  The decision is in ``placement_tuning``."""
        import time
        torch = self.torch
        if frames is None:
            frames = torch.randint(0, 256, (self.batch,) + self.frame_shape, device=self.tdev).to(torch.float32)
        self._placement_pending = False
        self.wait()                                # nothing of an overlapped step in flight on the side streams
        torch.cuda.synchronize(self.tdev)
        was = self.overlap
        self.overlap = False
        t_start = time.perf_counter()
        for _ in range(20):                        # past the idle -> load transient of the chip
            self.step(frames)
        names = ("pyr", "cs", "end") if self.mode == "gray" else ("pyr", "orient", "line_end", "value")
        cur = {k: (self._pyrs[0] if k == "pyr" else getattr(self, k)) for k in names}
        if len(self._pyrs) > 1:                    # an overlapped pipeline alternates between two pyramid buffers: both are drawn
            cur["pyr1"] = self._pyrs[1]
        cur = {k: v for k, v in cur.items() if v is not None}
        nbytes = {k: v.numel() * v.element_size() for k, v in cur.items()}
        big = max(reversed([k for k in names if k in cur]), key=lambda k: nbytes[k])    # ties: the last written map (end / line_end)
        small = [k for k in cur if k != big]
        total_maps = sum(nbytes.values())
        spacer = total_maps if spacer_gib is None else min(int(spacer_gib * (1 << 30)), total_maps)
        free, total = torch.cuda.mem_get_info(self.tdev)
        cap = min(16 * total_maps, total // 4) if max_held_gib is None else int(max_held_gib * (1 << 30))
        best = (self._time_step(frames, steps), dict(cur))
        tried, drawn = [round(best[0], 4)], ["first"]
        held, held_bytes, stop, flat = [], 0, "tries", 0
        for i in range(tries - 1):
            if time.perf_counter() - t_start > budget_s:
                stop = "budget_s"
                break
            # no contrast in the last three draws: the small maps have not left their class -- move the big map too
            flat = flat + 1 if max(tried[-3:]) < 1.02 * min(tried[-3:]) else 0
            redraw = list(cur) if (flat >= 3 and len(tried) >= 3) else small
            if redraw is not small:
                flat = 0
            need = spacer + sum(nbytes[k] for k in redraw)
            free, total = torch.cuda.mem_get_info(self.tdev)
            if held_bytes + need > cap or free - need < total // 4:
                stop = "memory cap"
                break
            try:
                if spacer:
                    held.append(_RawBlock(self.ctx, spacer))
                cand = dict(best[1])
                cand.update({k: self._raw_tensor(cur[k]) for k in redraw})
            except RuntimeError:
                stop = "allocation failed"
                break
            held_bytes += need
            self._adopt_maps(cand)
            for _ in range(3):
                self.step(frames)
            t = self._time_step(frames, steps)
            tried.append(round(t, 4))
            drawn.append("all" if redraw is not small else "small")
            if t < best[0]:
                held.append(best[1])               # (tensors shared with ``cand`` stay alive through it)
                best = (t, cand)
            else:
                held.append(cand)
            del cand
        self._adopt_maps(best[1])
        torch.cuda.synchronize(self.tdev)
        held.clear()                               # spacers and losing draws: back to the driver (silent_free), not to a cache
        del cur
        self.overlap = was
        self.placement_tuning = {"tries_ms": tried, "drawn": drawn, "first_draw_ms": tried[0], "chosen_ms": round(best[0], 4),
                                 "big_map": big, "small_maps": small, "spacer_gib": round(spacer / (1 << 30), 2), "stopped_by": stop,
                                 "peak_held_gib": round(held_bytes / (1 << 30), 2), "seconds": round(time.perf_counter() - t_start, 2)}
        return self.placement_tuning

    def close(self):
        """Order the caller's stream and the host behind everything the pipeline has in flight on its own streams (overlap, ingest)
        before its buffers go back to the allocator: they were allocated on the constructor's stream, and the caching allocator
        would hand them to the next tensor of that stream while a side stream still writes them."""
        torch = getattr(self, "torch", None)
        if torch is None:
            return
        for st in (self._chain_stream, self._walk_stream, self._copy_stream):
            if st is not None:
                try:
                    st.synchronize()
                except Exception:          # (interpreter shutdown: the runtime may be gone)
                    pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _new_stream_pair(self, flip=False):
        """flip: the FIRST half's stream (pyramid / stream kernel) at the higher priority instead of the second half's."""
        torch = self.torch
        self._chain_stream = torch.cuda.Stream(self.tdev, priority=0 if flip else self._chain_priority)
        self._walk_stream, self.overlap_verified = pick_concurrent_stream(torch, self.tdev, [self._chain_stream],
                                                                         priority=self._chain_priority if flip else 0)
        self._pyr_ready = [torch.cuda.Event() for _ in range(2)]
        self._pyr_free = [None, None]
        self._steps = 0

    def tune_overlap(self, frames=None, candidates=8, steps=10, budget_s=3.0):
        """MEASURE whether two streams pay on this device, in this process, with these streams -- and with which.
        HIP multiplexes streams onto hardware queues (and those onto the command processor's pipes); which pair of streams a
        pipeline draws from the pool decides whether the pyramid of batch n + 1 really runs beside the chain of batch n (config 3:
        -10 %) or mostly waits on it through the cross-stream events (+0 ... +40 % on the small reference layout), and no static
        rule -- priorities, a concurrency check with independent busy-wait kernels -- predicts it (scripts/ab_overlap_pool.py).  So
        up to ``candidates`` pairs are timed on synthetic noise frames (``steps`` steps each, at most ``budget_s`` seconds in all)
        against the one-stream step; then the best pair and the one-stream step are timed AGAIN, alternately, three windows each
        (the minimum of eight candidates against a baseline timed twice is biased towards two streams), and the pair is kept only
        if that second measurement still wins by more than 2 %.  Otherwise the pipeline stays on one stream and the second
        pyramid buffer is released.  The decision is in ``overlap_tuning``.  Results never depend on the choice (bit-identical
        paths)."""
        import time
        torch = self.torch
        pending = self._placement_pending
        if pending and frames is not None:
            self.tune_placement(frames)            # first things first: which pair of streams pays depends on where the maps lie
            pending = False
        self._placement_pending = False            # (frames=None, the constructor's call: the placement waits for the caller's first batch)
        if frames is None:
            frames = torch.randint(0, 256, (self.batch,) + self.frame_shape, device=self.tdev).to(torch.float32)
        t_start = time.perf_counter()
        self.wait()                                # (an overlapped step still in flight on the side streams: order it, then let it finish)
        torch.cuda.synchronize(self.tdev)
        if len(self._pyrs) < 2:
            self._pyrs.append(torch.empty_like(self._pyrs[0]))
        self.overlap = False
        for _ in range(30):                        # (past the idle -> load transient of the chip: a cold baseline flatters every candidate)
            self.step(frames)
        serial = self._time_step(frames, steps)
        tried, best = [], (None, None)
        for i in range(candidates):
            if time.perf_counter() - t_start > budget_s:
                break
            self._new_stream_pair(flip=bool(i & 1))         # (every other candidate with the priorities the other way round)
            self.overlap = True
            for _ in range(3):
                self.step(frames)
            t = self._time_step(frames, steps)
            tried.append(round(t, 4))
            if best[0] is None or t < best[0]:
                best = (t, (self._chain_stream, self._walk_stream, self.overlap_verified))
            torch.cuda.synchronize(self.tdev)
        # the decision: winner and baseline once more, alternately
        self.overlap = False
        self.pyr = self._pyrs[0]
        again_one, again_two = [], []
        if best[1] is not None and best[0] < 0.98 * serial:
            for _ in range(3):
                self.overlap = False
                self.pyr = self._pyrs[0]
                again_one.append(self._time_step(frames, steps, windows=1))
                self._new_stream_pair()
                self._chain_stream, self._walk_stream, self.overlap_verified = best[1]
                self.overlap = True
                for _ in range(3):
                    self.step(frames)
                again_two.append(self._time_step(frames, steps, windows=1))
                torch.cuda.synchronize(self.tdev)
            serial = min(again_one)
        keep = bool(again_two) and min(again_two) < 0.98 * min(again_one)
        if keep:
            self.overlap = True
        else:
            self.overlap = False
            self.pyr = self._pyrs[0]
            del self._pyrs[1:]                     # one stream: the second pyramid buffer (0.7 GB at config 2) is not held for nothing
            self._chain_stream = self._walk_stream = None
        torch.cuda.synchronize(self.tdev)
        self.overlap_tuning = {"one_stream_ms": round(serial, 4), "two_stream_candidates_ms": tried,
                               "remeasured_ms": {"one_stream": [round(t, 4) for t in again_one], "two_streams": [round(t, 4) for t in again_two]},
                               "chosen": "two streams" if self.overlap else "one stream",
                               "chosen_ms": round(min(again_two), 4) if self.overlap else round(serial, 4),
                               "stream_pair_verified_concurrent": self.overlap_verified if self.overlap else None,
                               "seconds": round(time.perf_counter() - t_start, 2)}
        self._placement_pending = pending
        return self.overlap_tuning

    # -- byte accounting -------------------------------------------------------
    def algorithmic_bytes_per_frame(self):
        """4*[H*W*C (frame read) + P*C (pyramid written) + P*C (pyramid read) + P*sum(C_out returned)]; for crop layouts the
        frame read is the largest crop any level resamples (the part of the frame the pyramid depends on)."""
        h, w, c = self.frame_shape
        if self.crop_px is not None:
            h, w = 1, self.crop_px
        outs = (1 + self.n_orient) if self.mode == "gray" else (3 + (3 if self.orient_map else 0) + (1 if self.value_map else 0))
        return 4 * (h * w * c + 2 * self.frame_px * c + self.frame_px * outs)

    def filter_bytes_per_frame(self):
        """The filter pass alone: pyramid read once + every returned map written once."""
        c = self.channels
        outs = (1 + self.n_orient) if self.mode == "gray" else (3 + (3 if self.orient_map else 0) + (1 if self.value_map else 0))
        return 4 * self.frame_px * (c + outs)

    def pyramid_bytes_per_frame(self):
        h, w, c = self.frame_shape
        return 4 * c * (h * w + self.frame_px)

    def dominant_kernel_name(self):
        """Substring of the rocprofv3 kernel name of the launch that moves most bytes (bench.py matches PMC rows by it)."""
        return ("gray_stream_kernel<%d," % self.n_orient) if self.mode == "gray" else "rgb_line_end2_kernel"

    def launch_summary(self):
        if self.mode == "gray":
            return ("gray_stream_kernel (whole pyramid + level-0 CS/line-end, frame read once) + "
                    "gray_line_end_kernel (levels >= 1)")
        return ("single-read RGB pyramid (pyramid_walk3_kernel), fused RGB chain, max/min + fused selection "
                "(top 10 % > NMS > value), cell-max / count / scan / write keypoint kernels")

    # -- launches --------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.tdev).cuda_stream)

    def _check_frames(self, frames):
        if tuple(frames.shape) != (self.batch,) + self.frame_shape or frames.dtype != self.torch.float32 \
                or not frames.is_cuda or not frames.is_contiguous():
            raise ValueError("frames must be a contiguous float32 GPU tensor of shape %s" %
                             ((self.batch,) + self.frame_shape,))

    def run_pyramid(self, frames, stream=None):
        self._check_frames(frames)
        self.ctx.check(self._lib.silent_pyramid_dev(self.ctx.handle, self.plan.handle, C.c_void_p(frames.data_ptr()),
                                                    self.batch, C.c_void_p(self.pyr.data_ptr()),
                                                    stream or self._stream()))

    def run_filters(self, stream=None):
        s = stream or self._stream()
        if self.mode == "gray":
            self.ctx.check(self._lib.silent_gray_line_end_dev(
                self.ctx.handle, C.c_void_p(self.pyr.data_ptr()), self.levels_c, self.n_levels, self.batch,
                C.c_void_p(self.consts["cs"].ctypes.data), C.c_void_p(self.consts["end"].ctypes.data), self.n_orient,
                self.clip_hi, C.c_void_p(self.cs.data_ptr()), C.c_void_p(self.end.data_ptr()), s))
        else:
            self.ctx.check(self._lib.silent_rgb_line_end_dev(
                self.ctx.handle, C.c_void_p(self.pyr.data_ptr()), self.levels_c, self.n_levels, self.batch,
                C.byref(self._params), C.c_void_p(self.orient.data_ptr()) if self.orient is not None else None,
                C.c_void_p(self.line_end.data_ptr()),
                C.c_void_p(self.value.data_ptr()) if self.value is not None else None, s))

    def run_filters_keypoints(self, stream=None):
        """rgb, selection without the intermediate colour maps: chain + a-10 -> a-9 -> a-8 -> a-11 in one C-ABI call
        (silent_rgb_keypoints_dev): the chain kernel accumulates the per-level extrema of a-10 itself."""
        s = stream or self._stream()
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        self.ctx.check(self._lib.silent_rgb_keypoints_dev(
            self.ctx.handle, p(self.pyr), self.levels_c, self.n_levels, self.batch, C.byref(self._params), self.top_percent,
            self.regions, p(self.orient), p(self.line_end), p(self.value), p(self.peak_value), p(self.kp_idx), self.kp_cap,
            p(self.kp_counts), s))

    def sparse_tail_stats(self):
        """What the sparse keypoint tail of the last fused step did: dict(ran, pairs, dense_pairs, candidates)."""
        st = (C.c_int64 * 5)()
        self.ctx.check(self._lib.silent_sparse_tail_stats(self.ctx.handle, st))
        return {"ran": bool(st[0]), "pairs": int(st[1]), "dense_pairs": int(st[2]), "zero_map_pairs": int(st[4]),
                "candidates": int(st[3])}

    def run_keypoints(self, stream=None):
        s = stream or self._stream()
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None   # (no value map: the selection takes it from line_end)
        geom = (self.levels_c, self.n_levels, self.batch)
        if self.selection and not self.keep_selection_maps:
            # a-10 -> a-9 -> a-8 -> a-11 as one composite: the keypoint search's cell maxima come out of the selection pass
            self.ctx.check(self._lib.silent_select_keypoints_dev(
                self.ctx.handle, p(self.line_end), p(self.value), *geom, 3, self.top_percent, self.regions,
                p(self.peak_value), p(self.kp_idx), self.kp_cap, p(self.kp_counts), s))
            return
        value = self.value
        if self.selection:
            # the same with the two intermediate colour maps kept (tests, visualisation)
            self.ctx.check(self._lib.silent_select_peaks_dev(
                self.ctx.handle, p(self.line_end), p(self.value), *geom, 3, self.top_percent, p(self.top), p(self.peaks),
                p(self.peak_value), s))
            value = self.peak_value
        self.ctx.check(self._lib.silent_max_value_indices_region_dev(
            self.ctx.handle, p(value), *geom, self.regions, p(self.kp_idx), self.kp_cap, p(self.kp_counts), s))

    def run_gray_pass(self, frames, stream=None, parts=3):
        """Whole grayscale hot path in one C-ABI call (silent_gray_pass_dev): region kernel for the non-unit
        levels, fused pyramid + CS + end kernel for the unit levels, filter kernel for the rest.  ``parts``: 1 = pyramid + unit
        levels only, 2 = the filter of the remaining levels only (silent_gray_pass_parts_dev; the halves of an overlapped step)."""
        self._check_frames(frames)
        self.ctx.check(self._lib.silent_gray_pass_parts_dev(
            self.ctx.handle, self.plan.handle, C.c_void_p(frames.data_ptr()), self.batch,
            C.c_void_p(self.consts["cs"].ctypes.data), C.c_void_p(self.consts["end"].ctypes.data), self.n_orient,
            self.clip_hi, C.c_void_p(self.pyr.data_ptr()), C.c_void_p(self.cs.data_ptr()),
            C.c_void_p(self.end.data_ptr()), int(parts), stream or self._stream()))

    def set_profiling(self, every=1):
        """Bracket the dominant kernel with HIP events on every ``every``-th step (0 / False: off)."""
        self.ctx.check(self._lib.silent_set_profiling(self.ctx.handle, int(every)))

    def profiled_kernel(self):
        """(milliseconds, pixels) of the dominant kernel of the last run_gray_pass (HIP events on its stream)."""
        ms, px = C.c_float(0), C.c_int64(0)
        self.ctx.check(self._lib.silent_profile_elapsed_ms(self.ctx.handle, C.byref(ms), C.byref(px)))
        return ms.value, px.value

    def _step_overlapped(self, frames):
        torch = self.torch
        k = self._steps & 1
        self._steps += 1
        self.pyr = self._pyrs[k]
        ws, cs = self._walk_stream, self._chain_stream
        ws.wait_stream(torch.cuda.current_stream(self.tdev))     # the frames were produced on the caller's stream
        if self._pyr_free[k] is not None:
            ws.wait_event(self._pyr_free[k])                      # the chain of two steps ago has read this buffer
        if self.mode == "gray":
            self.run_gray_pass(frames, C.c_void_p(ws.cuda_stream), parts=_lib.GRAY_PART_PYRAMID)
        else:
            self.run_pyramid(frames, C.c_void_p(ws.cuda_stream))
        self._pyr_ready[k].record(ws)
        cs.wait_event(self._pyr_ready[k])
        # the caller's stream is ordered behind the READ of its frames (not behind the chain): whatever it enqueues next may
        # overwrite them, exactly as after a one-stream step
        if self._order_caller:
            torch.cuda.current_stream(self.tdev).wait_event(self._pyr_ready[k])
        s = C.c_void_p(cs.cuda_stream)
        if self.mode == "gray":
            self.run_gray_pass(frames, s, parts=_lib.GRAY_PART_FILTER)
        elif self.selection and not self.keep_selection_maps:
            self.run_filters_keypoints(s)
        else:
            self.run_filters(s)
            self.run_keypoints(s)
        if self._pyr_free[k] is None:
            self._pyr_free[k] = torch.cuda.Event()
        self._pyr_free[k].record(cs)

    def wait(self):
        """Order the caller's current stream behind everything step() has enqueued (a no-op without ``overlap``)."""
        if self.overlap:
            cur = self.torch.cuda.current_stream(self.tdev)
            cur.wait_stream(self._chain_stream)
            cur.wait_stream(self._walk_stream)

    # -- ingest: host frames -> pinned ring -> copy stream -> widening cast -> step -----------------------------
    def _ingest_slot(self, dtype, pinned_source):
        """Ring of two slots per dtype: [pinned staging buffer (None when the caller's frames are pinned already), device buffer
        of the source dtype, float32 frame buffer, events].  Two batches are in flight: while batch n computes, batch n + 1 is
        copied (and, for NumPy sources, batch n + 2 staged by the host)."""
        torch = self.torch
        ring = self._ingest.setdefault(dtype, {"slots": [], "next": 0})
        if not ring["slots"]:
            shape = (self.batch,) + self.frame_shape
            for _ in range(2):
                ring["slots"].append({
                    "pinned": None,
                    "raw": torch.empty(shape, dtype=dtype, device=self.tdev) if dtype != torch.float32 else None,
                    "f32": torch.empty(shape, dtype=torch.float32, device=self.tdev),
                    "h2d_done": torch.cuda.Event(), "raw_free": None, "f32_free": None})
        slot = ring["slots"][ring["next"]]
        ring["next"] ^= 1
        if not pinned_source and slot["pinned"] is None:
            slot["pinned"] = torch.empty((self.batch,) + self.frame_shape, dtype=dtype, pin_memory=True)
        return slot

    def step_host(self, frames):
        """One pass over a batch of HOST frames -- the reference's per-call ingest, recognition_testing.py:141-143
        (np.asarray(frame, float32) -> zoom.from_image -> session.run feed), batched: ``frames`` [batch, H, W, C] as a NumPy array
        or a CPU torch tensor (uint8 as a camera delivers it, or int16 / uint16 / int32 / float32 / float64).  The batch goes
        through a pinned staging buffer (skipped when ``frames`` is a pinned torch tensor already), an asynchronous host-to-device
        copy on the pipeline's COPY stream, the library's widening cast (silent_cast_interleave_dev: uint8 -> float32 on the GPU,
        a quarter of the PCIe bytes) and step().  Returns at once; two batches are in flight (ring of two slots), so the copy of
        batch n + 1 overlaps the compute of batch n.  Results are bit-identical to step() on the same frames resident as float32."""
        torch = self.torch
        if isinstance(frames, np.ndarray):
            src = torch.from_numpy(np.ascontiguousarray(frames))
        else:
            src = frames.contiguous()
        if src.is_cuda or tuple(src.shape) != (self.batch,) + self.frame_shape:
            raise ValueError("step_host takes host frames of shape %s" % ((self.batch,) + self.frame_shape,))
        if src.dtype != torch.float32:
            _runtime._torch_dtype_code(src)         # TypeError for dtypes the cast kernel does not take
        if not hasattr(self, "_ingest"):
            # a stream of another priority than the compute stream gets a hardware queue of its own (two equal-priority streams
            # may share one, and then copy and compute do not overlap): -1 beside a caller's normal stream; with overlap=True the
            # chain stream holds -1 and the copies queue with the pyramid stream, which has to follow them anyway
            self._ingest = {}
            beside = [self._chain_stream, self._walk_stream] if self.overlap else [torch.cuda.current_stream(self.tdev)]
            self._copy_stream, self.copy_stream_verified = pick_concurrent_stream(torch, self.tdev, beside, priority=0 if self.overlap else -1)
        pinned_source = src.is_pinned()
        slot = self._ingest_slot(src.dtype, pinned_source)
        cur = torch.cuda.current_stream(self.tdev)
        cs = self._copy_stream
        if not pinned_source:
            slot["h2d_done"].synchronize()          # the staging buffer's previous copy has left the host
            _staging_copy(slot["pinned"], src)
            src = slot["pinned"]
        dst = slot["f32"] if slot["raw"] is None else slot["raw"]
        for ev in ((slot["f32_free"],) if slot["raw"] is None else (slot["raw_free"],)):
            if ev is not None:
                cs.wait_event(ev)                   # the previous batch of this slot has been consumed on the device
        with torch.cuda.stream(cs):
            dst.copy_(src, non_blocking=True)
        slot["h2d_done"].record(cs)
        cur.wait_event(slot["h2d_done"])
        if slot["raw"] is not None:
            if slot["f32_free"] is not None:
                cur.wait_event(slot["f32_free"])
            c = self.channels
            _runtime.cast_interleave(slot["raw"], slot["f32"], c, 0, c, c, 0, self.batch * self.frame_shape[0] * self.frame_shape[1])
            if slot["raw_free"] is None:
                slot["raw_free"] = torch.cuda.Event()
            slot["raw_free"].record(cur)
        self.step(slot["f32"])
        # the float32 frames are read by the pyramid kernel only: on the walk stream with overlap, else on the caller's stream
        if slot["f32_free"] is None:
            slot["f32_free"] = torch.cuda.Event()
        slot["f32_free"].record(self._walk_stream if self.overlap else cur)

    def step(self, frames):
        """One pass of the hot path over one batch of frames (asynchronous)."""
        if self._placement_pending:
            self.tune_placement(frames)            # once, on the first batch (placement="auto")
        if self.overlap:
            return self._step_overlapped(frames)
        s = self._stream()
        if self.mode == "gray":
            self.run_gray_pass(frames, s)
            return
        self.run_pyramid(frames, s)
        if self.selection and not self.keep_selection_maps:
            self.run_filters_keypoints(s)
            return
        self.run_filters(s)
        self.run_keypoints(s)

    # -- results as PackedPyramids over the pipeline's buffers ---------------------------------------
    def outputs(self, allow_truncated=False):
        """Views over the pipeline's buffers (maps) and host copies of the keypoints.  Raises ValueError when a frame
        produced more keypoints than ``max_keypoints_per_frame`` (like the host entry point's SILENT_E_CAPACITY) unless
        ``allow_truncated``; ``keypoint_counts`` always holds the true counts."""
        P = _runtime.PackedPyramid
        self.wait()
        out = {"pyramid": P(self.pyr, self.extents, self.channels, self.batch)}
        if self.mode == "gray":
            out["cs"] = P(self.cs, self.extents, 1, self.batch)
            out["end"] = P(self.end, self.extents, self.n_orient, self.batch)
        else:
            if self.orient is not None:
                out["orient"] = P(self.orient, self.extents, 3, self.batch)
            out["line_end"] = P(self.line_end, self.extents, 3, self.batch)
            if self.value_map:
                out["value"] = P(self.value, self.extents, 1, self.batch)
            if self.selection:
                if self.keep_selection_maps:
                    out["top"] = P(self.top, self.extents, 3, self.batch)
                    out["peaks"] = P(self.peaks, self.extents, 3, self.batch)
                if self.peak_value is not None:
                    out["peak_value"] = P(self.peak_value, self.extents, 1, self.batch)
            counts = self.kp_counts.cpu().numpy()
            if not allow_truncated and (counts > self.kp_cap).any():
                # the asynchronous *_dev entry points cannot return SILENT_E_CAPACITY: counts[f] > cap IS the overflow flag
                raise ValueError("keypoint capacity exceeded: frame %d produced %d rows, max_keypoints_per_frame is %d "
                                 "(only the first %d were written; pass allow_truncated=True to take them)"
                                 % (int(np.argmax(counts)), int(counts.max()), self.kp_cap, self.kp_cap))
            # (only the rows each frame produced: the buffer holds kp_cap rows per frame -- every pyramid pixel by default)
            out["keypoints"] = [self.kp_idx[f, :min(int(counts[f]), self.kp_cap)].cpu().numpy() for f in range(self.batch)]
            out["keypoint_counts"] = counts
        return out
