"""The ingest leg of the per-frame path (recognition_testing.py:141-143: np.asarray(frame, float32) -> zoom.from_image -> feed),
batched: LineEndPipeline.step_host (pinned ring, copy stream, widening cast on the GPU) against step() on the same frames
resident as float32 -- bit-identical maps and keypoints, over several consecutive batches so that both ring slots are reused
while the previous batch is still in flight; and overlap=True (two internal streams) against the one-stream step."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b, keys):
    import torch
    for k in keys:
        assert torch.equal(a[k].data.view(torch.int32), b[k].data.view(torch.int32)), k
    if "keypoints" in a:
        np.testing.assert_array_equal(a["keypoint_counts"], b["keypoint_counts"])
        for x, y in zip(a["keypoints"], b["keypoints"]):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("mode,overlap", [("gray", False), ("gray", "force"), ("rgb", False), ("rgb", "force")])
@pytest.mark.parametrize("source", ["numpy_u8", "pinned_u8", "numpy_f32", "numpy_i16"])
def test_step_host_equals_the_resident_step(mode, overlap, source):
    import torch
    from pysilent_amd.pipeline import LineEndPipeline
    h, w, B = 135, 240, 3
    c = 1 if mode == "gray" else 3
    kw = dict(selection=True, value_map=False, peak_value_map=False) if mode == "rgb" else {}
    host = LineEndPipeline((h, w), mode=mode, n_levels=4, batch=B, overlap=overlap, **kw)
    ref = LineEndPipeline((h, w), mode=mode, n_levels=4, batch=B, **kw)
    keys = ("pyramid", "cs", "end") if mode == "gray" else ("pyramid", "orient", "line_end")
    rng = np.random.default_rng(11)
    batches = []
    for _ in range(5):
        u8 = rng.integers(0, 256, (B, h, w, c)).astype(np.uint8)
        if source == "numpy_i16":
            batches.append((u8.astype(np.int16) - 100))
        elif source == "numpy_f32":
            batches.append(u8.astype(np.float32) + np.float32(0.25))
        else:
            batches.append(u8)
    srcs = [torch.from_numpy(b).pin_memory() for b in batches] if source == "pinned_u8" else batches
    # all five batches enqueued back to back (two in flight at any time), only the last one's results are looked at ...
    for s in srcs:
        host.step_host(s)
    ref.step(torch.from_numpy(batches[-1].astype(np.float32)).cuda())
    torch.cuda.synchronize()
    _same(host.outputs(), ref.outputs(), keys)
    # ... and once more batch by batch with the results read in between
    for s, b in zip(srcs[:2], batches[:2]):
        host.step_host(s)
        ref.step(torch.from_numpy(b.astype(np.float32)).cuda())
        _same(host.outputs(), ref.outputs(), keys)


def test_step_host_rejects_what_it_cannot_take():
    import torch
    from pysilent_amd.pipeline import LineEndPipeline
    pipe = LineEndPipeline((32, 48), mode="gray", n_levels=2, batch=2)
    with pytest.raises(ValueError):
        pipe.step_host(np.zeros((1, 32, 48, 1), np.uint8))                  # wrong batch
    with pytest.raises(ValueError):
        pipe.step_host(torch.zeros((2, 32, 48, 1), dtype=torch.uint8).cuda())     # device frames go to step()
    with pytest.raises(TypeError):
        pipe.step_host(np.zeros((2, 32, 48, 1), np.complex64))


@pytest.mark.parametrize("center", [None, (72, 48)])
def test_overlap_mode_is_bit_identical_over_many_steps(center):
    """overlap=True: pyramid of batch n + 1 on a second stream beside the chain + keypoint tail of batch n, double-buffered
    pyramid.  Ten steps with DIFFERENT frames, results read after some of them, against the one-stream pipeline."""
    import torch
    from pysilent_amd.pipeline import LineEndPipeline
    h, w, B = 216, 384, 4
    kw = dict(selection=True, value_map=False, peak_value_map=False)
    if center is not None:
        kw.update(center_dimensions=center, scale=np.e ** .5)
    a = LineEndPipeline((h, w), mode="rgb", n_levels=4, batch=B, overlap="force", **kw)
    b = LineEndPipeline((h, w), mode="rgb", n_levels=4, batch=B, **kw)
    rng = np.random.default_rng(3)
    frames = [torch.from_numpy(rng.integers(0, 256, (B, h, w, 3)).astype(np.float32)).cuda() for _ in range(10)]
    for i, f in enumerate(frames):
        a.step(f)
        b.step(f)
        if i in (0, 3, 4, 9):
            _same(a.outputs(), b.outputs(), ("pyramid", "orient", "line_end"))


def test_overlap_auto_measures_and_stays_bit_identical():
    """overlap="auto": the pipeline times a few stream pairs against its one-stream step when it is built (which streams of the
    process's pool a pipeline draws decides whether two streams pay) and keeps what wins; whatever it chose, the results are the
    one-stream results, and step_host works on top."""
    import torch
    from pysilent_amd.pipeline import LineEndPipeline
    h, w, B = 216, 384, 4
    kw = dict(selection=True, value_map=False, peak_value_map=False)
    a = LineEndPipeline((h, w), mode="rgb", n_levels=4, batch=B, overlap="auto", **kw)
    b = LineEndPipeline((h, w), mode="rgb", n_levels=4, batch=B, **kw)
    t = a.overlap_tuning
    assert t and t["chosen"] in ("one stream", "two streams") and len(t["two_stream_candidates_ms"]) >= 1 and t["one_stream_ms"] > 0
    assert a.overlap == (t["chosen"] == "two streams")
    rng = np.random.default_rng(5)
    for i in range(6):
        u8 = rng.integers(0, 256, (B, h, w, 3)).astype(np.uint8)
        f = torch.from_numpy(u8.astype(np.float32)).cuda()
        if i % 2:
            a.step_host(u8)
        else:
            a.step(f)
        b.step(f)
        if i in (0, 1, 5):
            _same(a.outputs(), b.outputs(), ("pyramid", "orient", "line_end"))


@pytest.mark.parametrize("mode", ["gray", "rgb"])
def test_placement_tuning_keeps_the_results(mode):
    """placement="auto", the default (LineEndPipeline.tune_placement): on the FIRST batch the big map stays and the small maps are
    drawn again a few times behind spacers; the fastest relation is kept.  Where the maps lie relative to each other moves the
    step time, never the results; the record says what was tried (first draw beside the chosen one), the tuner stops at its time
    budget and at its memory cap, nothing it allocated stays behind, and the overlap tuner and step_host work on top."""
    import torch
    from pysilent_amd.pipeline import LineEndPipeline
    h, w, B = 216, 384, 4
    kw = dict(selection=True, value_map=False, peak_value_map=False) if mode == "rgb" else dict(n_orient=8)
    a = LineEndPipeline((h, w), mode=mode, n_levels=4, batch=B, overlap="auto", **kw)          # placement="auto" is the default
    b = LineEndPipeline((h, w), mode=mode, n_levels=4, batch=B, placement=None, **kw)
    assert a.placement_tuning is None and a._placement_pending and not b._placement_pending    # tuned on the first batch, not here
    rng = np.random.default_rng(9)
    c = 1 if mode == "gray" else 3
    names = ("pyramid", "cs", "end") if mode == "gray" else ("pyramid", "orient", "line_end")
    free0 = None
    for i in range(4):
        u8 = rng.integers(0, 256, (B, h, w, c)).astype(np.uint8)
        f = torch.from_numpy(u8.astype(np.float32)).cuda()
        if i % 2:
            a.step_host(u8)
        else:
            a.step(f)
        b.step(f)
        _same(a.outputs(), b.outputs(), names)
        if i == 0:
            t = a.placement_tuning
            big = "end" if mode == "gray" else "line_end"
            assert t and 1 <= len(t["tries_ms"]) <= 10 and t["chosen_ms"] == min(t["tries_ms"]) and t["first_draw_ms"] == t["tries_ms"][0]
            assert t["big_map"] == big and big not in t["small_maps"] and "pyr" in t["small_maps"] and t["seconds"] < 5.0
            assert len(t["drawn"]) == len(t["tries_ms"]) and t["drawn"][0] == "first" and set(t["drawn"][1:]) <= {"small", "all"}
            assert b.placement_tuning is None and not a._placement_pending
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info()[0]
    # an explicit call: budget spent -> the current set stays; a memory cap of nothing -> the same; views of an earlier outputs()
    # keep the OLD buffers, a new outputs() shows the new ones
    assert len(a.tune_placement(f, tries=3, budget_s=0.0)["tries_ms"]) == 1
    t = a.tune_placement(f, tries=3, max_held_gib=0.0)
    assert len(t["tries_ms"]) == 1 and t["stopped_by"] == "memory cap"
    before = a.outputs()
    t = a.tune_placement(f, tries=4, budget_s=5.0)
    assert len(t["tries_ms"]) == 4 and t["stopped_by"] == "tries"
    a.step(f)
    b.step(f)
    _same(a.outputs(), b.outputs(), names)
    if t["chosen_ms"] < t["first_draw_ms"]:
        assert before["pyramid"].data.data_ptr() != a.outputs()["pyramid"].data.data_ptr()
    # spacers and losing draws went back to the driver (silent_free), not into a cache: the device has as much free memory as
    # after the first tuning, give or take the small maps themselves
    torch.cuda.synchronize()
    del before
    assert abs(torch.cuda.mem_get_info()[0] - free0) < 256 << 20
    with pytest.raises(ValueError):
        LineEndPipeline((h, w), mode=mode, n_levels=4, batch=B, placement="yes", **kw)
    # a map the tuner drew is a silent_malloc block wrapped as a tensor: the block lives exactly as long as the tensor (and its views)
    import gc
    import weakref
    t = a._raw_tensor(a.pyr)
    owner = weakref.ref([o for o in gc.get_objects() if type(o).__name__ == "_RawBlock" and o.ptr == t.data_ptr()][0])
    view = t[16:32]
    del t
    gc.collect()
    assert owner() is not None and owner().ptr != 0               # the view keeps the storage, the storage keeps the block
    view.fill_(3.0)
    torch.cuda.synchronize()
    assert float(view.sum()) == 48.0
    del view
    gc.collect()
    assert owner() is None                                        # ... and silent_free ran with the last reference
    a.close()


@pytest.mark.parametrize("overlap", ["force", "auto"])
@pytest.mark.parametrize("shape,levels,K", [((135, 240), 4, 4), ((216, 384), 5, 8), ((64, 96), 1, 3)])
def test_gray_overlap_is_bit_identical(overlap, shape, levels, K):
    """Gray pipelines on two streams: the stream kernel (pyramid + level 0's CS / line-end) of batch n + 1 beside the filter kernel
    of batch n's smaller levels (silent_gray_pass_parts_dev), double-buffered pyramid -- eight different batches, results read in
    between, against the one-call pass; a single-level pyramid (nothing for the second half to do) included."""
    import torch
    from pysilent_amd.pipeline import LineEndPipeline
    h, w = shape
    B = 3
    a = LineEndPipeline((h, w), mode="gray", n_levels=levels, n_orient=K, batch=B, overlap=overlap)
    b = LineEndPipeline((h, w), mode="gray", n_levels=levels, n_orient=K, batch=B)
    rng = np.random.default_rng(8)
    for i in range(8):
        f = torch.from_numpy(rng.integers(0, 256, (B, h, w, 1)).astype(np.float32)).cuda()
        a.step(f)
        b.step(f)
        if i in (0, 2, 3, 7):
            _same(a.outputs(), b.outputs(), ("pyramid", "cs", "end"))
