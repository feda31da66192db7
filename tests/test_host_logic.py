"""Host-side logic of the product (no GPU): level geometry, packed pyramids, constant packing."""
import math

import numpy as np
import numpy.testing as npt
import pytest

import silent_oracle as so


def test_reference_levels_match_oracle_geometry():
    from pysilent_amd.util.zoom.from_image import reference_levels
    for hw, center, scale in [((480, 640), (288, 192), math.e ** .5), ((480, 640), (160, 120), math.e ** .5),
                              ((1080, 1920), (192, 108), math.e ** .5), ((1080, 1920), (120, 68), math.e ** .5),
                              ((97, 131), (32, 24), 1.7)]:
        lv = reference_levels(hw, center, scale)
        center_hw = list(reversed(center))
        assert len(lv) == so.ref_num_scales(hw, center_hw, scale)
        for s, l in enumerate(lv):
            y0, x0, ch, cw, zh, zw, _ = so.ref_level_geometry(hw, center_hw, scale, s)
            assert l == (y0, x0, ch, cw, zh, zw, center_hw[0], center_hw[1])
    # level counts obtainable by choosing center_dimensions
    assert len(reference_levels((480, 640), (160, 120), math.e ** .5)) == 3
    assert len(reference_levels((1080, 1920), (192, 108), math.e ** .5)) == 5
    assert len(reference_levels((1080, 1920), (120, 68), math.e ** .5)) == 6


def test_classic_levels_match_oracle_extents():
    from pysilent_amd.util.zoom.from_image import classic_levels
    for hw, scale, n in [((1080, 1920), 2.0, 5), ((2160, 3840), 2.0, 8), ((480, 640), 2.0, 3), ((270, 480), math.e ** .5, 6)]:
        lv = classic_levels(hw, scale, n)
        assert [(l[6], l[7]) for l in lv] == so.classic_extents(hw[0], hw[1], scale, n)
        assert all(l[:4] == (0, 0, hw[0], hw[1]) for l in lv)
    with pytest.raises(ValueError):
        classic_levels((4, 4), 2.0, 6)


def test_from_image_argument_checks_mirror_reference_asserts():
    from pysilent_amd.util import zoom
    img = np.zeros((8, 8, 3), np.float32)
    with pytest.raises(AssertionError, match="Scale must be greater than one"):
        zoom.from_image(img, 3, (4, 4), 1.0)
    with pytest.raises(AssertionError, match="colors"):
        zoom.from_image(img, 0, (4, 4), 2.0)
    with pytest.raises(AssertionError, match="dimension"):
        zoom.from_image(img, 3, (0, 4), 2.0)
    with pytest.raises(TypeError):
        zoom.from_image([[1, 2]], 3, (4, 4), 2.0)


def test_packed_pyramid_roundtrip():
    from pysilent_amd import PackedPyramid
    rng = np.random.default_rng(0)
    levels = [rng.standard_normal((3, h, w, 2)).astype(np.float32) for h, w in [(5, 7), (3, 4), (1, 1)]]
    p = PackedPyramid.from_levels(levels)
    assert p.n_frames == 3 and p.channels == 2 and p.frame_px == 35 + 12 + 1 and not p.on_device
    for l, lev in enumerate(levels):
        npt.assert_array_equal(p.level(l), lev)
    # layout: frames outermost, then levels (what the C ABI documents)
    npt.assert_array_equal(p.data[:70], levels[0][0].reshape(-1))
    npt.assert_array_equal(p.data[70:94], levels[1][0].reshape(-1))
    q = p.like(4)
    assert q.data.shape[0] == 3 * 48 * 4 and q.extents == p.extents
    with pytest.raises(ValueError):
        PackedPyramid(np.zeros(5, np.float32), [(2, 2)], 1, 1)


def test_constants_pack_unpack_roundtrip():
    from pysilent_amd.pipeline import default_constants, pack_constants, unpack_constants
    for mode in ("gray", "rgb"):
        c = default_constants(mode, 8)
        blob, layout = pack_constants(c)
        assert blob.dtype == np.float32 and blob.nbytes < 8192          # "< 8 KB" 
        back = unpack_constants(blob, layout)
        assert sorted(back) == sorted(c)
        for k in c:
            npt.assert_array_equal(back[k], c[k].astype(np.float32))
    with pytest.raises(ValueError):
        default_constants("bogus")


def test_frame_sharding_is_a_partition():
    from pysilent_amd.distributed import shard_frame_indices
    for total, world in [(512, 8), (10, 4), (3, 8), (64, 1)]:
        shards = [shard_frame_indices(total, r, world) for r in range(world)]
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(total))
        assert all(i % world == r for r, s in enumerate(shards) for i in s)


def test_get_dimensions_and_selection_argument_checks():
    from pysilent_amd.util.get_dimensions import get_dimensions
    from pysilent_amd.util.selection import pad_inwards, _regions_for
    assert get_dimensions(np.zeros((1, 4, 4, 3))) == 2 and get_dimensions(np.zeros((1, 4, 4, 4, 1))) == 3
    with pytest.raises(TypeError, match="must either be tensor or numpy array"):
        get_dimensions(3.0)
    with pytest.raises(ValueError):
        pad_inwards(np.zeros((1, 4, 4, 3), np.float32), [[1, 0], [2, 2], [2, 2], [0, 0]])
    assert _regions_for([(192, 288)], [1, 96.0, 144.0, 3]) == [(96, 144)]      # the reference passes floats
    assert _regions_for([(8, 8), (4, 4)], [(4, 4), (2, 2)]) == [(4, 4), (2, 2)]


def test_index_tensor_matches_reference_test_literal():
    # reference tests/test_index_tensor.py:7-30: from_shape([4,2,2,3]) == [[[0,0],[1,0]],[[0,1],[1,1]]] (x, y order)
    from pysilent_amd.util import index_tensor
    assert index_tensor.from_shape([4, 2, 2, 3]).tolist() == [[[0, 0], [1, 0]], [[0, 1], [1, 1]]]
    assert index_tensor.from_tensor(np.ones((4, 2, 2, 3))).tolist() == [[[0, 0], [1, 0]], [[0, 1], [1, 1]]]
    npt.assert_array_equal(index_tensor.from_shape([1, 5, 7, 1]), so.index_tensor_from_shape([1, 5, 7, 1]))


def test_stream_sharding_keeps_streams_whole():
    from pysilent_amd import distributed as d
    owners = {}
    for rank in range(3):
        for s_ in d.shard_stream_indices(8, rank, 3):
            assert s_ not in owners
            owners[s_] = rank
    assert sorted(owners) == list(range(8)) and all(owners[s_] == s_ % 3 for s_ in owners)


def test_recovery_mode_mirrors_reference_errors():
    from pysilent_amd.util.energy.recovery import recovery_mode
    assert recovery_mode(False, True) == 1 and recovery_mode(True, False) == 2 and recovery_mode(True, True) == 3
    with pytest.raises(ValueError, match="You must choose a type of recovery"):
        recovery_mode(False, False)


def _chain_structure(consts):
    import ctypes as C
    from pysilent_amd import _lib
    lib = _lib.load()
    fp = C.POINTER(C.c_float)
    arrs = {k: np.ascontiguousarray(consts[k], np.float32) for k in ("rgc", "rgby", "stripe", "blur", "end")}
    params = _lib.RgbChainParams(*[arrs[k].ctypes.data_as(fp) for k in ("rgc", "rgby", "stripe", "blur", "end")],
                                 1.0, 0.1, 0, 255.0, 2)
    flags, masks = C.c_uint(0), (C.c_uint * 6)()
    assert lib.silent_rgb_chain_structure(C.byref(params), C.byref(flags), masks) == 0
    return flags.value, list(masks)


def _chain_stream(consts, knobs=0):
    import ctypes as C
    from pysilent_amd import _lib
    lib = _lib.load()
    fp = C.POINTER(C.c_float)
    arrs = {k: np.ascontiguousarray(consts[k], np.float32) for k in ("rgc", "rgby", "stripe", "blur", "end")}
    params = _lib.RgbChainParams(*[arrs[k].ctypes.data_as(fp) for k in ("rgc", "rgby", "stripe", "blur", "end")],
                                 1.0, 0.1, 0, 255.0, 2)
    stream = np.full(384, np.nan, np.float32)
    n, variant = C.c_int(-1), C.c_int(-1)
    assert lib.silent_rgb_chain_stream(C.byref(params), knobs, stream.ctypes.data_as(fp), C.byref(n), C.byref(variant)) == 0
    return stream, n.value, variant.value, arrs


def test_rgb_weight_stream_is_the_kernels_consumption_order():
    """The pair kernel (csrc/silent_rgb2.h) reads its weights as a stream in the order it consumes them.  This walks the
    stream the way the kernel does -- stage by stage, pending rows of one output side by side -- and rebuilds from it the
    3 x 3 (x 3 x 3) kernels the stages apply; they must be the kernels that went in.  Dense form: every one of the 373
    weights, in place; basic form: the diagonal of rgc and the channel-0 slice of the stripe bank; two-group form: the
    scale x mix products reproduce rgby and the end bank."""
    from pysilent_amd.pipeline import default_constants
    consts = default_constants("rgb")

    def take_conv(it, pairs):          # for o: for active (dx, i): for dy = 2, 1, 0   ->  K[dy, dx, i, o]
        k = np.zeros((3, 3, 3, 3), np.float32)
        for o in range(3):
            for dx in range(3):
                for i in range(3):
                    if pairs >> (o * 3 + i) & 1:
                        for dy in (2, 1, 0):
                            k[dy, dx, i, o] = next(it)
        return k

    def take_two(it, masks):           # for (dx, i): for dy = 2, 1, 0 (scale); then for term (group, i): for o (mix)
        scale = np.zeros((3, 3, 3), np.float32)
        for dx in range(3):
            for i in range(3):
                for dy in (2, 1, 0):
                    scale[dy, dx, i] = next(it)
        mix = np.array([next(it) for _ in range(18)], np.float32).reshape(2, 3, 3)     # [group][i][o]
        k = np.zeros((3, 3, 3, 3), np.float64)
        for dy in range(3):
            for dx in range(3):
                for i in range(3):
                    grp = 0 if masks[i] >> (dy * 3 + dx) & 1 else 1
                    k[dy, dx, i, :] = np.float64(scale[dy, dx, i]) * mix[grp, i, :]
        return k

    def take_sum(it):                  # for o: for dx: for dy = 2, 1, 0
        k = np.zeros((3, 3, 3), np.float32)
        for o in range(3):
            for dx in range(3):
                for dy in (2, 1, 0):
                    k[dy, dx, o] = next(it)
        return k

    def take_blur(it):                 # for dx: for k = 0..6 (pending row k takes kernel row 6 - k)
        b = np.zeros((7, 7), np.float32)
        for dx in range(7):
            for k in range(7):
                b[6 - k, dx] = next(it)
        return b

    def take_blur_folded(it):          # mirror-symmetric form: for j = min(dx, 6 - dx) = 0..3: for d = |dy| = 0..3
        q = np.zeros((4, 4), np.float32)
        for j in range(4):
            for d in range(4):
                q[d, j] = next(it)
        b = np.zeros((7, 7), np.float32)
        for dy in range(7):
            for dx in range(7):
                b[dy, dx] = q[abs(dy - 3), min(dx, 6 - dx)]
        return b

    # dense
    stream, n, variant, a = _chain_stream(consts, knobs=1)
    assert (n, variant) == (373, 0) and not np.isnan(stream).any() and not stream[n:].any()
    it = iter(stream[:n])
    np.testing.assert_array_equal(take_conv(it, 0x1ff), a["rgc"])
    np.testing.assert_array_equal(take_conv(it, 0x1ff), a["rgby"])
    np.testing.assert_array_equal(take_conv(it, 0x1ff), a["stripe"])
    np.testing.assert_array_equal(take_blur(it), a["blur"][:, :, 0, 0])
    np.testing.assert_array_equal(take_conv(it, 0x1ff), a["end"])
    assert next(it, None) is None
    # basic: diagonal rgc, stripe as a filter of the channel sum
    stream, n, variant, a = _chain_stream(consts, knobs=2)
    assert (n, variant) == (27 + 81 + 27 + 49 + 81, 1) and not stream[n:].any()
    it = iter(stream[:n])
    np.testing.assert_array_equal(take_conv(it, 0x111), a["rgc"] * np.eye(3, dtype=np.float32))
    np.testing.assert_array_equal(take_conv(it, 0x1ff), a["rgby"])
    np.testing.assert_array_equal(take_sum(it), a["stripe"][:, :, 0, :])
    np.testing.assert_array_equal(take_blur(it), a["blur"][:, :, 0, 0])
    np.testing.assert_array_equal(take_conv(it, 0x1ff), a["end"])
    assert next(it, None) is None
    # two-group (the reference's kernels with the symmetric forms switched off: knob bit 6)
    stream, n, variant, a = _chain_stream(consts, knobs=64)
    assert (n, variant) == (27 + 45 + 27 + 16 + 45, 2) and len(stream) % 32 == 0 and not stream[n:].any()
    _, masks = _chain_structure(consts)
    it = iter(stream[:n])
    np.testing.assert_array_equal(take_conv(it, 0x111), a["rgc"] * np.eye(3, dtype=np.float32))
    np.testing.assert_allclose(take_two(it, masks[:3]), a["rgby"], rtol=2e-6, atol=1e-9)
    np.testing.assert_array_equal(take_sum(it), a["stripe"][:, :, 0, :])
    np.testing.assert_array_equal(take_blur_folded(it), a["blur"][:, :, 0, 0])      # 16 folded weights rebuild all 49
    np.testing.assert_allclose(take_two(it, masks[3:]), a["end"], rtol=2e-6, atol=1e-9)
    assert next(it, None) is None
    # symmetric forms (what the reference's kernels get): rgc per channel (corner, edge_h, edge_v, centre); rgby = S (x) A around the
    # centre + B at the centre; the stripe bank as (left, right) SGPR pairs, centres, (right, left) pairs per output
    stream, n, variant, a = _chain_stream(consts)
    assert (n, variant) == (12 + 27 + 1 + 48 + 16 + 45, 3) and not stream[n:].any()
    it = iter(stream[:n])
    q = np.array([next(it) for _ in range(12)], np.float32).reshape(4, 3)            # [corner, edge_v, edge_h, centre][channel]
    rgc = np.zeros((3, 3, 3, 3), np.float32)
    for c in range(3):
        corner, edge_v, edge_h, centre = q[:, c]
        rgc[:, :, c, c] = [[corner, edge_h, corner], [edge_v, centre, edge_v], [corner, edge_h, corner]]
    np.testing.assert_array_equal(rgc, a["rgc"])
    A = np.array([next(it) for _ in range(9)], np.float32).reshape(3, 3)             # [i][o]
    prof = np.array([next(it) for _ in range(9)], np.float32).reshape(3, 3)          # [corner, edge_v, edge_h][o]: one value per output
    assert (prof == prof[:, :1]).all()
    corner, edge_v, edge_h = prof[:, 0]
    S = np.array([[corner, edge_h, corner], [edge_v, 0, edge_v], [corner, edge_h, corner]], np.float64)
    B = np.array([next(it) for _ in range(9)], np.float32).reshape(3, 3)
    rgby = S[:, :, None, None] * A[None, None].astype(np.float64)
    rgby[1, 1] += B
    np.testing.assert_allclose(rgby, a["rgby"], rtol=2e-6, atol=1e-9)
    assert next(it) == 0.0                                                            # the pair block starts on an even position
    stripe = np.zeros((3, 3, 3), np.float32)                                          # [dy][dx][o]
    for o in range(3):
        blk = [next(it) for _ in range(16)]
        for k, dy in enumerate((2, 1, 0)):
            l, r, c = blk[2 * k], blk[2 * k + 1], blk[6 + k]
            assert (blk[10 + 2 * k], blk[10 + 2 * k + 1]) == (r, l)
            stripe[dy, :, o] = (l, c, r)
        assert blk[9] == 0.0
    np.testing.assert_array_equal(stripe, a["stripe"][:, :, 0, :])
    np.testing.assert_array_equal(take_blur_folded(it), a["blur"][:, :, 0, 0])
    np.testing.assert_allclose(take_two(it, masks[3:]), a["end"], rtol=2e-6, atol=1e-9)
    assert next(it, None) is None
    # an rgc that is not mirror-symmetric, an rgby whose surround is not one profile: the two-group stream
    lop = {k: np.array(v, np.float32) for k, v in consts.items()}
    lop["rgc"][0, 0, 1, 1] *= 1.5
    assert _chain_stream(lop)[2] == 2
    lop = {k: np.array(v, np.float32) for k, v in consts.items()}
    lop["rgby"][0, 1, 1, 2] *= 1.5                     # (still two-group per input channel? no: a third direction for channel 1)
    assert _chain_stream(lop)[2] in (1, 2)
    # a blur that is channel-uniform but NOT mirror-symmetric: no folded form, the basic instantiation
    skew = {k: np.array(v, np.float32) for k, v in consts.items()}
    skew["blur"][0, 0] *= 2.0
    stream, n, variant, a = _chain_stream(skew)
    assert (n, variant) == (27 + 81 + 27 + 49 + 81, 1)
    # generic weights: the dense stream whatever the knobs say
    rng = np.random.default_rng(3)
    noise = {k: rng.standard_normal(np.shape(v)).astype(np.float32) for k, v in consts.items()}
    noise["blur"] = np.repeat(np.repeat(noise["blur"][:, :, :1, :1], 3, axis=2), 3, axis=3)
    stream, n, variant, a = _chain_stream(noise)
    assert (n, variant) == (373, 0)
    it = iter(stream[:n])
    for name in ("rgc", "rgby", "stripe"):
        np.testing.assert_array_equal(take_conv(it, 0x1ff), a[name])


def test_rgb_chain_structure_of_the_reference_kernels():
    """The structure the fused RGB kernel exploits is DETECTED in the weights, and the reference's generators have it:
    diagonal rgc, channel-sum stripe, two-group rgby (centre tap | 8 surround taps) and end bank (per orientation the
    taps on either side of the facet).  The masks are what the specialised kernel is compiled for."""
    from pysilent_amd.pipeline import default_constants
    consts = default_constants("rgb")
    flags, masks = _chain_structure(consts)
    assert flags == 0b111111           # + rgc mirror-symmetric per channel, rgby = profile (x) channel mix + a centre mix
    assert masks == [0x010, 0x010, 0x010, 0x1f9, 0x119, 0x11f]
    # the end bank's groups are the signs of the three orientation profiles (oriented_end_detector.py:47-53)
    end = consts["end"].astype(np.float64)
    for i in range(3):
        a = [t for t in range(9) if masks[3 + i] >> t & 1]
        va = end.reshape(9, 3, 3)[a, i, :]
        assert np.linalg.matrix_rank(va, tol=1e-6) == 1
    # generic weights: nothing is found, the dense kernel runs
    rng = np.random.default_rng(0)
    noise = {k: rng.standard_normal(np.shape(v)).astype(np.float32) for k, v in consts.items()}
    assert _chain_structure(noise)[0] == 0
    mixed = dict(consts, rgby=noise["rgby"])
    assert _chain_structure(mixed)[0] == 0b011011


def test_additive_filter_mirrors_the_reference_constant():
    """centroids.py:9-18: identity over the two index channels at every tap of the region; the literal 2 x 2 identity makes
    every other channel count a broadcast error, as in the reference."""
    from pysilent_amd.util import additive_filter
    k = additive_filter([3, 4], 2)
    assert k.shape == (3, 4, 2, 2) and k.dtype == np.float32
    assert (k == np.eye(2, dtype=np.float32)).all()
    with pytest.raises(ValueError):
        additive_filter([3, 3], 3)


def test_generate_recovery_mirrors_the_reference_on_host_arrays():
    """slam_recognition/util/energy/recovery.py:12-22 under its own name; the constant branch needs no GPU."""
    import numpy as np
    import pytest
    from pysilent_amd.util.energy import generate_recovery
    from pysilent_amd.util.energy.recovery import generate_recovery as gr2
    assert generate_recovery is gr2
    x = np.array([[1.0, np.nan], [300.0, 0.0]], np.float32)
    r = generate_recovery(x)                                   # constant: ones_like * 10, NaN input does not show
    assert r.dtype == np.float32 and (r == 10.0).all() and r.shape == x.shape
    with pytest.raises(ValueError, match="You must choose a type of recovery"):
        generate_recovery(x, is_input_based=False, is_constant=False)


def test_staging_copy_splits_a_batch_over_threads():
    """LineEndPipeline.step_host's host leg for pageable sources: the batch is copied into the staging buffer by a small thread pool
    (one memcpy thread is a tenth of the link rate); every byte arrives, for batch sizes that do not divide by the worker count."""
    import torch
    from pysilent_amd import pipeline as P
    rng = np.random.default_rng(4)
    for batch, dtype in ((1, np.uint8), (5, np.uint8), (16, np.float32), (37, np.int16)):
        src = torch.from_numpy(rng.integers(0, 200, (batch, 33, 47, 3)).astype(dtype))
        dst = torch.zeros_like(src)
        P._staging_copy(dst, src, min_bytes=0, workers=4)
        assert torch.equal(dst, src)


def test_bench_line_carries_every_config_in_flat_keys_and_a_last_summary():
    """a reader who keeps only the headline keys (the driver's record keeps ``config``'s scalars) or only the tail
    of the line still finds every BASELINE config: flat ``config.side_*`` scalars, the first draw beside the tuned step, and one
    compact ``summary`` string as the LAST key of the JSON line."""
    import json
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    def wl(ms, frac, first):
        return {"ms_per_step": ms, "whole_pass_frac_of_hbm_peak": frac, "dominant_kernel_frac_of_hbm_peak": frac + 0.05,
                "placement_tuning": {"tries_ms": [first, ms], "chosen_ms": ms}}

    out = {"ms_per_step": 0.97, "config": {"workload": "1080p gray, 5-level pyramid", "whole_pass_frac_of_hbm_peak": 0.707,
                                            "placement_tuning": {"tries_ms": [0.998, 0.97], "chosen_ms": 0.97}},
           "roofline": {"kernel": "gray_stream_kernel<4>", "avg_launch_ms": 0.777, "frac": 0.626, "traffic": 4435420275},
           "latency": {"640x480": {"native_ms_p50": 0.38, "native_views_ms_p50": 0.23, "native_in_place_ms_p50": 0.21, "gpu_busy_ms": 0.18}},
           "other_workloads": {k: wl(1.0 + 0.1 * i, 0.3 + 0.05 * i, 1.2 + 0.1 * i) for i, k in enumerate(bench.SIDE_KEYS)},
           "cpu_baseline": None}
    out["other_workloads"]["config3_dense_tail"] = wl(1.6, 0.3, 1.7)           # not a BASELINE config: stays out of the summary
    bench.record_side(out)
    cfg = out["config"]
    assert cfg["first_draw_ms"] == 0.998 and cfg["chosen_ms"] == 0.97
    for i, k in enumerate(bench.SIDE_KEYS):
        assert cfg["side_%s_ms" % k] == out["other_workloads"][k]["ms_per_step"]
        assert cfg["side_%s_first_draw_ms" % k] == out["other_workloads"][k]["placement_tuning"]["tries_ms"][0]
        assert cfg["side"][k] == [cfg["side_%s_ms" % k], cfg["side_%s_frac" % k], cfg["side_%s_first_draw_ms" % k]]
    assert "side_config3_dense_tail_ms" not in cfg
    assert cfg["latency_480p_ms"] == 0.38 and cfg["latency_480p_views_ms"] == 0.23 and cfg["latency_480p_in_place_ms"] == 0.21
    assert list(out)[-1] == "summary" and len(out["summary"]) <= 1500 and out["summary"].startswith("SUMMARY")
    for k in ("config3 ", "config5 ", "reference_layout ", "reference_layout_gray ", "first draw 0.998", "latency 640x480", "frac 0.6260"):
        assert k in out["summary"], k
    assert json.loads(json.dumps(out))["summary"] == out["summary"] and json.dumps(out).rstrip("}").endswith('"')
    # a run without side workloads / latency (N > 1, --no-side-workloads): the keys that exist, nothing else, no exception
    bare = {"ms_per_step": 1.0, "config": {"workload": "w", "whole_pass_frac_of_hbm_peak": 0.5, "placement_tuning": None},
            "roofline": {"kernel": "k", "avg_launch_ms": 0.8, "frac": 0.6, "traffic": None}, "latency": None, "other_workloads": None}
    bench.record_side(bare)
    assert bare["config"]["first_draw_ms"] is None and bare["config"]["side"] == {} and "first draw None" in bare["summary"]
