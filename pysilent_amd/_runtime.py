"""Host-side runtime above the C ABI: contexts, tensor adapters, one thin wrapper per entry point.

Tensors the wrappers accept (the reference accepts ``tf.Tensor`` or ``np.ndarray``, get_dimensions.py:9-12):
  * ``np.ndarray`` rank 4 NHWC  -> host-pointer entry points, result is a fresh float32 ndarray
  * ``torch.Tensor`` on a ROCm device, rank 4 NHWC -> ``*_dev`` entry points on torch's current stream,
    result is a torch tensor on the same device (PyTorch is only the allocator / stream provider)
  * ``PackedPyramid`` (ragged levels, host or device) -> same, result is a PackedPyramid
Anything else raises the reference's TypeError.
"""
import ctypes as C
import os

import weakref

import numpy as np

from . import _lib

TYPE_ERROR_MESSAGE = "Input to orientation filter must either be tensor or numpy array."


# ----------------------------------------------------------------------------- context

class Context(object):
    """One silent_ctx: one per (host thread, GPU); not thread-safe."""

    def __init__(self, device=0):
        lib = _lib.load()
        self._lib = lib
        self.device = int(device)
        self.handle = C.c_void_p()
        rc = lib.silent_create(self.device, C.byref(self.handle))
        if rc != _lib.SILENT_OK:
            msg = _lib.last_error(None)
            self.handle = C.c_void_p()
            raise RuntimeError("pysilent_amd needs an MI355X (gfx950) GPU and has no CPU fallback: " + msg)

    @property
    def name(self):
        buf = C.create_string_buffer(256)
        self._lib.silent_device_name(self.handle, buf, 256)
        return buf.value.decode()

    def check(self, rc):
        _lib.check(rc, self.handle)

    def set_tuning(self, which, value):
        """Kernel-selection knob of this context (silent_set_tuning): _lib.TUNE_GRAY / TUNE_RGB / TUNE_PYRAMID."""
        self.check(self._lib.silent_set_tuning(self.handle, int(which), C.c_uint(int(value))))

    def get_tuning(self, which):
        v = C.c_uint(0)
        self.check(self._lib.silent_get_tuning(self.handle, int(which), C.byref(v)))
        return int(v.value)

    def close(self):
        if self.handle:
            self._lib.silent_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_contexts = {}


def default_device():
    if "SILENT_DEVICE" in os.environ:
        return int(os.environ["SILENT_DEVICE"])
    return int(os.environ.get("LOCAL_RANK", "0"))


_override = []      # stack of private contexts (use_context)


class use_context(object):
    """``with use_context(ctx): ...`` -- every op inside runs on ``ctx`` instead of the shared per-device context.
    A caller that bakes device pointers into a HIP graph needs this: the workspace of its private context only
    grows for its own calls, so what the graph captured stays valid."""

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        _override.append(self.ctx)
        return self.ctx

    def __exit__(self, *exc):
        _override.pop()
        return False


def get_context(device=None):
    device = default_device() if device is None else int(device)
    if _override and _override[-1].device == device:
        return _override[-1]
    ctx = _contexts.get(device)
    if ctx is None:
        ctx = _contexts[device] = Context(device)
    return ctx


class tuning(object):
    """``with tuning(TUNE_RGB, 1): ...`` -- run the ops inside with a kernel-selection knob of the current context set."""

    def __init__(self, which, value, device=None):
        self.which, self.value, self.device = which, value, device

    def __enter__(self):
        self.ctx = get_context(self.device)
        self.old = self.ctx.get_tuning(self.which)
        self.ctx.set_tuning(self.which, self.value)
        return self.ctx

    def __exit__(self, *exc):
        self.ctx.set_tuning(self.which, self.old)
        return False


def device_count():
    n = C.c_int(0)
    _lib.load().silent_device_count(C.byref(n))
    return n.value


# ----------------------------------------------------------------------------- tensors

def is_torch_tensor(x):
    return type(x).__module__.split(".")[0] == "torch" and hasattr(x, "data_ptr")


def _torch_dtype_code(t):
    import torch
    table = {torch.uint8: _lib.DT_U8, torch.float32: _lib.DT_F32, torch.float64: _lib.DT_F64, torch.int32: _lib.DT_I32,
             torch.int16: _lib.DT_I16, torch.int64: _lib.DT_I64}
    if hasattr(torch, "uint16"):
        table[torch.uint16] = _lib.DT_U16
    if t.dtype not in table:
        raise TypeError("unsupported tensor dtype %s (uint8, int16, int32, int64, float32, float64)" % t.dtype)
    return table[t.dtype]


def cast_interleave(src, out, in_stride, in_offset, count, out_stride, out_offset, n_pixels):
    """out[p * out_stride + out_offset + k] = float32(src[p * in_stride + in_offset + k]) on the GPU (silent_cast_interleave_dev,
    torch's current stream): dtype widening, cutting a colour plane out of an interleaved image, interleaving planes."""
    import torch
    ctx = get_context(src.device.index or 0)
    stream = C.c_void_p(torch.cuda.current_stream(src.device).cuda_stream)
    ctx.check(_lib.load().silent_cast_interleave_dev(ctx.handle, C.c_void_p(src.data_ptr()), _torch_dtype_code(src), int(n_pixels),
                                                    int(in_stride), int(in_offset), int(count), C.c_void_p(out.data_ptr()),
                                                    int(out_stride), int(out_offset), stream))
    return out


def as_float32(t):
    """A contiguous float32 GPU tensor with the values of ``t`` (np.asarray(frame, dtype=float32) of
    recognition_testing.py:141 for device tensors: any layout, any dtype).  float32 contiguous tensors pass through; contiguous
    uint8 / int16 / uint16 / int32 / int64 / float64 tensors are widened by the library's own cast kernel (the fast path: no
    torch kernel); anything else -- a strided view such as ``x.permute(...)`` or ``x[..., :3]``, float16 / bfloat16 / bool /
    int8 -- is made contiguous float32 by torch first, like ``tensor.to(float32).contiguous()`` did before round 3."""
    import torch
    if t.dtype == torch.float32 and t.is_contiguous():
        return t
    try:
        code_ok = t.is_contiguous() and _torch_dtype_code(t) is not None
    except TypeError:
        code_ok = False
    if not code_ok:
        return t.to(torch.float32).contiguous()
    out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    if t.numel():
        cast_interleave(t, out, 1, 0, 1, 1, 0, t.numel())
    return out


class PackedPyramid(object):
    """A batch of pyramids with per-level extents, packed as the C ABI wants it.

    ``data`` is a flat float32 buffer (np.ndarray, or torch tensor on the GPU) of
    ``n_frames * sum(h_l * w_l) * channels`` elements: frames outermost, then levels, NHWC inside."""

    def __init__(self, data, extents, channels, n_frames=1):
        self.extents = [(int(h), int(w)) for h, w in extents]
        self.channels = int(channels)
        self.n_frames = int(n_frames)
        self.frame_px = sum(h * w for h, w in self.extents)
        n = self.n_frames * self.frame_px * self.channels
        if is_torch_tensor(data):
            data = data.reshape(-1)
        else:
            data = np.ascontiguousarray(data, dtype=np.float32).reshape(-1)
        if data.shape[0] != n:
            raise ValueError("PackedPyramid: buffer holds %d floats, extents need %d" % (data.shape[0], n))
        self.data = data

    @property
    def on_device(self):
        return is_torch_tensor(self.data)

    @classmethod
    def from_levels(cls, levels):
        """levels: list of [n_frames, h_l, w_l, C] float32 ndarrays."""
        n_frames, c = levels[0].shape[0], levels[0].shape[-1]
        ext = [l.shape[1:3] for l in levels]
        per_frame = [np.concatenate([np.ascontiguousarray(l[f], np.float32).reshape(-1) for l in levels])
                     for f in range(n_frames)]
        return cls(np.concatenate(per_frame), ext, c, n_frames)

    def level(self, l):
        """[n_frames, h_l, w_l, C] view (host: NumPy view; device: torch view)."""
        h, w = self.extents[l]
        off = sum(eh * ew for eh, ew in self.extents[:l]) * self.channels
        n = h * w * self.channels
        return self.data.reshape(self.n_frames, -1)[:, off:off + n].reshape(self.n_frames, h, w, self.channels)

    def levels(self):
        return [self.level(l) for l in range(len(self.extents))]

    def like(self, channels, data=None):
        """Same geometry, other channel count; allocates when data is None."""
        n = self.n_frames * self.frame_px * channels
        if data is None:
            if self.on_device:
                import torch
                data = torch.empty(n, dtype=torch.float32, device=self.data.device)
            else:
                data = np.empty(n, dtype=np.float32)
        return PackedPyramid(data, self.extents, channels, self.n_frames)


class _Operand(object):
    """Normalised view of an input tensor: geometry + pointer + how to allocate results."""

    def __init__(self, x, channels=None):
        self.packed = isinstance(x, PackedPyramid)
        if self.packed:
            self.src = x
            buf = x.data
            self.extents, self.n_frames, self.c = x.extents, x.n_frames, x.channels
            self.dev = x.on_device
        elif isinstance(x, np.ndarray) or is_torch_tensor(x):
            if x.ndim != 4:
                raise ValueError("expected a rank-4 NHWC tensor, got shape %s" % (tuple(x.shape),))
            n, h, w, c = (int(s) for s in x.shape)
            if min(n, h, w, c) < 1:
                raise ValueError("empty tensor: shape %s" % (tuple(x.shape),))
            self.extents, self.n_frames, self.c = [(h, w)], n, c
            self.dev = is_torch_tensor(x)
            buf = x
        else:
            raise TypeError(TYPE_ERROR_MESSAGE)
        if channels is not None and self.c != channels:
            raise ValueError("tensor has %d channels, expected %d" % (self.c, channels))
        if self.dev:
            import torch
            if not buf.is_cuda:
                raise TypeError("torch tensors must live on the GPU (use a numpy array for host data)")
            buf = as_float32(buf)
            self.device = buf.device.index or 0
            self.stream = C.c_void_p(torch.cuda.current_stream(buf.device).cuda_stream)
            self.ptr = C.c_void_p(buf.data_ptr())
            self._torch_device = buf.device
        else:
            buf = np.ascontiguousarray(buf, dtype=np.float32)
            self.device = None
            self.stream = None
            self.ptr = C.c_void_p(buf.ctypes.data)
        self.buf = buf                                  # keep alive
        self.frame_px = sum(h * w for h, w in self.extents)
        self.levels = (_lib.Extent * len(self.extents))(*[_lib.Extent(h, w) for h, w in self.extents])
        self.n_levels = len(self.extents)
        self.ctx = get_context(self.device)

    def alloc(self, channels, dtype=np.float32):
        n = self.n_frames * self.frame_px * channels
        if self.dev:
            import torch
            tdt = {np.float32: torch.float32, np.int64: torch.int64}[dtype]
            out = torch.empty(n, dtype=tdt, device=self._torch_device)
            return out, C.c_void_p(out.data_ptr())
        out = np.empty(n, dtype=dtype)
        return out, C.c_void_p(out.ctypes.data)

    def wrap(self, flat, channels):
        if self.packed:
            return PackedPyramid(flat, self.extents, channels, self.n_frames)
        h, w = self.extents[0]
        return flat.reshape(self.n_frames, h, w, channels)

    def geom(self):
        return (self.levels, self.n_levels, self.n_frames)


def _kernel_arg(k, c_in=None):
    k = np.ascontiguousarray(np.asarray(k, dtype=np.float64).astype(np.float32))   # tf.constant(k, float32)
    if k.ndim != 4:
        raise ValueError("filter must be HWIO rank 4, got shape %s" % (k.shape,))
    if c_in is not None and k.shape[2] != c_in:
        raise ValueError("filter C_in %d does not match tensor channels %d" % (k.shape[2], c_in))
    return k


def _same_geometry(a, b, what):
    if a.extents != b.extents or a.n_frames != b.n_frames or a.dev != b.dev:
        raise ValueError("%s: tensors differ in geometry or placement" % what)


# ----------------------------------------------------------------------------- op wrappers

def conv2d_same(x, kernel_hwio, relu=False, clip_hi=None):
    op = _Operand(x)
    k = _kernel_arg(kernel_hwio, op.c)
    kh, kw, _, co = k.shape
    out, optr = op.alloc(co)
    flags = (_lib.RELU if relu else 0) | (_lib.CLIP if clip_hi is not None else 0)
    lib, ctx = _lib.load(), op.ctx
    args = (ctx.handle, op.ptr) + op.geom() + (op.c, C.c_void_p(k.ctypes.data), kh, kw, co, flags,
                                                  float(clip_hi or 0.0), optr)
    ctx.check(lib.silent_conv2d_same_dev(*(args + (op.stream,))) if op.dev else lib.silent_conv2d_same(*args))
    return op.wrap(out, co)


def regulate(x, blur_hwio, regulation_value, regulation_root=0.5, flat_policy="ieee"):
    op = _Operand(x)
    k = _kernel_arg(blur_hwio, op.c)
    if k.shape[3] != op.c:
        raise ValueError("blur tensor must be [kh, kw, C, C]")
    try:
        pol = {"ieee": _lib.FLAT_IEEE, "zero": _lib.FLAT_ZERO}[flat_policy]
    except KeyError:
        raise ValueError("flat_policy must be 'ieee' or 'zero'")
    out, optr = op.alloc(op.c)
    lib, ctx = _lib.load(), op.ctx
    args = (ctx.handle, op.ptr) + op.geom() + (op.c, C.c_void_p(k.ctypes.data), k.shape[0], k.shape[1],
                                                  float(regulation_value), float(regulation_root), pol, optr)
    ctx.check(lib.silent_regulate_dev(*(args + (op.stream,))) if op.dev else lib.silent_regulate(*args))
    return op.wrap(out, op.c)


def gray_line_end(x, cs_kernel, end_bank, clip_hi=255.0, want_cs=True, want_end=True):
    op = _Operand(x, channels=1)
    cs = _kernel_arg(cs_kernel, 1)
    eb = _kernel_arg(end_bank, 1)
    if cs.shape != (3, 3, 1, 1) or eb.shape[:3] != (3, 3, 1):
        raise ValueError("gray_line_end needs a [3,3,1,1] CS kernel and a [3,3,1,K] end bank")
    K = eb.shape[3]
    cs_out, cs_ptr = op.alloc(1) if want_cs else (None, None)
    end_out, end_ptr = op.alloc(K) if want_end else (None, None)
    lib, ctx = _lib.load(), op.ctx
    args = (ctx.handle, op.ptr) + op.geom() + (C.c_void_p(cs.ctypes.data), C.c_void_p(eb.ctypes.data), K,
                                                  float(clip_hi), cs_ptr, end_ptr)
    ctx.check(lib.silent_gray_line_end_dev(*(args + (op.stream,))) if op.dev else lib.silent_gray_line_end(*args))
    return (op.wrap(cs_out, 1) if want_cs else None, op.wrap(end_out, K) if want_end else None)


def pad_inwards(x, pt, pb, pl, pr):
    op = _Operand(x)
    out, optr = op.alloc(op.c)
    lib, ctx = _lib.load(), op.ctx
    args = (ctx.handle, op.ptr) + op.geom() + (op.c, int(pt), int(pb), int(pl), int(pr), optr)
    ctx.check(lib.silent_pad_inwards_dev(*(args + (op.stream,))) if op.dev else lib.silent_pad_inwards(*args))
    return op.wrap(out, op.c)


def value_from_color(x):
    op = _Operand(x)
    out, optr = op.alloc(1)
    lib, ctx = _lib.load(), op.ctx
    args = (ctx.handle, op.ptr) + op.geom() + (op.c, optr)
    ctx.check(lib.silent_value_from_color_dev(*(args + (op.stream,))) if op.dev else lib.silent_value_from_color(*args))
    return op.wrap(out, 1)


def bw_from_color(x):
    op = _Operand(x)
    out, optr = op.alloc(1)
    lib, ctx = _lib.load(), op.ctx
    args = (ctx.handle, op.ptr) + op.geom() + (op.c, optr)
    ctx.check(lib.silent_bw_from_color_dev(*(args + (op.stream,))) if op.dev else lib.silent_bw_from_color(*args))
    return op.wrap(out, 1)


def nms3x3(x, mode="product"):
    op = _Operand(x)
    try:
        m = {"product": _lib.NMS_PRODUCT, "fired": _lib.NMS_FIRED}[mode]
    except KeyError:
        raise ValueError("mode must be 'product' or 'fired'")
    out, optr = op.alloc(op.c)
    lib, ctx = _lib.load(), op.ctx
    args = (ctx.handle, op.ptr) + op.geom() + (op.c, m, optr)
    ctx.check(lib.silent_nms3x3_dev(*(args + (op.stream,))) if op.dev else lib.silent_nms3x3(*args))
    return op.wrap(out, op.c)


def top_value_points(color, top_percent=0.1, value=None):
    op = _Operand(color)
    vptr = None
    if value is not None:
        vop = _Operand(value, channels=1)
        _same_geometry(op, vop, "top_value_points")
        vptr = vop.ptr
    out, optr = op.alloc(op.c)
    lib, ctx = _lib.load(), op.ctx
    args = (ctx.handle, op.ptr, vptr) + op.geom() + (op.c, float(top_percent), optr)
    ctx.check(lib.silent_top_value_points_dev(*(args + (op.stream,))) if op.dev else lib.silent_top_value_points(*args))
    return op.wrap(out, op.c)


def max_value_indices_region(value, regions, cap_per_frame=None):
    """value: 1-channel tensor.  regions: one (rH, rW) per level.  Returns (idx [n_frames, cap, 4] int64,
    counts [n_frames] int64) -- host arrays for host input, torch tensors for device input."""
    op = _Operand(value, channels=1)
    if len(regions) != op.n_levels:
        raise ValueError("need one (rH, rW) region per level")
    reg = (_lib.Extent * op.n_levels)(*[_lib.Extent(int(rh), int(rw)) for rh, rw in regions])
    cap = op.frame_px if cap_per_frame is None else int(cap_per_frame)
    lib, ctx = _lib.load(), op.ctx
    if op.dev:
        import torch
        idx = torch.empty((op.n_frames, cap, 4), dtype=torch.int64, device=op._torch_device)
        counts = torch.empty(op.n_frames, dtype=torch.int64, device=op._torch_device)
        ctx.check(lib.silent_max_value_indices_region_dev(ctx.handle, op.ptr, op.levels, op.n_levels, op.n_frames, reg,
                                                          C.c_void_p(idx.data_ptr()), cap,
                                                          C.c_void_p(counts.data_ptr()), op.stream))
        return idx, counts
    idx = np.empty((op.n_frames, cap, 4), dtype=np.int64)
    counts = np.empty(op.n_frames, dtype=np.int64)
    ctx.check(lib.silent_max_value_indices_region(ctx.handle, op.ptr, op.levels, op.n_levels, op.n_frames, reg,
                                                  C.c_void_p(idx.ctypes.data), cap, C.c_void_p(counts.ctypes.data)))
    return idx, counts


def select_peaks(color, top_percent=0.1, value=None, want=("top", "peaks", "peak_value")):
    """silent_select_peaks: top_value_points -> nms3x3 (product) -> value in one pass.  Returns a dict."""
    op = _Operand(color)
    vptr = None
    if value is not None:
        vop = _Operand(value, channels=1)
        _same_geometry(op, vop, "select_peaks")
        vptr = vop.ptr
    outs, ptrs = {}, {}
    for name, ch in (("top", op.c), ("peaks", op.c), ("peak_value", 1)):
        if name in want:
            outs[name], ptrs[name] = op.alloc(ch)
        else:
            ptrs[name] = None
    lib, ctx = _lib.load(), op.ctx
    args = (ctx.handle, op.ptr, vptr) + op.geom() + (op.c, float(top_percent), ptrs["top"], ptrs["peaks"], ptrs["peak_value"])
    ctx.check(lib.silent_select_peaks_dev(*(args + (op.stream,))) if op.dev else lib.silent_select_peaks(*args))
    return {n: op.wrap(o, 1 if n == "peak_value" else op.c) for n, o in outs.items()}


def centroids(value, region_h, region_w):
    """silent_centroids: (L1 distance map like ``value``, total_pool as a tensor of cells)."""
    op = _Operand(value, channels=1)
    rh, rw = int(region_h), int(region_w)
    if rh < 1 or rw < 1:
        raise ValueError("region extents must be >= 1")
    cell_ext = [(-(-h // rh), -(-w // rw)) for h, w in op.extents]
    n_cells = op.n_frames * sum(a * b for a, b in cell_ext)
    dist, dptr = op.alloc(1)
    lib, ctx = _lib.load(), op.ctx
    if op.dev:
        import torch
        tot = torch.empty(n_cells, dtype=torch.float32, device=op._torch_device)
        tptr = C.c_void_p(tot.data_ptr())
    else:
        tot = np.empty(n_cells, dtype=np.float32)
        tptr = C.c_void_p(tot.ctypes.data)
    args = (ctx.handle, op.ptr) + op.geom() + (rh, rw, dptr, tptr)
    ctx.check(lib.silent_centroids_dev(*(args + (op.stream,))) if op.dev else lib.silent_centroids(*args))
    if op.packed:
        return op.wrap(dist, 1), PackedPyramid(tot, cell_ext, 1, op.n_frames)
    return op.wrap(dist, 1), tot.reshape(op.n_frames, cell_ext[0][0], cell_ext[0][1], 1)


def _require_inplace(state):
    """The boosting state is updated in place: it has to be float32 and contiguous already (no hidden copy)."""
    data = state.data if isinstance(state, PackedPyramid) else state
    if isinstance(data, np.ndarray):
        ok = data.dtype == np.float32 and data.flags["C_CONTIGUOUS"] and data.flags["WRITEABLE"]
    elif is_torch_tensor(data):
        import torch
        ok = data.dtype == torch.float32 and data.is_contiguous()
    else:
        raise TypeError(TYPE_ERROR_MESSAGE)
    if not ok:
        raise ValueError("the boosting state must be a writable, contiguous float32 tensor (it is updated in place)")


def boosting_step(x, energy, exhaustion_max=1.0, excitation_max=1.0, recovery_mode=_lib.RECOVERY_CONSTANT,
                  visualize=False, recovery_amount=10.0, recovery_percentage=0.8):
    """silent_boosting_step: advances ``energy`` in place, returns (fired map, energy map) -- 3 channels each when
    ``visualize``."""
    _require_inplace(energy)
    op, st = _Operand(x, channels=1), _Operand(energy, channels=1)
    _same_geometry(op, st, "boosting_step")
    c = 3 if visualize else 1
    fired, fptr = op.alloc(c)
    eout, eptr = op.alloc(c)
    params = _lib.BoostingParams(float(exhaustion_max), float(excitation_max), int(recovery_mode),
                                 float(recovery_amount), float(recovery_percentage), 1 if visualize else 0)
    lib, ctx = _lib.load(), op.ctx
    args = (ctx.handle, op.ptr) + op.geom() + (C.byref(params), st.ptr, fptr, eptr)
    ctx.check(lib.silent_boosting_step_dev(*(args + (op.stream,))) if op.dev else lib.silent_boosting_step(*args))
    return op.wrap(fired, c), op.wrap(eout, c)


def affine_clip(x, mul=1.0, add=0.0, lo=-float("inf"), hi=float("inf"), post_add=0.0, div=1.0):
    """silent_affine_clip: clip(x * mul / div + add, lo, hi) + post_add, any channel count."""
    op = _Operand(x)
    out, optr = op.alloc(op.c)
    params = _lib.AffineParams(float(mul), float(div), float(add), float(lo), float(hi), float(post_add))
    lib, ctx = _lib.load(), op.ctx
    n = op.n_frames * op.frame_px * op.c
    args = (ctx.handle, op.ptr, n, C.byref(params), optr)
    ctx.check(lib.silent_affine_clip_dev(*(args + (op.stream,))) if op.dev else lib.silent_affine_clip(*args))
    return op.wrap(out, op.c)


def gather_to_host(tensors):
    """Contiguous float32 GPU tensors (torch) -> one flat float32 np.ndarray holding them back to back
    (silent_gather_d2h: n async device-to-host copies on the tensors' current stream, ONE synchronisation)."""
    import torch
    n = len(tensors)
    if n == 0:
        return np.empty(0, np.float32)
    dev = tensors[0].device
    for t in tensors:
        if not (t.is_cuda and t.device == dev and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("gather_to_host: contiguous float32 tensors on one GPU expected")
    ctx = get_context(dev.index)
    host = np.empty(sum(t.numel() for t in tensors), np.float32)
    srcs = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
    sizes = (C.c_size_t * n)(*[t.numel() * 4 for t in tensors])
    ctx.check(ctx._lib.silent_gather_d2h(ctx.handle, C.c_void_p(host.ctypes.data), srcs, sizes, n,
                                          C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return host


def resize_nearest(x, out_extents):
    """silent_resize_nearest.  ``out_extents``: one (h, w) for an NHWC tensor, one per level for a PackedPyramid."""
    op = _Operand(x)
    if not op.packed:
        out_extents = [tuple(out_extents)] if np.ndim(out_extents[0]) == 0 else list(out_extents)
    out_extents = [(int(h), int(w)) for h, w in out_extents]
    if len(out_extents) != op.n_levels or min(min(e) for e in out_extents) < 1:
        raise ValueError("resize_nearest: need one positive (h, w) per level")
    out_levels = (_lib.Extent * op.n_levels)(*[_lib.Extent(h, w) for h, w in out_extents])
    n = op.n_frames * sum(h * w for h, w in out_extents) * op.c
    lib, ctx = _lib.load(), op.ctx
    if op.dev:
        import torch
        out = torch.empty(n, dtype=torch.float32, device=op._torch_device)
        optr = C.c_void_p(out.data_ptr())
    else:
        out = np.empty(n, dtype=np.float32)
        optr = C.c_void_p(out.ctypes.data)
    args = (ctx.handle, op.ptr) + op.geom() + (op.c, out_levels, optr)
    ctx.check(lib.silent_resize_nearest_dev(*(args + (op.stream,))) if op.dev else lib.silent_resize_nearest(*args))
    if op.packed:
        return PackedPyramid(out, out_extents, op.c, op.n_frames)
    return out.reshape(op.n_frames, out_extents[0][0], out_extents[0][1], op.c)


def rgb_line_end(x, kernels, regulation_value=1.0, regulation_root=0.1, flat_policy="ieee", clip_hi=255.0, pad=2,
                 want=("orient", "line_end", "value")):
    """The reference graph recognition_testing.py:69-77 on 3-channel levels.  ``kernels``: dict with
    rgc, rgby, stripe, end ([3,3,3,3]) and blur ([7,7,3,3]).  Returns dict of the requested outputs."""
    op = _Operand(x, channels=3)
    ks = {}
    for name, shape in (("rgc", (3, 3, 3, 3)), ("rgby", (3, 3, 3, 3)), ("stripe", (3, 3, 3, 3)),
                        ("blur", (7, 7, 3, 3)), ("end", (3, 3, 3, 3))):
        k = _kernel_arg(kernels[name], 3)
        if k.shape != shape:
            raise ValueError("kernel %s must have shape %s, got %s" % (name, shape, k.shape))
        ks[name] = k
    fp = C.POINTER(C.c_float)
    params = _lib.RgbChainParams(*[ks[n].ctypes.data_as(fp) for n in ("rgc", "rgby", "stripe", "blur", "end")],
                                 float(regulation_value), float(regulation_root),
                                 {"ieee": _lib.FLAT_IEEE, "zero": _lib.FLAT_ZERO}[flat_policy], float(clip_hi), int(pad))
    outs, ptrs = {}, {}
    for name, ch in (("orient", 3), ("line_end", 3), ("value", 1)):
        if name in want:
            outs[name], ptrs[name] = op.alloc(ch)
        else:
            ptrs[name] = None
    lib, ctx = _lib.load(), op.ctx
    args = (ctx.handle, op.ptr) + op.geom() + (C.byref(params), ptrs["orient"], ptrs["line_end"], ptrs["value"])
    ctx.check(lib.silent_rgb_line_end_dev(*(args + (op.stream,))) if op.dev else lib.silent_rgb_line_end(*args))
    return {n: op.wrap(o, 1 if n == "value" else 3) for n, o in outs.items()}


# ----------------------------------------------------------------------------- pyramid plans

class FrameDisplayer(object):
    """silent_displayer: one camera frame -> the six fetched tensors of the reference's graph, ONE library call per frame
    (recognition_testing.py:106-144 as a replayed HIP graph; include/silent_hip.h).  ``frame_shape`` (H, W, 3), ``dtype`` the
    frames' NumPy dtype, ``output_size`` (w, h) and ``zoom_ratio`` as PyramidDisplayer takes them."""

    _DT = {"uint8": _lib.DT_U8, "float32": _lib.DT_F32, "float64": _lib.DT_F64, "int32": _lib.DT_I32, "uint16": _lib.DT_U16,
           "int16": _lib.DT_I16, "int64": _lib.DT_I64}

    def __init__(self, frame_shape, dtype, output_size, zoom_ratio, kernels, centroid_region=(3, 3), recovery_mode=_lib.RECOVERY_CONSTANT,
                 flat_policy="ieee", clip_hi=255.0, pad=2, device=None):
        from .util.zoom.from_image import reference_levels
        h, w, c = (int(v) for v in frame_shape)
        if c != 3:
            raise ValueError("FrameDisplayer takes [H, W, 3] frames")
        self.dtype = np.dtype(dtype)
        if self.dtype.name not in self._DT:
            raise TypeError("frames of dtype %s are not supported" % self.dtype)
        self.frame_shape = (h, w, 3)
        self.ctx = get_context(device)
        levels = reference_levels((h, w), output_size, zoom_ratio)
        if not levels:
            raise ValueError("the frame is not larger than output_size: image_to_zoom_tensor has no level (from_image.py:45-46)")
        arr = (_lib.PyrLevel * len(levels))(*[_lib.PyrLevel(*[int(v) for v in l]) for l in levels])
        self._kernels = {k: np.ascontiguousarray(kernels[k], np.float32) for k in ("rgc", "rgby", "stripe", "blur", "end")}
        fp = C.POINTER(C.c_float)
        chain = _lib.RgbChainParams(*[self._kernels[k].ctypes.data_as(fp) for k in ("rgc", "rgby", "stripe", "blur", "end")],
                                    1.0, 0.1, {"ieee": _lib.FLAT_IEEE, "zero": _lib.FLAT_ZERO}[flat_policy], float(clip_hi), int(pad))
        boost = _lib.BoostingParams(1.0, 1.0, int(recovery_mode), 10.0, 0.8, 1)
        prm = _lib.DisplayerParams(h, w, self._DT[self.dtype.name], int(centroid_region[0]), int(centroid_region[1]), chain, boost)
        self.handle = C.c_void_p()
        lib = self._lib = _lib.load()
        self.ctx.check(lib.silent_displayer_create(self.ctx.handle, C.byref(prm), arr, len(levels), C.byref(self.handle)))
        shape, floats = (C.c_int32 * 7)(), (C.c_size_t * 6)()
        self.ctx.check(lib.silent_displayer_shape(self.handle, shape, floats))
        L, lh, lw, ch, cw, hh, hw = (int(v) for v in shape)
        self.shapes = [(L, lh, lw, 3), (L, lh, lw, 1), (L, hh, hw, 1), (L, ch, cw, 3), (L, ch, cw, 3), (L, lh, lw, 3)]
        self.state_shape = (L, ch, cw, 1)
        assert [int(np.prod(sh)) for sh in self.shapes] == [int(f) for f in floats]
        self._res = (C.c_void_p * 6)()
        self._ms = C.c_float(0)
        self.gpu_ms = 0.0
        self._exports = []                  # weak references to the ctypes buffers handed out as views (frame_buffer, step results)
        ptr, nbytes = C.c_void_p(), C.c_size_t(0)
        self.ctx.check(lib.silent_displayer_input(self.handle, C.byref(ptr), C.byref(nbytes)))
        buf = (C.c_char * nbytes.value).from_address(ptr.value)
        buf._owner = self
        self._exports.append(weakref.ref(buf))
        #: the displayer's pinned input buffer as an [H, W, 3] array: a capture loop that grabs INTO it (``cap.read(disp.frame_buffer)``,
        #: ``np.copyto``) and passes it to ``step`` is uploaded without the staging copy
        self.frame_buffer = np.frombuffer(buf, self.dtype).reshape(self.frame_shape)

    #: result slots a displayer may hold for ``step(hold=True)`` before it falls back to copying (3.55 MB each at 640 x 480)
    MAX_HELD_SLOTS = 8

    def step(self, frame, copy=False, timing=False, hold=False):
        """frame: [H, W, 3] ndarray of the displayer's dtype.  Returns the six float32 arrays, in one of three ways:

        ``hold=True`` -- what ``LineEndDisplayer.callback`` hands out: arrays that stay valid FOR AS LONG AS THEY ARE REFERENCED, like the
        fresh arrays of the reference's session.run (recognition_testing.py:132), without a copy: the frame is stepped into a pinned
        result slot nobody references any more (the displayer keeps weak references to what it handed out; slots are added on demand,
        at most MAX_HELD_SLOTS); when every slot is still held the frame goes through the two alternating slots and is COPIED out.
        ``copy=False`` (default) -- the raw form: views of one of two alternating slots, valid until the second next raw step.
        ``copy=True`` -- fresh arrays by one memcpy of the slot.
        All views (and ``frame_buffer``) keep the displayer alive; ``close`` defers the free while they exist.
        ``timing``: also measure the device time of the frame (``gpu_ms``; HIP events around the graph and a stream synchronisation
        instead of the poll of the frame's completion word: ~0.02 ms more wall time)."""
        if not getattr(self, "handle", None):
            raise RuntimeError("the displayer is closed")
        if not isinstance(frame, np.ndarray) or frame.dtype != self.dtype or tuple(frame.shape) != self.frame_shape:
            raise ValueError("frame must be a %s ndarray of shape %s" % (self.dtype, self.frame_shape,))
        f = frame if frame.flags["C_CONTIGUOUS"] else np.ascontiguousarray(frame)
        ms = C.byref(self._ms) if timing else None
        slot = None
        if hold:
            slot = self._free_slot()
            if slot is None:
                copy = True                     # every slot is still referenced by somebody: this frame is copied out
        if slot is None:
            self.ctx.check(self._lib.silent_displayer_step(self.handle, C.c_void_p(f.ctypes.data), self._res, ms))
        else:
            self.ctx.check(self._lib.silent_displayer_step_slot(self.handle, C.c_void_p(f.ctypes.data), slot, self._res, ms))
        if timing:
            self.gpu_ms = float(self._ms.value)
        out = []
        if copy and slot is None:
            # ONE copy of the whole slot (the six results lie back to back, 64-byte steps), then views of the copy
            ptrs = [int(p) for p in self._res]
            total = (max(ptrs) - min(ptrs)) // 4 + int(np.prod(self.shapes[ptrs.index(max(ptrs))]))
            whole = np.frombuffer((C.c_float * total).from_address(min(ptrs)), np.float32).copy()
            for ptr, sh in zip(ptrs, self.shapes):
                o = (ptr - min(ptrs)) // 4
                out.append(whole[o:o + int(np.prod(sh))].reshape(sh))
            return out
        refs = self._exports if slot is None else self._held[slot]
        for ptr, sh in zip(self._res, self.shapes):
            buf = (C.c_float * int(np.prod(sh))).from_address(ptr)
            buf._owner = self               # the views keep the displayer (and with it the pinned slot) alive
            refs.append(weakref.ref(buf))
            out.append(np.frombuffer(buf, np.float32).reshape(sh))
        if len(self._exports) > 64:
            self._exports = [r for r in self._exports if r() is not None]
        return out

    def _free_slot(self):
        """A held-results slot (numbers 2, 3, ...: 0 and 1 are the raw alternating pair) nobody references any more; a new one while
        fewer than MAX_HELD_SLOTS exist; None when all are taken."""
        held = self.__dict__.setdefault("_held", {})
        for slot, refs in held.items():
            if not any(r() is not None for r in refs):
                del refs[:]
                return slot
        if len(held) >= self.MAX_HELD_SLOTS:
            return None
        idx = C.c_int(-1)
        self.ctx.check(self._lib.silent_displayer_add_slot(self.handle, C.byref(idx)))
        held[int(idx.value)] = []
        return int(idx.value)

    def get_state(self):
        st = np.empty(self.state_shape, np.float32)
        self.ctx.check(self._lib.silent_displayer_get_state(self.handle, C.c_void_p(st.ctypes.data)))
        return st

    def set_state(self, state):
        st = np.ascontiguousarray(state, np.float32)
        if tuple(st.shape) != self.state_shape:
            raise ValueError("state shape %s does not match the compiled pyramid %s" % (st.shape, self.state_shape))
        self.ctx.check(self._lib.silent_displayer_set_state(self.handle, C.c_void_p(st.ctypes.data)))

    def close(self):
        """Free the displayer (graphs, device buffers, the pinned frame buffer and result slots).  While views handed out by
        ``step(copy=False)`` or ``frame_buffer`` are still referenced the free is DEFERRED: every view keeps the displayer alive
        (``_owner``), and it is destroyed when the last of them goes -- never under an array somebody still holds.  Further
        ``step`` calls raise either way."""
        if not getattr(self, "handle", None):
            return
        fb = self.__dict__.pop("frame_buffer", None)      # our own reference to the input view does not count
        del fb
        live = list(getattr(self, "_exports", ())) + [r for refs in getattr(self, "_held", {}).values() for r in refs]
        if any(r() is not None for r in live):
            self._deferred, self.handle = self.handle, C.c_void_p()
            return
        self._lib.silent_displayer_destroy(self.handle)
        self.handle = C.c_void_p()

    def __del__(self):
        try:
            h = getattr(self, "_deferred", None) or getattr(self, "handle", None)
            if h:                           # nobody references the displayer any more, hence no view either
                self._lib.silent_displayer_destroy(h)
                self.handle = self._deferred = C.c_void_p()
        except Exception:
            pass


class PyramidPlan(object):
    """Tap tables of one (frame size, level geometry) on the device.  ``levels`` is a list of dicts / tuples
    (src_y0, src_x0, src_h, src_w, zoom_h, zoom_w, out_h, out_w)."""

    def __init__(self, frame_h, frame_w, channels, levels, device=None):
        self.ctx = get_context(device)
        self.frame_shape = (int(frame_h), int(frame_w), int(channels))
        self.levels = [tuple(int(v) for v in l) for l in levels]
        self.extents = [(l[6], l[7]) for l in self.levels]
        arr = (_lib.PyrLevel * len(self.levels))(*[_lib.PyrLevel(*l) for l in self.levels])
        self.handle = C.c_void_p()
        self.ctx.check(_lib.load().silent_pyramid_plan_create(self.ctx.handle, self.frame_shape[0], self.frame_shape[1],
                                                              self.frame_shape[2], arr, len(self.levels),
                                                              C.byref(self.handle)))
        self.frame_px = sum(h * w for h, w in self.extents)

    def run(self, frames):
        """frames: [n, H, W, C] ndarray (host) or torch GPU tensor.  Returns a PackedPyramid."""
        h, w, c = self.frame_shape
        lib = _lib.load()
        if is_torch_tensor(frames):
            import torch
            if tuple(frames.shape[1:]) != (h, w, c):
                raise ValueError("frames must be [n, %d, %d, %d]" % (h, w, c))
            f = as_float32(frames)
            n = int(f.shape[0])
            out = torch.empty(n * self.frame_px * c, dtype=torch.float32, device=f.device)
            stream = C.c_void_p(torch.cuda.current_stream(f.device).cuda_stream)
            self.ctx.check(lib.silent_pyramid_dev(self.ctx.handle, self.handle, C.c_void_p(f.data_ptr()), n,
                                                  C.c_void_p(out.data_ptr()), stream))
            return PackedPyramid(out, self.extents, c, n)
        if not isinstance(frames, np.ndarray):
            raise TypeError(TYPE_ERROR_MESSAGE)
        if tuple(frames.shape[1:]) != (h, w, c):
            raise ValueError("frames must be [n, %d, %d, %d], got %s" % (h, w, c, frames.shape))
        f = np.ascontiguousarray(frames, dtype=np.float32)
        n = f.shape[0]
        out = np.empty(n * self.frame_px * c, dtype=np.float32)
        self.ctx.check(lib.silent_pyramid(self.ctx.handle, self.handle, C.c_void_p(f.ctypes.data), n,
                                          C.c_void_p(out.ctypes.data)))
        return PackedPyramid(out, self.extents, c, n)

    @property
    def streamable(self):
        """True when silent_gray_pass takes the single-read stream kernel for this plan."""
        return bool(_lib.load().silent_pyramid_plan_is_streamable(self.handle))

    @property
    def walk_plans(self):
        """(number of walk plans, pixels per consumer wave) of a 3-channel plan; (0, 0): unit + region kernels."""
        px = C.c_int(0)
        n = _lib.load().silent_pyramid_plan_walk_plans(self.handle, C.byref(px))
        return int(n), int(px.value)

    def gray_pass(self, frames, cs_kernel, end_bank, clip_hi=255.0):
        """Whole grayscale hot path (silent_gray_pass): frames [n,H,W,1] -> (pyramid, cs, end) PackedPyramids.
        Same results as run() + gray_line_end(), one pass less over level 0."""
        h, w, c = self.frame_shape
        if c != 1:
            raise ValueError("gray_pass needs a single-channel plan")
        cs = _kernel_arg(cs_kernel, 1)
        eb = _kernel_arg(end_bank, 1)
        if cs.shape != (3, 3, 1, 1) or eb.shape[:3] != (3, 3, 1):
            raise ValueError("gray_pass needs a [3,3,1,1] CS kernel and a [3,3,1,K] end bank")
        K = eb.shape[3]
        lib = _lib.load()
        if is_torch_tensor(frames):
            import torch
            if tuple(frames.shape[1:]) != (h, w, c):
                raise ValueError("frames must be [n, %d, %d, 1]" % (h, w))
            f = as_float32(frames)
            n = int(f.shape[0])
            mk = lambda ch: torch.empty(n * self.frame_px * ch, dtype=torch.float32, device=f.device)
            pyr, cso, endo = mk(1), mk(1), mk(K)
            stream = C.c_void_p(torch.cuda.current_stream(f.device).cuda_stream)
            self.ctx.check(lib.silent_gray_pass_dev(self.ctx.handle, self.handle, C.c_void_p(f.data_ptr()), n,
                                                    C.c_void_p(cs.ctypes.data), C.c_void_p(eb.ctypes.data), K,
                                                    float(clip_hi), C.c_void_p(pyr.data_ptr()),
                                                    C.c_void_p(cso.data_ptr()), C.c_void_p(endo.data_ptr()), stream))
        else:
            if not isinstance(frames, np.ndarray):
                raise TypeError(TYPE_ERROR_MESSAGE)
            if tuple(frames.shape[1:]) != (h, w, c):
                raise ValueError("frames must be [n, %d, %d, 1], got %s" % (h, w, frames.shape))
            f = np.ascontiguousarray(frames, dtype=np.float32)
            n = f.shape[0]
            mk = lambda ch: np.empty(n * self.frame_px * ch, dtype=np.float32)
            pyr, cso, endo = mk(1), mk(1), mk(K)
            self.ctx.check(lib.silent_gray_pass(self.ctx.handle, self.handle, C.c_void_p(f.ctypes.data), n,
                                                C.c_void_p(cs.ctypes.data), C.c_void_p(eb.ctypes.data), K,
                                                float(clip_hi), C.c_void_p(pyr.ctypes.data),
                                                C.c_void_p(cso.ctypes.data), C.c_void_p(endo.ctypes.data)))
        P = PackedPyramid
        return P(pyr, self.extents, 1, n), P(cso, self.extents, 1, n), P(endo, self.extents, K, n)

    def close(self):
        if self.handle:
            _lib.load().silent_pyramid_plan_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
