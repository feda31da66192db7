"""``LineEndDisplayer``: the reference's end-to-end application graph on the GPU.

Mirror of slam_recognition/recognition_testing.py:21-144: ``callback(frame)`` builds
the zoom pyramid (zoom.from_image, :141-142) and ``run`` evaluates what the reference fetches from its TF graph
(:99-100, :132):

    [orient_tensor, 255 - centroids * 255, 255 - centroids2 * 255, fired_importants * 255, update_importances,
     padded_line_end_tensor]

every one as a float32 [levels, ...] array; ``callback`` returns ``[frame] + 6 lists of per-level images`` (:144).
The boosting state (``energy_values``, a tf.Variable in the reference, :56) is ``self.energy_values``, a GPU tensor
that every ``run`` advances; like the reference (:108-117) it is re-initialised to 8 when the pyramid shape changes.
All intermediate maps stay on the GPU; only the six results are copied back.
"""
import math as m

import numpy as np

from . import _runtime
from .pipeline import default_constants
from .pyramid_displayer import PyramidDisplayer
from .util import zoom
from .util.energy.recovery import recovery_mode


class LineEndDisplayer(PyramidDisplayer):
    def __init__(self, n_dimensions=2, use_graph=False, native=True, **argv):
        """``native`` (default): ``callback`` hands the camera frame to ONE library call (silent_displayer_step: upload, cast,
        pyramid, the whole graph, download -- a HIP graph replayed per frame, buffers and boosting state owned by the library;
        _runtime.FrameDisplayer).  Like the reference's session.run (recognition_testing.py:132) ``callback`` returns arrays that
        are the caller's for as long as it holds them -- without a copy: the frame is computed into a pinned result slot nobody
        references any more (``FrameDisplayer.step(hold=True)``; up to eight slots, then copies).  ``callback(frame, copy=False)``
        returns the raw views of two alternating slots instead, valid until the second next frame.  ``native=False``: the
        per-op path below (``run`` / ``run_device`` always take it: they start from a pyramid, not from a frame).
        ``use_graph`` (per-op path): capture the ~20 launches of one frame into a HIP graph the first time a pyramid shape is
        seen and replay it per frame (torch.cuda.CUDAGraph is only the capture / replay plumbing; every node is one
        of this library's kernels)."""
        super(LineEndDisplayer, self).__init__(**argv)
        self.use_graph = bool(use_graph)
        self.native = bool(native)
        self._native = None                 # (frame shape, dtype) -> FrameDisplayer
        self._graph = None
        self._ctx = None        # graph mode: a private silent_ctx, so that no other caller regrows the captured workspace
        if n_dimensions != 2:
            raise ValueError("only 2-D images are supported")
        self.kernels = default_constants("rgb")
        self.simplex_end_stop = self.kernels["end"]
        self.constant_recovery = True
        self.input_based_recovery = False
        self.centroid_region_shape = [1, 3, 3]
        self.pyramid_tensor_shape = None
        self.energy_values = None
        self.device_index = _runtime.default_device()

    # -- state -------------------------------------------------------------------------------------------
    def pre_compile(self, pyramid_tensor):
        """Initial boosting state: 8 in every centroid cell (recognition_testing.py:45-58)."""
        import torch
        n, h, w = (int(s) for s in pyramid_tensor.shape[:3])
        rh, rw = self.centroid_region_shape[1], self.centroid_region_shape[2]
        self.energy_values = torch.full((n, -(-h // rh), -(-w // rw), 1), 8.0, dtype=torch.float32,
                                        device=torch.device("cuda", self.device_index))
        self.pyramid_tensor_shape = tuple(pyramid_tensor.shape)
        self._graph = None

    def get_state(self):
        """Host copy of the boosting state (checkpoint); None before the first run."""
        if self._native is not None:
            return self._native[1].get_state()
        return None if self.energy_values is None else self.energy_values.cpu().numpy()

    def set_state(self, state):
        import torch
        if self._native is not None:
            return self._native[1].set_state(state)
        state = np.ascontiguousarray(state, np.float32)
        if self.energy_values is None or tuple(state.shape) != tuple(self.energy_values.shape):
            raise ValueError("state shape %s does not match the compiled pyramid" % (state.shape,))
        self.energy_values.copy_(torch.from_numpy(state))

    # -- the graph -----------------------------------------------------------------------------------------
    def run_device(self, pyramid_tensor):
        """The six fetched tensors as GPU tensors."""
        import torch
        rt = _runtime
        if _runtime.is_torch_tensor(pyramid_tensor):
            x = rt.as_float32(pyramid_tensor)
        else:
            x = torch.from_numpy(np.ascontiguousarray(pyramid_tensor, np.float32)).to(
                torch.device("cuda", self.device_index))
        if x.ndim != 4 or x.shape[3] != 3:
            raise ValueError("pyramid tensor must be [levels, h, w, 3]")
        if self.pyramid_tensor_shape != tuple(x.shape):
            self.pre_compile(x)
        if self.use_graph:
            if self._ctx is None:
                self._ctx = _runtime.Context(self.device_index)
            with _runtime.use_context(self._ctx):
                return self._replay(x)
        return self._launch(x)

    def _replay(self, x):
        import torch
        if self._graph is None:
            static_in = torch.empty_like(x)
            static_in.copy_(x)
            saved = self.energy_values.clone()
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side):          # eager warm-up: grows the library's workspace outside the capture
                self._launch(static_in)
            torch.cuda.current_stream(x.device).wait_stream(side)
            self.energy_values.copy_(saved)        # the warm-up advanced the boosting state: put it back
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self._launch(static_in)
            self._graph = (graph, static_in, static_out)
        graph, static_in, static_out = self._graph
        static_in.copy_(x)
        graph.replay()
        return static_out

    def _launch(self, x):
        rt = _runtime
        ch = rt.rgb_line_end(x, self.kernels)
        gray = ch["value"]
        centroids, importances = rt.centroids(rt.affine_clip(gray, div=255.0), *self.centroid_region_shape[1:])
        importances = rt.affine_clip(importances, 255 / 4.0, 0.0, 1.0, 256.0, -1.0)
        root_e = np.float32(m.e ** .5)
        half = (int(np.float32(x.shape[1]) / root_e), int(np.float32(x.shape[2]) / root_e))
        im2 = rt.resize_nearest(gray, half)
        centroids2, _ = rt.centroids(rt.affine_clip(im2, div=255.0), *self.centroid_region_shape[1:])
        fired, update = rt.boosting_step(importances, self.energy_values, 1.0, 1.0,
                                         recovery_mode(self.input_based_recovery, self.constant_recovery), True)
        return [ch["orient"], rt.affine_clip(centroids, -255.0, 255.0), rt.affine_clip(centroids2, -255.0, 255.0),
                rt.affine_clip(fired, 255.0), update, ch["line_end"]]

    def run(self, pyramid_tensor):
        import torch
        outs = self.run_device(pyramid_tensor)
        if any(t.dtype != torch.float32 for t in outs):
            return [t.cpu().numpy() for t in outs]
        # the six maps come back through ONE library call (silent_gather_d2h: six async copies into one host buffer,
        # one synchronisation) -- no torch kernel anywhere on this path
        host = _runtime.gather_to_host(outs)
        res, o = [], 0
        for t in outs:
            res.append(host[o:o + t.numel()].reshape(tuple(t.shape)))
            o += t.numel()
        return res

    def _native_for(self, frame):
        key = (tuple(frame.shape), frame.dtype.str)
        if self._native is None or self._native[0] != key:
            # (like the reference, recognition_testing.py:108-117: a new frame shape compiles anew and re-initialises the state)
            # (the old displayer is DROPPED, not closed: arrays a consumer still holds from callback(copy=False) / frame_buffer keep
            # it -- and the pinned memory under them -- alive; it is destroyed with the last of them)
            self._native = (key, _runtime.FrameDisplayer(
                frame.shape, frame.dtype, self.output_size, self.zoom_ratio, self.kernels, self.centroid_region_shape[1:],
                recovery_mode(self.input_based_recovery, self.constant_recovery), device=self.device_index))
        return self._native[1]

    def callback(self, frame, cam_id=None, depth=2, copy=True):
        import torch
        if self.native and isinstance(frame, np.ndarray) and frame.ndim == 3 and frame.shape[2] == 3 and self.output_colors == 3 \
                and frame.dtype.name in _runtime.FrameDisplayer._DT:
            tensors = self._native_for(frame).step(frame, hold=bool(copy))
            return [frame] + [[tensors[x][y] for y in range(len(tensors[x]))] for x in range(6)]
        # frame -> GPU once; the zoom pyramid stays on the device between from_image and the graph
        dev = torch.device("cuda", self.device_index)
        if isinstance(frame, np.ndarray) and frame.dtype == np.uint8:
            # camera frames: 1 byte per sample over PCIe, widened on the device (exact)
            z_tensor = _runtime.as_float32(torch.from_numpy(np.ascontiguousarray(frame)).to(dev))
        else:
            z_tensor = torch.from_numpy(np.ascontiguousarray(frame, dtype=np.float32)).to(dev)
        z_tensor = zoom.from_image(z_tensor, self.output_colors, self.output_size, self.zoom_ratio)
        tensors = self.run(z_tensor)
        return [frame] + [[tensors[x][y] for y in range(len(tensors[x]))] for x in range(6)]
