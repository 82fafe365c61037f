"""PathTracer — thin Python handle on a pt_ctx (include/ptrace.h).

Mirrors how the reference drives its GPU boundary: set_geometry once (src/lib.rs:57), run the
uniform setters and render every frame (src/lib.rs:92-102), read the result out.  Device memory
and streams can come from PyTorch (`use_torch=True`): the accumulation buffer is then a torch
tensor bound with pt_bind_accum and kernels run on torch's current stream, which is what the
multi-GPU gather (dist.py) and bench.py's timing need.  PyTorch is plumbing here, not compute.
"""
import ctypes as C

import numpy as np

from . import abi
from ._lib import load


class PtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libptrace error %d: %s" % (code, message))
        self.code = code


class PathTracer:
    def __init__(self, width, height, device=0, use_torch=False):
        self.lib = load()
        self.width, self.height = int(width), int(height)
        self.device = int(device)
        self._ctx = C.c_void_p()
        self.use_torch = bool(use_torch)
        stream_handle = None
        if self.use_torch:
            import torch

            self._torch = torch
            try:
                stream = torch.cuda.current_stream(self.device)
            except RuntimeError as e:  # e.g. "No HIP GPUs are available": a second HIP runtime in this process (_lib.py)
                raise PtError(abi.PT_ERR_NO_DEVICE, "PyTorch cannot see the GPU libptrace is using (%s): two HIP runtimes in one "
                              "process?  `import torch` before anything of ray_tracer_webgl_amd" % e) from e
            # torch's default stream is the NULL stream, and NULL means "the context's own stream" in this
            # ABI: name it as hipStreamLegacy (1).  Kernels must run on the stream torch orders its own work
            # on, or a gather / .cpu() right after render_passes reads the buffer before they have written it.
            # The context is CREATED on that stream (pt_create_on_stream): it never makes a stream — an HSA queue,
            # 80-150 ms when it is the process's first — that it would not use.
            stream_handle = C.c_void_p(stream.cuda_stream or abi.PT_STREAM_LEGACY)
            rc = self.lib.pt_create_on_stream(C.byref(self._ctx), self.device, self.width, self.height, stream_handle)
        else:
            rc = self.lib.pt_create(C.byref(self._ctx), self.device, self.width, self.height)
        if rc != abi.PT_OK:
            msg = self.lib.pt_last_error(None)
            self._ctx = C.c_void_p()
            raise PtError(rc, msg.decode() if msg else "pt_create failed")
        self.params = None
        self.local_rows = self.height
        self.accum_tensor = None
        self._keep = None

    # -- plumbing -------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != abi.PT_OK:
            msg = self.lib.pt_last_error(self._ctx)
            raise PtError(rc, msg.decode() if msg else "")
        return rc

    def close(self):
        if self._ctx:
            self.lib.pt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- scene / uniforms -----------------------------------------------------------------------
    def set_spheres(self, spheres):
        ptr, n, keep = abi.spheres_as_ctypes(spheres)
        self._check(self.lib.pt_set_spheres(self._ctx, ptr, n))
        self.n_spheres = n

    def set_params(self, params):
        self.params = params.copy()
        rows = abi.local_rows(self.height, params.band_rows, params.band_index, params.band_count)
        if self.use_torch and self.accum_tensor is not None and self.accum_tensor.shape[0] != max(rows, 1):
            # a different row partition: give the old tensor back before the context resizes
            self._check(self.lib.pt_bind_accum(self._ctx, None, 0))
            self.accum_tensor = None
        self._check(self.lib.pt_set_params(self._ctx, C.byref(self.params)))
        self.local_rows = rows
        if self.use_torch:
            self._bind_torch_accum()

    def _bind_torch_accum(self):
        torch = self._torch
        shape = (max(self.local_rows, 1), self.width, 4)
        if self.accum_tensor is None or tuple(self.accum_tensor.shape) != shape:
            self.accum_tensor = torch.zeros(shape, dtype=torch.float32, device="cuda:%d" % self.device)
            nbytes = self.accum_tensor.numel() * 4
            self._check(self.lib.pt_bind_accum(self._ctx, C.c_void_p(self.accum_tensor.data_ptr()), nbytes))

    def set_geometry_path(self, path):
        """abi.PT_GEOM_AUTO (default: measure the usable paths once per scene) / PT_GEOM_LDS /
        PT_GEOM_SCALAR / PT_GEOM_BVH / PT_GEOM_GRID."""
        self._check(self.lib.pt_set_option(self._ctx, abi.PT_OPT_GEOMETRY_PATH, int(path)))

    def set_count_work(self, on=True):
        """Walk kernels: launch the measuring twin, which fills stats().work (executed iterations
        and active lanes per phase).  Slower; never time it."""
        self._check(self.lib.pt_set_option(self._ctx, abi.PT_OPT_COUNT_WORK, int(on)))  # (2: dev tools, + the grid twins' gather histogram)

    def set_carry_lanes(self, n):
        """Walk kernels: move on to shading when fewer than n lanes (and less than half the wave)
        still walk; 0 = lockstep.  Scheduling only: images do not depend on it."""
        self._check(self.lib.pt_set_option(self._ctx, abi.PT_OPT_CARRY_LANES, int(n)))

    def set_refill_min(self, n):
        """Lanes of a busy wave that wait for a new item before the item decode runs (1 = at once).
        Scheduling only: images do not depend on it."""
        self._check(self.lib.pt_set_option(self._ctx, abi.PT_OPT_REFILL_MIN, int(n)))

    def set_russian_roulette(self, min_depth):
        """Opt-in perf mode (0 = off, the default): after `min_depth` bounces a path survives each further
        bounce with probability min(max(throughput), 1) and is re-weighted.  Same expectation per pixel
        as the reference's estimator, different samples: never bit-comparable with the oracle."""
        self._check(self.lib.pt_set_option(self._ctx, abi.PT_OPT_RUSSIAN_ROULETTE, int(min_depth)))

    def set_grid_fit(self, unmeasured):
        """How tune() chooses the grid's margin class: False (default) it times the candidates, True it takes the smallest
        class that covers the camera without launching anything (PT_OPT_GRID_FIT).  Speed only."""
        self._check(self.lib.pt_set_option(self._ctx, abi.PT_OPT_GRID_FIT, 1 if unmeasured else 0))

    def tune(self, n_passes):
        """Fit the context to scene and uniforms: the grid is rebuilt for the margin class that renders this view fastest
        (measured: one timed launch of n_passes passes per candidate class; speed only), then PT_GEOM_AUTO is settled now
        (one cold + one untimed launch of n_passes passes per usable path); clears the accumulation."""
        self._check(self.lib.pt_tune(self._ctx, int(n_passes)))
        if self.accum_tensor is not None:
            self.accum_tensor.zero_()

    def refit_grid(self, only_if_stale=False):
        """Rebuild the grid for the margin class the current camera needs (pt_refit_grid): what a frame loop does when
        grid_fit() says 1.  No measuring launches; accumulation, textures and statistics stay."""
        self._check(self.lib.pt_refit_grid(self._ctx, 1 if only_if_stale else 0))

    def grid_fit(self):
        """0: the grid fits the camera of the last set_params (or there is none / another path is in use); 1: the camera
        stands outside the region whose rays walk the cells — every primary ray is tested against the whole list;
        2: a smaller margin class would do.  Host arithmetic, no synchronisation (PtStats.grid_fit_stale)."""
        rc = self.lib.pt_grid_fit(self._ctx)
        if rc < 0:
            raise PtError(rc, "pt_grid_fit failed")
        return rc

    def reserve_passes(self, n):
        self._check(self.lib.pt_reserve_passes(self._ctx, int(n)))

    # -- rendering ------------------------------------------------------------------------------
    def render(self):
        self._check(self.lib.pt_render(self._ctx))

    def render_passes(self, n):
        self._check(self.lib.pt_render_passes(self._ctx, int(n)))

    def reset(self):
        self._check(self.lib.pt_reset_accum(self._ctx))
        if self.accum_tensor is not None:
            self.accum_tensor.zero_()

    def synchronize(self):
        self._check(self.lib.pt_synchronize(self._ctx))

    def wait(self, timeout_s):
        """Dev tools' watchdog (include/ptrace_dev.h pt_debug_wait): True when the stream went idle within
        timeout_s, False on timeout.  Polls an event, never blocks in the driver."""
        fn = self.lib.pt_debug_wait
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_uint]
        rc = fn(self._ctx, int(timeout_s * 1000))
        if rc < 0:
            raise PtError(rc, "pt_debug_wait failed")
        return rc == 0

    # -- the reference's frame on device-resident textures ---------------------------------------
    def clear_textures(self):
        self._check(self.lib.pt_clear_textures(self._ctx))

    def render_frame(self, even_odd_count):
        """One animation tick (src/lib.rs:92-102) with the current uniforms: trace one pass, blend
        with texture[(even_odd_count + 1) % 2], draw to the canvas (and, when averaging, to the
        other texture).  Asynchronous; nothing crosses PCIe."""
        self._check(self.lib.pt_render_frame(self._ctx, int(even_odd_count) & 0xFFFFFFFF))

    def render_frames(self, even_odd_count, max_render_count, n_frames):
        """n ticks at the constant frame interval params.time_step, replayed from hipGraphs (groups of 64, 16 and 4 frames, then singles) with
        the per-frame state (u_time, render_count, even/odd) counted on the device."""
        self._check(self.lib.pt_render_frames(self._ctx, int(even_odd_count) & 0xFFFFFFFF, int(max_render_count), int(n_frames)))

    def read_canvas(self):
        out = np.empty((self.local_rows, self.width, 4), dtype=np.uint8)
        if out.size:
            self._check(self.lib.pt_read_canvas(self._ctx, out.ctypes.data_as(C.c_void_p)))
        return out

    def read_texture(self, index):
        out = np.empty((self.local_rows, self.width, 4), dtype=np.uint8)
        if out.size:
            self._check(self.lib.pt_read_texture(self._ctx, int(index), out.ctypes.data_as(C.c_void_p)))
        return out

    def write_texture(self, index, rgba8):
        a = np.ascontiguousarray(rgba8, dtype=np.uint8)
        if a.shape != (self.local_rows, self.width, 4):
            raise ValueError("texture is %s, this context holds %s" % (a.shape, (self.local_rows, self.width, 4)))
        self._check(self.lib.pt_write_texture(self._ctx, int(index), a.ctypes.data_as(C.c_void_p)))

    # -- read-out -------------------------------------------------------------------------------
    def resolve(self, gamma=True):
        out = np.empty((self.local_rows, self.width, 4), dtype=np.float32)
        if out.size:
            self._check(self.lib.pt_resolve(self._ctx, out.ctypes.data_as(C.c_void_p), 1 if gamma else 0))
        return out

    def resolve_rgba8(self, gamma=True):
        out = np.empty((self.local_rows, self.width, 4), dtype=np.uint8)
        if out.size:
            self._check(self.lib.pt_resolve_rgba8(self._ctx, out.ctypes.data_as(C.c_void_p), 1 if gamma else 0))
        return out

    def blend_rgba8(self, prev):
        prev = np.ascontiguousarray(prev, dtype=np.uint8)
        out = np.empty_like(prev)
        self._check(self.lib.pt_blend_rgba8(self._ctx, prev.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        return out

    def accum(self):
        """Raw accumulation buffer (local_rows, width, 4) fp32: rgb sums, a = spp."""
        self.synchronize()
        if self.accum_tensor is not None:
            self._torch.cuda.current_stream(self.device).synchronize()
            return self.accum_tensor[: self.local_rows].cpu().numpy()
        out = np.empty((self.local_rows, self.width, 4), dtype=np.float32)
        if out.size:
            self._check(self.lib.pt_read_accum(self._ctx, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def load_accum(self, accum):
        """Resume from a checkpoint: `accum` is what accum() returned earlier (rgb sums, a = spp)
        for the same image size and row partition (image_io.load_accum reads one from disk)."""
        a = np.ascontiguousarray(accum, dtype=np.float32)
        if a.shape != (self.local_rows, self.width, 4):
            raise ValueError("accumulation checkpoint is %s, this context holds %s"
                             % (a.shape, (self.local_rows, self.width, 4)))
        self._check(self.lib.pt_load_accum(self._ctx, a.ctypes.data_as(C.c_void_p), a.nbytes))

    def stats(self):
        st = abi.PtStats()
        self._check(self.lib.pt_get_stats(self._ctx, C.byref(st)))
        return st

    def probe(self, kind, inp, out_per_item, n):
        inp = np.ascontiguousarray(inp, dtype=np.float32)
        out = np.zeros(int(n) * out_per_item, dtype=np.float32)
        self._check(
            self.lib.pt_probe(self._ctx, int(kind), inp.ctypes.data_as(C.c_void_p), inp.size,
                              out.ctypes.data_as(C.c_void_p), out.size, int(n))
        )
        return out


def render_scene(scene, device=0, use_torch=False, passes_per_launch=None, band=None, geometry_path=None, tune=None):
    """Render a scenes.Scene completely; returns (PathTracer, accum ndarray).  `tune` = n: pt_tune(n) after scene and
    uniforms are up, as bench.py does before it times anything — the grid is refitted to the camera and PT_GEOM_AUTO
    settled with n-pass launches (the as-benchmarked path: tests that pin what the bench line times pass it)."""
    p = scene.params.copy()
    if band is not None:
        p.band_rows, p.band_index, p.band_count = band
    pt = PathTracer(p.width, p.height, device=device, use_torch=use_torch)
    if geometry_path is not None:
        pt.set_geometry_path(geometry_path)
    pt.set_spheres(scene.spheres)
    pt.set_params(p)
    per = passes_per_launch or scene.n_passes
    pt.reserve_passes(max(per, int(tune or 0)))
    if tune:
        pt.tune(int(tune))
    done = 0
    while done < scene.n_passes:
        k = min(per, scene.n_passes - done)
        q = p.copy()
        q.first_pass = scene.params.first_pass + done  # u_time = time + float(first_pass + p) * time_step
        pt.set_params(q)
        pt.render_passes(k)
        done += k
    return pt, pt.accum()
