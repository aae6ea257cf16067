"""ctypes binding of librays1.so (include/rays1.h).  No compute happens in Python and
nothing here imports oracle/: this is the product's host-side mirror of the reference's
interface for the hot path (src/step13/rayweek1.cpp:552-719 scene builders, :845 benchmark,
src/common/common.h:36-122 RESULT / log_results / tga_write_rgb24)."""
import ctypes as C
import os
import sys
import subprocess
import time

import numpy as np

_lib = None
HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")

R1_OK, R1_EINVAL, R1_ENODEVICE, R1_EHIP, R1_ENOMEM, R1_ELIMIT = 0, -1, -2, -3, -4, -5
SCENE_SMALL, SCENE_MEDIUM, SCENE_LARGE, SCENE_GRID = 0, 1, 2, 3
VARIANT_DEFAULT, VARIANT_REFERENCE, VARIANT_PREFILTER, VARIANT_STATS, VARIANT_BVH, VARIANT_BVH_STATS, VARIANT_WAVEFRONT = 0, 1, 2, 3, 4, 5, 6



def _stream_arg(stream_ptr):
    """hipStream_t argument of the device-pointer entry points.  0 / None means the CONTEXT'S OWN stream (a non-blocking stream: the
    library never touches the null stream), which is not ordered with torch's default stream — where a caller who passes
    torch.cuda.current_stream().cuda_stream (= 0 for the default stream) has just filled the buffers it hands over.  So for that case
    the pending work of torch's current stream is waited for here (tests and smoke only: bench.py passes streams of its own)."""
    if stream_ptr:
        return C.c_void_p(stream_ptr)
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
        torch.cuda.current_stream().synchronize()
    return None


class R1Error(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"librays1 error {code}: {text}")
        self.code = code


class CScene(C.Structure):
    _fields_ = [("count", C.c_uint32)] + [
        (n, C.POINTER(C.c_float)) for n in ("center_x", "center_y", "center_z", "radius_sq", "inv_radius")
    ] + [("mat_type", C.POINTER(C.c_uint8))] + [
        (n, C.POINTER(C.c_float)) for n in ("albedo_r", "albedo_g", "albedo_b", "mat_param")
    ]


class CCamera(C.Structure):
    _fields_ = [(n, C.c_float * 3) for n in ("origin", "lower_left", "horizontal", "vertical", "u", "v", "w")] + [
        ("lens_radius", C.c_float)
    ]


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("max_bounces", C.c_int32),
        ("seed", C.c_uint32), ("tile_w", C.c_int32), ("tile_h", C.c_int32),
        ("shard", C.c_int32), ("num_shards", C.c_int32), ("variant", C.c_int32),
    ]


class MultiLayout(C.Structure):
    _fields_ = [(k, C.c_size_t) for k in ("block_bytes", "record_bytes", "count_offset", "send_bytes", "gathered_bytes", "frame_record_bytes",
                                          "frame_count_offset", "host_bytes", "counts_pitch")]


class LaunchInfo(C.Structure):
    _fields_ = [("compute_units", C.c_int32), ("blocks", C.c_int32), ("threads_per_block", C.c_int32),
                ("spheres_active", C.c_int32), ("spheres_padded", C.c_int32), ("groups", C.c_int32), ("samples", C.c_uint64),
                ("kernel", C.c_int32), ("bvh_nodes", C.c_int32), ("bvh_leaves", C.c_int32), ("bvh_depth", C.c_int32),
                ("tiles_in_kernel", C.c_int32)]


class BvhInfo(C.Structure):
    _fields_ = [("nodes", C.c_int32), ("leaves", C.c_int32), ("depth", C.c_int32), ("stack_entries", C.c_int32), ("spheres", C.c_int32),
                ("pairs", C.c_int32), ("centre", C.c_float * 3), ("pad_local", C.c_int32), ("root_leaf", C.c_int32)]


def make_params(width, height, spp, seed=10001, max_bounces=50, tile_w=32, tile_h=32, shard=0, num_shards=1, variant=0):
    return Params(width, height, spp, max_bounces, seed, tile_w, tile_h, shard, num_shards, variant)


_lib_path = os.path.join(LIBDIR, "librays1.so")


def lib_path():
    return _lib_path


def set_lib_path(path):
    """Explicit choice of another build of the library (bench.py --lib, tools/: e.g. lib/librays1_tuning.so, the
    -DR1_TUNING build that reads the R1_* knobs).  Must be called before the first lib(); nothing is read from the
    environment here, and the shipped file is never overwritten by an experiment."""
    global _lib_path
    if _lib is not None:
        raise R1Error(R1_EINVAL, "set_lib_path after the library was loaded")
    _lib_path = os.path.abspath(path)


def build(verbose=False):
    """Compiles librays1.so and rayweek1_hip in-tree with hipcc --offload-arch=gfx950."""
    subprocess.check_call(["make", "-j4", "-C", CSRC] + ([] if verbose else ["-s"]))


# every symbol include/rays1.h declares: (name, restype, argtypes)
_u8p, _f32p, _u64p, _i32p, _dblp = (C.POINTER(t) for t in (C.c_uint8, C.c_float, C.c_uint64, C.c_int32, C.c_double))
_ctx = C.c_void_p
SYMBOLS = [
    ("r1_abi_version", C.c_int, []),
    ("r1_create", C.c_int, [C.c_int, C.POINTER(_ctx)]),
    ("r1_destroy", None, [_ctx]),
    ("r1_last_error", C.c_char_p, []),
    ("r1_device_count", C.c_int, []),
    ("r1_set_scene", C.c_int, [_ctx, C.POINTER(CScene), C.POINTER(CCamera)]),
    ("r1_render", C.c_int, [_ctx, C.POINTER(Params), _u8p, _u64p, _dblp]),
    ("r1_render_samples", C.c_int, [_ctx, C.POINTER(Params), _u8p, _u64p, _f32p]),
    ("r1_tile_count", C.c_int, [C.POINTER(Params), _i32p, _i32p]),
    ("r1_shard_block_bytes", C.c_size_t, [C.POINTER(Params)]),
    ("r1_shard_record_bytes", C.c_size_t, [C.POINTER(Params)]),
    ("r1_render_async", C.c_int, [_ctx, C.POINTER(Params), _u8p, _u64p, C.c_void_p]),
    ("r1_frame_record_bytes", C.c_size_t, [C.POINTER(Params)]),
    ("r1_render_batch_async", C.c_int, [_ctx, C.POINTER(Params), C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("r1_render_shard_device_batch", C.c_int, [_ctx, C.POINTER(Params), C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("r1_assemble_device_records_batch", C.c_int, [_ctx, C.POINTER(Params), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("r1_host_alloc", C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    ("r1_host_free", None, [C.c_void_p]),
    ("r1_render_shard_device", C.c_int, [_ctx, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p]),
    ("r1_set_pixel_mode", C.c_int, [_ctx, C.c_int32]),
    ("r1_render_shard_device_once", C.c_int, [_ctx, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p]),
    ("r1_multi_create", C.c_int, [C.c_int32, _i32p, C.POINTER(C.c_void_p)]),
    ("r1_multi_destroy", None, [C.c_void_p]),
    ("r1_multi_set_scene", C.c_int, [C.c_void_p, C.POINTER(CScene), C.POINTER(CCamera)]),
    ("r1_multi_render", C.c_int, [C.c_void_p, C.POINTER(Params), _u8p, _u64p, _dblp]),
    ("r1_multi_render_async", C.c_int, [C.c_void_p, C.POINTER(Params), C.c_void_p]),
    ("r1_multi_render_batch_async", C.c_int, [C.c_void_p, C.POINTER(Params), C.c_int32, C.c_uint32, C.c_void_p]),
    ("r1_multi_sync", C.c_int, [C.c_void_p]),
    ("r1_multi_layout", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ("r1_multi_info", C.c_int, [C.c_void_p, _i32p, _i32p, C.POINTER(LaunchInfo)]),
    ("r1_assemble_device", C.c_int, [_ctx, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p]),
    ("r1_assemble_device_strided", C.c_int, [_ctx, C.POINTER(Params), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    ("r1_assemble_device_records", C.c_int, [_ctx, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("r1_sync", C.c_int, [_ctx]),
    ("r1_last_timing", C.c_int, [_ctx, _dblp, _dblp]),
    ("r1_timing_begin", C.c_int, [_ctx, C.c_int32]),
    ("r1_timing_end", C.c_int, [_ctx, _dblp, _dblp, _i32p]),
    ("r1_last_stats", C.c_int, [_ctx, _u64p]),
    ("r1_last_launch_info", C.c_int, [_ctx, C.POINTER(LaunchInfo)]),
    ("r1_last_wave_log", C.c_int, [_ctx, _u64p, C.c_size_t, C.POINTER(C.c_uint32)]),
    ("r1_host_scene_create", C.c_int, [C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    ("r1_host_scene_destroy", None, [C.c_void_p]),
    ("r1_host_scene_spheres", C.POINTER(CScene), [C.c_void_p]),
    ("r1_host_scene_camera", C.POINTER(CCamera), [C.c_void_p]),
    ("r1_bvh_describe", C.c_int, [C.POINTER(CScene), C.c_int32, C.POINTER(BvhInfo), _f32p, C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t]),
    ("r1_tga_write_rgb24", C.c_int, [C.c_char_p, C.c_int32, C.c_int32, _u8p]),
    ("r1_log_results", C.c_int, [C.c_char_p, C.c_char_p, _dblp, _u64p, C.c_int32]),
]


def lib():
    """Loads librays1.so (fails loudly if it has not been built: no fallback)."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise R1Error(R1_ENODEVICE, f"{p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                        "(there is no CPU fallback)")
        L = C.CDLL(p)
        for name, res, args in SYMBOLS:
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _check(rc):
    if rc != R1_OK:
        raise R1Error(rc, lib().r1_last_error().decode(errors="replace"))


def device_count():
    n = lib().r1_device_count()
    return max(n, 0)


class Scene:
    """Host scene = what create_*_scene() returns in the reference (Scene{camera, hitables},
    rayweek1.cpp:539-549), flattened; owns the native r1_host_scene."""

    def __init__(self, kind, width, height, grid_w=0, grid_h=0, name=None):
        self._h = C.c_void_p()
        _check(lib().r1_host_scene_create(kind, width, height, grid_w, grid_h, C.byref(self._h)))
        self.kind, self.width, self.height = kind, width, height
        self.name = name or {0: "small", 1: "medium", 2: "large", 3: "grid"}[kind]
        self.spheres = lib().r1_host_scene_spheres(self._h)
        self.camera = lib().r1_host_scene_camera(self._h)

    @property
    def count(self):
        return int(self.spheres.contents.count)

    def arrays(self):
        s = self.spheres.contents
        n = s.count
        out = {k: np.ctypeslib.as_array(getattr(s, k), shape=(n,)).copy() for k in
               ("center_x", "center_y", "center_z", "radius_sq", "inv_radius", "albedo_r", "albedo_g", "albedo_b", "mat_param")}
        out["mat_type"] = np.ctypeslib.as_array(s.mat_type, shape=(n,)).copy()
        return out

    def camera_array(self):
        c = self.camera.contents
        return np.array(sum([list(getattr(c, f)) for f in ("origin", "lower_left", "horizontal", "vertical", "u", "v", "w")], [])
                        + [c.lens_radius], dtype=np.float32)

    def close(self):
        if self._h:
            lib().r1_host_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def create_small_scene(width=1280, height=720):
    return Scene(SCENE_SMALL, width, height)


def create_medium_scene(width=1280, height=720):
    return Scene(SCENE_MEDIUM, width, height)


def create_large_scene(width=1280, height=720):
    return Scene(SCENE_LARGE, width, height)


def create_grid_scene(width, height, grid_w, grid_h):
    return Scene(SCENE_GRID, width, height, grid_w, grid_h)


class Renderer:
    """One r1_context (device, stream, cached scene + workspace)."""

    def __init__(self, device=0):
        self._c = _ctx()
        _check(lib().r1_create(device, C.byref(self._c)))
        self.device = device

    def set_scene(self, scene):
        _check(lib().r1_set_scene(self._c, scene.spheres, scene.camera))

    def set_scene_raw(self, cscene, ccamera):
        _check(lib().r1_set_scene(self._c, C.byref(cscene), C.byref(ccamera)))

    def render(self, params):
        img = np.zeros((params.height, params.width, 3), np.uint8)
        rays, secs = C.c_uint64(), C.c_double()
        _check(lib().r1_render(self._c, C.byref(params), img.ctypes.data_as(_u8p), C.byref(rays), C.byref(secs)))
        return img, int(rays.value), float(secs.value)

    def render_into(self, params, img):
        rays, secs = C.c_uint64(), C.c_double()
        _check(lib().r1_render(self._c, C.byref(params), img.ctypes.data_as(_u8p), C.byref(rays), C.byref(secs)))
        return int(rays.value), float(secs.value)

    def render_samples(self, params):
        img = np.zeros((params.height, params.width, 3), np.uint8)
        samples = np.zeros((params.height * params.width * params.spp, 4), np.float32)
        rays = C.c_uint64()
        _check(lib().r1_render_samples(self._c, C.byref(params), img.ctypes.data_as(_u8p), C.byref(rays), samples.ctypes.data_as(_f32p)))
        return img, int(rays.value), samples

    def render_async(self, params, host_frame, stream_ptr=None):
        """r1_render_async: enqueue one frame (throughput kernels) whose pixels + ray count land in `host_frame`
        (a HostFrame) once the stream is idle."""
        _check(lib().r1_render_async(self._c, C.byref(params), C.cast(host_frame.ptr, _u8p),
                                     C.cast(host_frame.ptr + host_frame.rays_offset, _u64p), _stream_arg(stream_ptr)))

    def render_batch_async(self, params, n_frames, host_frames, seed_stride=0, stream_ptr=None):
        """r1_render_batch_async: n_frames frames in one launch; their records land in `host_frames` (a HostFrames, or None
        to leave them on the device) once the stream is idle."""
        _check(lib().r1_render_batch_async(self._c, C.byref(params), n_frames, seed_stride, C.c_void_p(host_frames.ptr) if host_frames else None,
                                           _stream_arg(stream_ptr)))

    def render_shard_device_batch(self, params, n_frames, d_records_ptr, seed_stride=0, stream_ptr=None):
        _check(lib().r1_render_shard_device_batch(self._c, C.byref(params), n_frames, seed_stride, C.c_void_p(d_records_ptr),
                                                  _stream_arg(stream_ptr)))

    def assemble_device_records_batch(self, params, n_frames, d_gathered_ptr, d_frames_ptr, stream_ptr=None):
        _check(lib().r1_assemble_device_records_batch(self._c, C.byref(params), n_frames, C.c_void_p(d_gathered_ptr), C.c_void_p(d_frames_ptr),
                                                      _stream_arg(stream_ptr)))

    def render_frame_device(self, params, stream_ptr=None):
        """r1_render_async without host buffers: the frame stays in the context's device buffers (what the copies cost)."""
        _check(lib().r1_render_async(self._c, C.byref(params), None, None, _stream_arg(stream_ptr)))

    def render_shard_device(self, params, d_block_ptr, d_rays_ptr, stream_ptr=None):
        _check(lib().r1_render_shard_device(self._c, C.byref(params), C.c_void_p(d_block_ptr), C.c_void_p(d_rays_ptr),
                                            _stream_arg(stream_ptr)))

    def assemble_device(self, params, d_blocks_ptr, d_rgb_ptr, stream_ptr=None):
        _check(lib().r1_assemble_device(self._c, C.byref(params), C.c_void_p(d_blocks_ptr), C.c_void_p(d_rgb_ptr),
                                        _stream_arg(stream_ptr)))

    def assemble_device_strided(self, params, d_blocks_ptr, shard_stride_bytes, d_rgb_ptr, stream_ptr=None):
        _check(lib().r1_assemble_device_strided(self._c, C.byref(params), C.c_void_p(d_blocks_ptr), shard_stride_bytes,
                                                C.c_void_p(d_rgb_ptr), _stream_arg(stream_ptr)))

    def assemble_device_records(self, params, d_records_ptr, d_rgb_ptr, d_total_rays_ptr, stream_ptr=None):
        _check(lib().r1_assemble_device_records(self._c, C.byref(params), C.c_void_p(d_records_ptr), C.c_void_p(d_rgb_ptr),
                                                C.c_void_p(d_total_rays_ptr), _stream_arg(stream_ptr)))

    def set_pixel_mode(self, on):
        _check(lib().r1_set_pixel_mode(self._c, 1 if on else 0))

    def sync(self):
        _check(lib().r1_sync(self._c))

    def last_timing(self):
        a, b = C.c_double(), C.c_double()
        _check(lib().r1_last_timing(self._c, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def timing_begin(self, max_frames):
        _check(lib().r1_timing_begin(self._c, max_frames))

    def timing_end(self):
        a, b, n = C.c_double(), C.c_double(), C.c_int32()
        _check(lib().r1_timing_end(self._c, C.byref(a), C.byref(b), C.byref(n)))
        return float(a.value), float(b.value), int(n.value)

    def last_stats(self):
        out = (C.c_uint64 * 16)()
        _check(lib().r1_last_stats(self._c, out))
        names = ["wave_iterations", "alive_lanes", "candidate_loop_trips", "overflow_lanes", "cycles_refill", "cycles_pass1",
                 "cycles_candidates", "cycles_shade", "cycles_wave", "candidates"]
        d = {n: int(out[i]) for i, n in enumerate(names)}
        m = (1 << 64) - 1
        d["longest_wave_cycles"] = int(out[10])
        d["shortest_wave_cycles"] = m - int(out[11])
        d["span_cycles"] = int(out[12]) - (m - int(out[13]))
        d["leaf_lane_trips"] = int(out[14])  # tree diagnostic build: leaf trips summed over lanes
        d["root_steps"] = int(out[15])       # tree diagnostic build: root steps (one box test + the root's leaf, outside the walk's loops)
        return d

    def wave_log(self):
        n = C.c_uint32()
        _check(lib().r1_last_wave_log(self._c, None, 0, C.byref(n)))
        out = np.zeros((n.value, 4), np.uint64)
        if n.value:
            _check(lib().r1_last_wave_log(self._c, out.ctypes.data_as(_u64p), n.value, C.byref(n)))
        return out

    def launch_info(self):
        li = LaunchInfo()
        _check(lib().r1_last_launch_info(self._c, C.byref(li)))
        return {k: int(getattr(li, k)) for k, _ in LaunchInfo._fields_}

    def close(self):
        if self._c:
            lib().r1_destroy(self._c)
            self._c = _ctx()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostFrame:
    """Page-locked host memory (r1_host_alloc) for one frame's results: height x width x 3 pixels followed by an
    8-byte-aligned uint64 ray count — the target of Renderer.render_async."""

    def __init__(self, width, height):
        self.nbytes = width * height * 3
        self.rays_offset = (self.nbytes + 7) & ~7
        p = C.c_void_p()
        _check(lib().r1_host_alloc(self.rays_offset + 8, C.byref(p)))
        self.ptr = p.value
        buf = (C.c_uint8 * (self.rays_offset + 8)).from_address(self.ptr)
        self._all = np.frombuffer(buf, np.uint8)
        self._all[:] = 0
        self.image = self._all[:self.nbytes].reshape(height, width, 3)
        self._rays = self._all[self.rays_offset:].view(np.uint64)

    @property
    def rays(self):
        return int(self._rays[0])

    def close(self):
        if self.ptr:
            self.image = self._rays = self._all = None
            lib().r1_host_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostFrames:
    """Page-locked host memory for the n frame records of a batch (include/rays1.h r1_frame_record_bytes: image, padded
    to 8 bytes, + uint64 ray count) — the target of Renderer.render_batch_async and of bench.py's copies."""

    def __init__(self, width, height, n):
        self.n, self.nbytes = n, width * height * 3
        self.record = ((self.nbytes + 7) & ~7) + 8
        p = C.c_void_p()
        _check(lib().r1_host_alloc(self.record * n, C.byref(p)))
        self.ptr = p.value
        buf = (C.c_uint8 * (self.record * n)).from_address(self.ptr)
        self._all = np.frombuffer(buf, np.uint8)
        self._all[:] = 0
        self._shape = (height, width, 3)
        self.rays_offset = self.record - 8  # (a one-frame HostFrames is also a valid target of Renderer.render_async)

    def image(self, i):
        return self._all[i * self.record:i * self.record + self.nbytes].reshape(self._shape)

    def rays(self, i):
        return int(self._all[(i + 1) * self.record - 8:(i + 1) * self.record].view(np.uint64)[0])

    def close(self):
        if self.ptr:
            self._all = None
            lib().r1_host_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiRenderer:
    """r1_multi: one process, N GPUs, tile split + one RCCL all-gather per frame (include/rays1.h)."""

    def __init__(self, devices):
        devs = (C.c_int32 * len(devices))(*devices)
        self._m = C.c_void_p()
        _check(lib().r1_multi_create(len(devices), devs, C.byref(self._m)))

    def set_scene(self, scene):
        _check(lib().r1_multi_set_scene(self._m, scene.spheres, scene.camera))

    def render(self, params):
        img = np.zeros((params.height, params.width, 3), np.uint8)
        rays, secs = C.c_uint64(), C.c_double()
        _check(lib().r1_multi_render(self._m, C.byref(params), img.ctypes.data_as(_u8p), C.byref(rays), C.byref(secs)))
        return img, int(rays.value), float(secs.value)

    def render_into(self, params, img):
        rays, secs = C.c_uint64(), C.c_double()
        _check(lib().r1_multi_render(self._m, C.byref(params), img.ctypes.data_as(_u8p), C.byref(rays), C.byref(secs)))
        return int(rays.value), float(secs.value)

    def render_async(self, params, host_frames):
        """r1_multi_render_async: enqueue one frame over the N GPUs; its record lands in `host_frames` (a HostFrames of 1)."""
        _check(lib().r1_multi_render_async(self._m, C.byref(params), C.c_void_p(host_frames.ptr)))

    def render_batch_async(self, params, n_frames, host_frames, seed_stride=0):
        _check(lib().r1_multi_render_batch_async(self._m, C.byref(params), n_frames, seed_stride, C.c_void_p(host_frames.ptr)))

    def sync(self):
        _check(lib().r1_multi_sync(self._m))

    def info(self):
        n, v, li = C.c_int32(), C.c_int32(), LaunchInfo()
        _check(lib().r1_multi_info(self._m, C.byref(n), C.byref(v), C.byref(li)))
        return {"devices": int(n.value), "rccl_version": int(v.value), "first_device": {k: int(getattr(li, k)) for k, _ in LaunchInfo._fields_}}

    def close(self):
        if self._m:
            lib().r1_multi_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_block_bytes(params):
    return int(lib().r1_shard_block_bytes(C.byref(params)))


def frame_record_bytes(params):
    return int(lib().r1_frame_record_bytes(C.byref(params)))


def shard_record_bytes(params):
    return int(lib().r1_shard_record_bytes(C.byref(params)))


def tile_count(params):
    a, b = C.c_int32(), C.c_int32()
    _check(lib().r1_tile_count(C.byref(params), C.byref(a), C.byref(b)))
    return int(a.value), int(b.value)


def multi_layout(params, n_devices, n_frames=1):
    """r1_multi_layout: the buffer layout of an n_devices-device frame / batch (pure arithmetic, no device needed)."""
    out = MultiLayout()
    _check(lib().r1_multi_layout(C.byref(params), n_devices, n_frames, C.byref(out)))
    return {k: int(getattr(out, k)) for k, _ in MultiLayout._fields_}


def bvh_describe(cscene, leaf_max=0):
    """Host-side build of the R1_VARIANT_BVH index: (info dict, nodes float32[n,16], ids uint32[2*pairs], 0xFFFFFFFF = empty slot)."""
    info = BvhInfo()
    _check(lib().r1_bvh_describe(C.byref(cscene), leaf_max, C.byref(info), None, 0, None, 0))
    nodes = np.zeros((info.nodes, 16), np.float32)
    ids = np.zeros(max(2 * info.pairs, 2), np.uint32)
    _check(lib().r1_bvh_describe(C.byref(cscene), leaf_max, C.byref(info), nodes.ctypes.data_as(_f32p), nodes.size,
                                 ids.ctypes.data_as(C.POINTER(C.c_uint32)), ids.size))
    d = {k: int(getattr(info, k)) for k, _ in BvhInfo._fields_ if k != "centre"}
    d["centre"] = np.array(list(info.centre), np.float32)
    return d, nodes, ids[:2 * info.pairs]


class RESULT:
    """common.h:36-45"""

    def __init__(self, elapsed_seconds=0.0, num_rays=0):
        self.elapsed_seconds = elapsed_seconds
        self.num_rays = num_rays

    def get_mrays_per_sec(self):
        return self.num_rays / self.elapsed_seconds / 1000000.0 if self.elapsed_seconds else 0


def benchmark(scene, pixels, write_tga, scene_name, spp=10, seed=10001, renderer=None, quiet=False):
    """Python twin of RESULT benchmark(Scene*, Pix*, bool, const char*) (rayweek1.cpp:845-927):
    times dispatch -> pixels + ray count on the host (timer span :848 -> :891), prints the
    report block (:895-902), consumes the scene (:905) and optionally writes out_<scene>.tga."""
    own = renderer is None
    r = renderer or Renderer(0)
    r.set_scene(scene)
    p = make_params(scene.width, scene.height, spp, seed)
    t0 = time.perf_counter()
    rays, dev_s = r.render_into(p, pixels)
    res = RESULT(time.perf_counter() - t0, rays)
    if not quiet:
        info = r.launch_info()
        print(scene_name)
        print("elapsed time:   %.3fs" % res.elapsed_seconds)
        print("total samples:  %d" % (scene.width * scene.height * spp))
        print("total rays:     %d" % res.num_rays)
        print("mrays/s:        %0.2f" % res.get_mrays_per_sec())
        print("device:         hip:%d, %d CUs, %d workgroups x %d" % (r.device, info["compute_units"], info["blocks"], info["threads_per_block"]))
        print("device time:    %.3fms" % (dev_s * 1e3))
        print()
    scene.close()
    if write_tga:
        tga_write_rgb24("out_%s.tga" % scene_name, scene.width, scene.height, pixels)
    if own:
        r.close()
    return res


def tga_write_rgb24(filename, width, height, pixels):
    _check(lib().r1_tga_write_rgb24(filename.encode(), width, height, pixels.ctypes.data_as(_u8p)))
    return True


def log_results(version, scene, results):
    n = len(results)
    el = (C.c_double * n)(*[r.elapsed_seconds for r in results])
    ry = (C.c_uint64 * n)(*[r.num_rays for r in results])
    _check(lib().r1_log_results(version.encode(), scene.encode(), el, ry, n))
