"""Host-side mirror of the reference's class surface (namespace fa) over the C ABI.

Reference: cvsteer/SteerableFilters.h:41-50, SteerableFiltersG2.h:35-67,
SteerableFiltersG4.h:35-57.  Method names and argument meaning follow the reference; where the
reference fills ``cv::Mat1f&`` out-parameters, these methods return the planes.

Planes may be
  * numpy float32 arrays (host memory; results come back as numpy arrays), or
  * torch CUDA float32 tensors (device memory; zero-copy in, results are torch tensors on the
    same device, work is enqueued on torch's current stream).
All arithmetic happens in libcvsteer_hip.so on the GPU; this module only marshals pointers.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import CvsError, Plane, lib

SETUP_BASIS, SETUP_ORIENT, SETUP_FULL = 1, 2, 3

try:  # torch is plumbing (device memory + streams), optional for host-plane use
    import torch
except Exception:  # pragma: no cover
    torch = None


# numpy mirror of `struct cvs_plane` (include/cvsteer_hip.h) for arrays of descriptors
_PLANE_DTYPE = np.dtype({"names": ["data", "rows", "cols", "step", "mem"], "formats": ["u8", "i4", "i4", "u8", "i4"],
                         "offsets": [Plane.data.offset, Plane.rows.offset, Plane.cols.offset, Plane.step.offset, Plane.mem.offset],
                         "itemsize": C.sizeof(Plane)})


def _is_torch(a):
    return torch is not None and isinstance(a, torch.Tensor)


KIND_G2, KIND_G4 = L.KIND_G2, L.KIND_G4


def alloc_planes(n, rows, cols, device=None):
    """n output planes as rows of ONE block, [row][plane][column] -- the layout the engine gives its own state planes
    (CVS_OPT_STATE_LAYOUT): a launch that writes all of them streams one linear sweep instead of n streams far apart.
    Returns n strided (rows, cols) views, ordinary planes for every entry point (like cv::Mat ROIs: step = n * cols * 4).
    device=None: numpy (host) planes."""
    if device is None:
        blk = np.empty((rows, n, cols), np.float32)
    else:
        blk = torch.empty((rows, n, cols), dtype=torch.float32, device=device)
    return [blk[:, k, :] for k in range(n)]


def num_basis(kind):
    return lib().cvs_num_basis(kind)


def make_taps(kind, idx, width, spacing):
    """SteerableFilters::create on the idx-th tap function (host math, no GPU needed)."""
    out = np.empty(2 * width + 1, np.float32)
    rc = lib().cvs_make_taps(kind, idx, width, spacing, out.ctypes.data_as(C.POINTER(C.c_float)))
    if rc:
        raise CvsError(rc, "cvs_make_taps")
    return out


def basis_taps(kind, p):
    a, b = C.c_int(), C.c_int()
    rc = lib().cvs_basis_taps(kind, p, C.byref(a), C.byref(b))
    if rc:
        raise CvsError(rc, "cvs_basis_taps")
    return a.value, b.value


def steer_weights(kind, theta):
    out = np.empty(num_basis(kind), np.float32)
    rc = lib().cvs_steer_weights(kind, theta, out.ctypes.data_as(C.POINTER(C.c_float)))
    if rc:
        raise CvsError(rc, "cvs_steer_weights")
    return out


def _plane(a):
    """cvs_plane view of a 2-D float32 numpy array or torch CUDA tensor (no copy); 8-bit arrays /
    tensors are accepted for input images (CVS_DEPTH_U8)."""
    if _is_torch(a) and a.dtype == torch.uint8:
        if a.dim() != 2 or (a.numel() and a.shape[1] > 1 and a.stride(1) != 1):
            raise ValueError("8-bit image must be 2-D with unit column stride")
        mem = (L.MEM_DEVICE if a.is_cuda else L.MEM_HOST) | L.DEPTH_U8
        return Plane(a.data_ptr(), a.shape[0], a.shape[1], a.stride(0) if a.shape[0] > 1 else a.shape[1], mem)
    if isinstance(a, np.ndarray) and a.dtype == np.uint8:
        if a.ndim != 2 or (a.size and a.shape[1] > 1 and a.strides[1] != 1):
            raise ValueError("8-bit image must be 2-D with unit column stride")
        return Plane(a.ctypes.data, a.shape[0], a.shape[1], a.strides[0] if (a.shape[0] > 1 and a.size) else a.shape[1],
                     L.MEM_HOST | L.DEPTH_U8)
    if _is_torch(a):
        if a.dtype != torch.float32 or a.dim() != 2 or (a.numel() and a.shape[1] > 1 and a.stride(1) != 1):
            raise ValueError("torch plane must be 2-D float32 with unit column stride")
        mem = L.MEM_DEVICE if a.is_cuda else L.MEM_HOST
        return Plane(a.data_ptr(), a.shape[0], a.shape[1], a.stride(0) * 4 if a.shape[0] > 1 else a.shape[1] * 4, mem)
    if not isinstance(a, np.ndarray) or a.dtype != np.float32 or a.ndim != 2:
        raise ValueError("numpy plane must be 2-D float32")
    if a.size and a.shape[1] > 1 and a.strides[1] != 4:
        raise ValueError("numpy plane must have unit column stride")
    step = a.strides[0] if (a.shape[0] > 1 and a.size) else a.shape[1] * 4
    return Plane(a.ctypes.data, a.shape[0], a.shape[1], step, L.MEM_HOST)


def _as_input(a):
    """the reference converts any Mat to Mat1f unscaled (Mat1f(const Mat&)); do the same on the host side"""
    if _is_torch(a):
        return a if a.dtype in (torch.float32, torch.uint8) else a.to(torch.float32)
    a = np.asarray(a)
    return a if a.dtype in (np.float32, np.uint8) else a.astype(np.float32)


def pyramid_setup(handles, image, level_images=None, flags=SETUP_BASIS):
    """BASELINE config 3 in one call (cvs_pyramid_setup): handles[l].setup(level l) for every level of the Gaussian pyramid of
    `image`, the pyramid built on the way (the filter launch of a level writes the next level).  level_images: the levels - 1 planes that receive levels 1.. (allocated when None).
    Returns [image, level 1, ...]."""
    n = len(handles)
    image = _as_input(image)
    if level_images is None:
        level_images, shape = [], tuple(image.shape)
        for _ in range(n - 1):
            shape = ((shape[0] + 1) // 2, (shape[1] + 1) // 2)
            level_images.append(torch.empty(shape, dtype=torch.float32, device=image.device) if _is_torch(image) else np.empty(shape, np.float32))
    for hnd in handles:
        hnd._bind_stream(image, *level_images)
    arr = (C.c_void_p * n)(*[hnd._h for hnd in handles])
    planes = (Plane * max(1, n - 1))(*[_plane(l) for l in level_images])
    pi = _plane(image)
    rc = lib().cvs_pyramid_setup(arr, n, C.byref(pi), int(flags), planes)
    if rc:
        raise CvsError(rc, "cvs_pyramid_setup", lib().cvs_last_error(handles[0]._h).decode())
    for hnd, l in zip(handles, [image] + list(level_images)):
        hnd._like = l
        hnd._image_keepalive = l
    return [image] + list(level_images)


class SteerableFilters:
    """fa::SteerableFilters (SteerableFilters.h:41-50): setup(image), steer(theta) -> (g, h)."""

    KIND = None
    DEFAULT_WIDTH = None
    DEFAULT_SPACING = None

    def __init__(self, image=None, width=None, spacing=None, device=None, setup_flags=None):
        width = self.DEFAULT_WIDTH if width is None else width
        spacing = self.DEFAULT_SPACING if spacing is None else spacing
        if device is None:
            device = image.device.index if (_is_torch(image) and image.is_cuda and image.device.index is not None) else 0
        self._h = C.c_void_p()
        rc = lib().cvs_create(self.KIND, int(width), float(spacing), int(device), C.byref(self._h))
        if rc:
            self._h = None
            raise CvsError(rc, "cvs_create", "no usable HIP device -- there is no CPU fallback" if rc == L.E_HIP else "")
        self.device = int(device)
        self.width, self.spacing = int(width), float(spacing)
        self._like = None
        self._setup_flags = setup_flags
        if image is not None:
            self.setup(image)

    # -- plumbing --
    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().cvs_destroy(h)
            except Exception:
                pass

    def _check(self, rc, where):
        if rc:
            raise CvsError(rc, where, lib().cvs_last_error(self._h).decode())

    def _bind_stream(self, *planes):
        if torch is not None and any(_is_torch(p) and p.is_cuda for p in planes):
            lib().cvs_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))

    def _new(self, shape=None):
        shape = self.shape if shape is None else shape
        if _is_torch(self._like) and self._like.is_cuda:
            return torch.empty(shape, dtype=torch.float32, device=self._like.device)
        return np.empty(shape, np.float32)

    def _new_like(self, a):
        if _is_torch(a):
            return torch.empty(tuple(a.shape), dtype=torch.float32, device=a.device)
        return np.empty(a.shape, np.float32)

    def _new_block_like(self, a, n):
        """n fresh output planes shaped like `a`: device planes come as rows of one block (alloc_planes), host planes dense"""
        if _is_torch(a) and a.is_cuda:
            return alloc_planes(n, int(a.shape[0]), int(a.shape[1]), device=a.device)
        return [self._new_like(a) for _ in range(n)]

    def set_option(self, option, value):
        self._check(lib().cvs_set_option(self._h, option, int(value)), "cvs_set_option")

    def get_option(self, option):
        v = C.c_int(0)
        self._check(lib().cvs_get_option(self._h, option, C.byref(v)), "cvs_get_option")
        return v.value

    def launch_info(self):
        """cvs_get_launch_info as a dict: configuration of the last basis launch (the engine's default or its tuner's decision)"""
        li = L.LaunchInfo()
        li.struct_size = C.sizeof(L.LaunchInfo)
        self._check(lib().cvs_get_launch_info(self._h, C.byref(li)), "cvs_get_launch_info")
        return {k: getattr(li, k) for k, _ in L.LaunchInfo._fields_}

    def set_atan_mode(self, exact):
        self.set_option(L.OPT_ATAN_MODE, 1 if exact else 0)

    def set_strip_rows(self, rows):
        self.set_option(L.OPT_STRIP_ROWS, rows)

    def sync(self):
        self._check(lib().cvs_sync(self._h), "cvs_sync")

    @property
    def shape(self):
        r, c = C.c_int(), C.c_int()
        self._check(lib().cvs_shape(self._h, C.byref(r), C.byref(c)), "cvs_shape")
        return (r.value, c.value)

    def taps(self, idx):
        out = np.empty(2 * self.width + 1, np.float32)
        self._check(lib().cvs_taps(self._h, idx, out.ctypes.data_as(C.POINTER(C.c_float))), "cvs_taps")
        return out

    def _state(self, which):
        out = self._new()
        self._bind_stream(out)
        p = _plane(out)
        self._check(lib().cvs_read_state(self._h, which, C.byref(p)), "cvs_read_state")
        return out

    def basis(self, p):
        """p-th separable basis plane (the reference's protected m_g2a.. / m_g4a.. members)."""
        return self._state(L.PLANE_BASIS0 + p)

    def basis_view(self, p):
        """zero-copy (ptr, rows, cols, step_bytes) of a device-resident basis plane"""
        v = Plane()
        self._check(lib().cvs_state_plane(self._h, L.PLANE_BASIS0 + p, C.byref(v)), "cvs_state_plane")
        return v.data, v.rows, v.cols, v.step

    # -- reference surface --
    def setup(self, image, flags=None):
        """virtual setup(const Mat1f&)"""
        image = _as_input(image)
        if flags is None:
            flags = self._setup_flags if self._setup_flags is not None else self._DEFAULT_FLAGS
        self._like = image
        self._bind_stream(image)
        p = _plane(image)
        self._image_keepalive = image
        self._check(lib().cvs_setup(self._h, C.byref(p), flags), "cvs_setup")

    def setup_steer(self, image, theta, flags=SETUP_BASIS, out=None):
        """setup(image) + steer(float theta) in one kernel launch -> (g, h)"""
        image = _as_input(image)
        self._like = image
        g, h = out if out is not None else self._new_block_like(image, 2)
        self._bind_stream(image, g, h)
        pi, pg, ph = _plane(image), _plane(g), _plane(h)
        self._check(lib().cvs_setup_steer(self._h, C.byref(pi), flags, float(theta), C.byref(pg), C.byref(ph)),
                    "cvs_setup_steer")
        return g, h

    def _steer(self, theta, full, out=None):
        n = 5 if full else 2
        outs = list(out) if out is not None else [self._new() for _ in range(n)]
        planes = [_plane(o) for o in outs] + [None] * (5 - n)
        ptrs = [C.byref(p) if p is not None else None for p in planes]
        if isinstance(theta, (int, float, np.floating)):
            self._bind_stream(*outs)
            self._check(lib().cvs_steer_scalar(self._h, float(theta), *ptrs), "cvs_steer_scalar")
        else:
            if theta is not None:
                theta = _as_input(theta)
                self._bind_stream(theta, *outs)
                pt = _plane(theta)
                tptr = C.byref(pt)
            else:
                self._bind_stream(*outs)
                tptr = None
            self._check(lib().cvs_steer_map(self._h, tptr, *ptrs), "cvs_steer_map")
        return tuple(outs)

    def steer(self, theta, full=False, out=None):
        """steer(float theta, g, h) / steer(const Mat1f& theta, g, h); theta=None steers at the
        dominant orientation.  full=True adds (e, magnitude, phase) (G2 only)."""
        return self._steer(theta, full, out)


    # -- adjacent component: Gaussian pyramid, not part of the reference --
    def pyrDown(self, image):
        """one pyramid level with cv::pyrDown semantics: ((rows+1)//2, (cols+1)//2)"""
        image = _as_input(image)
        shape = ((image.shape[0] + 1) // 2, (image.shape[1] + 1) // 2)
        if _is_torch(image):
            dst = torch.empty(shape, dtype=torch.float32, device=image.device)
        else:
            dst = np.empty(shape, np.float32)
        self._bind_stream(image, dst)
        ps, pd = _plane(image), _plane(dst)
        self._check(lib().cvs_pyr_down(self._h, C.byref(ps), C.byref(pd)), "cvs_pyr_down")
        return dst

    def pyramid(self, image, levels):
        """[image, pyrDown(image), ...] -- `levels` planes"""
        out = [_as_input(image)]
        for _ in range(levels - 1):
            out.append(self.pyrDown(out[-1]))
        return out

    def setup_pyr(self, image, flags=None, out=None):
        """setup(image) and pyrDown(image) in one pass over the image -> the next pyramid level"""
        image = _as_input(image)
        if flags is None:
            flags = self._setup_flags if self._setup_flags is not None else self._DEFAULT_FLAGS
        shape = ((image.shape[0] + 1) // 2, (image.shape[1] + 1) // 2)
        if out is not None:
            dst = out
        elif _is_torch(image):
            dst = torch.empty(shape, dtype=torch.float32, device=image.device)
        else:
            dst = np.empty(shape, np.float32)
        self._like = image
        self._bind_stream(image, dst)
        self._image_keepalive = image
        ps, pd = _plane(image), _plane(dst)
        self._check(lib().cvs_setup_pyr(self._h, C.byref(ps), flags, C.byref(pd)), "cvs_setup_pyr")
        return dst


class SteerableFiltersG2(SteerableFilters):
    """fa::SteerableFiltersG2 (SteerableFiltersG2.h:35-67)."""

    KIND = L.KIND_G2
    DEFAULT_WIDTH = 4
    DEFAULT_SPACING = 0.67
    _DEFAULT_FLAGS = SETUP_FULL

    def getDominantOrientationAngle(self):
        return self._state(L.PLANE_THETA)

    def getDominantOrientationStrength(self):
        return self._state(L.PLANE_STRENGTH)

    def coefficients(self):
        """(C1, C2, C3) -- the reference's protected m_c1..m_c3"""
        return tuple(self._state(w) for w in (L.PLANE_C1, L.PLANE_C2, L.PLANE_C3))

    def steer_point(self, p, theta, full=False):
        """steer(const cv::Point& p, theta, ...): p = (x, y) = (col, row)"""
        out = (C.c_float * 5)()
        self._check(lib().cvs_steer_point(self._h, int(p[0]), int(p[1]), float(theta), out), "cvs_steer_point")
        vals = tuple(float(v) for v in out)
        return vals if full else vals[:2]

    def computeMagnitudeAndPhase(self, g2, h2):
        mag, phase = self._new_like(g2), self._new_like(g2)
        self._bind_stream(g2, h2, mag, phase)
        pg, ph, pm, pp = _plane(g2), _plane(h2), _plane(mag), _plane(phase)
        self._check(lib().cvs_mag_phase(self._h, C.byref(pg), C.byref(ph), C.byref(pm), C.byref(pp)), "cvs_mag_phase")
        return mag, phase

    def wrap(self, angle):
        """SteerableFilters::wrap (SteerableFilters.cpp:46-51)"""
        out = self._new_like(angle)
        self._bind_stream(angle, out)
        pa, po = _plane(angle), _plane(out)
        self._check(lib().cvs_wrap(self._h, C.byref(pa), C.byref(po)), "cvs_wrap")
        return out

    def phaseWeights(self, phase, phi, signum, k=2.0):
        lam = self._new_like(phase)
        self._bind_stream(phase, lam)
        pp, pl = _plane(phase), _plane(lam)
        self._check(lib().cvs_phase_weights(self._h, C.byref(pp), C.byref(pl), float(phi), int(bool(signum)), float(k)),
                    "cvs_phase_weights")
        return lam

    def find(self, e, phase, which=(True, True, True)):
        """findEdges + findDarkLines + findBrightLines in one pass -> (edges, dark, bright)"""
        outs = [self._new_like(e) if w else None for w in which]
        self._bind_stream(e, phase, *[o for o in outs if o is not None])
        pe, pp = _plane(e), _plane(phase)
        planes = [_plane(o) if o is not None else None for o in outs]
        ptrs = [C.byref(p) if p is not None else None for p in planes]
        self._check(lib().cvs_find(self._h, C.byref(pe), C.byref(pp), *ptrs), "cvs_find")
        return tuple(outs)

    def findEdges(self, e, phase, k=2.0):
        return self.find(e, phase, (True, False, False))[0]

    def findDarkLines(self, e, phase, k=2.0):
        return self.find(e, phase, (False, True, False))[1]

    def findBrightLines(self, e, phase, k=2.0):
        return self.find(e, phase, (False, False, True))[2]

    def pipeline(self, image, out=None):
        """the callers' whole sequence (test/test.cpp:85-90) for one image ->
        (g2, h2, e, magnitude, phase, edges, dark, bright)"""
        image = _as_input(image)
        self._like = image
        outs = list(out) if out is not None else self._new_block_like(image, 8)
        self._bind_stream(image, *[o for o in outs if o is not None])
        pi = _plane(image)
        planes = [_plane(o) if o is not None else None for o in outs]
        arr = (C.POINTER(Plane) * 8)(*[C.pointer(p) if p is not None else None for p in planes])
        self._check(lib().cvs_pipeline(self._h, C.byref(pi), arr), "cvs_pipeline")
        return tuple(outs)

    def pipeline_batch(self, frames, out=None, outputs=None):
        """pipeline() for n same-size frames in one launch.  frames: [n, H, W] tensor/array (or a list
        of planes).  outputs: indices into (g2, h2, e, magnitude, phase, edges, dark, bright) to
        produce (default all 8); returns / fills out [n, len(outputs), H, W].  select_frame(i) then
        picks whose state the getters and steer() use (unless set_persist(False))."""
        sel = list(range(8)) if outputs is None else [int(k) for k in outputs]
        block = _is_torch(frames) and frames.dim() == 3 and frames.dtype in (torch.float32, torch.uint8) and frames.is_cuda
        if block and out is None:
            out = torch.empty((frames.shape[0], len(sel)) + tuple(frames.shape[1:]), dtype=torch.float32, device=frames.device)
        if block and _is_torch(out) and out.dim() == 4 and out.is_cuda and out.dtype == torch.float32 \
                and out.shape[0] == frames.shape[0] and out.shape[1] == len(sel) and frames.stride(2) == 1 and out.stride(3) == 1:
            # one [n, H, W] block in, one [n, K, H, W] block out: the plane descriptors are filled in arithmetically
            # (two numpy arrays laid out like `struct cvs_plane`), not one Python object per plane
            n, rows, cols = (int(v) for v in frames.shape)
            self._like = frames[0]
            self._bind_stream(frames, out)
            imgs = np.zeros(n, _PLANE_DTYPE)
            esz = 1 if frames.dtype == torch.uint8 else 4   # 8-bit frames are read as bytes by the kernel (CVS_DEPTH_U8)
            imgs["data"] = frames.data_ptr() + np.arange(n, dtype=np.uint64) * np.uint64(frames.stride(0) * esz)
            imgs["rows"], imgs["cols"], imgs["step"] = rows, cols, frames.stride(1) * esz
            imgs["mem"] = L.MEM_DEVICE | (L.DEPTH_U8 if esz == 1 else 0)
            outs = np.zeros((n, 8), _PLANE_DTYPE)  # data == NULL means "not requested"
            frame_off = np.arange(n, dtype=np.uint64) * np.uint64(out.stride(0) * 4)
            for j, k in enumerate(sel):
                outs["data"][:, k] = out.data_ptr() + frame_off + np.uint64(j * out.stride(1) * 4)
                outs["rows"][:, k], outs["cols"][:, k], outs["step"][:, k], outs["mem"][:, k] = rows, cols, out.stride(2) * 4, L.MEM_DEVICE
            self._check(lib().cvs_pipeline_batch(self._h, imgs.ctypes.data_as(L._PP), n, outs.ctypes.data_as(L._PP)), "cvs_pipeline_batch")
            self._batch_keepalive = (frames, out)
            return out
        planes = [_as_input(f) for f in frames]
        n = len(planes)
        self._like = planes[0]
        shape = tuple(planes[0].shape)
        if out is None:
            if _is_torch(planes[0]):
                out = torch.empty((n, len(sel)) + shape, dtype=torch.float32, device=planes[0].device)
            else:
                out = np.empty((n, len(sel)) + shape, np.float32)
        self._bind_stream(planes[0], out[0][0])
        imgs = (Plane * n)(*[_plane(p) for p in planes])
        outs = (Plane * (n * 8))()  # zero-initialised: data == NULL means "not requested"
        for i in range(n):
            for j, k in enumerate(sel):
                outs[i * 8 + k] = _plane(out[i][j])
        self._check(lib().cvs_pipeline_batch(self._h, imgs, n, outs), "cvs_pipeline_batch")
        self._batch_keepalive = (planes, out)
        return out

    def set_persist(self, on):
        """pipeline()/pipeline_batch(): keep the basis + orientation planes (default, like the reference
        object) or write the requested outputs only"""
        self.set_option(L.OPT_PERSIST_STATE, 1 if on else 0)

    def select_frame(self, i):
        self._check(lib().cvs_select_frame(self._h, int(i)), "cvs_select_frame")

    def _to_u8(self, plane, gain):
        self._bind_stream(plane)
        pp = _plane(plane)
        if _is_torch(plane) and plane.is_cuda:
            dst = torch.empty(tuple(plane.shape), dtype=torch.uint8, device=plane.device)
            ptr, step, mem = C.c_void_p(dst.data_ptr()), dst.stride(0), L.MEM_DEVICE
        else:
            dst = np.empty(plane.shape, np.uint8)
            ptr, step, mem = C.c_void_p(dst.ctypes.data), dst.strides[0], L.MEM_HOST
        if gain is None:
            rc = lib().cvs_normalize_u8(self._h, C.byref(pp), ptr, step, mem)
        else:
            rc = lib().cvs_convert_u8(self._h, C.byref(pp), float(gain), 0.0, ptr, step, mem)
        self._check(rc, "cvs_normalize_u8" if gain is None else "cvs_convert_u8")
        return dst

    def normalize_u8(self, plane):
        """cv::normalize(plane, dst, 0, 255, NORM_MINMAX, CV_8UC1) on the GPU"""
        return self._to_u8(plane, None)

    def convert_u8(self, plane, gain):
        """plane.convertTo(dst, CV_8UC1, gain) on the GPU"""
        return self._to_u8(plane, gain)


class SteerableFiltersG4(SteerableFilters):
    """fa::SteerableFiltersG4 (SteerableFiltersG4.h:35-57): setup + steer only.

    extensions=True switches on what the reference leaves unfinished (G4.h:55, G4.cpp:88-90): dominant
    orientation / strength / C1..C3 derived from the G4/H4 steering polynomials, steer(..., full=True)
    and a working computeMagnitudeAndPhase.  Off by default: then the class behaves exactly like the
    reference (empty getters, no-op computeMagnitudeAndPhase)."""

    KIND = L.KIND_G4
    DEFAULT_WIDTH = 6
    DEFAULT_SPACING = 0.5
    _DEFAULT_FLAGS = SETUP_BASIS

    def __init__(self, image=None, width=None, spacing=None, device=None, setup_flags=None, extensions=False):
        self.extensions = bool(extensions)
        super().__init__(None, width, spacing, device, setup_flags)
        if self.extensions:
            self.set_option(L.OPT_G4_EXTENSIONS, 1)
            if setup_flags is None:
                self._setup_flags = SETUP_FULL
        if image is not None:
            self.setup(image)

    def getDominantOrientationAngle(self):
        """never assigned in the reference (G4.h:55): an empty Mat -- unless extensions are on"""
        return self._state(L.PLANE_THETA) if self.extensions else np.empty((0, 0), np.float32)

    def getDominantOrientationStrength(self):
        return self._state(L.PLANE_STRENGTH) if self.extensions else np.empty((0, 0), np.float32)

    def coefficients(self):
        if not self.extensions:
            raise CvsError(L.E_UNSUPPORTED, "coefficients", "G4 orientation is an extension (extensions=True)")
        return tuple(self._state(w) for w in (L.PLANE_C1, L.PLANE_C2, L.PLANE_C3))

    def computeMagnitudeAndPhase(self, g4, h4, magnitude=None, phase=None):
        """empty body in the reference (G4.cpp:88-90): outputs untouched -- unless extensions are on"""
        if not self.extensions:
            return magnitude, phase
        mag, ph = self._new_like(g4), self._new_like(g4)
        self._bind_stream(g4, h4, mag, ph)
        pg, phh, pm, pp = _plane(g4), _plane(h4), _plane(mag), _plane(ph)
        self._check(lib().cvs_mag_phase(self._h, C.byref(pg), C.byref(phh), C.byref(pm), C.byref(pp)), "cvs_mag_phase")
        return mag, ph
