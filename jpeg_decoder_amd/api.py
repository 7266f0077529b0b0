"""ctypes binding of include/jpegblk.h (libjpegblk.so).  One Python function per C entry point;
no arithmetic happens on this side."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# JPEGBLK_LIB: experiment builds of the same ABI (tools/); the product is libjpegblk.so
_LIB_PATH = os.environ.get("JPEGBLK_LIB") or os.path.join(_HERE, "libjpegblk.so")

JB_OK = 0
STATUS_NAMES = {0: "JB_OK", -1: "JB_ERR_NULL", -2: "JB_ERR_GEOMETRY", -3: "JB_ERR_SAMPLING",
                -4: "JB_ERR_QTAB", -5: "JB_ERR_CAPACITY", -6: "JB_ERR_HIP", -7: "JB_ERR_STATE",
                -8: "JB_ERR_FORMAT", -9: "JB_ERR_UNSUPPORTED"}


class JbError(RuntimeError):
    def __init__(self, status, text=""):
        self.status = status
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {text}")


class ImageDesc(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("hs", ctypes.c_int32), ("vs", ctypes.c_int32),
                ("qtab_id", ctypes.c_int32 * 3), ("reserved", ctypes.c_int32)]


class Geometry(ctypes.Structure):
    _fields_ = [("mcu_w", ctypes.c_int32), ("mcu_h", ctypes.c_int32),
                ("mcu_w_real", ctypes.c_int32), ("mcu_h_real", ctypes.c_int32),
                ("mcus_x", ctypes.c_int32), ("mcus_y", ctypes.c_int32),
                ("blocks_per_mcu", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("n_coded_blocks", ctypes.c_int64), ("coef_bytes", ctypes.c_int64),
                ("rgb_bytes", ctypes.c_int64)]


class DeviceBatch(ctypes.Structure):
    _fields_ = [("desc", ImageDesc), ("n_images", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("d_coef", ctypes.c_void_p), ("coef_image_stride", ctypes.c_int64),
                ("d_qtabs", ctypes.c_void_p), ("qtab_image_stride", ctypes.c_int64),
                ("d_rgb", ctypes.c_void_p), ("rgb_image_stride", ctypes.c_int64),
                ("rgb_row_stride", ctypes.c_int64)]


def build_library():
    """Compile csrc/ for gfx950 into jpeg_decoder_amd/libjpegblk.so (hipcc cross-compiles
    without a GPU)."""
    subprocess.run(["make", "-C", os.path.join(_HERE, "csrc")], check=True, stdout=subprocess.DEVNULL)


def lib_path():
    return _LIB_PATH


_lib = None


def lib():
    """The loaded C-ABI library.  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH) and not os.environ.get("JPEGBLK_LIB"):
        # not built yet (fresh checkout): compile it -- building is not a fallback, there is none
        try:
            build_library()
        except Exception as e:  # hipcc missing, compile error ...
            raise ImportError(f"{_LIB_PATH} is missing and could not be built ({e}); run "
                              "`python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback for the block pipeline)") from e
    if not os.path.exists(_LIB_PATH):
        raise ImportError(f"{_LIB_PATH} is missing (there is no CPU fallback for the block pipeline)")
    # PyTorch bundles its own HIP/HSA runtime with the same sonames as /opt/rocm's; two copies
    # in one process cannot both open the GPU.  Load torch's first (when torch is present) so
    # that libjpegblk.so binds to the runtime torch tensors and streams live in.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(_LIB_PATH)
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    pd = ctypes.POINTER(ImageDesc)
    L.jb_abi_version.restype = ctypes.c_int
    L.jb_device_count.restype = ctypes.c_int
    L.jb_geometry_of.argtypes = [pd, ctypes.POINTER(Geometry)]
    L.jb_ctx_create.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(vp)]
    L.jb_ctx_destroy.argtypes = [vp]
    L.jb_ctx_destroy.restype = None
    L.jb_last_error.argtypes = [vp]
    L.jb_last_error.restype = ctypes.c_char_p
    L.jb_ctx_stream.argtypes = [vp]
    L.jb_ctx_stream.restype = vp
    L.jb_ctx_synchronize.argtypes = [vp]
    L.jb_blocks_to_rgb.argtypes = [vp, pd, vp, vp, vp, i64]
    L.jb_submit.argtypes = [vp, pd, vp, vp, vp, i64, ctypes.POINTER(ctypes.c_int)]
    L.jb_wait.argtypes = [vp, ctypes.c_int]
    L.jb_pinned_alloc.argtypes = [ctypes.c_size_t]
    L.jb_pinned_alloc.restype = vp
    L.jb_pinned_alloc_on.argtypes = [ctypes.c_int, ctypes.c_size_t]
    L.jb_pinned_alloc_on.restype = vp
    L.jb_device_numa_node.argtypes = [ctypes.c_int]
    L.jb_ctx_reserve.argtypes = [vp, ctypes.c_size_t, ctypes.c_size_t]
    L.jb_ctx_device.argtypes = [vp]
    L.jb_ctx_device_entropy_images.argtypes = [vp]
    L.jb_ctx_device_entropy_images.restype = ctypes.c_longlong
    L.jb_entropy_decode_device.argtypes = [vp, vp, ctypes.c_size_t, pd, vp, vp, ctypes.c_size_t]
    L.jb_pinned_free.argtypes = [vp]
    L.jb_pinned_free.restype = None
    L.jb_blocks_to_rgb_device.argtypes = [vp, ctypes.POINTER(DeviceBatch), vp]
    L.jb_resolve_qtabs.argtypes = [pd, vp, vp]
    L.jb_kernel_name.argtypes = [pd]
    L.jb_kernel_name.restype = ctypes.c_char_p
    L.jb_entropy_decode.argtypes = [vp, ctypes.c_size_t, pd, vp, vp, ctypes.c_size_t]
    L.jb_entropy_decode_mt.argtypes = [vp, ctypes.c_size_t, pd, vp, vp, ctypes.c_size_t, ctypes.c_int]
    L.jb_decode_file.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(vp), ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.jb_decode_memory.argtypes = [vp, vp, ctypes.c_size_t, ctypes.POINTER(vp), ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.jb_decode_batch.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, ctypes.c_int,
                                  ctypes.POINTER(vp), ctypes.POINTER(i32), ctypes.POINTER(i32),
                                  ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]
    L.jb_batch_decoder_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(vp)]
    L.jb_batch_decoder_create_multi.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.c_size_t,
                                                ctypes.c_size_t, ctypes.POINTER(vp)]
    L.jb_batch_decoder_device_entropy_images.argtypes = [vp]
    L.jb_batch_decoder_device_entropy_images.restype = ctypes.c_longlong
    L.jb_batch_decoder_run.argtypes = [vp] + L.jb_decode_batch.argtypes[1:3] + L.jb_decode_batch.argtypes[4:]
    L.jb_batch_decoder_destroy.argtypes = [vp]
    L.jb_batch_decoder_destroy.restype = None
    L.jb_batch_decoder_submit.argtypes = L.jb_batch_decoder_run.argtypes[:-1] + [ctypes.POINTER(ctypes.c_int)]
    L.jb_batch_decoder_collect.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    L.jb_poll.argtypes = [vp, ctypes.c_int]
    L.jb_submit_batch.argtypes = [vp, ctypes.POINTER(ImageDesc), ctypes.c_int, vp, vp, vp, ctypes.POINTER(ctypes.c_int)]
    L.jb_batch_decoder_set_arena.argtypes = [vp, ctypes.c_size_t]
    L.jb_batch_decoder_set_device_output.argtypes = [vp, vp, ctypes.c_size_t]
    L.jb_batch_decoder_set_device_output.restype = ctypes.c_int
    L.jb_batch_decoder_set_device_outputs.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t), ctypes.c_int]
    L.jb_batch_decoder_set_device_outputs.restype = ctypes.c_int
    L.jb_free.argtypes = [vp]
    L.jb_free.restype = None
    L.jb_write_ppm.argtypes = [ctypes.c_char_p, vp, i32, i32, i64]
    L.jb_write_bmp.argtypes = [ctypes.c_char_p, vp, i32, i32, i64]
    if L.jb_abi_version() != 1:
        raise ImportError("libjpegblk.so ABI version mismatch")
    _lib = L
    return L


def make_desc(width, height, hs, vs, qtab_id=(0, 1, 1)):
    d = ImageDesc()
    d.width, d.height, d.hs, d.vs = int(width), int(height), int(hs), int(vs)
    for i in range(3):
        d.qtab_id[i] = int(qtab_id[i])
    d.reserved = 0
    return d


def _check(rc, ctx=None):
    if rc != JB_OK:
        raise JbError(rc, lib().jb_last_error(ctx).decode(errors="replace"))


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def geometry_of(desc):
    g = Geometry()
    _check(lib().jb_geometry_of(ctypes.byref(desc), ctypes.byref(g)))
    return g


def resolve_qtabs(desc, qtabs):
    """(uint16 [4,64], desc.qtab_id) -> int32 [3,64], the layout the kernel reads."""
    q = np.ascontiguousarray(qtabs, dtype=np.uint16).reshape(4, 64)
    out = np.zeros((3, 64), np.int32)
    _check(lib().jb_resolve_qtabs(ctypes.byref(desc), _ptr(q), _ptr(out)))
    return out


def entropy_decode(jpeg_bytes, headers_only=False, n_threads=1):
    """Host front end: JFIF bytes -> (desc, qtabs uint16 [4,64], coef int16 [n,64] or None).
    n_threads > 1 decodes the restart intervals of the image in parallel."""
    buf = np.frombuffer(jpeg_bytes, dtype=np.uint8)
    desc = ImageDesc()
    q = np.zeros((4, 64), np.uint16)
    _check(lib().jb_entropy_decode(_ptr(buf), buf.size, ctypes.byref(desc), _ptr(q), None, 0))
    if headers_only:
        return desc, q, None
    g = geometry_of(desc)
    coef = np.zeros((g.n_coded_blocks, 64), np.int16)
    _check(lib().jb_entropy_decode_mt(_ptr(buf), buf.size, ctypes.byref(desc), _ptr(q), _ptr(coef), coef.nbytes, n_threads))
    return desc, q, coef


class Context:
    """jb_ctx: one device, one stream, a ring of staging slots."""

    def __init__(self, device=0, max_coef_bytes=0, max_rgb_bytes=0, n_slots=2):
        self._h = ctypes.c_void_p()
        _check(lib().jb_ctx_create(device, max_coef_bytes, max_rgb_bytes, n_slots, ctypes.byref(self._h)))

    @classmethod
    def for_image(cls, desc, device=0, n_slots=2):
        g = geometry_of(desc)
        return cls(device, g.coef_bytes, g.rgb_bytes, n_slots)

    def close(self):
        if self._h:
            lib().jb_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def reserve(self, max_coef_bytes, max_rgb_bytes):
        """jb_ctx_reserve: grow (or create) the staging ring."""
        _check(lib().jb_ctx_reserve(self._h, max_coef_bytes, max_rgb_bytes), self._h)

    @property
    def device(self):
        return lib().jb_ctx_device(self._h)

    @property
    def device_entropy_images(self):
        """Images this context decoded with the entropy stage on the device."""
        return lib().jb_ctx_device_entropy_images(self._h)

    def entropy_decode_device(self, jpeg_bytes):
        """jb_entropy_decode_device: JFIF bytes (with restart intervals) -> (desc, qtabs, coef int16
        [n, 64]) with the Huffman stage on the GPU; the coefficients come back through a torch
        tensor (plumbing).  Raises JbError(-9) when the stream is not eligible."""
        import torch
        buf = np.frombuffer(jpeg_bytes, dtype=np.uint8)
        desc, q, _ = entropy_decode(jpeg_bytes, headers_only=True)
        g = geometry_of(desc)
        t = torch.full((g.n_coded_blocks, 64), 0x5a5a, dtype=torch.int16, device=f"cuda:{self.device}")
        torch.cuda.synchronize()
        d2 = ImageDesc()
        q2 = np.zeros((4, 64), np.uint16)
        _check(lib().jb_entropy_decode_device(self._h, _ptr(buf), buf.size, ctypes.byref(d2), _ptr(q2), t.data_ptr(), t.numel() * 2), self._h)
        return d2, q2, t.cpu().numpy()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return lib().jb_ctx_stream(self._h)

    def synchronize(self):
        _check(lib().jb_ctx_synchronize(self._h), self._h)

    # -- the seam, host buffers --------------------------------------------------------------
    def blocks_to_rgb(self, desc, coef, qtabs, stride=None):
        coef = np.ascontiguousarray(coef, dtype=np.int16)
        q = np.ascontiguousarray(qtabs, dtype=np.uint16).reshape(4, 64)
        stride = stride or 3 * desc.width
        out = np.zeros((desc.height, stride), np.uint8)
        _check(lib().jb_blocks_to_rgb(self._h, ctypes.byref(desc), _ptr(coef), _ptr(q), _ptr(out), stride), self._h)
        return out[:, :3 * desc.width].reshape(desc.height, desc.width, 3)

    def submit(self, desc, coef, qtabs, out, stride=None):
        """coef / out: numpy arrays that stay alive until wait(ticket)."""
        q = np.ascontiguousarray(qtabs, dtype=np.uint16).reshape(4, 64)
        t = ctypes.c_int(-1)
        _check(lib().jb_submit(self._h, ctypes.byref(desc), _ptr(coef), _ptr(q), _ptr(out),
                               stride or 3 * desc.width, ctypes.byref(t)), self._h)
        return t.value

    def submit_batch(self, desc, coef, qtabs, out):
        """jb_submit_batch: n images of one geometry in one submission.  coef int16 [n, blocks, 64],
        qtabs uint16 [n, 4, 64], out uint8 [n, H, 3*W] (tight rows); all C-contiguous and alive
        until wait()."""
        n = coef.shape[0]
        assert coef.flags.c_contiguous and qtabs.flags.c_contiguous and out.flags.c_contiguous
        assert qtabs.shape == (n, 4, 64) and out.shape[0] == n
        t = ctypes.c_int()
        _check(lib().jb_submit_batch(self._h, ctypes.byref(desc), n, _ptr(coef), _ptr(qtabs), _ptr(out), ctypes.byref(t)), self._h)
        return t.value

    def poll(self, ticket):
        """True once the submission has completed (non-blocking)."""
        rc = lib().jb_poll(self._h, ticket)
        if rc == 1:
            return False
        _check(rc, self._h)
        return True

    def wait(self, ticket):
        _check(lib().jb_wait(self._h, ticket), self._h)

    # -- the seam, device buffers ------------------------------------------------------------
    def blocks_to_rgb_device(self, batch, stream=None):
        _check(lib().jb_blocks_to_rgb_device(self._h, ctypes.byref(batch), stream), self._h)

    # -- decode(path) -> RGB -----------------------------------------------------------------
    def decode_file(self, path):
        p, w, h = ctypes.c_void_p(), ctypes.c_int32(), ctypes.c_int32()
        _check(lib().jb_decode_file(self._h, os.fsencode(path), ctypes.byref(p), ctypes.byref(w), ctypes.byref(h)), self._h)
        try:
            n = w.value * h.value * 3
            arr = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(n,)).copy()
        finally:
            lib().jb_free(p)
        return arr.reshape(h.value, w.value, 3)

    def decode_memory(self, jpeg_bytes):
        """jb_decode_memory: a JFIF byte string -> RGB [H, W, 3] (front end + device seam)."""
        buf = np.frombuffer(jpeg_bytes, dtype=np.uint8)
        p, w, h = ctypes.c_void_p(), ctypes.c_int32(), ctypes.c_int32()
        _check(lib().jb_decode_memory(self._h, _ptr(buf), buf.size, ctypes.byref(p), ctypes.byref(w), ctypes.byref(h)), self._h)
        try:
            n = w.value * h.value * 3
            arr = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(n,)).copy()
        finally:
            lib().jb_free(p)
        return arr.reshape(h.value, w.value, 3)


class BatchDecoder:
    """jb_batch_decoder: n_threads host lanes (pinned buffers each) feeding one shared context per
    device, reusable.  devices=[...] (jb_batch_decoder_create_multi): one decoder over several
    devices, file i -> devices[i % len(devices)], the host threads split evenly."""

    def __init__(self, n_threads=8, device=0, max_coef_bytes=0, max_rgb_bytes=0, arena_bytes=0, devices=None):
        self._h = ctypes.c_void_p()
        if devices is not None:
            ids = (ctypes.c_int * len(devices))(*devices)
            _check(lib().jb_batch_decoder_create_multi(ids, len(devices), n_threads, max_coef_bytes, max_rgb_bytes, ctypes.byref(self._h)))
        else:
            _check(lib().jb_batch_decoder_create(device, n_threads, max_coef_bytes, max_rgb_bytes, ctypes.byref(self._h)))
        self._arena = False
        self._device_out = False
        self._flights = {}
        if arena_bytes:
            _check(lib().jb_batch_decoder_set_arena(self._h, arena_bytes))
            self._arena = True

    @property
    def device_entropy_images(self):
        return lib().jb_batch_decoder_device_entropy_images(self._h)

    def run(self, paths, keep_pixels=True, on_image=None):
        assert not self._device_out, "device output is set: use run_to_device"
        return decode_batch(paths, keep_pixels=keep_pixels, on_image=on_image, _decoder=self._h, _arena=self._arena)

    def set_device_output(self, d_base, nbytes):
        """jb_batch_decoder_set_device_output: decoded images stay in the caller's DEVICE memory
        [d_base, d_base + nbytes) (e.g. a torch uint8 CUDA tensor's data_ptr()); (0, 0) = host output again."""
        _check(lib().jb_batch_decoder_set_device_output(self._h, ctypes.c_void_p(d_base or None), nbytes))
        self._arena = bool(d_base)
        self._device_out = bool(d_base)

    def set_device_outputs(self, regions):
        """jb_batch_decoder_set_device_outputs: [(device pointer, bytes), ...], one per listed device of a
        multi-device decoder; [] = host output again."""
        n = len(regions)
        ptrs = (ctypes.c_void_p * max(n, 1))(*[r[0] for r in regions])
        sizes = (ctypes.c_size_t * max(n, 1))(*[r[1] for r in regions])
        _check(lib().jb_batch_decoder_set_device_outputs(self._h, ptrs, sizes, n))
        self._arena = self._device_out = n > 0

    def run_to_device(self, paths):
        """After set_device_output: -> (device pointers (int, 0 = failed), (width, height) per image, statuses, times)."""
        assert getattr(self, "_device_out", False), "call set_device_output first"
        n = len(paths)
        arr = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
        rgb = (ctypes.c_void_p * n)()
        w = (ctypes.c_int32 * n)()
        h = (ctypes.c_int32 * n)()
        st = (ctypes.c_int * n)()
        times = (ctypes.c_double * 4)()
        rc = lib().jb_batch_decoder_run(self._h, arr, n, rgb, w, h, st, times)
        t = {"wall_s": times[0], "entropy_s": times[1], "device_s": times[2], "read_s": times[3], "rc": rc,
             "error": lib().jb_last_error(None).decode(errors="replace") if rc else ""}
        return [int(rgb[i] or 0) for i in range(n)], [(w[i], h[i]) for i in range(n)], list(st), t

    # -- batches in a stream (jb_batch_decoder_submit / _collect): two in flight -----------------
    def submit(self, paths):
        """-> a ticket (keeps the batch's arrays alive); the batch runs while the caller prepares the next one."""
        n = len(paths)
        t = {"n": n, "paths": (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths]), "rgb": (ctypes.c_void_p * n)(),
             "w": (ctypes.c_int32 * n)(), "h": (ctypes.c_int32 * n)(), "st": (ctypes.c_int * n)(), "id": ctypes.c_int(-1)}
        _check(lib().jb_batch_decoder_submit(self._h, t["paths"], n, t["rgb"], t["w"], t["h"], t["st"], ctypes.byref(t["id"])))
        # the library writes into these arrays until the batch is collected (or the decoder destroyed): the decoder
        # object holds them as well, so a ticket the caller drops cannot free them under a running batch
        self._flights[t["id"].value] = t
        return t

    def collect(self, ticket, keep_pixels=True, on_image=None):
        """-> what run() returns (host output: arrays / None; with an arena the pixels are views' copies), or, with
        device output set, what run_to_device() returns."""
        times = (ctypes.c_double * 4)()
        rc = lib().jb_batch_decoder_collect(self._h, ticket["id"], times)
        if rc != -7:   # (JB_ERR_STATE: no such batch -- nothing was collected)
            self._flights.pop(ticket["id"].value, None)
        n, rgb, w, h, st = ticket["n"], ticket["rgb"], ticket["w"], ticket["h"], ticket["st"]
        t = {"wall_s": times[0], "entropy_s": times[1], "device_s": times[2], "read_s": times[3], "rc": rc,
             "error": lib().jb_last_error(None).decode(errors="replace") if rc else ""}
        if self._device_out:
            return [int(rgb[i] or 0) for i in range(n)], [(w[i], h[i]) for i in range(n)], list(st), t
        out = []
        for i in range(n):
            if rgb[i]:
                if on_image is not None or keep_pixels:
                    m = w[i] * h[i] * 3
                    view = np.ctypeslib.as_array(ctypes.cast(rgb[i], ctypes.POINTER(ctypes.c_uint8)), shape=(m,)).reshape(h[i], w[i], 3)
                    if on_image is not None:
                        on_image(i, view)
                out.append(view.copy() if keep_pixels else (w[i], h[i]))
                if not self._arena:
                    lib().jb_free(rgb[i])
            else:
                out.append(None)
        return out, list(st), t

    def close(self):
        if self._h:
            lib().jb_batch_decoder_destroy(self._h)   # (waits for batches still in flight)
            self._h = ctypes.c_void_p()
            self._flights.clear()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def decode_batch(paths, n_threads=8, device=0, keep_pixels=True, on_image=None, _decoder=None, _arena=False):
    """jb_decode_batch: -> (list of uint8 [H,W,3] arrays or None, statuses, times dict).
    on_image(i, view): called with a no-copy [H,W,3] view of every decoded image before its buffer
    is released (checks over batches too large to keep)."""
    n = len(paths)
    arr = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
    rgb = (ctypes.c_void_p * n)()
    w = (ctypes.c_int32 * n)()
    h = (ctypes.c_int32 * n)()
    st = (ctypes.c_int * n)()
    times = (ctypes.c_double * 4)()
    if _decoder is not None:
        rc = lib().jb_batch_decoder_run(_decoder, arr, n, rgb, w, h, st, times)
    else:
        rc = lib().jb_decode_batch(device, arr, n, n_threads, rgb, w, h, st, times)
    out = []
    for i in range(n):
        if rgb[i]:
            m = w[i] * h[i] * 3
            if on_image is not None:
                on_image(i, np.ctypeslib.as_array(ctypes.cast(rgb[i], ctypes.POINTER(ctypes.c_uint8)), shape=(m,)).reshape(h[i], w[i], 3))
            if keep_pixels:
                a = np.ctypeslib.as_array(ctypes.cast(rgb[i], ctypes.POINTER(ctypes.c_uint8)), shape=(m,)).copy()
                out.append(a.reshape(h[i], w[i], 3))
            else:
                out.append((w[i], h[i]))
            if not _arena:  # arena images belong to the decoder
                lib().jb_free(rgb[i])
        else:
            out.append(None)
    t = {"wall_s": times[0], "entropy_s": times[1], "device_s": times[2], "read_s": times[3], "rc": rc,
         "error": lib().jb_last_error(None).decode(errors="replace") if rc else ""}
    return out, list(st), t


def torch_batch(desc, n_images, coef_t, qtabs_t, rgb_t, rgb_row_stride=None, shared_qtabs=True):
    """DeviceBatch over torch CUDA tensors (plumbing): coef_t int16 [n_images, n_blocks, 64],
    qtabs_t int32 [3,64] (shared) or [n_images,3,64], rgb_t uint8 [n_images, H, row_stride]."""
    b = DeviceBatch()
    b.desc = desc
    b.n_images = n_images
    b.d_coef = coef_t.data_ptr()
    b.coef_image_stride = coef_t.stride(0) * 2 if n_images > 1 else coef_t.numel() * 2
    b.d_qtabs = qtabs_t.data_ptr()
    b.qtab_image_stride = 0 if shared_qtabs else 768
    b.d_rgb = rgb_t.data_ptr()
    b.rgb_row_stride = rgb_row_stride or rgb_t.stride(1)
    b.rgb_image_stride = rgb_t.stride(0)
    return b
