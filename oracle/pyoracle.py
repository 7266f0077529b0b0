"""ctypes bindings for the CHECKERS under oracle/ -- test infrastructure only.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this module;
nothing under jpeg_decoder_amd/ does.

  Oracle  -> oracle/liboracle.so      (scalar C restatement, jpegblk_oracle.c; travels to the GPU box)
  Ref     -> oracle/_ref/libjpegref.so (the genuine reference compiled in place through
             ref_harness.cpp; exists only where oracle/Makefile found /root/reference, or as the
             prebuilt file that travelled with the snapshot)
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class ImageDesc(ctypes.Structure):
    """jb_image_desc of include/jpegblk.h."""
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("hs", ctypes.c_int32), ("vs", ctypes.c_int32),
                ("qtab_id", ctypes.c_int32 * 3), ("reserved", ctypes.c_int32)]


class Geometry(ctypes.Structure):
    """jb_geometry of include/jpegblk.h."""
    _fields_ = [("mcu_w", ctypes.c_int32), ("mcu_h", ctypes.c_int32),
                ("mcu_w_real", ctypes.c_int32), ("mcu_h_real", ctypes.c_int32),
                ("mcus_x", ctypes.c_int32), ("mcus_y", ctypes.c_int32),
                ("blocks_per_mcu", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("n_coded_blocks", ctypes.c_int64), ("coef_bytes", ctypes.c_int64),
                ("rgb_bytes", ctypes.c_int64)]


def make_desc(width, height, hs, vs, qtab_id=(0, 1, 1)):
    d = ImageDesc()
    d.width, d.height, d.hs, d.vs = int(width), int(height), int(hs), int(vs)
    for i in range(3):
        d.qtab_id[i] = int(qtab_id[i])
    d.reserved = 0
    return d


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def build(with_ref=True):
    """(Re)build the checkers; building the checker is not using it."""
    target = "all" if with_ref else "liboracle.so"
    subprocess.run(["make", "-C", _HERE, target], check=True, stdout=subprocess.DEVNULL)


class Oracle:
    def __init__(self):
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(with_ref=False)
        self.lib = ctypes.CDLL(path)
        L = self.lib
        L.jbo_geometry_of.argtypes = [ctypes.POINTER(ImageDesc), ctypes.POINTER(Geometry)]
        L.jbo_blocks_to_rgb.argtypes = [ctypes.POINTER(ImageDesc), ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_int64]
        L.jbo_blocks_to_rgb_mt.argtypes = L.jbo_blocks_to_rgb.argtypes + [ctypes.c_int]
        L.jbo_time_blocks_to_rgb.argtypes = L.jbo_blocks_to_rgb.argtypes + [ctypes.c_int, ctypes.c_int]
        L.jbo_time_blocks_to_rgb.restype = ctypes.c_double
        L.jbo_dequant_block.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.jbo_idct_block.argtypes = [ctypes.c_void_p]
        L.jbo_constants.argtypes = [ctypes.c_void_p]

    def geometry(self, desc):
        g = Geometry()
        rc = self.lib.jbo_geometry_of(ctypes.byref(desc), ctypes.byref(g))
        if rc:
            raise ValueError(f"jbo_geometry_of -> {rc}")
        return g

    def blocks_to_rgb(self, desc, coef, qtabs, nthreads=1, stride=None):
        coef = np.ascontiguousarray(coef, dtype=np.int16)
        qtabs = np.ascontiguousarray(qtabs, dtype=np.uint16).reshape(4, 64)
        g = self.geometry(desc)
        assert coef.size == g.n_coded_blocks * 64, (coef.size, g.n_coded_blocks)
        stride = stride or 3 * desc.width
        out = np.zeros((desc.height, stride), np.uint8)
        if nthreads <= 1:
            rc = self.lib.jbo_blocks_to_rgb(ctypes.byref(desc), _ptr(coef), _ptr(qtabs), _ptr(out), stride)
        else:
            rc = self.lib.jbo_blocks_to_rgb_mt(ctypes.byref(desc), _ptr(coef), _ptr(qtabs), _ptr(out), stride, nthreads)
        if rc:
            raise ValueError(f"jbo_blocks_to_rgb -> {rc}")
        return out[:, :3 * desc.width].reshape(desc.height, desc.width, 3)

    def time_blocks_to_rgb(self, desc, coef, qtabs, nthreads, reps):
        coef = np.ascontiguousarray(coef, dtype=np.int16)
        qtabs = np.ascontiguousarray(qtabs, dtype=np.uint16).reshape(4, 64)
        out = np.zeros((desc.height, 3 * desc.width), np.uint8)
        s = self.lib.jbo_time_blocks_to_rgb(ctypes.byref(desc), _ptr(coef), _ptr(qtabs), _ptr(out),
                                            3 * desc.width, nthreads, reps)
        if s < 0:
            raise ValueError("jbo_time_blocks_to_rgb failed")
        return s

    def idct_block(self, blk):
        b = np.ascontiguousarray(blk, dtype=np.int32).copy().reshape(64)
        self.lib.jbo_idct_block(_ptr(b))
        return b

    def constants(self):
        c = np.zeros(18, np.uint32)
        self.lib.jbo_constants(_ptr(c))
        return c


class RefInfo(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int) for n in
                 "width height hs vs mcu_w mcu_h mcu_w_real mcu_h_real restart_interval".split()]
                + [("qtab_id", ctypes.c_int * 3), ("n_coded_blocks", ctypes.c_int),
                   ("coef_min", ctypes.c_int), ("coef_max", ctypes.c_int)]
                + [(n, ctypes.c_double) for n in
                   "ms_huffman ms_dequant ms_idct ms_colour ms_parse".split()])


class Ref:
    """The genuine reference (oracle/_ref/libjpegref.so)."""
    PATH = os.path.join(_HERE, "_ref", "libjpegref.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.PATH)

    def __init__(self):
        self.lib = ctypes.CDLL(self.PATH)
        L = self.lib
        L.ref_decode_file.argtypes = [ctypes.c_char_p, ctypes.POINTER(RefInfo), ctypes.c_void_p,
                                      ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
        L.ref_blocks_to_rgb.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p,
                                                              ctypes.c_void_p, ctypes.c_void_p,
                                                              ctypes.c_long, ctypes.c_void_p]
        L.ref_constants.argtypes = [ctypes.c_void_p]

    def decode_file(self, path):
        """-> (info, coef int16 [n,64], qtabs uint16 [4,64], rgb uint8 [H,W,3]).
        The reference exit(1)s on files it rejects: only call on files it accepts."""
        info = RefInfo()
        # first pass sizes everything (cheap images only)
        rc = self.lib.ref_decode_file(path.encode(), ctypes.byref(info), None, 0, None, None, 0)
        if rc:
            raise ValueError(f"ref_decode_file({path}) -> {rc}")
        coef = np.zeros((info.n_coded_blocks, 64), np.int16)
        q = np.zeros((4, 64), np.uint16)
        rgb = np.zeros((info.height, info.width, 3), np.uint8)
        rc = self.lib.ref_decode_file(path.encode(), ctypes.byref(info), _ptr(coef), coef.size,
                                      _ptr(q), _ptr(rgb), rgb.size)
        if rc:
            raise ValueError(f"ref_decode_file({path}) -> {rc}")
        return info, coef, q, rgb

    def blocks_to_rgb(self, desc, coef, qtabs, stage_ms=None):
        coef = np.ascontiguousarray(coef, dtype=np.int16)
        qtabs = np.ascontiguousarray(qtabs, dtype=np.uint16).reshape(4, 64)
        qid = (ctypes.c_int * 3)(*[desc.qtab_id[i] for i in range(3)])
        rgb = np.zeros((desc.height, desc.width, 3), np.uint8)
        ms = (ctypes.c_double * 3)()
        rc = self.lib.ref_blocks_to_rgb(desc.width, desc.height, desc.hs, desc.vs, _ptr(coef),
                                        _ptr(qtabs), qid, _ptr(rgb), 3 * desc.width, ms)
        if rc:
            raise ValueError(f"ref_blocks_to_rgb -> {rc}")
        if stage_ms is not None:
            stage_ms[:] = list(ms)
        return rgb

    def constants(self):
        c = np.zeros(18, np.uint32)
        self.lib.ref_constants(_ptr(c))
        return c
