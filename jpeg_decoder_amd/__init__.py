"""jpeg_decoder_amd -- MI355X-native JPEG block pipeline (dequantize -> 8x8 IDCT -> YCbCr->RGB).

The product is the C-ABI shared library ``libjpegblk.so`` (include/jpegblk.h, sources under
``csrc/``: hand-written HIP kernels for gfx950 + the C++ host side).  This Python package is a
thin ctypes binding over that ABI, used by tests/, bench.py and __graft_entry__.py; PyTorch is
used only as plumbing (device memory, streams, torch.distributed).

There is no CPU fallback: if the library is missing, or no HIP device is usable, calls fail
loudly (ImportError / JbError).
"""
from .api import (JbError, Context, ImageDesc, Geometry, DeviceBatch, lib, lib_path, make_desc,
                  geometry_of, resolve_qtabs, entropy_decode, decode_batch, BatchDecoder, build_library)

__all__ = ["JbError", "Context", "ImageDesc", "Geometry", "DeviceBatch", "lib", "lib_path",
           "make_desc", "geometry_of", "resolve_qtabs", "entropy_decode", "decode_batch", "BatchDecoder", "build_library"]
