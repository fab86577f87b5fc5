"""The reference's own LZO (ProcessedVideo/lzo/minilzo.c compiled unmodified into oracle/_ref/libminilzo.so by oracle/ref.mk): the checker of
this library's LZO1X encoder (trex_amd/csrc/pvfile.cpp) and of the oracle's decoder restatement (oracle/trex_pv.c).  TEST INFRASTRUCTURE ONLY.
The shared object is built on demand where the reference tree exists and travels with the snapshot elsewhere; without either, available() is False."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libminilzo.so")
_LIB = None
LZO1X_1_MEM_COMPRESS = 16384 * C.sizeof(C.c_void_p)          # minilzo.h: LZO1X_1_MEM_COMPRESS = 16384 * lzo_sizeof_dict_t


def available():
    if not os.path.exists(_SO) and os.path.exists("/root/reference/Application/src/ProcessedVideo/lzo/minilzo.c"):
        subprocess.call(["make", "-s", "-f", "oracle/ref.mk"], cwd=os.path.dirname(_HERE))
    return os.path.exists(_SO)


def lib():
    global _LIB
    if _LIB is None:
        assert available(), "oracle/_ref/libminilzo.so is not built (needs /root/reference: make -f oracle/ref.mk)"
        L = C.CDLL(_SO)
        for f in (L.lzo1x_1_compress, L.lzo1x_decompress):
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p]
        _LIB = L
    return _LIB


def decompress(data, out_len):
    """lzo1x_decompress(src, src_len, dst, &dst_len, NULL) as pv::Frame::read_from calls it (pv.cpp:331) -> bytes (LZO_E_OK asserted)"""
    src = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8))
    dst = np.zeros(out_len + 16, np.uint8)
    n = C.c_size_t(0)
    rc = lib().lzo1x_decompress(src.ctypes.data_as(C.c_void_p), len(src), dst.ctypes.data_as(C.c_void_p), C.byref(n), None)
    assert rc == 0, f"lzo1x_decompress returned {rc}"
    return dst[:n.value].copy()


def compress(data):
    """lzo1x_1_compress as pv::Frame::serialize calls it (pv.cpp:738) -> bytes"""
    src = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8))
    dst = np.zeros(len(src) + len(src) // 16 + 64 + 3, np.uint8)
    wrk = np.zeros(LZO1X_1_MEM_COMPRESS, np.uint8)
    n = C.c_size_t(0)
    rc = lib().lzo1x_1_compress(src.ctypes.data_as(C.c_void_p), len(src), dst.ctypes.data_as(C.c_void_p), C.byref(n), wrk.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return dst[:n.value].copy()
