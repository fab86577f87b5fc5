"""CPU-side checks of the drop-in boundary: libtrexhip.so loads and exports every symbol
include/trexhip.h declares; struct layouts agree between the C header and the ctypes mirror."""
import ctypes as C
import os
import re
import subprocess
import numpy as np
import pytest
from trex_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "trexhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(trexhip_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(capi.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = C.CDLL(capi.LIB_PATH)
    declared = header_symbols()
    assert declared, "no declarations parsed"
    for s in declared:
        assert hasattr(lib, s), f"libtrexhip.so does not export {s}"
    assert sorted(capi.SYMBOLS) == declared, "capi.SYMBOLS out of date with include/trexhip.h"
    import re
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "trexhip.h")).read()
    assert lib.trexhip_abi_version() == int(re.search(r"#define TREXHIP_ABI_VERSION (\d+)", header).group(1))


def test_struct_layouts_match_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "trexhip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(trexhip_params),sizeof(trexhip_run),sizeof(trexhip_blob),sizeof(trexhip_frame_info),'
                   'sizeof(trexhip_batch_result),offsetof(trexhip_params,cm_per_pixel),offsetof(trexhip_blob,m10),'
                   'sizeof(trexhip_split_params),offsetof(trexhip_split_params,size_ranges),sizeof(trexhip_split_info),'
                   'offsetof(trexhip_split_info,min_size_bound));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == [C.sizeof(capi.Params), capi.RUN_DTYPE.itemsize, capi.BLOB_DTYPE.itemsize, capi.INFO_DTYPE.itemsize,
                   C.sizeof(capi.BatchResult), capi.Params.cm_per_pixel.offset, capi.BLOB_DTYPE.fields["m10"][1],
                   C.sizeof(capi.SplitParams), capi.SplitParams.size_ranges.offset, capi.SPLIT_INFO_DTYPE.itemsize,
                   capi.SPLIT_INFO_DTYPE.fields["min_size_bound"][1]]


def test_oracle_and_product_share_table_layouts():
    from oracle import oracle
    assert oracle.BLOB_DTYPE == capi.BLOB_DTYPE and oracle.RUN_DTYPE == capi.RUN_DTYPE


def test_default_params_are_the_reference_defaults():
    p = capi.default_params(640, 480)
    # detect_threshold 15, threshold_maximum 255, detect_threshold_is_absolute true,
    # enable_difference true, image_invert false, dilation_size 0, use_closing false, closing_size 3
    assert (p.threshold, p.threshold_maximum, p.absolute_difference, p.enable_difference) == (15, 255, 1, 1)
    assert (p.image_invert, p.dilation_size, p.use_closing, p.closing_size, p.n_ranges) == (0, 0, 0, 3, 0)


def test_product_does_not_reference_the_oracle():
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "trex_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                t = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"\boracle\b", t) and "never imports" not in t:
                    bad.append(os.path.join(d, f))
    assert not bad, f"product sources mention the oracle: {bad}"


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.TrexHipError):
        capi.Segmenter(capi.default_params(64, 64))


@pytest.mark.parametrize("name", ["image_adjust", "blur_difference", "equalize_histogram", "correct_luminance", "use_adaptive_threshold"])
def test_unimplemented_settings_are_refused_not_ignored(name):
    # parameter validation happens before the device is touched, so this runs without a GPU
    p = capi.default_params(64, 64, **{name: 1})
    h = C.c_void_p()
    rc = capi.lib().trexhip_create(C.byref(p), C.byref(h))
    assert rc == -4 and h.value is None                                   # TREXHIP_E_UNSUPPORTED
    assert name.encode() in capi.lib().trexhip_last_error()


def test_host_colour_reduce_simd_paths_are_bit_exact():
    # hostcvt.cpp: the upload threads' BGRA -> gray reduction (cv::cvtColor's 8-bit fixed point, BackgroundSubtraction.cpp:162-180) has
    # explicit AVX2 / AVX-512 forms; every instruction set the CPU offers must give the scalar formula's bytes, for every length and alignment
    import ctypes as C
    import numpy as np
    from trex_amd import capi
    L = capi.lib()
    f = L.trexhip_host_reduce_row_isa
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rng = np.random.default_rng(0)
    ran = set()
    for n in (1, 15, 16, 17, 31, 33, 100, 1023, 1024, 1025, 4096, 2048 * 8 + 5):
        src = rng.integers(0, 256, (n, 4), dtype=np.uint8)
        src[:min(n, 3)] = [[255, 255, 255, 7], [0, 0, 0, 255], [255, 0, 255, 0]][:min(n, 3)]
        want = ((src[:, 0].astype(np.uint32) * 1868 + src[:, 1].astype(np.uint32) * 9617 + src[:, 2].astype(np.uint32) * 4899 + 8192) >> 14).astype(np.uint8)
        for off in (0, 5):
            for isa in (0, 1, 2):
                buf = np.full(n + 64, 77, np.uint8)
                a = (-buf.ctypes.data) % 16 + off
                rc = f(src.ctypes.data, buf.ctypes.data + a, n, isa)
                if rc != 0:
                    continue                      # this CPU lacks the instruction set
                ran.add(isa)
                assert np.array_equal(buf[a:a + n], want) and (buf[a + n:] == 77).all() and (buf[:a] == 77).all(), (n, isa, off)
    assert 0 in ran
