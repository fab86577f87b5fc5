"""Builds and runs the C++ host-adapter test (tests/cpp/test_host_adapter.cpp): TRex's
BackendHooks / TileImage / pv::Frame contract and the VINetwork facade, through the C ABI."""
import os
import subprocess
import numpy as np
import pytest
from oracle import cnn_oracle
from trex_amd import weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    exe = str(tmp_path / "test_host_adapter")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "test_host_adapter.cpp"), "-o", exe,
           "-L", os.path.join(ROOT, "trex_amd"), "-ltrexhip", "-L", os.path.join(ROOT, "oracle"), "-loracle",
           "-Wl,-rpath," + os.path.join(ROOT, "trex_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread"]
    subprocess.check_call(cmd)
    return exe


def test_host_adapter_compiles(tmp_path):
    # CPU-side: the adapter headers are valid C++17 against the stand-in types and link against the ABI
    build(tmp_path)


@pytest.mark.gpu
def test_host_adapter_runs(tmp_path):
    exe = build(tmp_path)
    st = weights.synthetic_state(8, 77)
    crops = weights.synthetic_crops(9, 5)
    probs, _ = cnn_oracle.predict(st, crops, threads=4)
    (tmp_path / "w.bin").write_bytes(weights.pack_blob(st, 8))
    (tmp_path / "c.bin").write_bytes(crops.tobytes())
    (tmp_path / "p.bin").write_bytes(np.ascontiguousarray(probs, np.float32).tobytes())
    out = subprocess.run([exe, str(tmp_path / "w.bin"), str(tmp_path / "c.bin"), str(tmp_path / "p.bin")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host adapter ok" in out.stdout and "identity facade ok" in out.stdout and "training facade ok" in out.stdout
