"""Runs the C++ host-mirror test binary (tests/cpp/test_mirror.cpp) on the GPU box."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "cpp", "_build", "test_mirror")


def test_cpp_mirror_header_compiles_on_cpu():
    # built by __graft_entry__.build(); compiling needs no GPU
    subprocess.check_call(["make", "-C", os.path.join(HERE, "cpp"), "-s"])
    assert os.path.exists(BIN)


@pytest.mark.gpu
def test_cpp_mirror_on_gpu():
    assert os.path.exists(BIN), "build it with make -C tests/cpp"
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout
