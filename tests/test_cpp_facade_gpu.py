"""GPU: the C++ facade (include/qrkit/QRKit.hpp) runs the reference's test_block_diagonal through the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_facade_block_diagonal():
    subprocess.check_call(["make", "-C", ROOT, "-s", "cpptest"])
    out = subprocess.run([os.path.join(ROOT, "build", "test_block_diagonal")], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Failed." not in out.stdout and out.stdout.count("Passed.") == 5


@pytest.mark.gpu
def test_cpp_facade_compositions():
    """The reference's test_banded_blocked (3 inputs), test_block_angular (banded and block-diagonal left solver),
    test_block_angular_denseblocked / _denseblocked_sparse (thin right solvers) and the thin solver on its own,
    written against the C++ facade classes BandedBlockedSparseQR / BlockAngularSparseQR / BlockedThin*QR."""
    subprocess.check_call(["make", "-C", ROOT, "-s", "cpptest"])
    out = subprocess.run([os.path.join(ROOT, "build", "test_compositions")], capture_output=True, text=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Failed." not in out.stdout and out.stdout.count("Passed.") == 16


@pytest.mark.gpu
def test_cpp_sharded_world1_over_rccl():
    """qrk_shard_ranges + ShardedBlockDiagonalSparseQR / qrk_gather_r with a real one-rank RCCL communicator (the C-level multi-GPU
    entry; the peers of an 8-GPU node are the driver's to run)."""
    subprocess.check_call(["make", "-C", ROOT, "-s", "cpptest"])
    out = subprocess.run([os.path.join(ROOT, "build", "test_sharded")], capture_output=True, text=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Failed." not in out.stdout and out.stdout.count("Passed.") == 4


def test_cpp_facade_compiles():
    """CPU: the facade and its test compile and link against the library (no GPU needed)."""
    subprocess.check_call(["make", "-C", ROOT, "-s", "cpptest"])
    assert os.path.exists(os.path.join(ROOT, "build", "test_block_diagonal"))
    assert os.path.exists(os.path.join(ROOT, "build", "test_compositions"))
    assert os.path.exists(os.path.join(ROOT, "build", "test_sharded"))
