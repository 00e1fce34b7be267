"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer builds of what runs on the host (GPU sanitizers are not
available on this pool): the oracle, exercised by its own test files through QRK_ORACLE_LIB, and the host-side integer logic of the
banded solver (qrkit_amd/csrc/banded_host.hip, plain C++) with the driver tests/san/banded_host_san.cpp on the reference's known
answers and on malformed patterns."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_logic_and_oracle_under_asan_ubsan():
    out = subprocess.run(["make", "-C", ROOT, "-s", "san"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "Passed." in out.stdout, out.stdout + out.stderr
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, QRK_ORACLE_LIB=os.path.join(ROOT, "build", "san", "libqrk_oracle_san.so"), LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle.py"), os.path.join(ROOT, "tests", "test_oracle_blockmap.py")],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
