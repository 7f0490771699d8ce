"""SURVEY.md §5 sanitizer leg, as far as this pool allows it (GPU AddressSanitizer and XNACK-on runs are refused by gpurun): the
HOST side of the C-ABI library — workspace carving, geometry / support predicates, argument checks, error reporting — built with
AddressSanitizer (csrc/Makefile: libdwiseneuro_hip_asan.so, -fsanitize=address -fno-gpu-sanitize) and driven on CPU under the
ASAN runtime by tests/asan_host_driver.py.  Any heap / stack / global overflow or use-after-free in those paths aborts the child."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
LIB = ROOT / "sensorium_amd" / "csrc" / "libdwiseneuro_hip_asan.so"


def _asan_runtime():
    for clang in ("/opt/rocm/lib/llvm/bin/clang", "clang"):
        try:
            out = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True, timeout=60)
        except (OSError, subprocess.TimeoutExpired):
            continue
        path = out.stdout.strip()
        if out.returncode == 0 and path and os.path.isabs(path) and os.path.exists(path):
            return path
    return None


def test_host_paths_are_clean_under_address_sanitizer():
    assert LIB.exists(), "run __graft_entry__.build() (csrc/Makefile builds the host-ASAN library)"
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("clang's ASAN runtime not found on this machine")
    env = dict(os.environ, LD_PRELOAD=rt, DWN_LIB_PATH=str(LIB), DWN_DETERMINISTIC="0",
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:verify_asan_link_order=0:exitcode=97")
    res = subprocess.run([sys.executable, str(ROOT / "tests" / "asan_host_driver.py")], cwd=str(ROOT), env=env,
                         capture_output=True, text=True, timeout=900)
    assert "AddressSanitizer" not in res.stderr, res.stderr[-4000:]
    assert res.returncode == 0 and "ASAN_HOST_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
