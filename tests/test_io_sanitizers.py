"""AddressSanitizer + UBSan run of the CPU-only part of the library (tscm_io.cpp: calibration YAML and corner-list
parsers; tscm_boards.cpp: chessboard structure recovery on random candidate sets) under deterministic fuzzing (tests/native/fuzz_io.cpp).  GPU sanitizers are not available on
the pool, so this is where the sanitizers earn their keep: malformed files must come back as error codes."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_yaml_and_corner_parsers_under_asan_ubsan(tmp_path):
    exe = tmp_path / "fuzz_io"
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "fuzz_io.cpp"),
           os.path.join(ROOT, "tscm_calib_amd", "csrc", "tscm_io.cpp"), os.path.join(ROOT, "tscm_calib_amd", "csrc", "tscm_boards.cpp"),
           "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "asan" in (r.stderr + r.stdout).lower():
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe), "4000", str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert "clean" in r.stdout
