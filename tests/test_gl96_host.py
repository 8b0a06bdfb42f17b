"""CPU tier: the lazy 96-bit arithmetic of the NTT rounds (0-kno-vectorx_amd/csrc/gl96.h) compiled for the host and checked
against 128-bit integer arithmetic -- shifts by every power of w_16, the fast / exact folds, whole radix-16 rounds in both
directions for every round width, on random and corner-case words (tests/host/gl96_check.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gl96_arithmetic_matches_integers(tmp_path):
    exe = tmp_path / "gl96_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tests", "host", "gl96_check.cpp")])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    tag, k_shift, k_fold, headroom = out.stdout.split()
    assert tag == "ok"
    assert int(k_shift) <= 7 and int(k_fold) <= 64 and int(headroom) >= 0  # the bounds gl96.h states
