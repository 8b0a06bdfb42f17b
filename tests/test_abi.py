"""The C-ABI library loads on a CPU-only host and exports every symbol include/vx.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vx_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(vx):
    assert declared_symbols() == sorted(vx.lib.SYMBOLS)


def test_library_exports_every_declared_symbol(vx):
    if not os.path.exists(vx.lib.LIB_PATH):
        vx.lib.build()
    L = ctypes.CDLL(vx.lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(L, name), f"libvxprove.so does not export {name}"
    L.vx_backend_name.restype = ctypes.c_char_p
    assert L.vx_backend_name() == b"hip-gfx950"


def test_no_cpu_fallback(vx):
    """Without a GPU the product path must fail loudly, never compute on the host."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(vx.VxError):
        vx.Context(0)


def test_product_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "0-kno-vectorx_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cuh", ".cpp")) or f == "Makefile":
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "vxo_" not in src and "libvxoracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_integration_md_rust_block_is_generated_from_the_header():
    """INTEGRATION.md's `extern "C"` block is generated from include/vx.h and names every declared symbol
    (it used to claim a one-to-one mirror while omitting a dozen entry points)."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_bindings.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in declared_symbols():
        assert f"pub fn {name}(" in text, name


def test_loading_the_library_raises_the_hardware_queue_limit():
    """The provers run 5-6 streams per proof; the HIP runtime's default of 4 hardware queues serialised them (profiles/README.md).
    Binding and library both set GPU_MAX_HW_QUEUES (without overriding a host's own choice) before the first HIP call."""
    import subprocess
    import sys

    code = ("import os, sys; sys.path.insert(0, %r); os.environ.pop('GPU_MAX_HW_QUEUES', None); os.environ['VX_NO_PY_ENV'] = '1'; "
            "import ctypes, vx_import; vx = vx_import.load(); ctypes.CDLL(vx.lib.LIB_PATH); "
            "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; print(libc.getenv(b'GPU_MAX_HW_QUEUES'))") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout
    assert out.strip() == "b'16'", out  # set by the library's constructor alone
    code2 = code.replace("os.environ.pop('GPU_MAX_HW_QUEUES', None)", "os.environ['GPU_MAX_HW_QUEUES'] = '7'")
    out = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, check=True).stdout
    assert out.strip() == "b'7'", out  # a host's own value is kept


def test_blob_format_named_in_the_header_is_the_bindings(vx):
    """include/vx.h is the boundary: the blob magic and header layout it states are the ones the binding splits blobs by
    (the header still described "HRRANGE3" with two proofs)."""
    text = open(os.path.join(ROOT, "include", "vx.h")).read()
    magic = int(re.search(r"#define VX_HR_BLOB_MAGIC (0x[0-9a-fA-F]+)ULL", text).group(1), 16)
    fixed = int(re.search(r"VX_HR_BLOB_FIXED_WORDS = (\d+)", text).group(1))
    assert magic == vx.lib.HR_MAGIC and magic.to_bytes(8, "little") == b"HRRANGE6" and "HRRANGE6" in text
    assert "HRRANGE3" not in text and "HRRANGE5" not in text and fixed == vx.lib.HR_FIXED
    import numpy as np

    # a blob of three map segments: header = fixed words + one length per segment; blob order = segments, commitment, Merkle, Ed25519, SHA-512
    blob = np.zeros(fixed + 3 + 6 + 7 + 8 + 1 + 2 + 3 + 4, dtype=np.uint64)
    blob[0], blob[16] = magic, 3
    blob[17:21] = [1, 2, 3, 4]
    blob[fixed:fixed + 3] = [6, 7, 8]
    segs, p_sha, p_tree, p_ed, p_h = vx.lib.split_blob_segments(blob)
    assert [len(p) for p in segs] == [6, 7, 8] and [len(p) for p in (p_sha, p_tree, p_ed, p_h)] == [1, 2, 3, 4]
    one = np.zeros(fixed + 1 + 5 + 1 + 2 + 3 + 4, dtype=np.uint64)
    one[0], one[16], one[fixed] = magic, 1, 5
    one[17:21] = [1, 2, 3, 4]
    assert [len(p) for p in vx.lib.split_blob(one)] == [5, 1, 2, 3, 4]
