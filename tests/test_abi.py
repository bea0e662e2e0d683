"""The C-ABI library loads and exports every symbol include/lt_gpu.h declares.
No compute calls: runs without a GPU."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "lt_gpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(lt_[a-z_0-9]+)\s*\(", hdr)))


def test_header_symbols_exported():
    from raytracing_amd import _lib
    assert os.path.exists(_lib.SO_PATH), "libltgpu.so not built (run __graft_entry__.build())"
    so = ctypes.CDLL(_lib.SO_PATH)
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(so, n), f"{n} declared in include/lt_gpu.h but not exported"
    assert set(_lib.EXPORTS) == set(names)


def test_no_cpu_fallback_in_product():
    """The product package never imports the checker."""
    pkg = os.path.join(ROOT, "raytracing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "lt_oracle" not in txt, f
