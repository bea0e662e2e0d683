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


def test_library_sets_hardware_queues_only_when_unset():
    """loaded into a process that has made no HIP call: GPU_MAX_HW_QUEUES becomes 8 if it was unset, and a value
    that was set stays (each case in a process of its own: the variable is read once, when the library is loaded)"""
    import subprocess
    import sys
    from raytracing_amd import _lib
    prog = ("import ctypes, sys; ctypes.CDLL(sys.argv[1]); libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; "
            "print((libc.getenv(b'GPU_MAX_HW_QUEUES') or b'unset').decode())")
    for given, expect in ((None, "8"), ("4", "4"), ("16", "16")):
        env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
        if given is not None:
            env["GPU_MAX_HW_QUEUES"] = given
        out = subprocess.run([sys.executable, "-c", prog, _lib.SO_PATH], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert out.stdout.strip() == expect, (given, out.stdout, out.stderr)
