"""CPU: the C-ABI library builds, loads, and exports exactly what include/deepsignal_hip.h declares;
without a GPU the product path fails LOUDLY (no CPU fallback). No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from deepsignal_amd import engine
    return engine.load_library()


def _declared():
    text = open(os.path.join(ROOT, "include", "deepsignal_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ds_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    from deepsignal_amd import engine
    names = _declared()
    assert len(names) >= 18
    assert sorted(engine.EXPORTED_SYMBOLS) == names
    for n in names:
        assert hasattr(lib, n), n


def test_version_and_config_layout(lib):
    from deepsignal_amd.engine import DsConfig
    assert b"gfx950" in lib.ds_version()
    assert ctypes.sizeof(DsConfig) == 16 * 4


def test_product_has_no_cpu_fallback(lib):
    """On a box without a GPU construction must raise; the oracle is never imported by the package."""
    import torch
    from deepsignal_amd.engine import Engine
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu suite")
    with pytest.raises(RuntimeError) as ei:
        Engine()
    assert "HIP" in str(ei.value) or "device" in str(ei.value)


def test_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "deepsignal_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                hit = re.search(r"import\s+oracle|from\s+oracle|libds_oracle|ds_oracle_|oracle[/.]_build|oracle\.oracle", src)
                assert hit is None, "%s uses the oracle (%s)" % (f, hit.group(0))


def test_invalid_variant_is_rejected(lib):
    """model.py:28-29: at least one of is_cnn / is_rnn."""
    from deepsignal_amd.engine import DsConfig
    cfg = DsConfig(17, 360, 2, 0, 0, 1, 0, 0, 8)
    h = ctypes.c_void_p()
    rc = lib.ds_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc == -1 and not h.value
    assert b"at least one" in lib.ds_last_error(None)


def test_header_is_plain_c99_and_client_links(tmp_path):
    """The boundary is a C ABI: the public header must compile as C99 and a plain C client must link against the
    library with nothing else on the command line (no HIP, no torch, no C++ runtime flags)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "deepsignal_hip.h"\nint main(void) { ds_config c; (void)c; return 0; }\n')
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I",
                    os.path.join(root, "include"), str(src)], check=True)
    lib = os.path.join(root, "deepsignal_amd", "libdeepsignal_hip.so")
    if not os.path.exists(lib):
        pytest.skip("native library not built")
    out = tmp_path / "abi_client"
    subprocess.run([gcc, "-std=c99", "-Wall", "-O1", os.path.join(root, "tests", "abi_client.c"), "-o", str(out),
                    "-L" + os.path.dirname(lib), "-ldeepsignal_hip", "-Wl,-rpath," + os.path.dirname(lib)], check=True)
    assert out.exists()
