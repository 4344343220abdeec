"""The C-ABI library loads without a GPU and exports every symbol include/flux_abi.h declares.
No compute entry point is exercised here (that is tests/test_gpu_parity.py, -m gpu)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "flux_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(flux_\w+)\s*\(", text)))


def test_header_functions_are_exported(flux):
    names = _declared_functions()
    assert len(names) >= 15 and "flux_ctx_create" in names and "flux_render_rows" in names
    lib = ctypes.CDLL(flux._lib.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # and the python binding covers exactly the header
    assert sorted(flux._lib.SYMBOLS) == names


def test_version_and_error_surface(flux):
    lib = flux._lib.lib
    assert lib.flux_abi_version() == 3 == flux._lib.ABI_VERSION
    header = open(os.path.join(ROOT, "include", "flux_abi.h")).read()
    assert re.search(r"#define\s+FLUX_ABI_VERSION\s+3\b", header)
    # introspection calls check their arguments without a device
    assert lib.flux_ctx_bvh_info(None, None, 16) == flux._lib.E_INVALID
    assert lib.flux_ctx_launch_plan(None, 0, 0, None) == flux._lib.E_INVALID
    assert lib.flux_device_count() >= 0
    assert lib.flux_ctx_create(None, None, 0, 0, None) == flux._lib.E_INVALID
    assert b"null" in lib.flux_last_error()
    lib.flux_ctx_destroy(None)  # no-op, like dropping nothing
    assert lib.flux_ctx_device_bytes(None) == 0
    assert lib.flux_ctx_last_kernel_ms(None) < 0
    # the multi-GPU entry points (ABI version 3) check their arguments before they look for a device or for RCCL
    assert lib.flux_multi_create(None, None, 0, None, 0, 0, None) == flux._lib.E_INVALID
    assert lib.flux_multi_render_frame(None, None) == flux._lib.E_INVALID
    assert lib.flux_multi_info(None, None) == flux._lib.E_INVALID
    assert lib.flux_render_frame_multi(None, None, 0, None, 0, 0, None) == flux._lib.E_INVALID
    lib.flux_multi_destroy(None)
    assert lib.flux_multi_release_comms() == 0  # nothing cached: no communicator was ever made
    assert lib.flux_ctx_create_timing(None, None) == flux._lib.E_INVALID


def test_struct_layout_matches_header(flux):
    """sizeof the ctypes mirrors == what a C compiler lays out for include/flux_abi.h."""
    import subprocess
    import tempfile
    src = '#include <stdio.h>\n#include "flux_abi.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n",sizeof(flux_material),' \
          'sizeof(flux_shape),sizeof(flux_scene_desc),sizeof(flux_job_cfg),sizeof(flux_work_unit));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        sizes = [int(x) for x in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    L = flux._lib
    assert sizes == [ctypes.sizeof(L.FluxMaterial), ctypes.sizeof(L.FluxShape), ctypes.sizeof(L.FluxSceneDesc),
                     ctypes.sizeof(L.FluxJobCfg), ctypes.sizeof(L.FluxWorkUnit)]


def test_product_never_touches_the_oracle():
    """Nothing under flux_amd/ may import, link or call anything under oracle/."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "flux_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".inc", ".cpp", ".c")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"\boracle\b|fxo_", text):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
