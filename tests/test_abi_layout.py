"""The boundary's struct layouts cannot drift: include/flux_abi.h (compiled here with gcc) against
 (1) the layout table documented in INTEGRATION.md, (2) the field order of INTEGRATION.md's Rust `#[repr(C)]` block
 (what a maintainer of the reference pastes next to fluxcore/src/workers.rs:46-60), (3) the ctypes mirror in
 flux_amd/_lib.py that every Python test goes through.  CPU only: no compute call."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

STRUCTS = {
    "flux_material": ["kind", "reserved", "color", "ambient", "k", "exponent"],
    "flux_shape": ["kind", "invert", "p", "n", "radius", "material"],
    "flux_mesh": ["num_vertices", "vertices", "num_triangles", "indices", "material"],
    "flux_scene_desc": ["scene_name", "image_width", "image_height", "pixel_size", "background", "eye", "look_at", "up",
                        "zoom_factor", "view_plane_distance", "focal_distance", "lens_radius", "num_shapes", "shapes",
                        "num_meshes", "meshes"],
    "flux_job_cfg": ["sample_root", "max_trace_depth", "rows_per_work_unit"],
    "flux_work_unit": ["row_start", "row_end"],
}
RUST_NAMES = {"flux_material": "FluxMaterial", "flux_shape": "FluxShape", "flux_mesh": "FluxMesh",
              "flux_scene_desc": "FluxSceneDesc", "flux_job_cfg": "FluxJobCfg", "flux_work_unit": "FluxWorkUnit"}


@pytest.fixture(scope="module")
def header_layout(tmp_path_factory):
    """sizeof / offsetof of every field as gcc lays out include/flux_abi.h."""
    d = tmp_path_factory.mktemp("abi")
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "flux_abi.h"', 'int main(void) {']
    for s, fields in STRUCTS.items():
        lines.append(f'  printf("{s} size %zu", sizeof({s}));')
        for f in fields:
            lines.append(f'  printf(" {f} %zu", offsetof({s}, {f}));')
        lines.append('  printf("\\n");')
    lines += ['  return 0;', '}']
    src = d / "layout.c"
    src.write_text("\n".join(lines))
    exe = d / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    return parse_table(out)


def parse_table(text):
    lay = {}
    for line in text.strip().splitlines():
        tok = line.split()
        assert tok[1] == "size"
        lay[tok[0]] = {"size": int(tok[2]), "fields": [(tok[k], int(tok[k + 1])) for k in range(3, len(tok), 2)]}
    return lay


def test_header_declares_every_struct_field(header_layout):
    # the header compiled, and every struct / field named here exists in it (offsetof would not compile otherwise)
    assert set(header_layout) == set(STRUCTS)
    hdr = open(os.path.join(ROOT, "include", "flux_abi.h")).read()
    for s, fields in STRUCTS.items():
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (s, s), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        declared = re.findall(r"(\w+)(?:\[\d+\])?;", body)
        assert declared == fields, (s, declared)      # the test's list IS the header's, in order, nothing missing


def test_integration_md_layout_table(header_layout):
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = parse_table(re.search(r"```abi-layout\n(.*?)```", md, re.S).group(1))
    assert table == header_layout


def test_integration_md_rust_block_field_order():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    rust = md[md.index("```rust"):]
    for s, fields in STRUCTS.items():
        m = re.search(r"pub struct %s \{(.*?)\}" % RUST_NAMES[s], rust, re.S)
        assert m, f"INTEGRATION.md has no #[repr(C)] struct {RUST_NAMES[s]}"
        assert re.findall(r"pub (\w+):", m.group(1)) == fields, s
        assert "#[repr(C)] pub struct %s" % RUST_NAMES[s] in rust


def test_ctypes_mirror_matches_header(header_layout):
    import ctypes as C
    from flux_amd import _lib
    mirror = {"flux_material": _lib.FluxMaterial, "flux_shape": _lib.FluxShape, "flux_mesh": _lib.FluxMesh,
              "flux_scene_desc": _lib.FluxSceneDesc, "flux_job_cfg": _lib.FluxJobCfg, "flux_work_unit": _lib.FluxWorkUnit}
    for s, cls in mirror.items():
        assert C.sizeof(cls) == header_layout[s]["size"], s
        assert [(n, getattr(cls, n).offset) for n, _ in cls._fields_] == header_layout[s]["fields"], s


# ---- functions: INTEGRATION.md's Rust `extern "C"` block against the header's prototypes ---------------
C_TO_RUST = {
    "int": "c_int", "double": "f64", "uint64_t": "u64", "int64_t": "i64", "uint32_t": "u32", "void": "()",
    "const char *": "*const c_char", "const double *": "*const f64", "double *": "*mut f64", "uint64_t *": "*mut u64",
    "int64_t *": "*mut i64", "int32_t *": "*mut i32", "const uint8_t *": "*const u8", "void *": "*mut c_void",
    "flux_ctx *": "*mut FluxCtx", "flux_ctx **": "*mut *mut FluxCtx", "const flux_scene_desc *": "*const FluxSceneDesc",
    "const flux_job_cfg *": "*const FluxJobCfg", "flux_work_unit *": "*mut FluxWorkUnit",
    "flux_multi *": "*mut FluxMulti", "flux_multi **": "*mut *mut FluxMulti", "const int *": "*const c_int",
    "const void **": "*mut *const c_void",
}


def _c_prototypes():
    """name -> (return type, [argument types]) of every function include/flux_abi.h declares (array parameters decay)."""
    hdr = open(os.path.join(ROOT, "include", "flux_abi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = {}
    for m in re.finditer(r"^([a-z_0-9]+(?:\s+[a-z_0-9]+)*\s*\**)\s*(flux_[a-z_0-9]+)\(([^;{]*?)\);", hdr, flags=re.M | re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        norm = lambda t: re.sub(r"\s*\*", " *", " ".join(t.split())).replace("* *", "**").strip()
        types = []
        if args != "void":
            for a in args.split(","):
                a = a.strip()
                arr = re.search(r"\[[A-Z_0-9a-z]*\]$", a)
                a = re.sub(r"\[[A-Z_0-9a-z]*\]$", "", a)
                mm = re.match(r"^(.*?)([a-z_0-9]+)$", a)
                t = norm(mm.group(1))
                if arr:
                    t = norm(t + " *")
                types.append(t)
        protos[name] = (norm(ret), types)
    return protos


def _rust_prototypes():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r'extern "C" \{(.*?)\n\}', md, flags=re.S)
    assert len(blocks) == 1, "ONE extern block binds the whole header"
    body = re.sub(r"//[^\n]*", "", blocks[0])
    protos = {}
    for m in re.finditer(r"pub fn (flux_[a-z_0-9]+)\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", body, flags=re.S):
        name, args, ret = m.group(1), " ".join(m.group(2).split()), (m.group(3) or "()").strip()
        types = [a.split(":", 1)[1].strip() for a in args.split(",") if a.strip()]
        protos[name] = (ret, types)
    return protos


def test_integration_md_rust_block_binds_every_function():
    c, rust = _c_prototypes(), _rust_prototypes()
    assert len(c) >= 38, sorted(c)                    # the parser saw the whole header
    assert set(rust) == set(c), (sorted(set(c) - set(rust)), sorted(set(rust) - set(c)))
    for name, (ret, args) in c.items():
        r_ret, r_args = rust[name]
        assert len(r_args) == len(args), (name, args, r_args)
        assert r_ret == C_TO_RUST[ret], (name, ret, r_ret)
        for k, (ca, ra) in enumerate(zip(args, r_args)):
            assert ra == C_TO_RUST[ca], (name, k, ca, ra)


def test_ctypes_binding_names_every_function():
    """flux_amd/_lib.py binds the same set (its SYMBOLS table is what tests/test_abi_symbols.py loads)."""
    src = open(os.path.join(ROOT, "flux_amd", "_lib.py")).read()
    for name in _c_prototypes():
        assert re.search(r"\b" + name + r"\b", src), name
