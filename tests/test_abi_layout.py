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
