"""Extension (no reference counterpart): the host-side BVH builder of flux_amd/csrc/bvh.cpp checked on the CPU -- the binary
SAH tree, its 16-bit quantisation and the 4-wide tree + quad leaf records the FAST traversal kernel walks.  The invariants
(tests/bvh_selftest.cpp) are the ones the device code relies on: containment at every level, every triangle reachable exactly
once, leaf records equal to the DevTri operands bit for bit, and the stack bound."""
import os
import subprocess

from conftest import ROOT


def test_bvh_builder_invariants(tmp_path):
    exe = str(tmp_path / "bvh_selftest")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "bvh_selftest.cpp"),
                    os.path.join(ROOT, "flux_amd", "csrc", "bvh.cpp")], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-3000:]
    assert "all ok" in out.stdout
    for name in ("one triangle", "soup 1000", "grid 200x100", "grid 40x30 at 1e6", "300 coincident triangles",
                 "grid 300x200 (threaded build)"):
        assert f"ok {name}" in out.stdout
    assert "ok threaded build equals the serial build" in out.stdout
