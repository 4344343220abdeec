#!/usr/bin/env python3
"""Rebuild the in-tree library (and optionally experiment variants) without importing the package (which refuses to
import when the built library lacks a symbol).  usage: scripts/rebuild.py [--host] [name=-Dflag,-Dflag ...]"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "flux_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
os.makedirs(os.path.join(ROOT, "flux_amd", "variants"), exist_ok=True)
for a in sys.argv[1:]:
    if "=" in a:
        name, flags = a.split("=", 1)
        b.build_variant(os.path.join(ROOT, "flux_amd", "variants", f"libflux_hip_{name}.so"), [f for f in flags.split(",") if f])
        print("built variant", name)
print(b.build_hip(force=True))
if "--host" in sys.argv:
    print(b.build_host(force=True))
