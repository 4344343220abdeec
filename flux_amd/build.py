"""Build the in-tree native library ``flux_amd/libflux_hip.so``: HIP kernels + the C ABI of
include/flux_abi.h, cross-compiled for gfx950 with hipcc (works without a GPU).

The ``.so`` is git-ignored but travels with the gpurun snapshot.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "flux_amd", "csrc")
HIP_SOURCES = ["abi.hip", "multi.hip", "tables.hip", "render.hip", "bvh.cpp"]
HIP_HEADERS = ["flux_ctx.h", "flux_device.h", "flux_rng.h", "flux_tables.h", "flux_bvh.h", "flux_math.h", "flux_math_coeffs.h",
               "render_body.inc"]
HIP_LIB = os.path.join(ROOT, "flux_amd", "libflux_hip.so")

# -ffp-contract=off is the translation units' default: host arithmetic, the table generator and the STRICT render kernels keep
# the reference's operation order and never fuse a*b+c (rustc does not contract).  The FAST render kernels -- the default
# arithmetic, FLUX_MATH_FAST -- are compiled under `#pragma clang fp contract(fast)` inside render.hip and DO use FMA; see
# DESIGN.md "Numerics" for what that changes (ulps) and the parity tests that bound it.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
             "-fno-fast-math", "-Wall", "-Wno-unused-function",
             # machine LICM hoists the ~35 f64 polynomial coefficients of flux_math.h into VGPRs for the
             # whole render loop (+60 VGPRs, one wave/SIMD less); rematerialising them at the use is free
             "-mllvm", "-disable-machine-licm"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


# the sources that become DEVICE code (the render / table kernels): their hash is the `kernels:` half of flux_build_id()
KERNEL_FILES = ["render.hip", "render_body.inc", "tables.hip", "flux_device.h", "flux_bvh.h", "flux_math.h", "flux_math_coeffs.h",
                "flux_rng.h", "flux_tables.h"]


def build_id(extra_flags=()):
    """"lib:<16 hex> kernels:<16 hex>": sha256 over (flags, every source) and over (flags, the device-code sources)."""
    import hashlib

    def digest(names):
        h = hashlib.sha256()
        h.update(" ".join(HIP_FLAGS + list(extra_flags)).encode())
        for n in names:
            path = os.path.join(ROOT, "include", n) if n == "flux_abi.h" else os.path.join(CSRC, n)
            h.update(n.encode() + b"\0")
            with open(path, "rb") as f:
                h.update(f.read())
        return h.hexdigest()[:16]
    return f"lib:{digest(sorted(set(HIP_SOURCES + HIP_HEADERS + ['flux_abi.h'])))} kernels:{digest(KERNEL_FILES)}"


def _id_flag(extra_flags=()):
    return "-DFLUX_BUILD_ID=\"" + build_id(extra_flags) + "\""


def build_variant(out_path, extra_flags, verbose=False):
    """Experiment builds (scripts/sweep_variants.py): same sources, extra -D/-f flags."""
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    flags = [f for f in HIP_FLAGS if not (f == "-ffp-contract=off" and any(x.startswith("-ffp-contract") for x in extra_flags))]
    cmd = [_hipcc()] + flags + list(extra_flags) + [_id_flag(extra_flags), "-o", out_path] + srcs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return out_path


def build_hip(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HIP_HEADERS] + [os.path.join(ROOT, "include", "flux_abi.h")]
    if not force and not _newer(HIP_LIB, deps):
        return HIP_LIB
    cmd = [_hipcc()] + HIP_FLAGS + [_id_flag(), "-o", HIP_LIB] + srcs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return HIP_LIB


HOST_DIR = os.path.join(ROOT, "flux_amd", "host")
HOST_SOURCES = ["flux_host.cpp", "yaml_lite.cpp", "cbor.cpp", "flux_net.cpp"]
HOST_BINARIES = {"flux": "flux_cli.cpp", "flux_host_test": "flux_host_test.cpp", "sampler_debug": "sampler_debug.cpp",
                 "flux_node": "flux_node.cpp"}


def build_host(force=False, verbose=False):
    """C++ host layer above the C ABI: the flag-compatible `flux` CLI and its CPU-only self test."""
    build_hip(force=False, verbose=verbose)
    outs = []
    common = [os.path.join(HOST_DIR, s) for s in HOST_SOURCES]
    hdrs = [os.path.join(HOST_DIR, h) for h in ("flux_host.hpp", "yaml_lite.hpp", "cbor.hpp", "flux_net.hpp")] + [os.path.join(ROOT, "include", "flux_abi.h")]
    for name, main in HOST_BINARIES.items():
        out = os.path.join(HOST_DIR, name)
        src = [os.path.join(HOST_DIR, main)] + common
        if force or _newer(out, src + hdrs + [HIP_LIB]):
            cmd = ["g++", "-O2", "-std=c++17", "-pthread", "-Wall", "-o", out] + src + [
                "-L" + os.path.join(ROOT, "flux_amd"), "-lflux_hip", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
        outs.append(out)
    return outs


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
