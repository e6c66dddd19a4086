"""Builds libmirhi.so (HIP kernels + C ABI) for gfx950 with hipcc. In-tree output: renderer-rs_amd/libmirhi.so."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmirhi.so")
SOURCES = ["mirhi_kernels.hip", "mirhi_api.hip"]
# every header of csrc/ (a glob: a new header cannot be forgotten -- round 2 missed mirhi_exact.hip.h, the exact-IEEE sequences the
# parity rests on, so neither needs_build() nor source_hash() saw it change) plus the public header
HEADERS = sorted(os.path.basename(h) for h in __import__("glob").glob(os.path.join(CSRC, "*.h"))) + [os.path.join("..", "..", "include", "mirhi.h")]
# -ffp-contract=off: coverage/depth arithmetic must match the oracle bit for bit
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X kernels cannot be built")


def needs_build() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(os.path.join(HERE, "libmirhi_kernels.hsaco")):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the kernel and C-ABI sources: tags measurements (profiles/*_hbm_traffic.json) with the
    build they were taken on, so that bench.py never quotes counters of another build."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES + HEADERS):
        path = os.path.join(CSRC, name)
        if os.path.exists(path):
            h.update(os.path.basename(path).encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str = LIB, suffix: str = "") -> str:
    if not force and out == LIB and not needs_build():
        return LIB
    objs = []
    for s in SOURCES:
        obj = os.path.join(CSRC, s + suffix + ".o")
        cmd = [hipcc(), *FLAGS, *extra_flags, f'-DMIRHI_SOURCE_HASH="{source_hash()}"', "-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, "-L/opt/rocm/lib", "-lhsa-runtime64"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    extract_code_object(objs[0], out[:-3] + "_kernels.hsaco", verbose)      # (libmirhi.so -> libmirhi_kernels.hsaco: what the native dispatcher of THAT library loads)
    return out


HSACO = os.path.join(HERE, "libmirhi_kernels.hsaco")


def extract_code_object(obj: str, dst: str, verbose: bool = False) -> None:
    """The gfx950 code object of the kernels, as a file beside the library: the native dispatcher (csrc/mirhi_native.h) loads it through
    ROCr.  Taken out of the object hipcc just produced (its .hip_fatbin section is a clang offload bundle) -- the same bits the HIP
    runtime loads, no second compilation."""
    llvm = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc()))), "lib", "llvm", "bin")
    if not os.path.exists(os.path.join(llvm, "llvm-objcopy")):
        llvm = "/opt/rocm/lib/llvm/bin"
    fat = dst + ".bundle"
    for cmd in ([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj],
                [os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + dst]):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    os.remove(fat)


def build_stamps(verbose: bool = False) -> str:
    """Diagnostic build with per-wave s_memtime stamps (never the product library)."""
    return build(force=True, verbose=verbose, extra_flags=["-DMIRHI_STAMPS"], out=os.path.join(HERE, "libmirhi_stamps.so"),
                 suffix=".stamps")


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        print(build_stamps(verbose=True))
        sys.exit(0)
    if "--variant" in sys.argv:          # build.py --variant NAME -DX=1 ...: libmirhi_NAME.so (+ its code object) beside the product, for tools/ab_bench.sh
        name = sys.argv[sys.argv.index("--variant") + 1]
        print(build(force=True, verbose=True, extra_flags=[a for a in sys.argv if a.startswith("-D")], out=os.path.join(HERE, f"libmirhi_{name}.so"), suffix="." + name))
        sys.exit(0)
    build(force="--force" in sys.argv, verbose=True,
          extra_flags=["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else ())
    print(LIB)
