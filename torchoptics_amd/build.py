"""
Builds torchoptics_amd/libtltrace.so (the C-ABI HIP library) for gfx950 with hipcc.

    python -m torchoptics_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects go to build/ (git-ignored); the .so is written
in-tree next to this file so it travels with the repo snapshot to the GPU box.
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(ROOT, "build", "tltrace")
LIB = os.path.join(HERE, "libtltrace.so")

ARCH = "gfx950"
# -fno-slp-vectorize: hipcc's SLP pass packs neighbouring scalar fp32 ops into v_pk_mul/add_f32,
#   which on CDNA4 issue no faster than two scalar ops but need 64-bit-aligned register pairs
#   (hundreds of extra v_mov, +49 VGPRs in the backward kernel): measured -11 % kernel time without it.
COMMON = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
          "-fno-slp-vectorize",
          # The fully unrolled backward kernels hold hundreds of LDS/global stores.  With LLVM's default
          # MemorySSA walk limit (100) the AMDGPU back end gives up proving that the wave-uniform loads of
          # the prescription (c, t, mu, kappa, poly) are never clobbered and emits them as VECTOR loads
          # (180 global_load per ray in trace_bwd_kernel<12,asph>) instead of scalar s_load.
          "-mllvm", "-memssa-check-limit=4000",
          # Long dependent FMA chains with independent side computations (the rounding fix-ups of sqrt, the
          # adjoint's parallel branches): the ILP-driven iterative scheduler interleaves them better than the
          # default (same instructions, bit-identical results; measured -1.5 % forward, -4..6 % walk-back).
          "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
# strict: no FMA contraction, HIP's default correctly rounded fp32 sqrt / divide
# fast  : contraction on; the kernels call v_rcp / v_sqrt explicitly
UNITS = {
    "tl_strict.hip": ["-ffp-contract=off"],
    "tl_fast.hip": ["-ffp-contract=fast"],
    "tl_api.hip": [],
    "tl_f64.hip": [],          # the double-precision twin (generic, untuned)
}
DEPS = ["tl_kernels.inc", "tl_common.h", os.path.join("..", "..", "include", "tl_trace.h")]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def _digest(paths, flags):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def build_library(force=False, verbose=True, tag=None, extra_flags=()):
    """Compile every translation unit and link libtltrace.so; returns its path.

    `tag` / `extra_flags` build an experimental variant libtltrace_<tag>.so next to the default
    one (used only by tools/ab_kernels.py for A/B timing)."""
    global OBJ, LIB
    obj_dir = OBJ if tag is None else OBJ + "_" + tag
    lib_path = LIB if tag is None else os.path.join(HERE, f"libtltrace_{tag}.so")
    return _build(obj_dir, lib_path, list(extra_flags), force, verbose)


def _build(OBJ, LIB, extra_flags, force, verbose):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    deps = [os.path.normpath(os.path.join(CSRC, d)) for d in DEPS]
    objs, jobs = [], []
    for src, extra in UNITS.items():
        spath = os.path.join(CSRC, src)
        opath = os.path.join(OBJ, src.replace(".hip", ".o"))
        stamp = opath + ".sha"
        flags = COMMON + extra + extra_flags
        dig = _digest([spath] + deps, flags)
        fresh = (not force and os.path.exists(opath) and os.path.exists(stamp)
                 and open(stamp).read() == dig)
        if not fresh:
            jobs.append(([hipcc] + flags + ["-c", spath, "-o", opath], stamp, dig))
        objs.append(opath)

    def compile_one(job):
        cmd, stamp, dig = job
        if verbose:
            print("[tltrace]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(stamp, "w") as f:
            f.write(dig)
    if jobs:        # the translation units are independent: compile them side by side (each hipcc is one process)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), 4)) as pool:
            list(pool.map(compile_one, jobs))
    rebuilt = bool(jobs)
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[tltrace]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


EXT = os.path.join(HERE, "_tlx.so")


def build_host_extension(force=False, verbose=True):
    """torchoptics_amd/_tlx.so: the eager host chain (argument normalisation, allocation, the calls into libtltrace.so's
    C ABI and the autograd nodes around them) as C++ torch::autograd::Functions (csrc/tl_torch.cpp).  Plain host C++ --
    no device code: compiled with g++ against PyTorch's headers, linked to libtltrace.so next to it (rpath $ORIGIN)."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    src = os.path.join(CSRC, "tl_torch.cpp")
    stamp = os.path.join(OBJ, "tl_torch.sha")
    os.makedirs(OBJ, exist_ok=True)
    flags = ["-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
             f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-DTORCH_EXTENSION_NAME=_tlx",
             "-DTORCH_API_INCLUDE_EXTENSION_H"]
    dig = _digest([src, os.path.join(ROOT, "include", "tl_trace.h")], flags + [torch.__version__])
    if not force and os.path.exists(EXT) and os.path.exists(stamp) and open(stamp).read() == dig:
        return EXT
    cxx = shutil.which("g++") or "g++"
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ([cxx] + flags + [src] + [f"-I{p}" for p in ce.include_paths()] +
           [f"-I{sysconfig.get_paths()['include']}", f"-I{rocm}/include"] +
           [f"-L{p}" for p in ce.library_paths()] + [f"-L{rocm}/lib", f"-L{HERE}"] +
           ["-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-ltorch_python", "-lamdhip64", "-ltltrace",
            "-Wl,-rpath,$ORIGIN"] + [f"-Wl,-rpath,{p}" for p in ce.library_paths()] + ["-o", EXT])
    if verbose:
        print("[tltrace]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(dig)
    return EXT


def build_cabi_demo(verbose=True):
    """examples/cabi_demo.bin: a plain C++ host program on the C ABI (no Python, no torch)."""
    src = os.path.join(ROOT, "examples", "cabi_demo.cpp")
    out = os.path.join(ROOT, "examples", "cabi_demo.bin")
    deps = [src, LIB, os.path.join(ROOT, "include", "tl_trace.h")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O2", "-std=c++17", src, "-I", os.path.join(ROOT, "include"),
           "-L", HERE, "-ltltrace", "-Wl,-rpath,$ORIGIN/../torchoptics_amd", "-o", out]
    if verbose:
        print("[tltrace]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
    print(build_host_extension(force="--force" in sys.argv))
    print(build_cabi_demo())
