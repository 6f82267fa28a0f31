"""Recipe for oracle/_ref/: the part of the REFERENCE that compiles here from its own source file.

    python -m oracle.build_ref            (also run by __graft_entry__.build() when /root/reference is present)

The reference's only native code on the path is the curope extension (uniception/models/libs/croco/curope/): curope.cpp holds the
dispatcher `rope_2d` and the CPU loop `rope_2d_cpu` (curope.cpp:11-46), kernels.cu the CUDA kernel.  curope.cpp is compiled AS IT LIES
under /root/reference (nothing is copied into this repository, nothing is written in its place) with g++ against the PyTorch headers
of this image into oracle/_ref/curope_ref.so.  kernels.cu needs nvcc and is not built: the shared object keeps `rope_2d_cuda` as an
undefined symbol, which a lazily bound load (`load()` below: RTLD_LAZY, link flag -z lazy) never resolves because CPU tensors take
the `rope_2d_cpu` branch.  Test infrastructure: only tests/ and __graft_entry__ use it — as the checker of oracle.rope2d and of
uc_rope2d (tests/test_oracle_golden.py, tests/test_ops_gpu.py) — never the product path.  oracle/_ref/ is git-ignored and travels to
the GPU box with the snapshot like the other built objects."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/uniception/models/libs/croco/curope/curope.cpp"
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "curope_ref.so")


def build(verbose: bool = True) -> str:
    "Compile the reference's curope.cpp (if /root/reference is here); returns the path of the shared object, or '' when it cannot be built."
    if not os.path.exists(SRC):
        return OUT if os.path.exists(OUT) else ""
    if os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    from torch.utils import cpp_extension
    os.makedirs(OUT_DIR, exist_ok=True)
    inc = [f"-I{p}" for p in cpp_extension.include_paths()] + [f"-I{sysconfig.get_paths()['include']}"]
    import torch
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-DTORCH_EXTENSION_NAME=curope_ref", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"] + inc + [SRC, "-o", OUT, f"-L{libdir}", "-ltorch_python", "-ltorch",
           "-ltorch_cpu", "-lc10", f"-Wl,-rpath,{libdir}", "-Wl,-z,lazy"]
    if verbose:
        print("[oracle.build_ref]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


def load():
    "Import oracle/_ref/curope_ref.so (lazy binding: its CUDA half stays an unresolved symbol that CPU tensors never reach); None if absent."
    if not os.path.exists(OUT):
        return None
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    flags = sys.getdlopenflags()
    sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_LOCAL)
    try:
        spec = importlib.util.spec_from_file_location("curope_ref", OUT)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.setdlopenflags(flags)
    return mod


if __name__ == "__main__":
    print(build() or "reference source not present: nothing built")
