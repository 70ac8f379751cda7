"""Build qserve_backend_ext/_C*.so: the compiled torch-extension form of the drop-in boundary (csrc/binding.cpp) with
torch.utils.cpp_extension, in-tree, linked against ../qserve_amd/libqserve_amd.so (rpath $ORIGIN/../qserve_amd).

    python -m qserve_backend_ext.build [--force]

Plain C++ (no device code here - the kernels live in libqserve_amd.so), so this also runs on a machine without a GPU."""
import glob
import os
import shutil
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "binding.cpp")
SUFFIX = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
OUT = os.path.join(HERE, "_C" + SUFFIX)


def build(force=False, verbose=True):
    from qserve_amd import build as libbuild
    lib = libbuild.build(verbose=False)
    deps = [SRC, os.path.join(ROOT, "include", "qserve_amd.h")]
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(d) for d in deps):
        if verbose:
            print(f"[qserve_backend_ext.build] up to date: {OUT}")
        return OUT
    from torch.utils import cpp_extension as ce
    rocm = os.environ.get("ROCM_PATH") or os.environ.get("ROCM_HOME") or "/opt/rocm"
    inc = [os.path.join(ROOT, "include"), os.path.join(rocm, "include")] + ce.include_paths()
    cxx = os.environ.get("CXX") or shutil.which("g++") or "g++"
    import torch
    abi = int(getattr(torch._C, "_GLIBCXX_USE_CXX11_ABI", True))          # the ABI the installed torch was built with
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=_C", "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={abi}",
           "-Wno-deprecated-declarations"]
    cmd += ["-I" + sysconfig.get_paths()["include"]] + ["-I" + i for i in inc]
    cmd += [SRC, "-o", OUT]
    tl = (ce.library_paths() or [os.path.abspath(os.path.join(os.path.dirname(ce.__file__), "..", "lib"))])[0]
    # the libraries this torch build ships (a ROCm wheel: torch_hip / c10_hip); only those that exist are named
    libs = [l for l in ("torch", "torch_cpu", "torch_hip", "c10", "c10_hip", "torch_python")
            if glob.glob(os.path.join(tl, "lib" + l + ".so*"))]
    if "torch_hip" not in libs:
        raise RuntimeError(f"the installed torch ({torch.__version__}) is not a ROCm build: libtorch_hip is missing from {tl}")
    cmd += ["-L" + tl] + ["-l" + l for l in libs] + [
            "-L" + os.path.dirname(lib), "-l:libqserve_amd.so",
            "-Wl,-rpath,$ORIGIN/../qserve_amd", "-Wl,-rpath," + tl]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building the torch extension failed:\n" + r.stderr[-4000:])
    if verbose:
        print(f"[qserve_backend_ext.build] built {OUT}")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
