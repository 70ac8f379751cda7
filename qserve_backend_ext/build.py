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
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=_C", "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=1",
           "-Wno-deprecated-declarations"]
    cmd += ["-I" + sysconfig.get_paths()["include"]] + ["-I" + i for i in inc]
    cmd += [SRC, "-o", OUT]
    tl = os.path.join(os.path.dirname(ce.__file__), "..", "lib")
    tl = os.path.abspath(tl)
    cmd += ["-L" + tl, "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", "-ltorch_python",
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
