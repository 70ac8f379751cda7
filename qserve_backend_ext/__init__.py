"""`qserve_backend` as a COMPILED torch extension (pybind11, torch::Tensor arguments) over libqserve_amd.so - the form of
the boundary the reference itself binds (kernels/setup.py:157-245, pybind.cpp:13-16): csrc/binding.cpp.

The seven modules of the reference are sub-modules of one shared object (`_C`); importing this package registers them as
`qserve_backend_ext.<module>`, and `install()` additionally under the reference's own import names `qserve_backend.<module>`
(replacing the ctypes mirror `qserve_backend/` for the process).  The default package `qserve_backend/` stays the ctypes
mirror: it needs no compiler at install time and carries the engine-side extras (plan-only entries, fusions)."""
import importlib
import sys

MODULES = ["qgemm_w4a8_per_chn", "qgemm_w4a8_per_group", "qgemm_w8a8", "fused_attention", "fused_kernels",
           "layernorm_ops", "activation_ops"]


_C = None


def load():
    """Import the shared object (built by `python -m qserve_backend_ext.build`) and register its seven sub-modules as
    `qserve_backend_ext.<module>`.  Called on package import when the object exists; raises ImportError otherwise."""
    global _C
    if _C is None:
        import os
        import torch  # noqa: F401  (libtorch must be loaded before the extension)
        alt = os.environ.get("QS_AMD_LIBRARY")
        if alt and os.path.realpath(alt) != os.path.realpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                                                          "qserve_amd", "libqserve_amd.so")):
            # the extension is linked (rpath) against qserve_amd/libqserve_amd.so; with QS_AMD_LIBRARY naming another build the
            # ctypes side would load a SECOND copy of the library - two sets of variant switches and scratch areas in one process
            raise ImportError(f"qserve_backend_ext is linked against qserve_amd/libqserve_amd.so, but QS_AMD_LIBRARY={alt} "
                              "selects another library for the ctypes side: unset it to use the compiled extension")
        try:
            _C = importlib.import_module(__name__ + "._C")
        except ImportError as e:
            raise ImportError(f"{__name__}._C is not built: run `python -m qserve_backend_ext.build` ({e})") from e
        for m in MODULES:
            mod = getattr(_C, m)
            sys.modules[__name__ + "." + m] = mod
            globals()[m] = mod
    return _C


def install():
    """Make `import qserve_backend.<module>` resolve to the compiled extension for this process."""
    c = load()
    pkg = importlib.import_module("qserve_backend")
    for m in MODULES:
        sys.modules["qserve_backend." + m] = getattr(c, m)
        setattr(pkg, m, getattr(c, m))


try:
    load()
except ImportError:      # not built yet (e.g. while `python -m qserve_backend_ext.build` itself is starting)
    pass
