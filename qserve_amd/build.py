"""Build libqserve_amd.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m qserve_amd.build [--force] [--debug-asm]

hipcc cross-compiles without a GPU, so this runs in the authoring container as well as on the MI355X box.
The .so lands in qserve_amd/ (git-ignored, but shipped to the GPU box by gpurun).
"""
import argparse
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libqserve_amd.so")
SOURCES = ["lib.hip", "gemm_w4a8.hip", "gemm_w4a8_lds.hip", "gemm_w4a8_tiled.hip", "gemm_w4a8_wide.hip", "gemm_w4a8_ring.hip", "gemm_w8a8.hip", "attention.hip", "attention_mfma.hip", "attention_mfma8.hip", "flash_prefill.hip", "fused_small.hip", "direct_allreduce.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm >= 7.0 with the gfx950 target)")
    return exe


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "qserve_amd.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force, extra, build_dir):
    obj = os.path.join(build_dir, src.replace(".hip", ".o"))
    srcp = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(srcp), _deps_mtime())):
        return obj, False
    cmd = [hipcc(), *FLAGS, *extra, "-c", srcp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
    return obj, True


def build(force=False, verbose=True, extra=(), timing=False):
    """timing=True: the measurement build (-DQS_TIMING: ablation switches that turn kernel parts off, timeline traces) ->
    libqserve_amd_timing.so from its own object directory; never loaded by the product (qserve_amd/_lib.py loads it only when
    QS_AMD_LIBRARY names it - scripts/)."""
    build_dir, lib = BUILD, LIB      # (locals: a timing build must not redirect later product builds of this process)
    if timing:
        build_dir, lib = os.path.join(HERE, "_build_timing"), os.path.join(HERE, "libqserve_amd_timing.so")
        extra = list(extra) + ["-DQS_TIMING"]
    os.makedirs(build_dir, exist_ok=True)
    extra = list(extra) + os.environ.get("QS_EXTRA_HIPCC_FLAGS", "").split()   # e.g. -DQS_RING_TRACE (timing tools)
    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        res = list(ex.map(lambda s: _compile(s, force, list(extra), build_dir), SOURCES))
    objs = [o for o, _ in res]
    rebuilt = any(r for _, r in res)
    if rebuilt or not os.path.exists(lib) or force:
        # (--no-undefined: a symbol one translation unit expects from another and does not get must fail HERE, not at dlopen)
        cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-Wl,--no-undefined", *objs, "-o", lib]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[qserve_amd.build] linked {lib}")
    elif verbose:
        print(f"[qserve_amd.build] up to date: {lib}")
    return lib


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--debug-asm", action="store_true",
                    help="keep the gfx950 assembly of every kernel next to the objects (qserve_amd/_build/*.s) and print "
                         "the per-kernel register / LDS / scratch usage (implies --force)")
    ap.add_argument("--timing", action="store_true", help="build libqserve_amd_timing.so (-DQS_TIMING) instead")
    a = ap.parse_args()
    extra = ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"] if a.debug_asm else []
    build(force=a.force or a.debug_asm, extra=extra, timing=a.timing)
    sys.exit(0)
