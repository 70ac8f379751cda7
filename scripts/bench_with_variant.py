#!/usr/bin/env python3
"""bench.py with a library A/B switch set first: V=<code for qs_set_gemm_variant> python scripts/bench_with_variant.py [bench args]."""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (torch first: its HIP runtime must be the one the library binds to)
from qserve_amd._lib import lib
lib.qs_set_gemm_variant(int(os.environ.get("V", "-1")))
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
