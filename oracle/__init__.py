"""CPU oracle for the W4A8KV4 hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy) of the reference algorithms on the hot path named by
BASELINE.json `north_star`.  Every function cites the reference file:line it follows (paths relative
to the upstream tree, mit-han-lab/qserve @ 2025-02-04).

It is a checker, never the product:
  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
  * nothing under `qserve_amd/` or `qserve_backend/` imports it, and the product path raises when the
    HIP library is missing instead of falling back to this code.

Pinning status (see DESIGN.md "Oracle"):
  * weight reorder / packing (`oracle.w4a8.pack_*`): PINNED against golden vectors produced by running
    the reference's own `from_linear` (tests/golden/make_golden.py -> tests/golden/*.npz);
  * GEMM arithmetic, KV-cache quantisation, decode attention: the reference has no tests, fixtures or
    golden vectors for them and its CUDA/PTX kernels cannot be built here (no nvcc) -> PARITY UNPINNED;
    these restatements follow the kernel sources line by line.
"""
