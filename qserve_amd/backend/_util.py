"""Helpers shared by the `qserve_backend` mirror modules: tensor -> device pointer lowering and checks."""
import torch

from .._lib import check, lib  # noqa: F401  (re-exported)


def ptr(t):
    return 0 if t is None else t.data_ptr()


def stream():
    """Current HIP stream as an integer handle (the reference GEMMs use the legacy default stream and the
    attention ops the current stream; the engine never switches streams, so 'current' is equivalent and
    makes the ops capturable in a hipGraph)."""
    return torch.cuda.current_stream().cuda_stream


class guard:
    """Device guard (the reference wraps its attention launch in at::cuda::CUDAGuard, fused_attention.cpp:203; the
    library's per-device scratch and the launch itself follow the CURRENT device): make the tensor's device current for
    the duration of the call.  One cached current_device() query when the device already matches."""
    __slots__ = ("idx", "prev")

    def __init__(self, t):
        self.idx = t.device.index

    def __enter__(self):
        self.prev = torch.cuda.current_device()
        if self.idx is not None and self.prev != self.idx:
            torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.idx is not None and self.prev != self.idx:
            torch.cuda.set_device(self.prev)
        return False


def on_device(t):
    return t.is_cuda


def expect(t, dtype, name, contiguous=True):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.dtype != dtype:
        # the reference's data_ptr<T>() throws on a dtype mismatch
        raise RuntimeError(f"expected scalar type {dtype} for {name} but found {t.dtype}")
    if not on_device(t):
        raise RuntimeError(f"{name} must be on CUDA")   # wording of the reference's CHECK_DEVICE
    if contiguous and not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
