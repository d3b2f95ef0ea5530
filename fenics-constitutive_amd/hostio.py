"""Host <-> device copies of the package: NumPy arrays of the caller on one side, ROCm tensors on the other.

``torch.from_numpy(a).to(device)`` / ``tensor.cpu()`` hand pageable memory to the HIP runtime, which (above 1 MiB)
page-locks it piecewise on the fly and remembers the locks in a cache keyed by address and size.  On this stack a
lock is an attribute of the process's pages: memory that appears later at a remembered address (an array that was
freed and allocated again, a heap that shrank and grew) carries none, the cache still calls it locked and the DMA
faults ("Memory access fault by GPU"; tools/hsa_lock_probe.c).  These helpers go through
``fcamd_copy_to_device`` / ``fcamd_copy_to_host`` instead: small copies through the context's own page-locked
scratch, large ones with the array page-locked for the duration of the copy.  They are synchronous and ordered
after the work queued on torch's current stream."""

from __future__ import annotations

import numpy as np

from . import _capi


def _ctx(device):
    import torch

    from .device import _current_stream_ptr

    dev = torch.device(device)
    index = _capi.default_device() if dev.index is None else dev.index
    ctx = _capi.get_context(index)
    ctx.set_stream(_current_stream_ptr(index))
    return ctx


def upload(dst, src: np.ndarray) -> None:
    """``dst`` (contiguous ROCm tensor) <- ``src`` (NumPy array of the same dtype and number of elements)."""
    a = np.ascontiguousarray(src)
    assert dst.is_cuda and dst.is_contiguous(), "upload: destination must be a contiguous device tensor"
    assert a.nbytes == dst.numel() * dst.element_size(), "upload: sizes differ"
    if a.nbytes:
        _ctx(dst.device).copy_to_device(dst.data_ptr(), a)


def to_device(src: np.ndarray, device, dtype=None):
    """New device tensor holding ``src`` (converted to ``dtype`` -- a NumPy dtype -- on the host first)."""
    import torch

    a = np.ascontiguousarray(src, dtype=dtype)
    out = torch.empty(a.shape, dtype=torch.from_numpy(a[:0]).dtype, device=device)
    upload(out, a)
    return out


def download(dst: np.ndarray, src) -> None:
    """``dst`` (C-contiguous NumPy array) <- ``src`` (contiguous ROCm tensor): no temporary, unlike ``dst[:] = src.cpu().numpy()``."""
    assert src.is_cuda and src.is_contiguous(), "download: source must be a contiguous device tensor"
    assert dst.flags.c_contiguous and dst.flags.writeable, "download: destination must be a writeable C-contiguous array"
    assert dst.nbytes == src.numel() * src.element_size(), "download: sizes differ"
    if dst.nbytes:
        _ctx(src.device).copy_to_host(dst, src.data_ptr())


def to_host(src) -> np.ndarray:
    """New NumPy array holding the (contiguous) device tensor ``src``."""
    import torch

    out = np.empty(tuple(src.shape), dtype=torch.empty(0, dtype=src.dtype).numpy().dtype)
    download(out, src.contiguous())
    return out


def assign(dst: np.ndarray, src) -> None:
    """``dst[:] = src`` for a device tensor ``src``: straight into ``dst`` when it is C-contiguous, through a temporary otherwise."""
    if dst.flags.c_contiguous and dst.flags.writeable and dst.dtype == np.float64 and src.is_contiguous():
        download(dst, src)
    else:
        dst[...] = to_host(src).reshape(dst.shape)


def copy_device(dst, src) -> None:
    """``dst`` <- ``src``, both contiguous device tensors of equal byte size (``fcamd_copy_device``: the access pattern of
    the evaluate kernels; asynchronous on torch's current stream)."""
    assert dst.is_cuda and src.is_cuda and dst.is_contiguous() and src.is_contiguous()
    nbytes = dst.numel() * dst.element_size()
    assert nbytes == src.numel() * src.element_size(), "copy_device: sizes differ"
    if nbytes:
        if dst.data_ptr() % 16 or src.data_ptr() % 16:
            dst.copy_(src)
        else:
            _ctx(dst.device).copy_device(dst.data_ptr(), src.data_ptr(), nbytes)
