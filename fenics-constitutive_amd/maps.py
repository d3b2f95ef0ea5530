"""Device form of the parent <-> submesh quadrature maps (reference: ``solver/maps.py:29-123``).

Multi-material problems gather the committed stress of a material's cells from the parent
array before ``evaluate`` and scatter stress and tangent back afterwards
(``solver/_lawonsubmesh.py:58-70``).  ``DeviceSubSpaceMap`` does both on the GPU through
``fcamd_map_rows_device`` so that device-resident state never returns to the host for it.
Only the index arithmetic of the reference classes is mirrored (no dolfinx objects, no ghost
update: quadrature data is cell-local).
"""

from __future__ import annotations

import ctypes as C

from . import _capi
from .device import _check_torch, _current_stream_ptr

__all__ = ["DeviceIdentityMap", "DeviceSubSpaceMap"]


def _rows(ctx, n_rows, row_size, src, src_idx, dst, dst_idx):
    _capi.check(ctx._lib.fcamd_map_rows_device(
        ctx.handle, int(n_rows), int(row_size), C.c_void_p(src.data_ptr()),
        C.c_void_p(0 if src_idx is None else src_idx.data_ptr()), C.c_void_p(dst.data_ptr()),
        C.c_void_p(0 if dst_idx is None else dst_idx.data_ptr())))


class DeviceSubSpaceMap:
    """``parent`` / ``sub``: equally long row-index arrays (``SubSpaceMap.parent`` / ``.sub``,
    solver/maps.py:75-79), given as NumPy int arrays or int32 device tensors."""

    def __init__(self, parent, sub, device=None):
        import torch

        dev = torch.device("cuda", _capi.default_device()) if device is None else torch.device(device)
        self.parent = torch.as_tensor(parent, dtype=torch.int32).to(dev).contiguous()
        self.sub = torch.as_tensor(sub, dtype=torch.int32).to(dev).contiguous()
        assert self.parent.numel() == self.sub.numel(), "index arrays must have equal length"
        self.device = dev

    def _ctx(self):
        ctx = _capi.get_context(self.device.index or 0)
        ctx.set_stream(_current_stream_ptr(self.device.index or 0))
        return ctx

    def map_to_parent(self, sub, parent, size: int) -> None:
        """``parent.reshape(-1, size)[self.parent] = sub.reshape(-1, size)[self.sub]`` (maps.py:82-101)."""
        _check_torch("sub", sub), _check_torch("parent", parent)
        _rows(self._ctx(), self.sub.numel(), size, sub, self.sub, parent, self.parent)

    def map_to_sub(self, parent, sub, size: int) -> None:
        """``sub.reshape(-1, size)[self.sub] = parent.reshape(-1, size)[self.parent]`` (maps.py:103-123)."""
        _check_torch("sub", sub), _check_torch("parent", parent)
        _rows(self._ctx(), self.sub.numel(), size, parent, self.parent, sub, self.sub)


class DeviceIdentityMap:
    """Single-material case: plain copies (``IdentityMap``, maps.py:29-59)."""

    def map_to_parent(self, sub, parent, size: int = 1) -> None:
        assert sub.numel() == parent.numel(), "Shapes do not match"
        parent.copy_(sub)

    def map_to_sub(self, parent, sub, size: int = 1) -> None:
        assert sub.numel() == parent.numel(), "Shapes do not match"
        sub.copy_(parent)
