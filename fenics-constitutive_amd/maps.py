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

__all__ = ["DeviceIdentityMap", "DeviceSubSpaceMap", "row_order", "ascending_order", "warn_if_rows_not_ascending"]

_warned_rows = False


def row_order(parent_rows) -> dict:
    """How the fused indexed kernel (``DeviceLaw.evaluate_indexed``: stress / tangent rows addressed through ``parent_rows``) will see
    a map: ``ascending`` -- the rows never go down; ``consecutive_tiles`` -- the share of 64-point tiles whose rows are one run
    (they take the coalesced tile body on shifted base pointers, csrc/fcamd_kernels.hip: run_full_tile)."""
    import numpy as np

    r = np.asarray(parent_rows).astype(np.int64, copy=False).ravel()
    if r.size < 2:
        return {"ascending": True, "consecutive_tiles": 1.0}
    d = np.diff(r)
    full = (r.size // 64) * 64
    run = 1.0
    if full:
        inside = np.ones(full, dtype=bool)
        inside[63::64] = False  # the step from one tile to the next does not count
        steps = np.concatenate([d, [1]])[:full]
        run = float(np.mean(np.all(((steps == 1) | ~inside).reshape(-1, 64), axis=1)))
    return {"ascending": bool(np.all(d >= 0)), "consecutive_tiles": run}


def ascending_order(parent_rows):
    """Permutation that visits a law's points in ascending parent-row order (stable).  Apply it ONCE, where the law's local arrays
    are laid out -- ``parent_rows[order]`` as the map, the gradient producer writing row ``k`` for point ``order[k]``, the history
    arrays permuted the same way -- and the indexed kernel meets cells as runs of consecutive rows."""
    import numpy as np

    return np.argsort(np.asarray(parent_rows), kind="stable")


def warn_if_rows_not_ascending(parent_rows, who: str) -> None:
    """The reference builds its submesh maps from the cells of a material in ascending order (solver/maps.py:127-178): rows rise,
    cells are runs.  A map whose rows go up and down makes every 336-byte stress + tangent row of a tile a scattered write:
    measured at 5e7 of 1e8 parent rows (profiles/r05_default_bench_rocprof.md, row `indexed_permuted`) 0.33-0.41 of the HBM roofline at 1.29 x
    the algorithmic bytes, against 0.76-0.82 for ascending maps.  Said once per process."""
    global _warned_rows
    if _warned_rows:
        return
    o = row_order(parent_rows)
    if not o["ascending"]:
        import warnings

        _warned_rows = True
        warnings.warn(
            f"{who}: the parent rows of this law are not ascending ({100.0 * o['consecutive_tiles']:.0f} % of its 64-point tiles are runs of "
            "consecutive rows).  The fused indexed kernel then scatters every 336-byte stress + tangent row: about 0.35 of the HBM "
            "roofline at 1.3 x the bytes instead of 0.76-0.82 (measured, MI355X).  Lay the law's points out in ascending parent-row "
            "order once (fenics_constitutive_amd.maps.ascending_order), as the reference's submesh maps are (solver/maps.py:127-178).",
            RuntimeWarning, stacklevel=3)


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
        if not hasattr(parent, "is_cuda") or not parent.is_cuda:  # (host index arrays: looked at once, at construction, as maps.py builds them once)
            warn_if_rows_not_ascending(parent, "DeviceSubSpaceMap")
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
