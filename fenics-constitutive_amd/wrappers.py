"""3D -> plane-strain / uniaxial-strain wrappers (reference: ``models/utils.py:211-412``).

Same contract as the reference classes: a FULL (3-D) model is driven with 1-D / 2-D arrays by
copying the mapped components into cached 3-D arrays, evaluating the 3-D model and copying the
mapped components back; the history is the 3-D model's.  Here the cached 3-D arrays live on
the GPU and the component maps are device kernels (``fcamd_convert_device``), so a 1-D/2-D
problem moves only its own small arrays over PCIe.  Around ``VonMises3D`` the three steps are one
fused kernel (``fcamd_evaluate_device_wrapped``): only the cached 3-D stress exists, no 3-D
gradient or tangent array:

* NumPy in  -> low-dimensional arrays are uploaded, expanded, evaluated, shrunk, downloaded;
* torch ROCm tensors in -> zero copies.

The history dict must be of the same kind as the other arrays (NumPy history is staged per
call like the reference's in-place arrays).
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from .device import _check_torch, _current_stream_ptr, _is_torch
from .interfaces import IncrSmallStrainModel, StressStrainConstraint

__all__ = ["UniaxialStrainFrom3D", "PlaneStrainFrom3D"]


class _From3D(IncrSmallStrainModel):
    _constraint: StressStrainConstraint
    _kinds: tuple[int, int, int, int]  # grad->3d, stress->3d, stress<-3d, tangent<-3d
    #: use the fused kernel where one exists (LinearElasticityModel, the plasticity laws); False forces map -> evaluate -> map
    fused = True

    def __init__(self, model: IncrSmallStrainModel) -> None:
        assert model.constraint.name == "FULL"
        self.model = model
        self.stress_3d = None
        self.tangent_3d = None
        self.grad_del_u_3d = None

    @property
    def constraint(self) -> StressStrainConstraint:
        return self._constraint

    @property
    def history_dim(self):
        return self.model.history_dim

    def update(self) -> None:
        self.model.update()

    def _convert(self, ctx, kind, n, src, dst):
        _capi.check(ctx._lib.fcamd_convert_device(ctx.handle, kind, n, C.c_void_p(src.data_ptr()),
                                                  C.c_void_p(dst.data_ptr())))

    def evaluate(self, t, del_t, grad_del_u, stress, tangent, history) -> None:
        import torch

        gd2, sd = self.geometric_dim**2, self.stress_strain_dim
        host = not _is_torch(grad_del_u)
        if host:
            if not torch.cuda.is_available():
                raise RuntimeError("the 3D wrappers evaluate on the GPU and no HIP device is available")
            dev = torch.device("cuda", _capi.default_device())
            from .hostio import to_device

            g_lo = to_device(grad_del_u, dev, np.float64)
            s_lo = to_device(stress, dev, np.float64)
            t_lo = torch.empty(tangent.size, dtype=torch.float64, device=dev)
            h_dev = None if history is None else {k: to_device(v, dev, np.float64) for k, v in history.items()}
        else:
            g_lo, s_lo, t_lo, h_dev = (_check_torch("grad_del_u", grad_del_u), _check_torch("stress", stress),
                                       _check_torch("tangent", tangent), history)
            dev = g_lo.device
        n = g_lo.numel() // gd2
        assert n == s_lo.numel() // sd == t_lo.numel() // (sd * sd)
        if self.fused and getattr(self.model, "_model_id", None) in (_capi.LINEAR_ELASTICITY, _capi.VON_MISES_3D, _capi.COMFE_MISES_PLASTICITY,
                                                                      _capi.COMFE_DRUCKER_PRAGER, _capi.COMFE_DRUCKER_PRAGER_HYPERBOLIC):
            # fused kernel (fcamd_evaluate_device_wrapped): only the cached 3-D stress exists
            if self.stress_3d is None or self.stress_3d.numel() != 6 * n or self.stress_3d.device != dev:
                self.stress_3d = torch.zeros(6 * n, dtype=torch.float64, device=dev)
            hist = self.model._history_arrays(h_dev)
            for h in hist:
                _check_torch("history", h)
            m = self.model._handle(dev.index or 0)
            m.ctx.set_stream(_current_stream_ptr(dev.index or 0))
            m.evaluate_device_wrapped(self._constraint.value, t, del_t, n, g_lo.data_ptr(), s_lo.data_ptr(),
                                      t_lo.data_ptr(), self.stress_3d.data_ptr(), [h.data_ptr() for h in hist])
            if host:
                self.model.device_stats(dev.index or 0)  # raises on Newton non-convergence like the reference
                self._download(stress, tangent, history, s_lo, t_lo, h_dev)
            return
        if self.grad_del_u_3d is None or self.grad_del_u_3d.numel() != 9 * n or self.grad_del_u_3d.device != dev:
            # cached 3-D arrays (utils.py:253-266): zero-initialised once, unmapped components persist
            self.grad_del_u_3d = torch.zeros(9 * n, dtype=torch.float64, device=dev)
            self.stress_3d = torch.zeros(6 * n, dtype=torch.float64, device=dev)
            self.tangent_3d = torch.zeros(36 * n, dtype=torch.float64, device=dev)
        ctx = _capi.get_context(dev.index or 0)
        ctx.set_stream(_current_stream_ptr(dev.index or 0))
        k_g, k_s, k_sb, k_tb = self._kinds
        self._convert(ctx, k_g, n, g_lo, self.grad_del_u_3d)
        self._convert(ctx, k_s, n, s_lo, self.stress_3d)
        self.model.evaluate(t, del_t, self.grad_del_u_3d, self.stress_3d, self.tangent_3d, h_dev)
        self._convert(ctx, k_tb, n, self.tangent_3d, t_lo)
        self._convert(ctx, k_sb, n, self.stress_3d, s_lo)
        if host:
            self._download(stress, tangent, history, s_lo, t_lo, h_dev)

    @staticmethod
    def _download(stress, tangent, history, s_lo, t_lo, h_dev):
        from .hostio import assign

        assign(stress, s_lo)
        assign(tangent, t_lo)
        if history is not None:
            for k in history:
                assign(history[k], h_dev[k])


class UniaxialStrainFrom3D(_From3D):
    """Drive a 3-D model under uniaxial strain (reference ``UniaxialStrainFrom3D``,
    utils.py:211-294): component 11 of gradient, stress and tangent."""

    _constraint = StressStrainConstraint.UNIAXIAL_STRAIN
    _kinds = (_capi.GRAD_1D_TO_3D, _capi.STRESS_1D_TO_3D, _capi.STRESS_3D_TO_1D, _capi.TANGENT_3D_TO_1D)


class PlaneStrainFrom3D(_From3D):
    """Drive a 3-D model under plane strain (reference ``PlaneStrainFrom3D``, utils.py:297-412):
    gradient components (0,1,2,3)->(0,1,3,4), Mandel components 0..3, tangent block 4x4."""

    _constraint = StressStrainConstraint.PLANE_STRAIN
    _kinds = (_capi.GRAD_2D_TO_3D, _capi.STRESS_2D_TO_3D, _capi.STRESS_3D_TO_2D, _capi.TANGENT_3D_TO_2D)
