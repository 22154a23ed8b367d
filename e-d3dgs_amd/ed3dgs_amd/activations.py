"""Fused activation op (csrc/activations.hip): rot = normalize, scales = exp, opacity = sigmoid, optionally with the
3D filter of scene/gaussian_model.py:594-603 -- one HIP launch per direction behind a torch.autograd.Function."""
import ctypes as C

import torch

from . import _lib


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class _Activations(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scales_log, rot_raw, opacity_logit, filter_3D):
        if not scales_log.is_cuda:
            raise RuntimeError("activations: tensors must be on the GPU (no CPU fallback)")
        L = _lib.lib()
        ctx.set_materialize_grads(False)   # None (= zero for the C ABI) for outputs the loss does not reach
        s = scales_log.detach().contiguous().float()
        r = rot_raw.detach().contiguous().float()
        o = opacity_logit.detach().contiguous().float()
        f = None if filter_3D is None else filter_3D.detach().contiguous().float()
        P = s.shape[0]
        so, ro, oo = torch.empty_like(s), torch.empty_like(r), torch.empty_like(o)
        rc = L.ed3dgs_activations_forward(C.c_int(P), _p(s), _p(r), _p(o), _p(f), _p(so), _p(ro), _p(oo),
                                          _lib.raw_stream(s.device))
        if rc < 0:
            raise RuntimeError(_lib.last_error())
        ctx.save_for_backward(s, r, o, *( [f] if f is not None else [] ))
        ctx.has_f = f is not None
        return so, ro, oo

    @staticmethod
    def backward(ctx, gs, gr, go):
        L = _lib.lib()
        saved = ctx.saved_tensors
        s, r, o = saved[:3]
        f = saved[3] if ctx.has_f else None
        c = lambda g: None if g is None else g.contiguous().float()
        gs, gr, go = c(gs), c(gr), c(go)
        gsl, grr, gol = torch.empty_like(s), torch.empty_like(r), torch.empty_like(o)
        rc = L.ed3dgs_activations_backward(C.c_int(s.shape[0]), _p(s), _p(r), _p(o), _p(f), _p(gs), _p(gr), _p(go),
                                           _p(gsl), _p(grr), _p(gol), _lib.raw_stream(s.device))
        if rc < 0:
            raise RuntimeError(_lib.last_error())
        return gsl, grr, gol, None


def fused_activations(scales_log, rot_raw, opacity_logit, filter_3D=None):
    """Returns (scales, rotations, opacity)."""
    return _Activations.apply(scales_log, rot_raw, opacity_logit, filter_3D)
