"""scene.deformation -- MI355X-native drop-in for the reference's deform_network (scene/deformation.py:15-148).

Same constructor, same parameter / state-dict names (`weight`, `offsets`, `feature_out_{c,f}.0.*`,
`{pos,scales,rotations,opacity,rgb}_deform_{c,f}.{1,3}.*`, so `deformation.pth` loads unchanged,
scene/gaussian_model.py:250-258) and the same `forward` signature / return structure (:108-141).  The arithmetic runs
in the fused HIP kernels of csrc/deform.hip through the C ABI (include/ed3dgs.h); torch only owns the parameters,
allocates outputs and links the call into autograd.  There is no CPU or eager fallback.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from ed3dgs_amd import _lib

HEADS = ("pos", "scales", "rotations", "opacity", "rgb")
# False: the backward re-forms the activations from the inputs (stateless C-ABI backward; less memory, more MFMA work)
KEEP_ACTIVATIONS = True
# True: the kept activations are released by the first backward (a second backward over a retained graph takes the stateless
# path).  False: they stay with the graph, and every backward over it reads them again.
RELEASE_KEPT_WORKSPACE = True


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _c32(t):
    return None if t is None else t.detach().contiguous().float()


class _DeformFn(torch.autograd.Function):
    """forward/backward of both stages as C-ABI calls.  Differentiable inputs: temporal table, offsets, the two packed
    parameter blocks, Gaussian embedding, and the five base tensors -- the SH one either whole (`sh` [P,n_sh,3],
    `sh_rest` None) or as the reference stores it (`sh` = _features_dc [P,1,3], `sh_rest` = _features_rest [P,n_sh-1,3])."""

    @staticmethod
    def forward(ctx, cfgd, want_sub, keep, act, table, offsets, flat_c, flat_f, emb, xyz, scales, rot, opacity, sh, sh_rest):
        """act: None, or (filter_3D or None,) -- also return render()'s activated scales / rotations / opacity
        (ed3dgs_deform_forward_activated: written by the MLP kernel's epilogue), after the other outputs."""
        L = _lib.lib()
        ctx.set_materialize_grads(False)   # unused outputs arrive as None in backward (NULL = zero for the C ABI)
        cfg = _lib.DeformCfg(**{k: v for k, v in cfgd.items() if k != "use_stage" and k != "n_rows"})
        cfg.use_stage[0], cfg.use_stage[1] = cfgd["use_stage"]
        cfg.n_rows[0], cfg.n_rows[1] = cfgd["n_rows"]
        dev = xyz.device
        if not xyz.is_cuda:
            raise RuntimeError("deform_network: tensors must be on the GPU (the MI355X path has no CPU fallback)")
        ins = [_c32(t) for t in (table, offsets, flat_c, flat_f, emb, xyz, scales, rot, opacity, sh, sh_rest)]
        table_, offsets_, fc, ff, emb_, xyz_, sc_, rot_, op_, sh_, shr_ = ins
        P = xyz_.shape[0]
        sh_shape = (P, cfgd["n_sh"], 3)
        outs = [torch.empty_like(t) for t in (xyz_, sc_, rot_, op_)] + [torch.empty(sh_shape, device=dev)]
        subs = ([torch.empty_like(t) for t in (xyz_, sc_, rot_, op_)] + [torch.empty(sh_shape, device=dev)]) if want_sub else [None] * 5
        # training: the forward keeps the hidden activations in the backward's workspace (as autograd does for the
        # reference's Linear/ReLU modules); inference: small workspace, nothing kept
        ws_bytes = L.ed3dgs_deform_workspace_bytes(C.byref(cfg), C.c_int(1 if keep else 0))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        params = (C.c_void_p * 2)(fc.data_ptr() if cfgd["use_stage"][0] else None, ff.data_ptr() if cfgd["use_stage"][1] else None)
        acts, filt = [], None
        if act is None:
            rc = L.ed3dgs_deform_forward(
                C.byref(cfg), _ptr(table_), _ptr(offsets_), params, _ptr(emb_), _ptr(xyz_), _ptr(sc_), _ptr(rot_), _ptr(op_),
                _ptr(sh_), _ptr(shr_), *[_ptr(t) for t in outs], *[_ptr(t) for t in subs], _ptr(ws), C.c_size_t(ws_bytes),
                C.c_int(1 if keep else 0), _lib.raw_stream(dev))
        else:
            filt = _c32(act[0])
            acts = [torch.empty_like(t) for t in (sc_, rot_, op_)]
            rc = L.ed3dgs_deform_forward_activated(
                C.byref(cfg), _ptr(table_), _ptr(offsets_), params, _ptr(emb_), _ptr(xyz_), _ptr(sc_), _ptr(rot_), _ptr(op_),
                _ptr(sh_), _ptr(shr_), *[_ptr(t) for t in outs], *[_ptr(t) for t in subs], _ptr(filt), *[_ptr(t) for t in acts],
                _ptr(ws), C.c_size_t(ws_bytes), C.c_int(1 if keep else 0), _lib.raw_stream(dev))
        if rc < 0:
            raise RuntimeError(_lib.last_error())
        ctx.kept_ws = ws if rc == 1 else None
        ctx.cfgd = cfgd
        ctx.want_sub = want_sub
        ctx.act = act is not None
        saved = [table_, offsets_, fc, ff, emb_]
        if ctx.act:   # the activation backward needs the raw final values (and the filter)
            saved += [outs[1], outs[2], outs[3]] + ([filt] if filt is not None else [])
        ctx.save_for_backward(*saved)
        ctx.shapes = [t.shape for t in (table, offsets, flat_c, flat_f, emb, xyz, scales, rot, opacity, sh)]
        ctx.split_sh = None if sh_rest is None else (sh.shape, sh_rest.shape)
        return tuple(outs) + (tuple(subs) if want_sub else ()) + tuple(acts)

    @staticmethod
    def backward(ctx, *gr):
        L = _lib.lib()
        cfgd = ctx.cfgd
        cfg = _lib.DeformCfg(**{k: v for k, v in cfgd.items() if k != "use_stage" and k != "n_rows"})
        cfg.use_stage[0], cfg.use_stage[1] = cfgd["use_stage"]
        cfg.n_rows[0], cfg.n_rows[1] = cfgd["n_rows"]
        table_, offsets_, fc, ff, emb_ = ctx.saved_tensors[:5]
        dev = emb_.device
        g_out = [_c32(g) for g in gr[:5]]
        g_sub = [_c32(g) for g in gr[5:10]] if ctx.want_sub else [None] * 5
        g_act = raw = filt = g_raw = None
        if ctx.act:
            n0 = 10 if ctx.want_sub else 5
            g_act = [_c32(g) for g in gr[n0:n0 + 3]]
            raw = list(ctx.saved_tensors[5:8])
            filt = ctx.saved_tensors[8] if len(ctx.saved_tensors) > 8 else None
            g_raw = [torch.empty_like(t) for t in raw]   # dL/d(raw final scales / rotations / opacity), from the C call
        gfc = torch.empty_like(fc)
        gff = torch.empty_like(ff)
        g_table = torch.empty_like(table_)
        g_off = torch.empty_like(offsets_)
        g_emb = torch.empty_like(emb_)
        # split SH storage: dL/d(_features_dc) and dL/d(_features_rest) come out of the pass that reads the upstream
        # gradients (contiguous tensors autograd can hand to the parameters as they are)
        g_dc = g_rest = None
        if ctx.split_sh is not None and (g_out[4] is not None or g_sub[4] is not None):
            g_dc = torch.empty(ctx.split_sh[0], device=dev)
            g_rest = torch.empty(ctx.split_sh[1], device=dev)
        ws_bytes = L.ed3dgs_deform_workspace_bytes(C.byref(cfg), C.c_int(1))
        kept = ctx.kept_ws is not None
        ws = ctx.kept_ws if kept else torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        if RELEASE_KEPT_WORKSPACE:
            ctx.kept_ws = None   # 6 KB per Gaussian go back to the allocator; a second backward (retain_graph) re-forms the activations
        params = (C.c_void_p * 2)(fc.data_ptr() if cfgd["use_stage"][0] else None, ff.data_ptr() if cfgd["use_stage"][1] else None)
        gparams = (C.c_void_p * 2)(gfc.data_ptr() if cfgd["use_stage"][0] else None, gff.data_ptr() if cfgd["use_stage"][1] else None)
        if not ctx.act:
            rc = L.ed3dgs_deform_backward(
                C.byref(cfg), _ptr(table_), _ptr(offsets_), params, _ptr(emb_), *[_ptr(g) for g in g_out],
                *[_ptr(g) for g in g_sub], gparams, _ptr(g_table), _ptr(g_off), _ptr(g_emb), _ptr(g_dc), _ptr(g_rest), _ptr(ws),
                C.c_size_t(ws_bytes), C.c_int(1 if kept else 0), _lib.raw_stream(dev))
        elif any(g is not None for g in g_out[1:4]):
            # the caller ALSO differentiated through the raw final scales / rotations / opacity: convert the activated-space
            # gradients with the stand-alone launch, add, and take the plain backward
            z = lambda g, like: torch.zeros_like(like) if g is None else g
            rc = L.ed3dgs_activations_backward(C.c_int(raw[0].shape[0]), _ptr(raw[0]), _ptr(raw[1]), _ptr(raw[2]), _ptr(filt),
                                               *[_ptr(g) for g in g_act], *[_ptr(g) for g in g_raw], _lib.raw_stream(dev))
            if rc >= 0:
                g_raw = [a + z(b, a) for a, b in zip(g_raw, g_out[1:4])]
                rc = L.ed3dgs_deform_backward(
                    C.byref(cfg), _ptr(table_), _ptr(offsets_), params, _ptr(emb_), _ptr(g_out[0]), *[_ptr(g) for g in g_raw],
                    _ptr(g_out[4]), *[_ptr(g) for g in g_sub], gparams, _ptr(g_table), _ptr(g_off), _ptr(g_emb), _ptr(g_dc),
                    _ptr(g_rest), _ptr(ws), C.c_size_t(ws_bytes), C.c_int(1 if kept else 0), _lib.raw_stream(dev))
        else:
            rc = L.ed3dgs_deform_backward_activated(
                C.byref(cfg), _ptr(table_), _ptr(offsets_), params, _ptr(emb_), _ptr(g_out[0]), _ptr(g_out[4]),
                *[_ptr(g) for g in g_sub], *[_ptr(t) for t in raw], _ptr(filt), *[_ptr(g) for g in g_act],
                *[_ptr(g) for g in g_raw], gparams, _ptr(g_table), _ptr(g_off), _ptr(g_emb), _ptr(g_dc), _ptr(g_rest), _ptr(ws),
                C.c_size_t(ws_bytes), C.c_int(1 if kept else 0), _lib.raw_stream(dev))
        if rc < 0:
            raise RuntimeError(_lib.last_error())
        if not cfgd["use_stage"][0]:
            gfc.zero_()
        if not cfgd["use_stage"][1]:
            gff.zero_()
        base = []
        for i in range(4 if ctx.split_sh is not None else 5):  # identity paths: out = base + ..., sub = base + ...
            a, b = gr[i], (gr[5 + i] if ctx.want_sub else None)
            if ctx.act and 1 <= i <= 3:
                a = g_raw[i - 1]   # dL/d(raw final value), the activation backward included (plus gr[i] where that was given)
            g = a if b is None else (b if a is None else a + b)
            base.append(None if g is None else g.reshape(ctx.shapes[5 + i]))
        if ctx.split_sh is not None:
            base += [g_dc, g_rest]
        else:
            base.append(None)
        sh = ctx.shapes
        return (None, None, None, None, g_table.reshape(sh[0]), g_off.reshape(sh[1]), gfc.reshape(sh[2]), gff.reshape(sh[3]),
                g_emb.reshape(sh[4]), *base)


class _PackedParams(torch.autograd.Function):
    """Identity on the packed parameter block; its backward hands each parameter its slice of the packed gradient."""

    @staticmethod
    def forward(ctx, flat, *parts):
        ctx.shapes = [p.shape for p in parts]
        return flat.view(-1)

    @staticmethod
    def backward(ctx, g):
        out, off = [], 0
        for sh in ctx.shapes:
            n = 1
            for d in sh:
                n *= d
            out.append(g[off:off + n].view(sh))
            off += n
        return (None, *out)


class deform_network(nn.Module):
    def __init__(self, D=8, W=256, min_embeddings=30, max_embeddings=150, num_frames=300, num_cam=None, args=None):
        super().__init__()
        self.D = D
        self.W = W
        self.args = args
        self.min_embeddings = min_embeddings
        self.max_embeddings = max_embeddings
        self.num_frames = num_frames
        self.temporal_embedding_dim = args.temporal_embedding_dim
        self.gaussian_embedding_dim = args.gaussian_embedding_dim
        self.c2f_temporal_iter = args.c2f_temporal_iter
        (self.feature_out_c, self.pos_deform_c, self.scales_deform_c, self.rotations_deform_c, self.opacity_deform_c,
         self.rgb_deform_c) = self.create_net()
        (self.feature_out_f, self.pos_deform_f, self.scales_deform_f, self.rotations_deform_f, self.opacity_deform_f,
         self.rgb_deform_f) = self.create_net()
        td = self.temporal_embedding_dim
        if args.zero_temporal:
            table = torch.zeros(max_embeddings, td)
        else:
            table = torch.normal(0., 0.01 / np.sqrt(td), size=(max_embeddings, td))
        self.weight = nn.Parameter(table)
        self.offsets = nn.Parameter(torch.zeros((30, 1)))  # per-camera time offsets (reference: hard-coded 30)

    def create_net(self):
        """Module layout of the reference (:38-51), kept so the state-dict keys are identical."""
        W = self.W
        trunk = [nn.Linear(self.temporal_embedding_dim + self.gaussian_embedding_dim, W)]
        for _ in range(self.D - 1):
            trunk += [nn.ReLU(), nn.Linear(W, W)]
        head = lambda n: nn.Sequential(nn.ReLU(), nn.Linear(W, W), nn.ReLU(), nn.Linear(W, n))
        return nn.Sequential(*trunk), head(3), head(3), head(4), head(1), head(3 * 16)

    def int_lininterp(self, t, init_val, final_val, until):
        return int(init_val + (final_val - init_val) * min(max(t, 0), until) / until)

    def get_mlp_parameters(self):
        return [p for n, p in self.named_parameters() if n != "offsets"]

    # ---- helpers of the fused path ----
    def _stage_parts(self, s):
        """The stage's 22 parameters in packed order (+ 2 per extra trunk layer when defor_depth > 1, appended at the end:
        include/ed3dgs.h).  The Linear modules are looked up once (nn.Sequential indexing is slow) and re-checked by identity
        against the module tree on every call; their parameters are read from the modules on every call, so a replaced module
        or Parameter is seen."""
        cache = self.__dict__.setdefault("_stage_linears", {})
        ent = cache.get(s)
        mods = self._modules
        if ent is not None:
            for name, idx, seq, lin in ent:
                if mods[name] is not seq or seq._modules[idx] is not lin:
                    ent = None
                    break
        if ent is None:
            ent = [(f"feature_out_{s}", "0", mods[f"feature_out_{s}"], mods[f"feature_out_{s}"][0])]
            for h in HEADS:
                seq = mods[f"{h}_deform_{s}"]
                ent += [(f"{h}_deform_{s}", "1", seq, seq[1]), (f"{h}_deform_{s}", "3", seq, seq[3])]
            trunk = mods[f"feature_out_{s}"]
            for i in range(max(self.D - 1, 0)):   # feature_out.{2, 4, ..} (scene/deformation.py:38-44)
                ent.append((f"feature_out_{s}", str(2 * (i + 1)), trunk, trunk[2 * (i + 1)]))
            cache[s] = ent
        parts = []
        for _, _, _, m in ent:
            pr = m._parameters
            parts.append(pr["weight"]); parts.append(pr["bias"])
        return parts

    def _flat_stage(self, s):
        """The stage's parameters packed as include/ed3dgs.h lays them out -- WITHOUT a copy per call: the first call (and any
        call after something re-bound a parameter's storage: .to(), a replaced Parameter) packs them once into one flat buffer
        and points every parameter's .data at its slice of it, so in-place updates (optimizer steps, load_state_dict) keep the
        packed block current.  Names, shapes and state-dict contents are untouched.  With autograd on, a view node routes the
        packed gradient back to the parameters as slices (no copies either way).  Visible side effects:
        the 22 parameters of a stage alias ONE storage (torch.save of a single parameter serialises the whole block; state_dict()
        round-trips unchanged), their .grad tensors are views of one packed gradient, and non-fp32 parameters are refused."""
        parts = self._stage_parts(s)
        store = self.__dict__.setdefault("_flat_store", {})
        flat = store.get(s)
        ok = flat is not None
        if ok:
            off, base = 0, flat.data_ptr()
            for p in parts:
                if p.data_ptr() != base + 4 * off or p.dtype != torch.float32 or not p.is_contiguous():
                    ok = False
                    break
                off += p.numel()
            ok = ok and off == flat.numel()
        if not ok:
            bad = [tuple(p.shape) for p in parts if p.dtype != torch.float32]
            if bad:   # re-binding .data to an fp32 slice would silently change the parameter's dtype
                raise TypeError("deform_network: the fused MI355X path computes in fp32 and packs the Linear parameters in place; "
                                f"found non-fp32 parameters of shapes {bad[:3]} -- keep the module in float32 (.float())")
            with torch.no_grad():
                flat = torch.cat([p.detach().reshape(-1) for p in parts])
                off = 0
                for p in parts:
                    p.data = flat[off:off + p.numel()].view(p.shape)
                    off += p.numel()
            store[s] = flat
        if torch.is_grad_enabled() and any(p.requires_grad for p in parts):
            return _PackedParams.apply(flat, *parts)
        return flat

    def _row_counts(self, it, num_down_emb_c, num_down_emb_f):
        """query_time (:72-80)"""
        a = self.args

        def c2f(nd):
            if a.no_c2f_temporal_embedding:
                return self.max_embeddings
            return self.int_lininterp(it, nd, self.max_embeddings, self.c2f_temporal_iter)
        n_c = num_down_emb_c if a.use_coarse_temporal_embedding else c2f(num_down_emb_c)
        return n_c, c2f(num_down_emb_f)

    def forward(self, point, scales=None, rotations=None, opacity=None, time_emb=None, cam_no=None, pc=None,
                embeddings=None, sh_coefs=None, iter=None, num_down_emb_c=30, num_down_emb_f=30, want_extras=True,
                sh_coefs_rest=None, activated=None):
        """The reference's signature (:108-141) plus two keywords.  `sh_coefs_rest`: pass the model's split SH storage
        as it is -- sh_coefs = _features_dc [P,1,3], sh_coefs_rest = _features_rest [P,n_sh-1,3] -- instead of their
        concatenation (get_features, scene/gaussian_model.py:128-131): the kernels read both and the backward writes both
        gradients, so neither the concatenation nor autograd's two slice copies run.  The last element of the returned
        `orig` tuple is then the pair (sh_coefs, sh_coefs_rest).
        `activated`: None (the reference's behaviour: raw values), or a 1-tuple (filter_3D or None,): the returned scales /
        rotations / opacity are then the ACTIVATED ones render() feeds the rasterizer (gaussian_renderer/__init__.py:77-83:
        exp / F.normalize / sigmoid, or the 3D-filter variant of scene/gaussian_model.py:594-603), written by the MLP kernel
        itself; the raw finals still exist inside the autograd node for the backward."""
        a = self.args
        pts, scales, rotations, opacity = point[:, :3], scales[:, :3], rotations[:, :4], opacity[:, :1]
        orig = (pts, scales, rotations, opacity, sh_coefs if sh_coefs_rest is None else (sh_coefs, sh_coefs_rest))
        emb = embeddings if pc is None else pc.get_embedding
        # the reference reads only time_emb[0, 0] (:58); a Python float avoids the device read-back
        time = float(time_emb) if not torch.is_tensor(time_emb) else float(time_emb.reshape(-1)[0])
        use_anneal = a.use_anneal
        coef = 1.0 if not use_anneal else float(np.clip(iter / 1000, 0, 1))
        coef_x = 1.0 if not use_anneal else float(np.clip((iter - a.deform_from_iter) / 1000, 0, 1))
        n_c, n_f = self._row_counts(iter, num_down_emb_c, num_down_emb_f)
        cfgd = dict(P=pts.shape[0], W=self.W, D=self.D, E=self.gaussian_embedding_dim, TD=self.temporal_embedding_dim,
                    n_sh=sh_coefs.shape[1] + (0 if sh_coefs_rest is None else sh_coefs_rest.shape[1]), max_embeddings=self.max_embeddings, num_offsets=self.offsets.shape[0],
                    use_stage=(int(not a.no_coarse_deform), int(not a.no_fine_deform)), n_rows=(int(n_c), int(n_f)),
                    no_ds=int(a.no_ds), no_dr=int(a.no_dr), no_do=int(a.no_do), no_dc=int(a.no_dc), coef=coef,
                    coef_c=coef_x, coef_o=coef_x, coef_s=coef_x, time=time, cam_no=-1 if cam_no is None else int(cam_no))
        args = (self.weight, self.offsets, self._flat_stage("c"), self._flat_stage("f"), emb, pts, scales, rotations,
                opacity, sh_coefs, sh_coefs_rest)
        # decided here: inside autograd.Function.forward grad mode is off
        keep = KEEP_ACTIVATIONS and torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in args)
        res = _DeformFn.apply(cfgd, bool(want_extras), bool(keep), None if activated is None else tuple(activated), *args)
        final = res[:5]
        sub = res[5:10] if want_extras else orig
        if activated is not None:
            acts = res[-3:]
            return final[0], acts[0], acts[1], acts[2], final[4], (tuple(sub), orig)
        return final[0], final[1], final[2], final[3], final[4], (tuple(sub), orig)


    supports_activated = True   # render() may ask for the activated outputs (a reference deform_network has no such attribute)


def initialize_weights(m):
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight, gain=1)
