"""pointnet2._mlp_ext -- ctypes binding of the fused BatchNorm(+ReLU)(+max-pool) kernels
(include/mlp_hip.h) used by pointnet2.pytorch_utils.SharedMLP on the GPU.

No reference counterpart: the reference runs nn.BatchNorm2d / nn.ReLU / F.max_pool2d here
(pointnet2/pytorch_utils.py:14-124, pointnet2/pointnet2_modules.py:256-262).
"""
import contextlib
import os

import torch

from pointnet2._ext import _L, _lib, _stream


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
        raise RuntimeError("%s must be a contiguous float32 GPU tensor" % name)


def _ws(y, b, c, r):
    n = int(_lib.mlp_bn_workspace_floats(b, c, r))
    return torch.empty(max(n, 1), dtype=torch.float32, device=y.device)


def new_tickets(channels, device):
    """The counters of the one-launch reductions (include/mlp_hip.h, `tickets`): `channels` ints,
    zero between launches.  Owned by whoever owns the layers -- the library keeps none."""
    return torch.zeros(max(int(channels), 1), dtype=torch.int32, device=device)


def tickets_of(module, channels, device):
    """The ticket array of `module`: a plain attribute `_pn2_tickets`, created on first use on the
    device of the module's input (again after a move), copied by deepcopy -- not a registered
    buffer, so named_buffers() / state_dict() stay the reference's.  One array per module: two
    modules -- a student and its teacher -- may run side by side on any two streams."""
    t = module.__dict__.get("_pn2_tickets")
    if t is None or t.device != device or t.numel() < channels:
        t = new_tickets(channels, device)
        module.__dict__["_pn2_tickets"] = t
    return t


def _tk(tickets, c, ref):
    if tickets is None:  # a caller without a module (tests, tools): a fresh zeroed array per call
        return new_tickets(c, ref.device)
    if tickets.dtype != torch.int32 or tickets.device != ref.device or tickets.numel() < c or \
            not tickets.is_contiguous():
        raise RuntimeError("tickets must be a contiguous int32 tensor of >= %d zeros on %s" % (c, ref.device))
    return tickets


def bn_coefficients(y, gamma, beta, running_mean, running_var, momentum, eps, training, tickets=None):
    """y (B,C,...) -> per-channel (mean, invstd, scale, shift); training=True uses (and folds
    into running_*) the batch statistics, otherwise the running statistics."""
    _f32c(y, "y")
    b, c = y.shape[0], y.shape[1]
    r = y.numel() // (b * c)
    out = torch.empty((4, c), dtype=torch.float32, device=y.device)
    mean, invstd, scale, shift = out[0], out[1], out[2], out[3]
    with torch.cuda.device(y.device):
        if training:
            ws = _ws(y, b, c, r)
            rm = running_mean.data_ptr() if running_mean is not None else None
            rv = running_var.data_ptr() if running_var is not None else None
            _L.check(_lib.mlp_bn_train_stats(b, c, r, y.data_ptr(), gamma.data_ptr(),
                                             beta.data_ptr(), float(eps), float(momentum), rm, rv,
                                             mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(),
                                             shift.data_ptr(), ws.data_ptr(),
                                             _tk(tickets, c, y).data_ptr(), _stream(y)),
                     "mlp_bn_train_stats")
        else:
            _L.check(_lib.mlp_bn_eval_coeff(c, gamma.data_ptr(), beta.data_ptr(), float(eps),
                                            running_mean.data_ptr(), running_var.data_ptr(),
                                            mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(),
                                            shift.data_ptr(), _stream(y)), "mlp_bn_eval_coeff")
    return mean, invstd, scale, shift


def bn_relu_apply(y, scale, shift):
    b, c = y.shape[0], y.shape[1]
    r = y.numel() // (b * c)
    z = torch.empty_like(y)
    with torch.cuda.device(y.device):
        _L.check(_lib.mlp_bn_relu_apply(b, c, r, y.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                        z.data_ptr(), _stream(y)), "mlp_bn_relu_apply")
    return z


def bn_relu_pool(y, scale, shift):
    """y (B,C,m,ns) -> pooled (B,C,m), argmax int32 (B,C,m), ymax (B,C,m)."""
    b, c, m, ns = y.shape
    pooled = torch.empty((b, c, m), dtype=torch.float32, device=y.device)
    ymax = torch.empty_like(pooled)
    argmax = torch.empty((b, c, m), dtype=torch.int32, device=y.device)
    with torch.cuda.device(y.device):
        _L.check(_lib.mlp_bn_relu_pool(b, c, m, ns, y.data_ptr(), scale.data_ptr(),
                                       shift.data_ptr(), pooled.data_ptr(), argmax.data_ptr(),
                                       ymax.data_ptr(), _stream(y)), "mlp_bn_relu_pool")
    return pooled, argmax, ymax


def bn_relu_backward(y, dz, gamma, scale, shift, mean, invstd, training, tickets=None):
    _f32c(dz, "dz")
    b, c = y.shape[0], y.shape[1]
    r = y.numel() // (b * c)
    dy = torch.empty_like(y)
    small = torch.empty((5, c), dtype=torch.float32, device=y.device)  # dgamma, dbeta, coef[3]
    with torch.cuda.device(y.device):
        ws = _ws(y, b, c, r)
        _L.check(_lib.mlp_bn_relu_backward(b, c, r, 1 if training else 0, y.data_ptr(),
                                           dz.data_ptr(), gamma.data_ptr(), scale.data_ptr(),
                                           shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                           dy.data_ptr(), small[0].data_ptr(), small[1].data_ptr(),
                                           small[2:].data_ptr(), ws.data_ptr(),
                                           _tk(tickets, c, y).data_ptr(), _stream(y)),
                 "mlp_bn_relu_backward")
    return dy, small[0], small[1]


def bn_relu_pool_backward(y, dpooled, argmax, ymax, gamma, scale, shift, mean, invstd, training,
                          tickets=None):
    _f32c(dpooled, "dpooled")
    b, c, m, ns = y.shape
    dy = torch.empty_like(y)
    small = torch.empty((5, c), dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
        ws = _ws(y, b, c, m)
        _L.check(_lib.mlp_bn_relu_pool_backward(b, c, m, ns, 1 if training else 0, y.data_ptr(),
                                                dpooled.data_ptr(), argmax.data_ptr(),
                                                ymax.data_ptr(), gamma.data_ptr(),
                                                scale.data_ptr(), shift.data_ptr(),
                                                mean.data_ptr(), invstd.data_ptr(), dy.data_ptr(),
                                                small[0].data_ptr(), small[1].data_ptr(),
                                                small[2:].data_ptr(), ws.data_ptr(),
                                                _tk(tickets, c, y).data_ptr(), _stream(y)),
                 "mlp_bn_relu_pool_backward")
    return dy, small[0], small[1]


def bn_relu_pool_backward_stats(y, dpooled, argmax, ymax, gamma, scale, shift, mean, invstd,
                                training, ns=None, tickets=None):
    """-> dgamma, dbeta, coef (C,3) of the pooled last layer; dy itself is formed inside
    gemm_dgrad / gemm_wgrad (pooled=...).  y is only asked for its shape (B,C,m,ns): with ns given
    it may be None (a layer whose raw output was never stored)."""
    _f32c(dpooled, "dpooled")
    if ns is not None:
        b, c, m = dpooled.shape
        y = dpooled  # (never read: the statistics come from the pooled tensors)
    else:
        b, c, m, ns = y.shape
    small = torch.empty((5, c), dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
        ws = _ws(y, b, c, m)
        _L.check(_lib.mlp_bn_relu_pool_backward(b, c, m, ns, 1 if training else 0, y.data_ptr(),
                                                dpooled.data_ptr(), argmax.data_ptr(),
                                                ymax.data_ptr(), gamma.data_ptr(),
                                                scale.data_ptr(), shift.data_ptr(),
                                                mean.data_ptr(), invstd.data_ptr(), None,
                                                small[0].data_ptr(), small[1].data_ptr(),
                                                small[2:].data_ptr(), ws.data_ptr(),
                                                _tk(tickets, c, y).data_ptr(), _stream(y)),
                 "mlp_bn_relu_pool_backward(stats)")
    return small[0], small[1], small[2:]


# ---- the 1x1 convolution on the matrix cores (include/mlp_hip.h: mlp_gemm_*) -------------------
def _ptr(t):
    return t.data_ptr() if t is not None else None


class WeightImages(object):
    """bf16 images of a set of weights for the SMALL layers' kernels (include/mlp_hip.h
    mlp_weight_images_build): the exact three-term split of every weight, written once per
    optimizer step by refresh() -- ONE launch -- in the two orders forward and data gradient read,
    instead of being redone by every workgroup of every small GEMM.  Inside `with
    weight_images(obj):` gemm_forward / gemm_backward_small find a weight's images by its address and
    use them where the layer runs on the small kernel; results are the same bit for bit.  The caller
    owns the freshness: refresh() after every update of the weights (the train step's optimizer
    writes them through raw pointers -- there is no version counter to key a cache on)."""

    def __init__(self, weights):
        import ctypes
        ws = [w for w in weights if w.is_cuda and w.dtype == torch.float32 and w.dim() >= 2
              and w.is_contiguous()]
        self.entries = {}
        self._keep = ws
        if not ws:
            self.n = 0
            return
        dev = ws[0].device
        dims = [(w.shape[0], w.numel() // w.shape[0]) for w in ws]
        elems = [int(_lib.mlp_weight_image_elems(m, k)) for m, k in dims]
        offs, total = [], 0
        for e in elems:
            offs.append(total)
            total += 2 * ((e + 7) // 8 * 8)
        self.buf = torch.empty(total, dtype=torch.int16, device=dev)
        base = self.buf.data_ptr()
        n = self.n = len(ws)
        self._w = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws])
        self._m = (ctypes.c_int * n)(*[m for m, _ in dims])
        self._k = (ctypes.c_int * n)(*[k for _, k in dims])
        img = [base + 2 * o for o in offs]
        img_t = [base + 2 * (o + (e + 7) // 8 * 8) for o, e in zip(offs, elems)]
        self._img = (ctypes.c_void_p * n)(*img)
        self._img_t = (ctypes.c_void_p * n)(*img_t)
        for w, (m, k), a, t in zip(ws, dims, img, img_t):
            self.entries[w.data_ptr()] = (m, k, a, t)
        self.device = dev

    def refresh(self):
        if self.n:
            with torch.cuda.device(self.device):
                _L.check(_lib.mlp_weight_images_build(self.n, self._w, self._m, self._k, self._img,
                                                      self._img_t, _stream(self.buf)),
                         "mlp_weight_images_build")


_ACTIVE_IMAGES = []


@contextlib.contextmanager
def weight_images(*sets):
    """gemm_forward / gemm_backward_small take the weights of `sets` (WeightImages) from their images."""
    global _ACTIVE_IMAGES
    previous = _ACTIVE_IMAGES
    _ACTIVE_IMAGES = previous + [s for s in sets if s is not None and s.n]
    try:
        yield
    finally:
        _ACTIVE_IMAGES = previous


def _image_of(w, b, r):
    """(forward image, transposed image) addresses of w, or None"""
    if not _ACTIVE_IMAGES:
        return None
    for s in _ACTIVE_IMAGES:
        e = s.entries.get(w.data_ptr())
        if e is not None and (e[0], e[1]) == (w.shape[0], w.numel() // w.shape[0]) and \
                _lib.mlp_gemm_image_supported(int(b), int(r)):
            return e[2], e[3]
    return None


def gemm_forward(w, x, coeff=None):
    """y (B,M,R) = w (M,K) @ x (B,K,R); with coeff=(scale, shift) the operand is
    relu(x*scale[k] + shift[k]) formed on the fly (x is then the previous layer's conv output)."""
    _f32c(x, "x"); _f32c(w, "w")
    b, k = x.shape[0], x.shape[1]
    r = x.numel() // (b * k)
    m = w.shape[0]
    y = torch.empty((b, m) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    scale, shift = coeff if coeff is not None else (None, None)
    image = _image_of(w, b, r)
    if image is not None:
        with torch.cuda.device(x.device):
            _L.check(_lib.mlp_gemm_forward_img(b, m, k, r, w.data_ptr(), image[0], x.data_ptr(),
                                               0 if coeff is None else 1, _ptr(scale), _ptr(shift),
                                               y.data_ptr(), _stream(x)), "mlp_gemm_forward_img")
        return y
    with torch.cuda.device(x.device):
        _L.check(_lib.mlp_gemm_forward(b, m, k, r, w.data_ptr(), x.data_ptr(),
                                       0 if coeff is None else 1, _ptr(scale), _ptr(shift),
                                       y.data_ptr(), _stream(x)), "mlp_gemm_forward")
    return y


def gemm_forward_bn(w, x, coeff, gamma, beta, running_mean, running_var, momentum, eps, pool=False,
                    tickets=None, store=True):
    """Training-mode layer: y = gemm_forward(w, x, coeff) and the BatchNorm coefficients of y
    (mean, invstd, scale, shift), with the batch statistics reduced in the GEMM epilogue when the
    shape allows (no second pass over y), else by bn_coefficients.
    pool=True (x is (B,K,m,ns), the layer is followed by the max over nsample): returns a sixth
    value, the extrema planes for pool_from_extrema -- or None when the shape has no such epilogue
    and the caller pools with bn_relu_pool.
    store=False (pool=True and forward_pool_supported only): y is None -- the raw output is not
    stored at all (its backward then runs from the Gram matrix of its input, pool_gram_backward)."""
    import ctypes
    _f32c(x, "x"); _f32c(w, "w")
    b, k = x.shape[0], x.shape[1]
    r = x.numel() // (b * k)
    m = w.shape[0]
    cols = ctypes.c_int(0)
    parts = int(_lib.mlp_gemm_forward_stats_parts(b, m, k, r, ctypes.byref(cols)))
    if parts <= 0:
        y = gemm_forward(w, x, coeff)
        return (y,) + tuple(bn_coefficients(y, gamma, beta, running_mean, running_var, momentum,
                                            eps, True, tickets)) + ((None,) if pool else ())
    ns = x.shape[3] if (pool and x.dim() == 4) else 0
    pooled = bool(ns) and coeff is not None and w.data_ptr() % 16 == 0 and \
        bool(_lib.mlp_gemm_forward_stats_pool_supported(b, m, k, r, ns))
    ext = torch.empty((2, b, m, r // ns), dtype=torch.float32, device=x.device) if pooled else None
    if not store and not pooled:
        raise RuntimeError("gemm_forward_bn(store=False): the pooled epilogue does not cover this layer")
    y = torch.empty((b, m) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device) if store else None
    pairs = torch.empty((parts, m, 2), dtype=torch.float32, device=x.device)
    out = torch.empty((4, m), dtype=torch.float32, device=x.device)
    scratch = torch.empty(int(_lib.mlp_bn_finalize_pairs_scratch_bytes(m)), dtype=torch.uint8,
                          device=x.device)
    scale, shift = coeff if coeff is not None else (None, None)
    with torch.cuda.device(x.device):
        if pooled:
            _L.check(_lib.mlp_gemm_forward_stats_pool(b, m, k, r, w.data_ptr(), x.data_ptr(),
                                                      scale.data_ptr(), shift.data_ptr(),
                                                      _ptr(y), pairs.data_ptr(), ns,
                                                      gamma.data_ptr(), ext.data_ptr(), _stream(x)),
                     "mlp_gemm_forward_stats_pool")
        else:
            _L.check(_lib.mlp_gemm_forward_stats(b, m, k, r, w.data_ptr(), x.data_ptr(),
                                                 0 if coeff is None else 1, _ptr(scale),
                                                 _ptr(shift), y.data_ptr(), pairs.data_ptr(),
                                                 _stream(x)), "mlp_gemm_forward_stats")
        rm = running_mean.data_ptr() if running_mean is not None else None
        rv = running_var.data_ptr() if running_var is not None else None
        _L.check(_lib.mlp_bn_finalize_pairs(m, parts, cols.value, pairs.data_ptr(),
                                            gamma.data_ptr(), beta.data_ptr(), float(eps),
                                            float(momentum), rm, rv, out[0].data_ptr(),
                                            out[1].data_ptr(), out[2].data_ptr(),
                                            out[3].data_ptr(), scratch.data_ptr(), _stream(x)),
                 "mlp_bn_finalize_pairs")
    if pool:
        return y, out[0], out[1], out[2], out[3], ext
    return y, out[0], out[1], out[2], out[3]


def forward_pool_supported(w, x, coeff):
    """Does gemm_forward_bn(w, x, coeff, ..., pool=True) leave the pooled extrema behind (x (B,K,m,ns))?"""
    if x.dim() != 4 or coeff is None or w.data_ptr() % 16 != 0:
        return False
    b, k, _, ns = x.shape
    r = x.numel() // (b * k)
    return bool(_lib.mlp_gemm_forward_stats_pool_supported(b, w.shape[0], k, r, ns))


def pool_from_extrema(ext, scale, shift):
    """(pooled, argmax, ymax) as bn_relu_pool(y, scale, shift) would return them, from the extrema
    planes (2,B,C,m) that gemm_forward_bn(pool=True) left behind: no pass over y."""
    _, b, c, m = ext.shape
    pooled = torch.empty((b, c, m), dtype=torch.float32, device=ext.device)
    ymax = torch.empty_like(pooled)
    argmax = torch.empty((b, c, m), dtype=torch.int32, device=ext.device)
    with torch.cuda.device(ext.device):
        _L.check(_lib.mlp_bn_pool_from_extrema(b, c, m, ext.data_ptr(), scale.data_ptr(),
                                               shift.data_ptr(), pooled.data_ptr(),
                                               argmax.data_ptr(), ymax.data_ptr(), _stream(ext)),
                 "mlp_bn_pool_from_extrema")
    return pooled, argmax, ymax


def bn_relu_backward_stats(y, dz, gamma, scale, shift, mean, invstd, training, tickets=None):
    """-> dgamma, dbeta, coef (C,3): everything the on-the-fly dy needs."""
    _f32c(dz, "dz")
    b, c = y.shape[0], y.shape[1]
    r = y.numel() // (b * c)
    small = torch.empty((5, c), dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
        ws = _ws(y, b, c, r)
        _L.check(_lib.mlp_bn_relu_backward_stats(b, c, r, 1 if training else 0, y.data_ptr(),
                                                 dz.data_ptr(), gamma.data_ptr(), scale.data_ptr(),
                                                 shift.data_ptr(), mean.data_ptr(),
                                                 invstd.data_ptr(), small[0].data_ptr(),
                                                 small[1].data_ptr(), small[2:].data_ptr(),
                                                 ws.data_ptr(), _tk(tickets, c, y).data_ptr(),
                                                 _stream(y)),
                 "mlp_bn_relu_backward_stats")
    return small[0], small[1], small[2:]


def gemm_dgrad(w, dy=None, fly=None, pooled=None):
    """dx (B,K,R) = w^T @ dy.  Either dy is a tensor, or fly = (y, dz, scale, shift, mean,
    invstd, coef), or pooled = (y (B,M,m,ns), dpooled, argmax, scale, shift, mean, invstd, coef)
    and dy is formed on the fly."""
    m, k = w.shape
    _f32c(w, "w")  # read as stored: the kernels take A transposed (no w.t().contiguous() copy)
    src = dy if dy is not None else (fly[0] if fly is not None else pooled[0])
    b = src.shape[0]
    r = src.numel() // (b * m)
    dx = torch.empty((b, k) + tuple(src.shape[2:]), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        if dy is not None:
            _f32c(dy, "dy")
            rc = _lib.mlp_gemm_dgrad_nt(b, m, k, r, w.data_ptr(), 0, dy.data_ptr(), None, None, None,
                                     None, None, None, None, dx.data_ptr(), _stream(src))
        elif pooled is not None:
            y, dpooled, argmax, scale, shift, mean, invstd, coef = pooled
            rc = _lib.mlp_gemm_dgrad_pooled_nt(b, m, k, y.shape[2], y.shape[3], w.data_ptr(),
                                            y.data_ptr(), dpooled.data_ptr(), argmax.data_ptr(),
                                            scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                            invstd.data_ptr(), coef.data_ptr(), dx.data_ptr(),
                                            _stream(src))
        else:
            y, dz, scale, shift, mean, invstd, coef = fly
            rc = _lib.mlp_gemm_dgrad_nt(b, m, k, r, w.data_ptr(), 2, None, y.data_ptr(),
                                     dz.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                     mean.data_ptr(), invstd.data_ptr(), coef.data_ptr(),
                                     dx.data_ptr(), _stream(src))
        _L.check(rc, "mlp_gemm_dgrad")
    return dx


_queued_workspaces = None  # not None: weight-gradient reductions are queued (see below)


def _keep_until_flush(*tensors):
    """While weight-gradient reductions are queued, both the partial-sum workspace AND the dw the
    queued reduction will write must stay allocated until the flush: autograd may drop a dw nobody
    accumulates (frozen weight) and the caching allocator would hand its block to a live tensor,
    which the flush would then overwrite."""
    if _queued_workspaces is not None:
        _queued_workspaces.extend(t for t in tensors if t is not None)


@contextlib.contextmanager
def deferred_weight_reductions(enabled=True):
    """Inside this context the reductions that finish gemm_wgrad / gemm_backward_fused are queued
    and run as ONE launch at exit (include/mlp_hip.h: mlp_defer_weight_reductions).  The returned
    dw tensors are undefined until then -- for a backward pass whose weight gradients nobody reads
    before the context ends (the train step: loss.backward(), then the gradient packing).
    Contract: inside the context every fused weight is used ONCE per backward pass and carries no
    gradient hook (autograd would sum / hand out undefined data); a weight that is frozen or whose
    gradient autograd drops is fine -- its dw stays allocated until the flush."""
    global _queued_workspaces
    if not enabled or _queued_workspaces is not None:
        yield
        return
    _queued_workspaces = []
    _L.check(_lib.mlp_defer_weight_reductions(1), "mlp_defer_weight_reductions")
    try:
        yield
    finally:
        held, device = _queued_workspaces, None
        if held:
            device = held[0].device
        try:
            if device is not None:
                with torch.cuda.device(device):
                    rc = _lib.mlp_defer_weight_reductions(0)
            else:
                rc = _lib.mlp_defer_weight_reductions(0)
        finally:
            _queued_workspaces = None
        _L.check(rc, "mlp_defer_weight_reductions")


def gemm_wgrad(m, k, x, xcoeff=None, dy=None, fly=None, pooled=None):
    """dw (M,K) = sum_b dy[b] @ x[b]^T; x direct or relu(bn(.)) via xcoeff=(scale, shift);
    dy direct or on the fly (fly as in gemm_dgrad)."""
    b = x.shape[0]
    r = x.numel() // (b * k)
    dw = torch.empty((m, k), dtype=torch.float32, device=x.device)
    xs, xh = xcoeff if xcoeff is not None else (None, None)
    with torch.cuda.device(x.device):
        ws = torch.empty(max(int(_lib.mlp_gemm_wgrad_workspace_floats(b, m, k, r)), 1),
                         dtype=torch.float32, device=x.device)
        _keep_until_flush(ws, dw)
        if dy is not None:
            rc = _lib.mlp_gemm_wgrad(b, m, k, r, 0, dy.data_ptr(), None, None, None, None, None,
                                     None, None, 0 if xcoeff is None else 1, x.data_ptr(),
                                     _ptr(xs), _ptr(xh), dw.data_ptr(), ws.data_ptr(), _stream(x))
        elif pooled is not None:
            y, dpooled, argmax, scale, shift, mean, invstd, coef = pooled
            rc = _lib.mlp_gemm_wgrad_pooled(b, m, k, y.shape[2], y.shape[3], y.data_ptr(),
                                            dpooled.data_ptr(), argmax.data_ptr(),
                                            scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                            invstd.data_ptr(), coef.data_ptr(),
                                            0 if xcoeff is None else 1, x.data_ptr(), _ptr(xs),
                                            _ptr(xh), dw.data_ptr(), ws.data_ptr(), _stream(x))
        else:
            y, dz, scale, shift, mean, invstd, coef = fly
            rc = _lib.mlp_gemm_wgrad(b, m, k, r, 2, None, y.data_ptr(), dz.data_ptr(),
                                     scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                     invstd.data_ptr(), coef.data_ptr(),
                                     0 if xcoeff is None else 1, x.data_ptr(), _ptr(xs), _ptr(xh),
                                     dw.data_ptr(), ws.data_ptr(), _stream(x))
        _L.check(rc, "mlp_gemm_wgrad")
    return dw


def gemm_backward_fused(w, x, xcoeff=None, fly=None, pooled=None, xstats=None, need_dx=True,
                        lin_w=None):
    """dgrad and wgrad of one layer in one pass: -> (dx (B,K,...), dw (M,K), below), or None when
    the layer's shape is outside the fused kernel (callers then use gemm_dgrad + gemm_wgrad).
    x (B,K,...) direct or relu(bn(.)) via xcoeff=(scale, shift); the gradient operand on the fly
    from fly / pooled as in gemm_dgrad.
    xstats = (mean, invstd, gamma, training) of the layer that produced x (required with xcoeff):
    `below` is then that layer's (dgamma, dbeta, coef), as bn_relu_backward_stats(x, dx, ...)
    would return them -- the sums come out of the dgrad epilogue, no pass over (x, dx).
    need_dx=False (a first layer whose input needs no gradient): dx is None.
    lin_w (64,4): the layer below is a 4 -> 64 first layer whose output was never stored; x is then
    THAT layer's input (B,4,...), xcoeff / xstats its BatchNorm.  The gradient w.r.t. its activated
    output is not written either: in its place comes a GatedSums -- what wgrad_first4_from_gated needs
    of it for that layer's weight gradient (`below` carries its BatchNorm sums)."""
    m, k = w.shape
    _f32c(w, "w"); _f32c(x, "x")
    b = x.shape[0]
    r = x.numel() // (b * (4 if lin_w is not None else k))
    if pooled is not None:
        y, dz, argmax, scale, shift, mean, invstd, coef = pooled
        pmode, ns = 3, y.shape[3]
    else:
        y, dz, scale, shift, mean, invstd, coef = fly
        argmax, pmode, ns = None, 2, 0
    qmode = 4 if lin_w is not None else (0 if xcoeff is None else 1)
    if not _lib.mlp_gemm_backward_fused_supported(b, m, k, r, pmode, qmode, ns):
        return None
    if not need_dx and (m, k) != (128, 259):  # the weight-gradient-only form exists for this shape
        return None
    if qmode != 0 and xstats is None:
        raise RuntimeError("xstats=(mean, invstd, gamma, training) is required with xcoeff")
    xs, xh = xcoeff if xcoeff is not None else (None, None)
    xmean, xinv, xgamma, xtraining = xstats if qmode != 0 else (None, None, None, False)
    parts = int(_lib.mlp_gemm_backward_fused_stats_parts(b, m, k, r)) if qmode != 0 else 0
    gated = lin_w is not None and bool(_lib.mlp_gemm_backward_fused_lin4_gated())
    if gated:
        _f32c(lin_w, "lin_w")
        if parts <= 0:
            raise RuntimeError("gemm_backward_fused(lin_w): no partials for this shape")
        dx = torch.empty((parts, k, 4), dtype=torch.float32, device=x.device)  # the gated sums' partials
    elif lin_w is not None:
        _f32c(lin_w, "lin_w")
        dx = torch.empty((b, k) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    else:
        dx = torch.empty_like(x) if need_dx else None
    dw = torch.empty((m, k), dtype=torch.float32, device=x.device)
    below = None
    with torch.cuda.device(x.device):
        ws = torch.empty(max(int(_lib.mlp_gemm_backward_fused_workspace_floats(b, m, k, r)), 1),
                         dtype=torch.float32, device=x.device)
        _keep_until_flush(ws, dw)
        sp = torch.empty((k, parts, 2), dtype=torch.float32, device=x.device) if parts else None
        _L.check(_lib.mlp_gemm_backward_fused(b, m, k, r, w.data_ptr(), pmode, y.data_ptr(),
                                              dz.data_ptr(), _ptr(argmax), ns, scale.data_ptr(),
                                              shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                              coef.data_ptr(), qmode, x.data_ptr(), _ptr(xs),
                                              _ptr(xh), _ptr(xmean), _ptr(xinv), _ptr(lin_w), _ptr(dx),
                                              dw.data_ptr(), ws.data_ptr(), _ptr(sp), _stream(x)),
                 "mlp_gemm_backward_fused")
        if parts:
            small = torch.empty((5, k), dtype=torch.float32, device=x.device)
            _L.check(_lib.mlp_bn_backward_finalize(k, parts, float(b) * float(r),
                                                   1 if xtraining else 0, sp.data_ptr(),
                                                   xgamma.data_ptr(), xinv.data_ptr(),
                                                   small[0].data_ptr(), small[1].data_ptr(),
                                                   small[2:].data_ptr(), _stream(x)),
                     "mlp_bn_backward_finalize")
            below = (small[0], small[1], small[2:])
    if gated:
        dx = GatedSums(dx)
    return dx, dw, below


class GatedSums(object):
    """What the one-pass backward of the layer above leaves of the gradient w.r.t. a VIRTUAL 4 -> 64
    first layer's output: parts x (64,4) partial sums G = sum_n [gate] dz x^T (the tensor itself,
    268 MB at SA1, is never written)."""

    def __init__(self, partials):
        self.partials = partials


def wgrad_first4_from_gated(w, gated, mean, invstd, coef, moments):
    """dw (64,4) of the virtual first layer from the GatedSums of gemm_backward_fused(lin_w=...), its
    BatchNorm's (mean, invstd), coef (64,3) and the moments of its input (first4_moments)."""
    _f32c(w, "w")
    g = gated.partials
    dw = torch.empty((64, 4), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        ws = torch.empty(256, dtype=torch.float32, device=w.device)
        _L.check(_lib.mlp_wgrad_first4_from_gated(int(g.shape[0]), g.data_ptr(), w.data_ptr(), mean.data_ptr(),
                                                  invstd.data_ptr(), coef.data_ptr(), moments.data_ptr(),
                                                  dw.data_ptr(), ws.data_ptr(), _stream(w)),
                 "mlp_wgrad_first4_from_gated")
    return dw


def small_backward_prefers_dy(w, y):
    """True when the backward of the layer y = w x (y (B,M,...) behind BatchNorm + ReLU) is faster
    with dy WRITTEN once (bn_relu_backward: sums + apply, two launches) and the pair launch reading
    it, than with dy formed inside the pair launch's operand loads: in the small regime every one
    of the K/64 row tiles of the data gradient and every tile of the weight gradient re-forms the
    same dy tile (6 instructions + the bf16 split per element, up to 8 times over).  Measured
    (tools/small_bwd_form_bench.py): 256 x 512 at 8 x 1024 columns 105 -> 83 us, 256 x 256 60 -> 48,
    at 8 x 512 columns 39 -> 37, 128 x 128 at 8 x 256 columns 19.5 = 19.4.  MLP_SMALL_BWD_DY_COLS:
    least number of columns (default 4096, 0 = never)."""
    m, k = w.shape
    b = y.shape[0]
    r = y.numel() // (b * m)
    least = int(os.environ.get("MLP_SMALL_BWD_DY_COLS", "4096"))
    return least > 0 and b * r >= least and bool(_lib.mlp_gemm_backward_small_supported(b, m, k, r, 0, 0))


def gemm_backward_small(w, x, xcoeff=None, dy=None, fly=None, need_dx=True):
    """Both backward GEMMs of a SMALL layer (FP modules, heads, pre-gather first layers) in one
    launch: -> (dx (B,K,...) or None, dw (M,K)), or None when the layer is outside the small
    regime (callers then use gemm_dgrad + gemm_wgrad).  The gradient operand is dy (B,M,...) or
    formed on the fly from fly = (y, dz, scale, shift, mean, invstd, coef); x direct or
    relu(bn(.)) via xcoeff = (scale, shift)."""
    m, k = w.shape
    b = x.shape[0]
    r = x.numel() // (b * k)
    pmode = 0 if dy is not None else 2
    qmode = 0 if xcoeff is None else 1
    if not _lib.mlp_gemm_backward_small_supported(b, m, k, r, pmode, qmode):
        return None
    _f32c(w, "w"); _f32c(x, "x")
    if dy is not None:
        _f32c(dy, "dy")
        p0, pdz, sc, sh, mean, invstd, coef = dy, None, None, None, None, None, None
    else:
        p0, pdz, sc, sh, mean, invstd, coef = fly
        _f32c(p0, "y"); _f32c(pdz, "dz")
    xs, xh = xcoeff if xcoeff is not None else (None, None)
    dx = torch.empty_like(x) if need_dx else None
    dw = torch.empty((m, k), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        ws = torch.empty(max(int(_lib.mlp_gemm_wgrad_workspace_floats(b, m, k, r)), 1),
                         dtype=torch.float32, device=x.device)
        _keep_until_flush(ws, dw)
        image = _image_of(w, b, r) if (pmode == 0 and need_dx) else None
        if image is not None:
            _L.check(_lib.mlp_gemm_backward_small_img(
                b, m, k, r, w.data_ptr(), image[1], pmode, p0.data_ptr(), _ptr(pdz), _ptr(sc), _ptr(sh),
                _ptr(mean), _ptr(invstd), _ptr(coef), qmode, x.data_ptr(), _ptr(xs), _ptr(xh), _ptr(dx),
                dw.data_ptr(), ws.data_ptr(), _stream(x)), "mlp_gemm_backward_small_img")
            return dx, dw
        _L.check(_lib.mlp_gemm_backward_small(b, m, k, r, w.data_ptr(), pmode, p0.data_ptr(),
                                              _ptr(pdz), _ptr(sc), _ptr(sh), _ptr(mean),
                                              _ptr(invstd), _ptr(coef), qmode, x.data_ptr(),
                                              _ptr(xs), _ptr(xh), _ptr(dx), dw.data_ptr(),
                                              ws.data_ptr(), _stream(x)), "mlp_gemm_backward_small")
    return dx, dw


def wgrad_first4(w, x, fly, moments=None):
    """dw (64,4) of a first layer y = w x with a 4-channel input x (B,4,...) behind BatchNorm +
    ReLU, from fly = (y, dz, scale, shift, mean, invstd, coef) as gemm_wgrad takes it -- y is
    ignored: the ReLU gate is recomputed from x and everything else follows from the second
    moments of x (`moments`: first4_moments(x) if the caller has them, e.g. from the forward).  None
    when the shape is not (64, 4) or the columns are not a multiple of 4."""
    m, k = w.shape
    b = x.shape[0]
    r = x.numel() // (b * x.shape[1])
    if (m, k) != (64, 4) or x.shape[1] != 4 or r % 4 != 0 or os.environ.get("MLP_WGRAD_FIRST4", "1") == "0":
        return None
    _f32c(w, "w"); _f32c(x, "x")
    _, dz, scale, shift, mean, invstd, coef = fly
    _f32c(dz, "dz")
    dw = torch.empty((m, k), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        ws = torch.empty(int(_lib.mlp_wgrad_first4_workspace_bytes(b, r)), dtype=torch.uint8, device=x.device)
        _L.check(_lib.mlp_wgrad_first4(b, r, w.data_ptr(), x.data_ptr(), dz.data_ptr(), scale.data_ptr(),
                                       shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                       coef.data_ptr(), _ptr(moments), dw.data_ptr(), ws.data_ptr(),
                                       _stream(x)), "mlp_wgrad_first4")
    return dw


def first4_moments(x):
    """The 14 moments (4 sums, 10 products) of a 4-channel tensor x (B,4,...) as partial rows of
    doubles: what BatchNorm statistics and weight gradients of a 4 -> 64 first layer need of x."""
    _f32c(x, "x")
    b = x.shape[0]
    r = x.numel() // (b * 4)
    mom = torch.empty(int(_lib.mlp_first4_moments_doubles()), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        _L.check(_lib.mlp_first4_moments(b, r, x.data_ptr(), mom.data_ptr(), _stream(x)), "mlp_first4_moments")
    return mom


def first4_bn(moments, count, w, gamma, beta, running_mean, running_var, momentum, eps):
    """Training-mode BatchNorm coefficients (mean, invstd, scale, shift) of y = w x for w (64,4),
    from the moments of x over `count` columns -- y itself is never formed; running statistics are
    updated like bn_coefficients does."""
    _f32c(w, "w")
    out = torch.empty((4, 64), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _L.check(_lib.mlp_first4_bn(moments.data_ptr(), float(count), w.data_ptr(), gamma.data_ptr(),
                                    beta.data_ptr(), float(eps), float(momentum), _ptr(running_mean),
                                    _ptr(running_var), out[0].data_ptr(), out[1].data_ptr(),
                                    out[2].data_ptr(), out[3].data_ptr(), _stream(w)), "mlp_first4_bn")
    return out[0], out[1], out[2], out[3]


def lin4_supported(w0, w1, x):
    """Can the first layer (w0 (64,4) on x (B,4,...)) stay virtual, i.e. be recomputed inside the
    kernels of the second layer (w1 (64,64))?"""
    if os.environ.get("MLP_FIRST4_VIRTUAL", "1") == "0":
        return False
    if os.environ.get("MLP_WGRAD_FIRST4", "1") == "0":
        return False  # the virtual layer's weight gradient only exists as mlp_wgrad_first4
    if tuple(w0.shape) != (64, 4) or tuple(w1.shape) != (64, 64) or x.shape[1] != 4:
        return False
    b = x.shape[0]
    r = x.numel() // (b * 4)
    return (w0.data_ptr() % 16 == 0 and w1.data_ptr() % 16 == 0
            and int(_lib.mlp_gemm_forward_stats_parts(b, 64, 64, r, None)) > 0
            and bool(_lib.mlp_gemm_backward_fused_supported(b, 64, 64, r, 2, 4, 0)))


def gemm_forward_bn_lin4(w, x4, w1, coeff1, gamma, beta, running_mean, running_var, momentum, eps):
    """gemm_forward_bn for the second layer (w (64,64)) when the first (w1 (64,4), BatchNorm
    coefficients coeff1 = (scale, shift)) is virtual: its activated output is recomputed from x4."""
    import ctypes
    _f32c(x4, "x4"); _f32c(w, "w"); _f32c(w1, "w1")
    b = x4.shape[0]
    r = x4.numel() // (b * 4)
    m = 64
    cols = ctypes.c_int(0)
    parts = int(_lib.mlp_gemm_forward_stats_parts(b, m, 64, r, ctypes.byref(cols)))
    y = torch.empty((b, m) + tuple(x4.shape[2:]), dtype=torch.float32, device=x4.device)
    pairs = torch.empty((parts, m, 2), dtype=torch.float32, device=x4.device)
    out = torch.empty((4, m), dtype=torch.float32, device=x4.device)
    scratch = torch.empty(int(_lib.mlp_bn_finalize_pairs_scratch_bytes(m)), dtype=torch.uint8,
                          device=x4.device)
    with torch.cuda.device(x4.device):
        _L.check(_lib.mlp_gemm_forward_stats_lin4(b, r, w.data_ptr(), x4.data_ptr(), w1.data_ptr(),
                                                  coeff1[0].data_ptr(), coeff1[1].data_ptr(),
                                                  y.data_ptr(), pairs.data_ptr(), _stream(x4)),
                 "mlp_gemm_forward_stats_lin4")
        _L.check(_lib.mlp_bn_finalize_pairs(m, parts, cols.value, pairs.data_ptr(), gamma.data_ptr(),
                                            beta.data_ptr(), float(eps), float(momentum),
                                            _ptr(running_mean), _ptr(running_var), out[0].data_ptr(),
                                            out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(),
                                            scratch.data_ptr(), _stream(x4)), "mlp_bn_finalize_pairs")
    return y, out[0], out[1], out[2], out[3]


def chain_lin4_supported(w0, w1, w2, x4, ns):
    """Can layers 2 and 3 of a 4 -> 64 -> 64 -> 128 module run as ONE register-chained kernel
    (csrc/mlp_chain.hip) on x4 (B,4,m,ns)?"""
    if os.environ.get("MLP_CHAIN_FWD", "1") == "0":
        return False
    if tuple(w0.shape) != (64, 4) or tuple(w1.shape) != (64, 64) or tuple(w2.shape) != (128, 64):
        return False
    if x4.dim() != 4 or x4.shape[1] != 4 or x4.shape[3] != ns:
        return False
    if any(w.data_ptr() % 16 for w in (w0, w1, w2)):
        return False
    b = x4.shape[0]
    r = x4.numel() // (b * 4)
    return int(_lib.mlp_chain_lin4_parts(b, r, 128, int(ns), None)) > 0


def chain_lin4_forward(x4, w0, coeff0, layer1, layer2, store=True, store_last=True):
    """Layers 2 and 3 of a 4 -> 64 -> 64 -> 128 set-abstraction MLP in training mode, chained in
    registers.  x4 (B,4,m,ns); w0 (64,4) and coeff0 = (scale, shift) of the virtual first layer;
    layerN = (w, gamma, beta, running_mean, running_var, momentum, eps).
    -> (y of layer 1 (B,64,m,ns), its (mean, invstd, scale, shift), y of layer 2 (B,128,m,ns), its
    coefficients, ext = the extrema planes for pool_from_extrema); store=False: the y are None;
    store_last=False: only the last layer's raw output is not stored (its backward then runs
    from the Gram matrix of its input, pool_gram_backward)."""
    _f32c(x4, "x4"); _f32c(w0, "w0")
    b, _, m, ns = x4.shape
    r = m * ns
    w1, g1, be1, rm1, rv1, mom1, eps1 = layer1
    w2, g2, be2, rm2, rv2, mom2, eps2 = layer2
    _f32c(w1, "w1"); _f32c(w2, "w2")
    import ctypes
    cols = ctypes.c_int(0)
    parts = int(_lib.mlp_chain_lin4_parts(b, r, 128, ns, ctypes.byref(cols)))
    if parts <= 0:
        raise RuntimeError("chain_lin4_forward: shape not covered")
    dev = x4.device
    pairs1 = torch.empty((parts, 64, 2), dtype=torch.float32, device=dev)
    pairs2 = torch.empty((parts, 128, 2), dtype=torch.float32, device=dev)
    img = torch.empty(int(_lib.mlp_chain_lin4_image_bytes()), dtype=torch.uint8, device=dev)
    out1 = torch.empty((4, 64), dtype=torch.float32, device=dev)
    out2 = torch.empty((4, 128), dtype=torch.float32, device=dev)
    y1 = torch.empty((b, 64, m, ns), dtype=torch.float32, device=dev) if store else None
    y2 = torch.empty((b, 128, m, ns), dtype=torch.float32, device=dev) if (store and store_last) else None
    ext = torch.empty((2, b, 128, m), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = _stream(x4)
        _L.check(_lib.mlp_chain_lin4_prepare(w0.data_ptr(), coeff0[0].data_ptr(), coeff0[1].data_ptr(),
                                             w1.data_ptr(), w2.data_ptr(), img.data_ptr(), st),
                 "mlp_chain_lin4_prepare")
        _L.check(_lib.mlp_chain_lin4_stats(b, r, ns, x4.data_ptr(), img.data_ptr(), pairs1.data_ptr(), st),
                 "mlp_chain_lin4_stats")
        _L.check(_lib.mlp_chain_finalize(64, parts, cols.value, pairs1.data_ptr(), g1.data_ptr(),
                                         be1.data_ptr(), float(eps1), float(mom1), _ptr(rm1), _ptr(rv1),
                                         out1[0].data_ptr(), out1[1].data_ptr(), out1[2].data_ptr(),
                                         out1[3].data_ptr(), st), "mlp_chain_finalize")
        _L.check(_lib.mlp_chain_lin4_forward(b, r, ns, x4.data_ptr(), img.data_ptr(), out1[2].data_ptr(),
                                             out1[3].data_ptr(), g2.data_ptr(), _ptr(y1), _ptr(y2),
                                             pairs2.data_ptr(), ext.data_ptr(), st), "mlp_chain_lin4_forward")
        _L.check(_lib.mlp_chain_finalize(128, parts, cols.value, pairs2.data_ptr(), g2.data_ptr(),
                                         be2.data_ptr(), float(eps2), float(mom2), _ptr(rm2), _ptr(rv2),
                                         out2[0].data_ptr(), out2[1].data_ptr(), out2[2].data_ptr(),
                                         out2[3].data_ptr(), st), "mlp_chain_finalize")
    return y1, (out1[0], out1[1], out1[2], out1[3]), y2, (out2[0], out2[1], out2[2], out2[3]), ext


def _gram_entry(w):
    """The C entry points for a pooled last layer of this shape: (128,64) csrc/mlp_pool_gram.hip,
    (256,128) csrc/mlp_pool_gram256.hip."""
    shape = tuple(w.shape)
    if shape == (128, 64):
        return (_lib.mlp_pool_gram_supported, _lib.mlp_pool_gram_parts, _lib.mlp_pool_gram_workspace_floats,
                _lib.mlp_pool_gram_backward, "mlp_pool_gram_backward")
    if shape == (256, 128):
        return (_lib.mlp_pool_gram256_supported, _lib.mlp_pool_gram256_parts,
                _lib.mlp_pool_gram256_workspace_floats, _lib.mlp_pool_gram256_backward,
                "mlp_pool_gram256_backward")
    return None


def pool_gram_supported(w, y_in, ns):
    """Can the backward of the pooled last layer (w (128,64) on y_in (B,64,m,ns), or w (256,128) on
    y_in (B,128,m,ns)) run without that layer's raw output (csrc/mlp_pool_gram.hip,
    csrc/mlp_pool_gram256.hip)?  Only the (B, *, m, ns) extent of y_in is looked at (the chained SA1
    form asks with its 4-channel input: the layer's own input never exists there)."""
    entry = _gram_entry(w)
    if os.environ.get("MLP_POOL_GRAM", "1") == "0" or entry is None or y_in.dim() != 4:
        return False
    b, r = y_in.shape[0], y_in.shape[2] * y_in.shape[3]
    return bool(entry[0](b, w.shape[0], w.shape[1], r, int(ns)))


def pool_gram_backward(w, y_in, in_coeff, in_gamma, coef, coeff, dpooled, argmax, ymax, ns, training):
    """Backward of a pooled last layer y = w . relu(bn(y_in)) from y_in and the pooled tensors alone.
    y_in (B,K,m,ns) raw output of the layer below, in_coeff = its (mean, invstd, scale, shift),
    in_gamma its BatchNorm weight; coef (M,3) and coeff = (mean, invstd, scale, shift) of THIS
    layer's BatchNorm; dpooled / argmax / ymax (B,M,m); (M,K) = (128,64) or (256,128).
    -> (d relu(bn(y_in)) (B,K,m,ns), dw (M,K), below = (dgamma, dbeta, coef) of the layer below)."""
    _f32c(w, "w"); _f32c(y_in, "y_in"); _f32c(dpooled, "dpooled"); _f32c(ymax, "ymax")
    entry = _gram_entry(w)
    if entry is None or y_in.shape[1] != w.shape[1]:
        raise RuntimeError("pool_gram_backward: layer shape %s not covered" % (tuple(w.shape),))
    _, parts_fn, ws_fn, backward_fn, name = entry
    mo, k = w.shape
    b = y_in.shape[0]
    r = y_in.numel() // (b * k)
    dev = y_in.device
    mean_i, invstd_i, scale_i, shift_i = in_coeff
    mean, invstd, scale, shift = coeff
    parts = int(parts_fn(b, r))
    dq = torch.empty_like(y_in)
    dw = torch.empty((mo, k), dtype=torch.float32, device=dev)
    sp = torch.empty((k, parts, 2), dtype=torch.float32, device=dev)
    small = torch.empty((5, k), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        ws_floats = ws_fn(b, r, int(ns)) if mo == 256 else ws_fn(b, r)
        ws = torch.empty(int(ws_floats), dtype=torch.float32, device=dev)
        st = _stream(y_in)
        _L.check(backward_fn(b, r, int(ns), w.data_ptr(), y_in.data_ptr(), scale_i.data_ptr(),
                             shift_i.data_ptr(), mean_i.data_ptr(), invstd_i.data_ptr(),
                             coef.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                             mean.data_ptr(), invstd.data_ptr(), argmax.data_ptr(),
                             dpooled.data_ptr(), ymax.data_ptr(), dq.data_ptr(),
                             dw.data_ptr(), sp.data_ptr(), ws.data_ptr(), st), name)
        _L.check(_lib.mlp_bn_backward_finalize(k, parts, float(b) * float(r), 1 if training else 0,
                                               sp.data_ptr(), in_gamma.data_ptr(), invstd_i.data_ptr(),
                                               small[0].data_ptr(), small[1].data_ptr(), small[2:].data_ptr(),
                                               st), "mlp_bn_backward_finalize")
    return dq, dw, (small[0], small[1], small[2:])


# ---- first layer of a set-abstraction module applied before the gather (csrc/mlp_pregather.hip) ----
def pregather_supported(b, c, n, m, ns):
    return bool(_lib.mlp_pregather_supported(int(b), int(c), int(n), int(m), int(ns)))


def pregather_pack(xyz, new_xyz, features, s):
    """src_ext (B, 3+C, N+m) = [xyz*s | new_xyz*s ; features | 0] from xyz (B,N,3), new_xyz (B,m,3),
    features (B,C,N)."""
    _f32c(xyz, "xyz"); _f32c(new_xyz, "new_xyz"); _f32c(features, "features")
    b, n, _ = xyz.shape
    m, c = new_xyz.shape[1], features.shape[1]
    out = torch.empty((b, 3 + c, n + m), dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _L.check(_lib.mlp_pregather_pack(b, n, m, c, float(s), xyz.data_ptr(), new_xyz.data_ptr(),
                                         features.data_ptr(), out.data_ptr(), _stream(xyz)),
                 "mlp_pregather_pack")
    return out


def pregather_unpack_grad(dsrc_ext, n, m):
    """d features (B, C, n) out of d src_ext (B, 3+C, n+m)."""
    _f32c(dsrc_ext, "dsrc_ext")
    b, c = dsrc_ext.shape[0], dsrc_ext.shape[1] - 3
    out = torch.empty((b, c, n), dtype=torch.float32, device=dsrc_ext.device)
    with torch.cuda.device(dsrc_ext.device):
        _L.check(_lib.mlp_pregather_unpack_grad(b, n, m, c, dsrc_ext.data_ptr(), out.data_ptr(),
                                                _stream(dsrc_ext)), "mlp_pregather_unpack_grad")
    return out


def pregather_forward(z_ext, idx, n, stats=None):
    """y (B, c, m, ns) = z_ext[:, :, idx] - z_ext[:, :, n + j].  stats = (gamma, beta, running_mean,
    running_var, momentum, eps): also the training-mode BatchNorm coefficients of y
    (mean, invstd, scale, shift), from the per-row moments the kernel leaves behind."""
    _f32c(z_ext, "z_ext")
    b, c, w = z_ext.shape
    m, ns = idx.shape[1], idx.shape[2]
    if idx.dtype != torch.int32 or not idx.is_contiguous() or w != n + m:
        raise RuntimeError("idx must be a contiguous int32 (B, m, ns) and z_ext (B, c, n + m)")
    y = torch.empty((b, c, m, ns), dtype=torch.float32, device=z_ext.device)
    pairs = torch.empty((b, c, 2), dtype=torch.float32, device=z_ext.device) if stats else None
    with torch.cuda.device(z_ext.device):
        _L.check(_lib.mlp_pregather_forward(b, c, n, m, ns, z_ext.data_ptr(), idx.data_ptr(),
                                            y.data_ptr(), _ptr(pairs), _stream(z_ext)),
                 "mlp_pregather_forward")
        if not stats:
            return y
        gamma, beta, running_mean, running_var, momentum, eps = stats
        out = torch.empty((4, c), dtype=torch.float32, device=z_ext.device)
        scratch = torch.empty(int(_lib.mlp_bn_finalize_pairs_scratch_bytes(c)), dtype=torch.uint8,
                              device=z_ext.device)
        _L.check(_lib.mlp_bn_finalize_pairs(c, b, m * ns, pairs.data_ptr(), gamma.data_ptr(),
                                            beta.data_ptr(), float(eps), float(momentum),
                                            _ptr(running_mean), _ptr(running_var),
                                            out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                            out[3].data_ptr(), scratch.data_ptr(), _stream(z_ext)),
                 "mlp_bn_finalize_pairs")
    return y, out[0], out[1], out[2], out[3]


def pregather_backward(fly, inverse, n):
    """dz_ext (B, c, n + m) from fly = (y, dz, scale, shift, mean, invstd, coef) as gemm_dgrad
    takes it and the inverse index of the layer's idx (_ext.group_inverse)."""
    y, dz, scale, shift, mean, invstd, coef = fly
    _f32c(y, "y"); _f32c(dz, "dz")
    b, c, m, ns = y.shape
    if (inverse.dtype != torch.int32 or not inverse.is_contiguous() or inverse.device != y.device
            or tuple(inverse.shape) != (b, int(_lib.pn2_group_inverse_entries(m, ns)))):
        raise RuntimeError("inverse is not the int32 group_inverse() of a (B,%d,%d) index array on %s"
                           % (m, ns, y.device))
    out = torch.empty((b, c, n + m), dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
        _L.check(_lib.mlp_pregather_backward(b, c, n, m, ns, y.data_ptr(), dz.data_ptr(),
                                             scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                             invstd.data_ptr(), coef.data_ptr(), inverse.data_ptr(),
                                             out.data_ptr(), _stream(y)), "mlp_pregather_backward")
    return out
