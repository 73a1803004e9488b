"""pointnet2._ext -- drop-in for the reference's compiled extension of the same name
(pybind surface: pointnet2/_ext_src/src/bindings.cpp:11-24), backed by the gfx950 HIP library
through its C ABI (include/pn2_hip.h).

Same nine functions, same argument order, same outputs.  Differences, all on the lenient
side:
  * errors are Python exceptions (the reference exit()s on a failed launch, cuda_utils.h:35-44);
  * CPU tensors raise RuntimeError("CPU not supported"), as the reference does
    (e.g. ball_query.cpp:33) -- there is no CPU path here either;
  * three_interpolate_grad computes the true gradient (the reference dispatches its forward
    kernel by mistake, interpolate.cpp:95).
"""
import importlib.util
import os
import sys

import torch


def _load_lib():
    name = "_3dioumatch_amd_lib"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                         "..", "..", "_lib.py"))
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        del sys.modules[name]
        raise
    return mod


_L = _load_lib()
_lib = _L.lib


def _chk_f32(t, name):
    if not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be a float tensor" % name)


def _chk_i32(t, name):
    if not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)
    if t.dtype != torch.int32:
        raise RuntimeError("%s must be an int tensor" % name)


def _chk_dev(first, *rest):
    if not first.is_cuda:
        raise RuntimeError("CPU not supported")
    for t, name in rest:
        if not t.is_cuda:
            raise RuntimeError("%s must be a CUDA tensor" % name)
        if t.device != first.device:
            raise RuntimeError("%s must be on %s" % (name, first.device))


def _stream(t):
    return _L.current_stream_ptr(t.device)


def gather_points(points, idx):
    """points (B,C,N) f32, idx (B,m) i32 -> (B,C,m).  sampling.cpp:20-44"""
    _chk_f32(points, "points"); _chk_i32(idx, "idx"); _chk_dev(points, (idx, "idx"))
    b, c, n = points.shape
    m = idx.shape[1]
    out = torch.empty((b, c, m), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        _L.check(_lib.pn2_gather_points(b, c, n, m, points.data_ptr(), idx.data_ptr(),
                                        out.data_ptr(), _stream(points)), "gather_points")
    return out


def gather_points_grad(grad_out, idx, n):
    """grad_out (B,C,m), idx (B,m) -> (B,C,n) scatter-add.  sampling.cpp:46-68"""
    _chk_f32(grad_out, "grad_out"); _chk_i32(idx, "idx"); _chk_dev(grad_out, (idx, "idx"))
    b, c, m = grad_out.shape
    out = torch.empty((b, c, int(n)), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        _L.check(_lib.pn2_gather_points_grad(b, c, int(n), m, grad_out.data_ptr(), idx.data_ptr(),
                                             out.data_ptr(), _stream(grad_out)),
                 "gather_points_grad")
    return out


# ---- cell lists behind the reference surface --------------------------------------------------
# The reference's set-abstraction layer calls furthest_point_sampling(xyz, npoint) and then
# ball_query(new_xyz, xyz, radius, nsample) on the SAME cloud (pointnet2_modules.py:236-250), and
# the pybind surface has nowhere to hand the cell lists from one call to the next.  So they are
# kept here, per cloud tensor:
#   * an entry lives exactly as long as its xyz tensor (weak reference) and is valid for one
#     in-place version of it (`_version`) and one radius;
#   * ball_query / query_and_group look the cloud up and only build on a miss; the radius they
#     were asked for is remembered per cloud size, and furthest_point_sampling of a large cloud of
#     that size leaves the lists for it behind (the sampling kernel streams the cloud anyway);
#   * while a stream is being captured into a graph the cache is bypassed altogether (a replay
#     may find other coordinates behind the same tensor) unless the caller opts in with
#     lists_cached_during_capture() because the captured inputs never change.
import contextlib
import weakref

_LISTS = {}          # id(xyz) -> (weakref(xyz), _version, radius, CellLists)
_RADIUS_HINT = {}    # (device index, n) -> radius of the last cell-list query on such a cloud
_CACHE_IN_CAPTURE = [False]
cache_stats = {"hits": 0, "misses": 0, "left_by_sampling": 0}


@contextlib.contextmanager
def lists_cached_during_capture():
    """Allow cache HITS while a HIP graph is being captured (never inserts): only for captures
    whose cloud tensors keep their contents for every replay (bench.py's operator timings)."""
    _CACHE_IN_CAPTURE[0] = True
    try:
        yield
    finally:
        _CACHE_IN_CAPTURE[0] = False


def _capturing(t):
    return t.is_cuda and torch.cuda.is_current_stream_capturing()


def _tensor_version(t):
    """In-place version of a tensor, or None where PyTorch keeps none (tensors created under
    torch.inference_mode(): such clouds simply bypass the caches -- no hit, no insert)."""
    if t.is_inference():
        return None
    try:
        return t._version
    except RuntimeError:
        return None


def _cached_lists(xyz, radius):
    if _capturing(xyz) and not _CACHE_IN_CAPTURE[0]:
        return None
    ent = _LISTS.get(id(xyz))
    if ent is None:
        return None
    ref, version, rad, lists, stream = ent
    if ref() is not xyz or version is None or version != _tensor_version(xyz) or rad != float(radius):
        return None
    cur = torch.cuda.current_stream(xyz.device)
    if stream != cur.cuda_stream:
        # built on another stream: order this stream after the build and keep the buffer alive
        # for it (the allocator would otherwise hand it back to the building stream's pool).
        # A build that has already completed needs no ordering; a capturing stream cannot wait
        # on an outside event, so an unfinished build is a miss there.
        if lists.event is None:
            return None
        done = lists.event.query()
        if _capturing(xyz):
            if not done:
                return None
        else:
            if not done:
                cur.wait_event(lists.event)
            lists.buf.record_stream(cur)
    return lists


def _remember_lists(xyz, radius, lists):
    if _capturing(xyz):
        return
    key = id(xyz)

    def _drop(_ref, key=key):
        ent = _LISTS.get(key)
        if ent is not None and ent[0] is _ref:
            del _LISTS[key]

    version = _tensor_version(xyz)
    if version is None:
        return
    cur = torch.cuda.current_stream(xyz.device)
    lists.event = torch.cuda.Event()
    lists.event.record(cur)
    _LISTS[key] = (weakref.ref(xyz, _drop), version, float(radius), lists, cur.cuda_stream)


def _lists_for(xyz, radius):
    """CellLists of xyz for radius through the cache (None: cloud outside the cell-list tier)."""
    b, n, _ = xyz.shape
    if not xyz.is_cuda or not grid_supported(b, n) or not (1e-6 < float(radius) < 1e6):
        return None
    _RADIUS_HINT[(xyz.device.index, n)] = float(radius)
    lists = _cached_lists(xyz, radius)
    if lists is not None:
        cache_stats["hits"] += 1
        return lists
    cache_stats["misses"] += 1
    lists = build_grid(xyz, radius)
    _remember_lists(xyz, radius, lists)
    return lists


def furthest_point_sampling(points, nsamples):
    """points (B,N,3) f32 -> (B,nsamples) i32, index-exact.  sampling.cpp:70-91"""
    _chk_f32(points, "points")
    if not points.is_cuda:
        raise RuntimeError("CPU not supported")
    b, n, _ = points.shape
    nsamples = int(nsamples)
    hint = _RADIUS_HINT.get((points.device.index, n))
    if hint is not None and not _capturing(points) and _lib.pn2_fps_grid_supported(n) and \
            _cached_lists(points, hint) is None:
        # a ball query of this radius followed the last sampling of such a cloud: leave its cell
        # lists behind (same kernel, same indices)
        out, lists = furthest_point_sampling_with_grid(points, nsamples, hint)
        if lists is not None:
            _remember_lists(points, hint, lists)
            cache_stats["left_by_sampling"] += 1
        return out
    out = torch.empty((b, nsamples), dtype=torch.int32, device=points.device)  # every index is written
    with torch.cuda.device(points.device):
        need = int(_lib.pn2_fps_workspace_bytes(b, n, nsamples))
        # private scratch per call (the reference allocates its `temp` per call as well)
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=points.device)
        _L.check(_lib.pn2_furthest_point_sampling_ws(b, n, nsamples, points.data_ptr(),
                                                     out.data_ptr(), ws.data_ptr(), need,
                                                     _stream(points)), "furthest_point_sampling")
    return out


def three_nn(unknowns, knows):
    """unknowns (B,n,3), knows (B,m,3) -> [dist2 (B,n,3) f32, idx (B,n,3) i32].
    interpolate.cpp:19-45"""
    _chk_f32(unknowns, "unknowns"); _chk_f32(knows, "knows")
    _chk_dev(unknowns, (knows, "knows"))
    b, n, _ = unknowns.shape
    m = knows.shape[1]
    dist2 = torch.empty((b, n, 3), dtype=torch.float32, device=unknowns.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=unknowns.device)
    with torch.cuda.device(unknowns.device):
        _L.check(_lib.pn2_three_nn(b, n, m, unknowns.data_ptr(), knows.data_ptr(),
                                   dist2.data_ptr(), idx.data_ptr(), _stream(unknowns)),
                 "three_nn")
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    """points (B,C,m), idx (B,n,3) i32, weight (B,n,3) -> (B,C,n).  interpolate.cpp:47-75"""
    _chk_f32(points, "points"); _chk_i32(idx, "idx"); _chk_f32(weight, "weight")
    _chk_dev(points, (idx, "idx"), (weight, "weight"))
    b, c, m = points.shape
    n = idx.shape[1]
    out = torch.empty((b, c, n), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        _L.check(_lib.pn2_three_interpolate(b, c, m, n, points.data_ptr(), idx.data_ptr(),
                                            weight.data_ptr(), out.data_ptr(), _stream(points)),
                 "three_interpolate")
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    """grad_out (B,C,n), idx, weight -> (B,C,m): the intended scatter-add
    (interpolate_gpu.cu:121-148); see module docstring."""
    _chk_f32(grad_out, "grad_out"); _chk_i32(idx, "idx"); _chk_f32(weight, "weight")
    _chk_dev(grad_out, (idx, "idx"), (weight, "weight"))
    b, c, n = grad_out.shape
    out = torch.empty((b, c, int(m)), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        _L.check(_lib.pn2_three_interpolate_grad(b, c, n, int(m), grad_out.data_ptr(),
                                                 idx.data_ptr(), weight.data_ptr(),
                                                 out.data_ptr(), _stream(grad_out)),
                 "three_interpolate_grad")
    return out


def three_nn_weights(dist2):
    """dist2 (B,n,3) squared distances as returned by three_nn -> (B,n,3) normalised inverse-distance
    weights 1 / (sqrt(d2) + 1e-8) / sum: the five tensor kernels of pointnet2_modules.py:395-398 /
    grid_conv_module.py:94-98 (and the sqrt of pointnet2_utils.py:147) in one."""
    _chk_f32(dist2, "dist2")
    if not dist2.is_cuda:
        raise RuntimeError("CPU not supported")
    out = torch.empty_like(dist2)
    with torch.cuda.device(dist2.device):
        _L.check(_lib.pn2_three_nn_weights(dist2.numel() // 3, dist2.data_ptr(), out.data_ptr(),
                                           _stream(dist2)), "three_nn_weights")
    return out


def group_inverse_supported(n, m, ns):
    """Does group_inverse() cover index arrays (B,m,ns) over n points?"""
    return bool(_lib.pn2_group_inverse_supported(int(n), int(m), int(ns)))


def group_inverse(idx, n):
    """Inverse of a grouping index array idx (B,m,ns) i32 with values < n: the positions of each
    cloud sorted by the point they refer to, packed (point << 16 | position) -> (B, entries) int32
    bits (entries = m*ns rounded up to 4096 << k, surplus slots -1, lane-interleaved storage), or None when the shape is outside the fast backward's range.  Built once per index
    array (it depends on coordinates only), consumed by group_points_grad(..., inverse=)."""
    _chk_i32(idx, "idx")
    if not idx.is_cuda:
        raise RuntimeError("CPU not supported")
    b, m, ns = idx.shape
    if not _lib.pn2_group_inverse_supported(int(n), m, ns):
        return None
    inv = torch.empty((b, _lib.pn2_group_inverse_entries(m, ns)), dtype=torch.int32, device=idx.device)
    with torch.cuda.device(idx.device):
        _L.check(_lib.pn2_group_inverse_build(b, int(n), m, ns, idx.data_ptr(), inv.data_ptr(),
                                              _stream(idx)), "group_inverse_build")
    return inv


def group_points_grad_sorted(grad_out, inverse, n, channel0=0):
    """group_points_grad through the inverse index: no float atomics per element.  channel0 > 0
    takes channels channel0.. of grad_out (the feature part of a grouped tensor's gradient) in
    place."""
    _chk_f32(grad_out, "grad_out"); _chk_i32(inverse, "inverse"); _chk_dev(grad_out, (inverse, "inverse"))
    b, c_total, m, ns = grad_out.shape
    c = c_total - int(channel0)
    if c <= 0 or channel0 < 0:
        raise RuntimeError("channel0 outside grad_out")
    if tuple(inverse.shape) != (b, _lib.pn2_group_inverse_entries(m, ns)) or not inverse.is_contiguous():
        raise RuntimeError("inverse is not the group_inverse() of a (B,%d,%d) index array" % (m, ns))
    out = torch.empty((b, c, int(n)), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        _L.check(_lib.pn2_group_points_grad_sorted(b, c, int(n), m, ns, grad_out.data_ptr(), c_total,
                                                   int(channel0), inverse.data_ptr(), out.data_ptr(),
                                                   _stream(grad_out)), "group_points_grad_sorted")
    return out


def three_interpolate_into(points, idx, weight, out, channel0):
    """three_interpolate written into channels [channel0, channel0 + C) of the preallocated
    contiguous (B, C_total, n) tensor `out` (no intermediate, no concatenation copy)."""
    _chk_f32(points, "points"); _chk_i32(idx, "idx"); _chk_f32(weight, "weight"); _chk_f32(out, "out")
    _chk_dev(points, (idx, "idx"), (weight, "weight"), (out, "out"))
    b, c, m = points.shape
    n = idx.shape[1]
    c_total = out.shape[1]
    if out.shape[0] != b or out.shape[2] != n or channel0 < 0 or channel0 + c > c_total:
        raise RuntimeError("out must be (B, C_total >= channel0 + C, n)")
    with torch.cuda.device(points.device):
        _L.check(_lib.pn2_three_interpolate_into(b, c, m, n, points.data_ptr(), idx.data_ptr(),
                                                 weight.data_ptr(),
                                                 out.data_ptr() + 4 * channel0 * n, c_total,
                                                 _stream(points)), "three_interpolate_into")
    return out


def three_interpolate_affine_supported(c, m, n):
    return bool(_lib.pn2_three_interpolate_affine_supported(int(c), int(m), int(n)))


def three_interpolate_affine(points, idx, weight, affine_w, affine_x):
    """three_interpolate(points (B,C,m), idx, weight) + affine_w (C,3) . affine_x (B,3,n) -> (B,C,n):
    the output of a 1x1 convolution over cat([3 coordinate rows, interpolated features]) when
    `points` is that convolution's feature part applied to the SOURCE points
    (include/pn2_hip.h pn2_three_interpolate_affine).  No gradient."""
    for t, name in ((points, "points"), (weight, "weight"), (affine_w, "affine_w"), (affine_x, "affine_x")):
        _chk_f32(t, name)
    _chk_i32(idx, "idx")
    _chk_dev(points, (idx, "idx"), (weight, "weight"), (affine_w, "affine_w"), (affine_x, "affine_x"))
    b, c, m = points.shape
    n = idx.shape[1]
    if tuple(affine_w.shape) != (c, 3) or tuple(affine_x.shape) != (b, 3, n) or \
            not three_interpolate_affine_supported(c, m, n):
        raise RuntimeError("three_interpolate_affine: affine_w (C,3), affine_x (B,3,n), m <= 2048, n % 4 == 0")
    out = torch.empty((b, c, n), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        _L.check(_lib.pn2_three_interpolate_affine(b, c, m, n, points.data_ptr(), idx.data_ptr(),
                                                   weight.data_ptr(), affine_w.data_ptr(),
                                                   affine_x.data_ptr(), out.data_ptr(), _stream(points)),
                 "three_interpolate_affine")
    return out


def three_interpolate_grad_from(grad, channel0, c, idx, weight, m):
    """three_interpolate_grad of channels [channel0, channel0 + c) of the contiguous
    (B, C_total, n) gradient `grad` (read in place, no slice copy) -> (B, c, m)."""
    _chk_f32(grad, "grad"); _chk_i32(idx, "idx"); _chk_f32(weight, "weight")
    _chk_dev(grad, (idx, "idx"), (weight, "weight"))
    b, c_total, n = grad.shape
    out = torch.empty((b, int(c), int(m)), dtype=torch.float32, device=grad.device)
    with torch.cuda.device(grad.device):
        _L.check(_lib.pn2_three_interpolate_grad_from(b, int(c), n, int(m),
                                                      grad.data_ptr() + 4 * channel0 * n, c_total,
                                                      idx.data_ptr(), weight.data_ptr(),
                                                      out.data_ptr(), _stream(grad)),
                 "three_interpolate_grad_from")
    return out


def _ball_ws(t, b, n, m, nsample):
    """Private scratch per call, from torch's stream-aware caching allocator: ball queries run
    concurrently on the prefetch stream and on the main stream, so a shared buffer would race."""
    need = int(_lib.pn2_ball_query_workspace_bytes(b, n, m, nsample))
    if need <= 0:
        return None, None, 0
    buf = torch.empty(need, dtype=torch.uint8, device=t.device)
    return buf, buf.data_ptr(), need


def ball_query(new_xyz, xyz, radius, nsample):
    """new_xyz (B,m,3), xyz (B,N,3) -> idx (B,m,nsample) i32 (note: new_xyz FIRST).
    ball_query.cpp:13-37"""
    _chk_f32(new_xyz, "new_xyz"); _chk_f32(xyz, "xyz"); _chk_dev(new_xyz, (xyz, "xyz"))
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    nsample = int(nsample)
    if nsample <= 256 and b > 0 and m > 0:
        lists = _lists_for(xyz, radius)
        if lists is not None:
            return ball_query_prebuilt(new_xyz, xyz, radius, nsample, lists)
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=new_xyz.device)
    with torch.cuda.device(new_xyz.device):
        ws_buf, ws, ws_size = _ball_ws(new_xyz, b, n, m, nsample)
        _L.check(_lib.pn2_ball_query(b, n, m, float(radius), nsample, new_xyz.data_ptr(),
                                     xyz.data_ptr(), idx.data_ptr(), ws, ws_size,
                                     _stream(new_xyz)), "ball_query")
    return idx


def group_points(points, idx):
    """points (B,C,N), idx (B,m,ns) i32 -> (B,C,m,ns).  group_points.cpp:17-40"""
    _chk_f32(points, "points"); _chk_i32(idx, "idx"); _chk_dev(points, (idx, "idx"))
    b, c, n = points.shape
    _, m, ns = idx.shape
    out = torch.empty((b, c, m, ns), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        _L.check(_lib.pn2_group_points(b, c, n, m, ns, points.data_ptr(), idx.data_ptr(),
                                       out.data_ptr(), _stream(points)), "group_points")
    return out


def group_points_grad(grad_out, idx, n):
    """grad_out (B,C,m,ns), idx -> (B,C,n) scatter-add.  group_points.cpp:42-65"""
    _chk_f32(grad_out, "grad_out"); _chk_i32(idx, "idx"); _chk_dev(grad_out, (idx, "idx"))
    b, c, m, ns = grad_out.shape
    out = torch.empty((b, c, int(n)), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        _L.check(_lib.pn2_group_points_grad(b, c, int(n), m, ns, grad_out.data_ptr(),
                                            idx.data_ptr(), out.data_ptr(), _stream(grad_out)),
                 "group_points_grad")
    return out


# ---- additions (not in the reference's pybind surface) -----------------------------------
class CellLists(object):
    """Cell lists of one (B,N,3) cloud for ball queries of one radius (include/pn2_hip.h
    pn2_grid_*): built once, queried by ball_query / query_and_group via `grid=`."""

    def __init__(self, buf, b, n, radius, picks=None):
        self.buf, self.b, self.n, self.radius = buf, b, n, float(radius)
        self.event = None  # recorded on the building stream when the lists enter _ext's cache
        # the index tensor of the sampling call that left these lists behind (its kernel also left
        # a query plan per pick, include/pn2_hip.h pn2_query_and_group_picks), and the centroid
        # tensor a caller has declared to be the cloud gathered at exactly those indices
        self.picks = picks
        self._centroids = None

    def mark_centroids(self, new_xyz, inds):
        """Declare that new_xyz (B,m,3) is the cloud gathered at `inds`, the picks of the sampling
        call that produced these lists (what a set-abstraction layer computes,
        pointnet2_modules.py:236-245): query_and_group(new_xyz, ..., grid=self) may then start
        every query from the plan the sampling kernel left.  Any other tensor, or one modified in
        place afterwards, takes the ordinary path."""
        import weakref
        if self.picks is None or inds is not self.picks or new_xyz.shape[1] != inds.shape[1]:
            return
        self._centroids = (weakref.ref(new_xyz), _tensor_version(new_xyz), new_xyz.data_ptr())

    def centroids_are_picks(self, new_xyz):
        if self._centroids is None:
            return False
        ref, version, ptr = self._centroids
        return ref() is new_xyz and version is not None and _tensor_version(new_xyz) == version and \
            new_xyz.data_ptr() == ptr

    def check(self, xyz, radius):
        if tuple(xyz.shape[:2]) != (self.b, self.n) or float(radius) != self.radius:
            raise RuntimeError("cell lists were built for another cloud shape / radius")
        if xyz.device != self.buf.device:
            raise RuntimeError("cell lists live on %s" % self.buf.device)

    def launch_order(self):
        """(order_for (B,) int32, order (B,N) int32, cell_start (B, cells + 1) int32): the order in
        which the query kernels answer the centroids the sampling kernel picked -- longest query
        first -- valid for `order_for` centroids per cloud, 0 = none
        (include/pn2_hip.h pn2_grid_launch_order).  Views of the object's own memory."""
        import ctypes
        so, oo = ctypes.c_size_t(), ctypes.c_size_t()
        stride, slot = ctypes.c_int(), ctypes.c_int()
        _L.check(_lib.pn2_grid_launch_order(self.b, self.n, ctypes.byref(so), ctypes.byref(stride),
                                            ctypes.byref(slot), ctypes.byref(oo)), "grid_launch_order")
        start = self.buf[so.value:so.value + 4 * stride.value * self.b].view(torch.int32)
        start = start.view(self.b, stride.value)
        order = self.buf[oo.value:oo.value + 4 * self.n * self.b].view(torch.int32).view(self.b, self.n)
        return start[:, slot.value], order, start[:, :slot.value]


def grid_supported(b, n):
    return int(_lib.pn2_grid_bytes(int(b), int(n))) > 0


def build_grid(xyz, radius):
    """CellLists of xyz (B,N,3) for `radius` (the stand-alone two-kernel build)."""
    _chk_f32(xyz, "xyz")
    if not xyz.is_cuda:
        raise RuntimeError("CPU not supported")
    b, n, _ = xyz.shape
    need = int(_lib.pn2_grid_bytes(b, n))
    if need <= 0:
        raise RuntimeError("no cell lists for clouds of %d points" % n)
    buf = torch.empty(need, dtype=torch.uint8, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _L.check(_lib.pn2_grid_build(b, n, float(radius), xyz.data_ptr(), buf.data_ptr(), need,
                                     _stream(xyz)), "grid_build")
    return CellLists(buf, b, n, radius)


def furthest_point_sampling_with_grid(points, nsamples, radius):
    """furthest_point_sampling that also leaves the CellLists of `points` for `radius` behind
    (one kernel; indices identical).  Returns (inds (B,nsamples) i32, CellLists or None when the
    cloud size is outside the by-product's range)."""
    _chk_f32(points, "points")
    if not points.is_cuda:
        raise RuntimeError("CPU not supported")
    b, n, _ = points.shape
    nsamples = int(nsamples)
    if not _lib.pn2_fps_grid_supported(n):
        return furthest_point_sampling(points, nsamples), None
    out = torch.empty((b, nsamples), dtype=torch.int32, device=points.device)  # every index is written
    with torch.cuda.device(points.device):
        need = int(_lib.pn2_fps_workspace_bytes(b, n, nsamples))
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=points.device)
        gbytes = int(_lib.pn2_grid_bytes(b, n))
        gbuf = torch.empty(gbytes, dtype=torch.uint8, device=points.device)
        _L.check(_lib.pn2_furthest_point_sampling_grid(b, n, nsamples, points.data_ptr(),
                                                       out.data_ptr(), ws.data_ptr(), need,
                                                       float(radius), gbuf.data_ptr(), gbytes,
                                                       _stream(points)),
                 "furthest_point_sampling_grid")
    return out, CellLists(gbuf, b, n, radius, picks=out)


def furthest_point_sampling_ties(points, nsamples, radius=None):
    """furthest_point_sampling that also reports, per cloud, the first round in which two points
    were equally far or nothing was left to take (nsamples if none): up to that round every pick
    was a STRICT maximum, and a new point.
    Returns (inds (B,nsamples) i32, CellLists or None, first_tie (B,) i32 or None); first_tie is
    None when the cloud size is outside the bucketed tier (then: a plain sampling).  radius: also
    leave the cell lists behind, as furthest_point_sampling_with_grid."""
    _chk_f32(points, "points")
    if not points.is_cuda:
        raise RuntimeError("CPU not supported")
    b, n, _ = points.shape
    nsamples = int(nsamples)
    if not _lib.pn2_fps_ties_supported(n):
        if radius is None:
            return furthest_point_sampling(points, nsamples), None, None
        inds, lists = furthest_point_sampling_with_grid(points, nsamples, radius)
        return inds, lists, None
    with_lists = radius is not None and bool(_lib.pn2_fps_grid_supported(n))
    out = torch.empty((b, nsamples), dtype=torch.int32, device=points.device)
    ties = torch.zeros((b,), dtype=torch.int32, device=points.device)  # (defined for nsamples == 0 too)
    with torch.cuda.device(points.device):
        need = int(_lib.pn2_fps_workspace_bytes(b, n, nsamples))
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=points.device)
        gbytes = int(_lib.pn2_grid_bytes(b, n)) if with_lists else 0
        gbuf = torch.empty(gbytes, dtype=torch.uint8, device=points.device) if with_lists else None
        _L.check(_lib.pn2_furthest_point_sampling_ties(
            b, n, nsamples, points.data_ptr(), out.data_ptr(), ws.data_ptr(), need,
            float(radius) if with_lists else 0.0, gbuf.data_ptr() if with_lists else None, gbytes,
            ties.data_ptr(), _stream(points)), "furthest_point_sampling_ties")
    return out, (CellLists(gbuf, b, n, radius, picks=out) if with_lists else None), ties


def furthest_point_sampling_prefix(points, nsamples, first_tie):
    """furthest_point_sampling(points, nsamples) for a cloud that is the HEAD -- the first N picks,
    in order -- of a sampling sequence whose ties were recorded by furthest_point_sampling_ties
    (or the head of such a head): clouds with first_tie >= N get 0..nsamples-1 without a single
    round (include/pn2_hip.h explains why that is the reference's answer), the others are sampled
    as usual.  first_tie None = furthest_point_sampling."""
    if first_tie is None:
        return furthest_point_sampling(points, nsamples)
    _chk_f32(points, "points")
    if not points.is_cuda:
        raise RuntimeError("CPU not supported")
    b, n, _ = points.shape
    nsamples = int(nsamples)
    if first_tie.dtype != torch.int32 or first_tie.numel() != b or first_tie.device != points.device:
        raise ValueError("first_tie must be (B,) int32 on the device of points")
    out = torch.empty((b, nsamples), dtype=torch.int32, device=points.device)
    with torch.cuda.device(points.device):
        need = int(_lib.pn2_fps_workspace_bytes(b, n, nsamples))
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=points.device)
        _L.check(_lib.pn2_furthest_point_sampling_prefix(b, n, nsamples, points.data_ptr(),
                                                         out.data_ptr(), ws.data_ptr(), need,
                                                         first_tie.contiguous().data_ptr(),
                                                         _stream(points)),
                 "furthest_point_sampling_prefix")
    return out


def ball_query_prebuilt(new_xyz, xyz, radius, nsample, grid):
    """ball_query on CellLists built earlier for (xyz, radius)."""
    _chk_f32(new_xyz, "new_xyz"); _chk_f32(xyz, "xyz"); _chk_dev(new_xyz, (xyz, "xyz"))
    grid.check(xyz, radius)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = torch.empty((b, m, int(nsample)), dtype=torch.int32, device=new_xyz.device)
    with torch.cuda.device(new_xyz.device):
        _L.check(_lib.pn2_ball_query_prebuilt(b, n, m, float(radius), int(nsample),
                                              new_xyz.data_ptr(), xyz.data_ptr(), idx.data_ptr(),
                                              grid.buf.data_ptr(), grid.buf.numel(),
                                              _stream(new_xyz)), "ball_query_prebuilt")
    return idx


def query_and_group(new_xyz, xyz, features, radius, nsample, normalize_xyz, idx=None, grid=None):
    """Fused QueryAndGroup front end (pointnet2_utils.py:335-358): returns
    (idx (B,m,ns) i32, grouped (B,3+C,m,ns) f32) with channels 0..2 = relative xyz
    (optionally / radius) and 3.. = gathered features (features may be None).
    A ball-query result computed earlier may be passed as `idx` (then only the gathers run);
    CellLists built earlier for (xyz, radius) as `grid` (then one kernel does everything)."""
    _chk_f32(new_xyz, "new_xyz"); _chk_f32(xyz, "xyz"); _chk_dev(new_xyz, (xyz, "xyz"))
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    c = 0
    fptr = None
    if features is not None:
        _chk_f32(features, "features"); _chk_dev(new_xyz, (features, "features"))
        c = features.shape[1]
        fptr = features.data_ptr()
    nsample = int(nsample)
    out = torch.empty((b, 3 + c, m, nsample), dtype=torch.float32, device=new_xyz.device)
    if idx is not None:
        _chk_i32(idx, "idx"); _chk_dev(new_xyz, (idx, "idx"))
        if tuple(idx.shape) != (b, m, nsample):
            raise RuntimeError("idx must have shape (B, npoint, nsample)")
        with torch.cuda.device(new_xyz.device):
            _L.check(_lib.pn2_group_concat(b, n, m, c, float(radius), nsample,
                                           1 if normalize_xyz else 0, new_xyz.data_ptr(),
                                           xyz.data_ptr(), fptr, idx.data_ptr(), out.data_ptr(),
                                           _stream(new_xyz)), "group_concat")
        return idx, out
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=new_xyz.device)
    if grid is not None and nsample <= 256:
        grid.check(xyz, radius)
        with torch.cuda.device(new_xyz.device):
            # (the layer's own centroids: every query starts from the plan its sampling kernel left)
            entry = _lib.pn2_query_and_group_picks if grid.centroids_are_picks(new_xyz) \
                else _lib.pn2_query_and_group_prebuilt
            _L.check(entry(b, n, m, c, float(radius), nsample, 1 if normalize_xyz else 0,
                           new_xyz.data_ptr(), xyz.data_ptr(), fptr, idx.data_ptr(), out.data_ptr(),
                           grid.buf.data_ptr(), grid.buf.numel(), _stream(new_xyz)),
                     "query_and_group_prebuilt")
        return idx, out
    with torch.cuda.device(new_xyz.device):
        ws_buf, ws, ws_size = _ball_ws(new_xyz, b, n, m, nsample)
        _L.check(_lib.pn2_query_and_group(b, n, m, c, float(radius), nsample,
                                          1 if normalize_xyz else 0, new_xyz.data_ptr(),
                                          xyz.data_ptr(), fptr, idx.data_ptr(), out.data_ptr(),
                                          ws, ws_size, _stream(new_xyz)), "query_and_group")
    return idx, out
