"""ctypes binding of oracle/liboracle.so (raster_oracle.c + knn_oracle.c).  TEST INFRASTRUCTURE ONLY.

numpy in / numpy out; builds the library on first use with oracle/Makefile (gcc).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OracleArgs(C.Structure):
    _fields_ = [("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("W", C.c_int), ("H", C.c_int),
                ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
                ("bg", C.c_float * 3), ("viewmatrix", C.c_float * 16), ("projmatrix", C.c_float * 16),
                ("campos", C.c_float * 3), ("prefiltered", C.c_int)]


def build(force=False):
    so = os.path.join(_DIR, "liboracle.so")
    srcs = [os.path.join(_DIR, f) for f in ("raster_oracle.c", "knn_oracle.c", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _DIR, "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        fp = C.POINTER(C.c_float)
        L.oracle_raster_forward.restype = C.c_void_p
        L.oracle_raster_forward.argtypes = [C.POINTER(OracleArgs)] + [C.c_void_p] * 7 + [C.c_void_p, C.c_void_p]
        L.oracle_raster_backward.restype = None
        L.oracle_raster_backward.argtypes = [C.c_void_p] + [C.c_void_p] * 10
        L.oracle_raster_free.argtypes = [C.c_void_p]
        L.oracle_raster_free.restype = None
        L.oracle_num_rendered.argtypes = [C.c_void_p]
        L.oracle_num_rendered.restype = C.c_int64
        for name, rt in [("depths", fp), ("xy", fp), ("conic_opacity", fp), ("rgb", fp), ("cov3D", fp),
                         ("clamped", C.POINTER(C.c_uint8)), ("tiles_touched", C.POINTER(C.c_uint32)),
                         ("point_list", C.POINTER(C.c_uint32)), ("ranges", C.POINTER(C.c_uint32)),
                         ("final_T", fp), ("n_contrib", C.POINTER(C.c_uint32))]:
            f = getattr(L, "oracle_" + name)
            f.argtypes = [C.c_void_p]
            f.restype = rt
        L.oracle_mark_visible.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_mark_visible.restype = None
        L.oracle_knn3_mean_dist2.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_knn3_mean_dist2.restype = None
        L.oracle_max_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class RasterOracle:
    """One forward (+ optional backward) of the CPU oracle.  Mirrors the argument meaning of
    GaussianRasterizer.forward; matrices in the reference's row-vector (transposed) layout."""

    def __init__(self, *, image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix, projmatrix,
                 sh_degree, campos, prefiltered=False, debug=False):
        self.H, self.W = int(image_height), int(image_width)
        self.tanfovx, self.tanfovy = float(tanfovx), float(tanfovy)
        self.bg = _f32(bg).reshape(3)
        self.scale_modifier = float(scale_modifier)
        self.viewmatrix = _f32(viewmatrix).reshape(16)
        self.projmatrix = _f32(projmatrix).reshape(16)
        self.sh_degree = int(sh_degree)
        self.campos = _f32(campos).reshape(3)
        self.prefiltered = bool(prefiltered)
        self._state = None
        self._keep = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def free(self):
        if self._state is not None:
            lib().oracle_raster_free(self._state)
            self._state = None

    def forward(self, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        self.free()
        means3D = _f32(means3D).reshape(-1, 3)
        P = means3D.shape[0]
        shs, colors_precomp = _f32(shs), _f32(colors_precomp)
        scales, rotations, cov3D_precomp = _f32(scales), _f32(rotations), _f32(cov3D_precomp)
        opacities = _f32(opacities).reshape(-1)
        M = 0 if shs is None else shs.reshape(P, -1, 3).shape[1]
        a = OracleArgs()
        a.P, a.D, a.M, a.W, a.H = P, self.sh_degree, M, self.W, self.H
        a.tanfovx, a.tanfovy, a.scale_modifier = self.tanfovx, self.tanfovy, self.scale_modifier
        a.bg[:] = self.bg.tolist()
        a.viewmatrix[:] = self.viewmatrix.tolist()
        a.projmatrix[:] = self.projmatrix.tolist()
        a.campos[:] = self.campos.tolist()
        a.prefiltered = int(self.prefiltered)
        color = np.zeros((3, self.H, self.W), np.float32)
        radii = np.zeros((P,), np.int32)
        self._keep = (means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp)
        self._state = lib().oracle_raster_forward(C.byref(a), _ptr(means3D), _ptr(shs), _ptr(colors_precomp),
                                                  _ptr(opacities), _ptr(scales), _ptr(rotations), _ptr(cov3D_precomp),
                                                  _ptr(color), _ptr(radii))
        self.P, self.M = P, M
        return color, radii

    @property
    def num_rendered(self):
        return int(lib().oracle_num_rendered(self._state))

    def saved(self):
        """Intermediate buffers (copies) for stage-by-stage parity checks."""
        L, s, P = lib(), self._state, self.P
        I = self.num_rendered
        tiles = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        npix = self.W * self.H

        def arr(fn, n, shape):
            if n == 0:
                return np.zeros(shape, dtype=np.ctypeslib.as_array(fn(s), (1,)).dtype)
            return np.ctypeslib.as_array(fn(s), (n,)).reshape(shape).copy()

        return dict(depths=arr(L.oracle_depths, P, (P,)), xy=arr(L.oracle_xy, 2 * P, (P, 2)),
                    conic_opacity=arr(L.oracle_conic_opacity, 4 * P, (P, 4)), rgb=arr(L.oracle_rgb, 3 * P, (P, 3)),
                    cov3D=arr(L.oracle_cov3D, 6 * P, (P, 6)), clamped=arr(L.oracle_clamped, 3 * P, (P, 3)),
                    tiles_touched=arr(L.oracle_tiles_touched, P, (P,)), point_list=arr(L.oracle_point_list, I, (I,)),
                    ranges=arr(L.oracle_ranges, 2 * tiles, (tiles, 2)), final_T=arr(L.oracle_final_T, npix, (self.H, self.W)),
                    n_contrib=arr(L.oracle_n_contrib, npix, (self.H, self.W)), num_rendered=I)

    def backward(self, dL_dpix):
        P, M = self.P, max(self.M, 1)
        dL_dpix = _f32(dL_dpix).reshape(3, self.H, self.W)
        g = dict(means2D=np.zeros((P, 3), np.float32), conic=np.zeros((P, 4), np.float32),
                 opacities=np.zeros((P, 1), np.float32), colors=np.zeros((P, 3), np.float32),
                 means3D=np.zeros((P, 3), np.float32), cov3D=np.zeros((P, 6), np.float32),
                 shs=np.zeros((P, M, 3), np.float32), scales=np.zeros((P, 3), np.float32),
                 rotations=np.zeros((P, 4), np.float32))
        lib().oracle_raster_backward(self._state, _ptr(dL_dpix), _ptr(g["means2D"]), _ptr(g["conic"]), _ptr(g["opacities"]),
                                     _ptr(g["colors"]), _ptr(g["means3D"]), _ptr(g["cov3D"]), _ptr(g["shs"]),
                                     _ptr(g["scales"]), _ptr(g["rotations"]))
        if self.M == 0:
            g["shs"] = np.zeros((P, 0, 3), np.float32)
        return g


def mark_visible(means3D, viewmatrix, projmatrix):
    means3D = _f32(means3D).reshape(-1, 3)
    out = np.zeros((means3D.shape[0],), np.uint8)
    v, p = _f32(viewmatrix).reshape(16), _f32(projmatrix).reshape(16)
    lib().oracle_mark_visible(means3D.shape[0], _ptr(means3D), _ptr(v), _ptr(p), _ptr(out))
    return out.astype(bool)


def knn3_mean_dist2(points):
    pts = _f32(points).reshape(-1, 3)
    out = np.zeros((pts.shape[0],), np.float32)
    lib().oracle_knn3_mean_dist2(pts.shape[0], _ptr(pts), _ptr(out))
    return out


def max_threads():
    return int(lib().oracle_max_threads())
