"""The deterministic math layer must track libm to a few ULP (it replaces libm/OpenCV
transcendentals on both sides of the parity comparison)."""
import ctypes

import numpy as np

from oracle.oracle import detmath_lib


def _un(lib, which, x):
    x = np.ascontiguousarray(x, np.float64)
    y = np.empty_like(x)
    lib.lfo_vec_unary(which, x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), x.size)
    return y


def _bi(lib, which, a, b):
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    y = np.empty_like(a)
    lib.lfo_vec_binary(which, a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
                       y.ctypes.data_as(ctypes.c_void_p), a.size)
    return y


def _ulp(y, ref):
    return np.max(np.abs(y - ref) / np.spacing(np.abs(ref)))


def test_detmath_vs_libm():
    lib = detmath_lib()
    rng = np.random.default_rng(0)
    x = rng.uniform(-700, 700, 100000)
    assert _ulp(_un(lib, 0, x), np.exp(x)) <= 2
    x = np.exp(rng.uniform(-700, 700, 100000))
    assert _ulp(_un(lib, 1, x), np.log(x)) <= 2
    x = rng.uniform(-20, 20, 200000)
    assert np.max(np.abs(_un(lib, 2, x) - np.sin(x))) <= 2.3e-16
    assert np.max(np.abs(_un(lib, 3, x) - np.cos(x))) <= 2.3e-16
    x = rng.uniform(-50, 50, 100000)
    assert _ulp(_un(lib, 4, x), np.arctan(x)) <= 2
    x = rng.uniform(-1, 1, 100000)
    assert _ulp(_un(lib, 5, x), np.arcsin(x)) <= 4
    a, b = rng.uniform(-5, 5, 100000), rng.uniform(-5, 5, 100000)
    assert _ulp(_bi(lib, 0, a, b), np.arctan2(a, b)) <= 2
    x = rng.uniform(1e-3, 1e5, 100000)
    assert _ulp(_un(lib, 6, x), np.log10(x)) <= 3
    x = rng.uniform(-0.1, 0.1, 100000)
    assert _ulp(_un(lib, 7, x), np.sinh(x)) <= 2
    a, b = rng.uniform(0.01, 30, 100000), rng.uniform(-5, 5, 100000)
    assert np.max(np.abs(_bi(lib, 1, a, b) / np.power(a, b) - 1)) < 1e-14


def test_special_values():
    lib = detmath_lib()
    assert _un(lib, 0, [0.0])[0] == 1.0
    assert _un(lib, 1, [1.0])[0] == 0.0
    assert _un(lib, 2, [0.0])[0] == 0.0 and _un(lib, 3, [0.0])[0] == 1.0
    assert _bi(lib, 0, [0.0, 1.0, -1.0], [1.0, 0.0, 0.0]).tolist() == [0.0, np.pi / 2, -np.pi / 2]
    assert np.isnan(_un(lib, 5, [1.5])[0])


def test_fast_atan2_is_opencv_polynomial():
    """fastAtan2 (OpenCV 3.x) is accurate to ~0.3 degrees and covers [0, 360)."""
    lib = detmath_lib()
    rng = np.random.default_rng(1)
    y = rng.uniform(-10, 10, 50000).astype(np.float32)
    x = rng.uniform(-10, 10, 50000).astype(np.float32)
    out = np.empty_like(x)
    lib.lfo_vec_fast_atan2(y.ctypes.data_as(ctypes.c_void_p), x.ctypes.data_as(ctypes.c_void_p),
                           out.ctypes.data_as(ctypes.c_void_p), x.size)
    ref = np.degrees(np.arctan2(y.astype(np.float64), x.astype(np.float64))) % 360.0
    err = np.abs(((out - ref) + 180) % 360 - 180)
    assert err.max() < 0.35
    assert out.min() >= 0 and out.max() <= 360
