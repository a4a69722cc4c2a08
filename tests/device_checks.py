"""Loader of tests/_build/libdbw_device_checks.so (tests/device_checks.hip): the product's shared host+device arithmetic evaluated on the
GPU on caller-supplied operands.  Checker side: built by build.py's build_device_checks (also by __graft_entry__.build()), here on demand."""
import ctypes
import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        spec = importlib.util.spec_from_file_location('_dbw_build', os.path.join(HERE, '..', 'differentiable-blocksworld_amd', 'build.py'))
        b = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b)
        _LIB = ctypes.CDLL(b.build_device_checks())
        p, i, f, ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong
        _LIB.dbwt_divcheck.argtypes = [p, p, ll, p, p]
        _LIB.dbwt_lane_merge.argtypes = [p, p, p, i, i, p, p, p]
        _LIB.dbwt_model_math.argtypes = [i, p, p, p, i, f, p, p]
        for fn in (_LIB.dbwt_divcheck, _LIB.dbwt_lane_merge, _LIB.dbwt_model_math):
            fn.restype = i
    return _LIB


def call(name, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise RuntimeError(f'{name} failed ({rc})')
