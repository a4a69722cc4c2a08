"""ctypes binding of libdbw_hip.so (C ABI declared in include/dbw_hip.h).

The library is the product: there is NO CPU fallback.  If it cannot be loaded, or a call fails, a RuntimeError is
raised (never a silent eager/PyTorch path)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libdbw_hip.so')
_lib = None


def _header_abi_version():
    """DBW_ABI_VERSION of include/dbw_hip.h: the one place the revision is written down (csrc/util.hip returns it, tests/test_abi.py and
    __graft_entry__.build() compare the loaded library with it)."""
    import re
    with open(os.path.join(_HERE, '..', '..', 'include', 'dbw_hip.h')) as f:
        return int(re.search(r'#define DBW_ABI_VERSION (\d+)', f.read()).group(1))


ABI_VERSION = _header_abi_version()

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_f = ctypes.c_float
c_i64 = ctypes.c_int64
c_d = ctypes.c_double
c_sz = ctypes.c_size_t



class TextureSet(ctypes.Structure):
    """dbw_texture_set of include/dbw_hip.h (one texture tensor of a multi-set launch)."""
    _fields_ = [('texture', c_p), ('n', c_i), ('h', c_i), ('w', c_i), ('decim', c_i), ('maps', c_p), ('sig', c_p), ('wrap_x', c_i),
                ('tv_scale', c_f), ('grad_sig_out', c_p), ('grad_maps', c_p), ('grad_sig', c_p), ('grad_texture', c_p)]


def texture_sets(sets):
    """list of dicts (missing fields = 0 / NULL) -> (ctypes array, count)"""
    arr = (TextureSet * len(sets))()
    for a, d in zip(arr, sets):
        for k, v in d.items():
            setattr(a, k, v)
    return arr, len(sets)


# name -> argtypes, exactly the prototypes of include/dbw_hip.h (checked by tests/test_abi.py)
SIGNATURES = {
    'dbw_project_clip_fwd': [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_i, c_f, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    'dbw_project_clip_bwd': [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    'dbw_rasterize_fwd': [c_p, c_p, c_p, c_p, c_i, c_i64, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_sz, c_p],
    'dbw_rasterize_bwd': [c_p, c_p, c_p, c_p, c_p, c_i, c_i64, c_i, c_i, c_i, c_i, c_i, c_p, c_p],
    'dbw_shade_blend_fwd': [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p],
    'dbw_shade_blend_bwd': [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p,
                            c_p, c_p, c_p, c_p, c_p, c_i, c_p],
    'dbw_render_fwd_fused': [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i64, c_i, c_i, c_i, c_i, c_f, c_f,
                             c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_sz, c_i, c_i, c_i, c_p],
    'dbw_render_fwd_fused_mse': [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i64, c_i, c_i, c_i, c_i, c_f, c_f,
                                 c_i, c_p, c_p, c_p, c_p, c_p, c_sz, c_p, c_p, c_f, c_p, c_p, c_p, c_i, c_i, c_p],
    'dbw_render_bwd_fused': [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p,
                             c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_i, c_p],
    'dbw_texbin_reduce': [c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_p],
    'dbw_bin_layout': [c_p, c_i64, c_d, c_i, c_p, c_p],
    'dbw_texture_prep_fwd': [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p],
    'dbw_texture_prep_bwd': [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p],
    'dbw_sq_blocks_fwd': [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_f, c_p, c_p, c_p, c_p],
    'dbw_sq_blocks_bwd': [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    'dbw_posed_mesh_fwd': [c_p, c_i, c_p, c_p, c_f, c_p, c_p, c_p, c_p],
    'dbw_posed_mesh_bwd': [c_p, c_i, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p],
    'dbw_composite_mse': [c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_p],
    'dbw_tv_l2sq': [c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p],
    'dbw_overlap_loss': [c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_f, c_f, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    'dbw_block_alpha_fwd': [c_p, c_p, c_f, c_f, c_i, c_p, c_p, c_p, c_p],
    'dbw_block_alpha_bwd': [c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_p],
    'dbw_sqrt_mean': [c_p, c_i, c_f, c_f, c_p, c_p, c_p],
    'dbw_adam_step': [c_p, c_p, c_p, c_p, c_i64, c_f, c_f, c_f, c_f, c_i, c_p],
    'dbw_debug_divcheck': [c_p, c_p, c_i64, c_p, c_p],
    'dbw_debug_model_math': [c_i, c_p, c_p, c_p, c_i, c_f, c_p, c_p],
    'dbw_debug_lane_merge': [c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p],
    'dbw_texture_prep_fwd_sets': [c_p, c_i, c_p],
    'dbw_texture_prep_bwd_sets': [c_p, c_i, c_p],
    'dbw_tv_l2sq_sets': [c_p, c_i, c_p, c_p],
    'dbw_adam_step_groups': [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_f, c_f, c_i, c_p, c_i64, c_p],
}


def load():
    """Load (building in-tree with hipcc if the .so is absent or stale and hipcc exists)."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location('_dbw_build', os.path.join(_HERE, '..', 'build.py'))
        b = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b)
        if b.needs_build() and os.path.exists(os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')):
            b.build()
    except Exception as e:                      # a stale-but-present library is still usable; a missing one is fatal below
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'libdbw_hip.so is missing and could not be built: {e}') from e
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f'{LIB_PATH} not found: run `python differentiable-blocksworld_amd/build.py` '
                           '(there is no CPU fallback for the render path)')
    # DBW_HIP_LIB: load an alternative build of the same sources (tools/variants.sh, kernel tuning sweeps)
    lib = ctypes.CDLL(os.environ.get('DBW_HIP_LIB') or LIB_PATH)
    lib.dbw_last_error.restype = ctypes.c_char_p
    lib.dbw_abi_version.restype = c_i
    if not os.environ.get('DBW_HIP_LIB') and lib.dbw_abi_version() != ABI_VERSION:
        raise RuntimeError(f'{LIB_PATH} was built for ABI {lib.dbw_abi_version()}, include/dbw_hip.h declares {ABI_VERSION}: rebuild '
                           '(python differentiable-blocksworld_amd/build.py --force)')
    if hasattr(lib, 'dbw_bin_subcursors'):      # (absent from tuning builds of older sources, tools/variants.sh: 16 there)
        lib.dbw_bin_subcursors.restype = c_i
    lib.dbw_rasterize_workspace_bytes.restype = c_sz
    lib.dbw_rasterize_workspace_bytes.argtypes = [c_i64]
    lib.dbw_rasterize_workspace_bytes_binned.restype = c_sz
    lib.dbw_rasterize_workspace_bytes_binned.argtypes = [c_i64, c_i, c_i, c_i]
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_i
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f'{name} failed (rc={rc}): {lib.dbw_last_error().decode()}')
