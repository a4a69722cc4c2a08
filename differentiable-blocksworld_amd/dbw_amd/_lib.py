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


class StepDesc(ctypes.Structure):
    """dbw_step_desc of include/dbw_hip.h (field order = the header's; tests/test_abi.py compares the two)."""
    _fields_ = ([(n, c_i) for n in ('H', 'W', 'faces_per_pixel', 'max_views', 'n_blocks', 'block_nv', 'block_nf', 'n_sky_verts', 'n_ground_verts',
                                    'n_sky_faces', 'n_ground_faces', 'txt_size', 'env_txt_size', 'decim_env', 'decim_blocks', 'coarse')]
                + [(n, c_f) for n in ('sigma', 'blur_radius', 'z_clip', 'cam_eps')] + [('perspective_correct', c_i)]
                + [('bg_fg', c_f * 3), ('bg_env', c_f * 3)]
                + [(n, c_f) for n in ('S_world', 'ratio_block_scene', 'scale_min', 'opacity_noise', 'mask_threshold', 'w_rgb', 'w_parsimony', 'w_tv_bkg',
                                      'w_tv_blocks', 'w_tv_ground', 'w_overlap')]
                + [('overlap_points', c_i), ('overlap_temperature', c_f), ('overlap_n_blocks', c_f)]
                + [(n, c_p) for n in ('R_world', 'T_world', 'Kmat', 'ground_base', 'env_verts', 'env_faces', 'env_face_uvs', 'env_face_map', 'env_map_desc',
                                      'trig', 'block_faces', 'block_face_uvs', 'block_face_map', 'block_map_desc', 'block_bin_base', 'block_bin_info')]
                + [('n_bins', c_i)]
                + [(n, c_p) for n in ('sq_eps', 'S', 'R6', 'T', 'alpha_logit', 'R6_ground', 'T_ground', 'texture_bkg', 'texture_ground', 'textures',
                                      'g_sq_eps', 'g_S', 'g_R6', 'g_T', 'g_alpha_logit', 'g_R6_ground', 'g_T_ground', 'g_texture_bkg', 'g_texture_ground',
                                      'g_textures', 'flat_param', 'flat_grad', 'exp_avg', 'exp_avg_sq')]
                + [('group_end', c_i64 * 2), ('small_grads', c_p), ('n_small_grads', c_i), ('fuse', c_i), ('backward_order', c_i),
                   ('binned_concurrent', c_i), ('serial_setup_max_views', c_i), ('seed', ctypes.c_uint64), ('sync_events', c_i), ('tv_value_scale', ctypes.c_float)])


class StepInputs(ctypes.Structure):
    """dbw_step_inputs of include/dbw_hip.h."""
    _fields_ = [('imgs', c_p), ('imgs_tiled', c_i), ('R', c_p), ('T', c_p), ('B', c_i), ('global_count', c_d), ('noise_override', c_p),
                ('overlap_u_override', c_p), ('with_adam', c_i), ('adam_step', c_i), ('lr', c_f * 2), ('beta1', c_f), ('beta2', c_f), ('adam_eps', c_f),
                ('read_losses', c_i), ('phase', c_i), ('rec_out', c_p), ('grad_rec', c_p), ('single_stream', c_i), ('arena_is_clean', c_i), ('rng_step', ctypes.c_uint64),
                ('defer_textures', c_i)]


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
    'dbw_texture_prep_fwd_sets': [c_p, c_i, c_p],
    'dbw_texture_prep_bwd_sets': [c_p, c_i, c_p],
    'dbw_tv_l2sq_sets': [c_p, c_i, c_p, c_p],
    'dbw_adam_step_groups': [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_f, c_f, c_i, c_p, c_i64, c_p, c_p],
    'dbw_lpips_head_fwd': [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p],
    'dbw_lpips_head_bwd': [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p],
    'dbw_debug_train_step_last_timeout': [c_p, c_p],
    'dbw_debug_train_step_counters': [c_p, c_p, c_p],
    'dbw_bias_relu': [c_p, c_p, c_i, c_i, c_i, c_p, c_p],
    'dbw_maxpool2_fwd': [c_p, c_i, c_i, c_i, c_p, c_p],
    'dbw_maxpool2_bwd': [c_p, c_p, c_i, c_i, c_i, c_p, c_p],
    'dbw_train_step_run': [c_p, c_p, c_p, c_p],
    'dbw_train_step_finish': [c_p, c_p, c_p],
    'dbw_train_step_losses': [c_p, c_p],
    'dbw_train_step_wait_blocks_ready': [c_p, c_p],
    'dbw_train_step_sync_timeouts': [c_p],
    'dbw_debug_train_step_force_timeout': [c_p],
    'dbw_debug_train_step_hasty_prologue_wait': [c_p],
    'dbw_train_step_profile': [c_p, c_i],
    'dbw_train_step_kernel_times': [c_p, c_p],
}
# entry points that do not return an error code: name -> (restype, argtypes)
OTHER_SIGNATURES = {
    'dbw_train_step_workspace_bytes': (c_sz, [c_p]),
    'dbw_train_step_create': (c_p, [c_p, c_p, c_sz]),
    'dbw_train_step_destroy': (None, [c_p]),
    'dbw_train_step_offset': (c_i64, [c_p, c_i]),
    'dbw_train_step_void_flag_offset': (c_i64, [c_p]),
    'dbw_train_step_voided_runs': (c_i, [c_p]),
    'dbw_lpips_head_blocks': (c_i, [c_i, c_i]),
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
        if name.startswith('dbw_train_step') and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_i
    for name, (restype, argtypes) in OTHER_SIGNATURES.items():
        if hasattr(lib, name):                  # (absent from tuning builds of older sources)
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = argtypes, restype
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f'{name} failed (rc={rc}): {lib.dbw_last_error().decode()}')
