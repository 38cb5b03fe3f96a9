"""ctypes binding of libhnrf.so (the C ABI declared in include/hnrf.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C humannerf_amd/csrc``.  There is NO fallback: if the shared object is
missing or a symbol cannot be resolved, importing/using the ops raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('HNRF_LIB_PATH', os.path.join(_HERE, 'libhnrf.so'))   # override: diagnostic builds only

_vp, _i64, _int, _sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/hnrf.h one to one
SIGNATURES = {
    'hnrf_abi_version': (_int, []),
    'hnrf_last_error': (ctypes.c_char_p, []),
    'hnrf_sample_warp_fwd': (_int, [_vp] * 10 + [_i64, _int, _int, _int] + [_vp] * 4 + [_vp]),
    'hnrf_nonrigid_packed_bytes': (_sz, [_int]),
    'hnrf_canonical_status_offset': (_sz, [_int]),
    'hnrf_nonrigid_status_offset': (_sz, [_int]),
    'hnrf_nonrigid_pack': (_int, [_vp, _vp, _vp, _int, _vp, _vp]),
    'hnrf_nonrigid_fwd': (_int, [_vp, _vp, _vp, _int, _i64, _vp, _vp, _vp]),
    'hnrf_canonical_packed_bytes': (_sz, [_int]),
    'hnrf_canonical_pack': (_int, [_vp, _vp, _int, _vp, _vp]),
    'hnrf_canonical_fwd': (_int, [_vp, _vp, _int, _i64, _vp, _vp]),
    'hnrf_composite_fwd': (_int, [_vp] * 6 + [_i64, _int, ctypes.c_float] + [_vp] * 8 + [_vp]),
    'hnrf_compact_samples': (_int, [_vp, ctypes.c_float, _i64, _vp, _vp, _vp]),
    'hnrf_canonical_fwd_sparse': (_int, [_vp, _vp, _int, _i64, _vp, _vp, _vp, _vp]),
    'hnrf_nonrigid_fwd_sparse': (_int, [_vp, _vp, _vp, _int, _i64, _vp, _vp, _vp, _vp, _vp]),
    'hnrf_canonical_fwd_train': (_int, [_vp, _vp, _int, _i64, _vp, _vp, _vp, _vp, _vp]),
    'hnrf_nonrigid_fwd_train': (_int, [_vp, _vp, _vp, _int, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    'hnrf_composite_bwd': (_int, [_vp] * 8 + [_i64, _int, _vp, _vp, _vp]),
    'hnrf_pe_bwd': (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _vp, _vp]),
    'hnrf_sample_warp_bwd': (_int, [_vp] * 12 + [_i64, _int, _int, _int, _vp, _vp, _vp, _vp]),
    'hnrf_mlp_dw_workspace_bytes': (_sz, [_i64, _int, _int]),
    'hnrf_mlp_dw': (_int, [_vp, _i64, _vp, _i64, _i64, _int, _int, _int, _vp, _int, _vp, _i64, _vp, _vp, _sz, _vp]),
    'hnrf_motion_basis_saved_bytes': (_sz, []),
    'hnrf_motion_basis_fwd': (_int, [_vp, _vp, _vp, _int, _vp, _vp, _vp, _vp]),
    'hnrf_motion_basis_bwd': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp]),
    'hnrf_refined_motion_basis_fwd': (_int, [_vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp]),
    'hnrf_refined_motion_basis_bwd': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp]),
    'hnrf_pose_mlp_saved_bytes': (_sz, [_int]),
    'hnrf_pose_mlp_fwd': (_int, [_vp, _vp, _vp, _vp, _int, _vp, _vp, _vp]),
    'hnrf_pose_mlp_bwd': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp]),
    'hnrf_mlp_dw_h_workspace_bytes': (_sz, [_i64, _int, _int]),
    'hnrf_mlp_dw_h': (_int, [_vp, _i64, _vp, _i64, _i64, _int, _int, _int, _vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    'hnrf_canonical_bwd_packed_bytes': (_sz, [_int]),
    'hnrf_nonrigid_bwd_packed_bytes': (_sz, [_int]),
    'hnrf_canonical_bwd_pack': (_int, [_vp, _int, _vp, _vp]),
    'hnrf_nonrigid_bwd_pack': (_int, [_vp, _int, _vp, _vp]),
    'hnrf_canonical_bwd': (_int, [_vp, _vp, _vp, _vp, _int, _vp, _i64, _vp, _vp, _vp, _vp]),
    'hnrf_nonrigid_bwd': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _vp, _i64, _vp, _vp, _vp, _vp]),
    'hnrf_gen_rays_workspace_bytes': (_sz, [_int, _int]),
    'hnrf_gen_rays': (_int, [_vp] * 5 + [_int, _int] + [_vp] * 7 + [_sz, _vp]),
    'hnrf_undistort_image': (_int, [_vp, _int, _int, _int, _vp, _vp, _int, _int, _vp, _vp]),
    'hnrf_composite_windows': (_int, [_vp, _vp, _int, _int, _vp, _int, _vp, _vp, _vp, _vp, _int, _int, _vp, _int, _int, _int, _vp, _vp]),
    'hnrf_resize_mask': (_int, [_vp, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp]),
    'hnrf_deconv_fold': (_int, [_vp, _vp, _int, _int, _int, _int, _vp, _vp]),
    'hnrf_render_workspace_bytes': (_sz, [_i64, _int]),
    'hnrf_render_frame_workspace_bytes': (_sz, [_i64, _int]),
    'hnrf_render_frame_fwd': (_int, [_vp] * 14 + [_int, ctypes.c_float, _i64, _int, _int, _int, _i64, _vp, _sz] + [_vp] * 11 + [_vp, _vp, _vp, _vp]),
    'hnrf_render_term_workspace_bytes': (_sz, [_i64, _int]),
    'hnrf_render_rays_term_fwd': (_int, [_vp] * 14 + [_int, ctypes.c_float, ctypes.c_float, _i64, _int, _int, _int, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    'hnrf_render_rays_fwd': (_int, [_vp] * 14 + [_int, ctypes.c_float, _i64, _int, _int, _int, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


class HnrfError(RuntimeError):
    pass


def load():
    """Load libhnrf.so and type every exported entry point.  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise HnrfError(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'or `make -C humannerf_amd/csrc`. There is no CPU/PyTorch fallback for the hot path.')
    # One HIP runtime per process: torch bundles its own libamdhip64 (soname
    # libamdhip64.so.7).  Import torch FIRST so that libhnrf's NEEDED entry resolves to
    # that already-loaded copy; loading libhnrf first would pull in /opt/rocm's copy and
    # torch would then load a second runtime whose allocations ours cannot see.
    import torch  # noqa: F401
    tl = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
    if os.path.isfile(tl):
        ctypes.CDLL(tl, mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().hnrf_last_error()
        raise HnrfError(f'{what} failed ({rc}): {msg.decode() if msg else "?"}')
