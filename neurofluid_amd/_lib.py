"""ctypes binding of libneurofluid_hip.so (include/neurofluid_hip.h).

The product path has NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NF_LIB_PATH", os.path.join(_HERE, "lib", "libneurofluid_hip.so"))   # env override: A/B kernel experiments

c_void_p, c_int, c_float, c_size_t, c_int64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_int64


class NerfParams(ctypes.Structure):
    _fields_ = [("w", c_void_p * 12), ("b", c_void_p * 12)]


class TransStep(ctypes.Structure):
    """nf_trans_step_t (include/neurofluid_hip.h)."""
    _fields_ = [(k, c_void_p) for k in ("k_fluid", "b_fluid", "k_obst", "b_obst", "dense0_w", "dense0_b", "wp1", "bc1", "bd1", "wp2",
                                        "bc2", "bd2", "wp3", "bc3", "bd3", "box_grid", "box_feats", "grid_ws")] + \
               [("grid_ws_bytes", c_size_t)] + \
               [(k, c_void_p) for k in ("pos_new", "vel_new", "feats", "counts2", "idx_f", "d2_f", "roff", "ent", "a0", "a1", "a1r", "g3", "y3",
                                        "scratch", "overflow2", "done_counter")] + \
               [(k, c_int) for k in ("n", "pitch_f", "pitch_b", "use_window", "max_wg")] + \
               [(k, c_float) for k in ("radius", "extent", "dt", "scale")] + [("gravity", c_float * 3), ("bbox", c_float * 6), ("split", c_int), ("search", c_int)]


# name -> (restype, argtypes); mirrors include/neurofluid_hip.h one-to-one (tests check the list)
PROTOTYPES = {
    "nf_version": (c_int, []),
    "nf_last_error": (ctypes.c_char_p, []),
    "nf_grid_workspace_bytes": (c_size_t, [c_int, c_float, ctypes.POINTER(c_float)]),
    "nf_grid_cells": (c_int, [c_int, c_float, ctypes.POINTER(c_float)]),
    "nf_grid_points_aabb_offset": (c_size_t, []),
    "nf_grid_build": (c_int, [c_void_p, c_int, c_float, ctypes.POINTER(c_float), c_void_p, c_size_t, c_int, c_void_p]),
    "nf_ball_query_firstk": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nf_radius_scan_workspace_bytes": (c_size_t, [c_int]),
    "nf_radius_count": (c_int, [c_void_p, c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nf_radius_fill": (c_int, [c_void_p, c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "nf_csr_clamp": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "nf_nearest": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "nf_get_rays": (c_int, [c_int, c_int, c_float, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "nf_get_rays_chunks": (c_int, [c_int, c_int, c_float, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "nf_render_classify": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    "nf_render_search": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_int, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nf_render_features": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_int, c_void_p, c_int,
                                   c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "nf_render_features_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_int, c_void_p, c_int,
                                       c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "nf_render_feature_dims": (c_int, [c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                       ctypes.POINTER(c_int)]),
    "nf_nerf_packed_floats": (c_size_t, [c_int, c_int]),
    "nf_nerf_pack": (c_int, [ctypes.POINTER(NerfParams), c_int, c_int, c_void_p, c_void_p]),
    "nf_nerf_mlp_fwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nf_nerf_mlp_fwd_n": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nf_nerf_mlp_fwd_n2": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nf_nerf_amask_words": (c_size_t, [c_int]),
    "nf_composite_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "nf_composite_fwd_noise": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "nf_composite_bwd_noise": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "nf_importance_sample": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "nf_coarse_perturb": (c_int, [c_void_p, c_void_p, c_float, c_int, c_int, c_void_p, c_void_p]),
    "nf_importance_sample_rays": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "nf_nerf_stream_floats": (c_size_t, [c_int, c_int]),
    "nf_nerf_pack_stream": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "nf_nerf_mlp_fwd_l": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "nf_nerf_stream_a_floats": (c_size_t, [c_int, c_int]),
    "nf_nerf_pack_stream_a": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "nf_nerf_mlp_fwd_a": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "nf_nerf_packed_h2_bytes": (c_size_t, []),
    "nf_nerf_pack_h2": (c_int, [ctypes.POINTER(NerfParams), c_int, c_int, c_void_p, c_void_p]),
    "nf_nerf_mlp_fwd_h2": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "nf_nerf_packed_ha_bytes": (c_size_t, []),
    "nf_nerf_pack_ha": (c_int, [c_void_p, c_void_p, c_void_p]),
    "nf_nerf_mlp_fwd_ha": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "nf_nerf_packed_s_bytes": (c_size_t, []),
    "nf_nerf_pack_s": (c_int, [ctypes.POINTER(NerfParams), c_int, c_int, c_void_p, c_void_p]),
    "nf_nerf_mlp_fwd_s": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "nf_nerf_wgrad_floats": (c_size_t, [c_int, c_int]),
    "nf_nerf_wgrad_workspace_floats": (c_size_t, [c_int, c_int, c_int]),
    "nf_nerf_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nf_nerf_wgrad_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nf_composite_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                 c_void_p, c_void_p, c_int, c_void_p]),
    "nf_nerf_packed_bwd_floats": (c_size_t, []),
    "nf_nerf_pack_bwd": (c_int, [ctypes.POINTER(NerfParams), c_int, c_int, c_void_p, c_void_p]),
    "nf_nerf_mlp_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    "nf_nerf_pack_n": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "nf_nerf_pack_bwd_n": (c_int, [c_void_p, c_void_p, c_void_p]),
    "nf_nerf_mlp_bwd_n": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
    "nf_nerf_mlp_bwd_n2": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p]),
    "nf_nerf_mlp_bwd_n3": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    "nf_embed_fwd": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "nf_embed_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "nf_gemm_f32_workspace_floats": (c_size_t, [c_int, c_int, c_int]),
    "nf_gemm_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_int64, c_int64, c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                            c_int, c_int, c_void_p, c_void_p]),
    "nf_trans_integrate": (c_int, [c_void_p, c_void_p, ctypes.POINTER(c_float), c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nf_trans_update": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_int, c_void_p, c_void_p, c_void_p]),
    "nf_cconv_pairs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "nf_cconv_gather_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "nf_cconv_small_bwd_filter_workspace_floats": (c_size_t, [c_int, c_int]),
    "nf_cconv_small_bwd_filter": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                          c_void_p, c_void_p]),
    "nf_cconv_small_bwd_feat": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "nf_cconv_small": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                               c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "nf_image_ssim_workspace_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "nf_image_ssim": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, ctypes.POINTER(c_float), c_float, c_void_p, c_void_p,
                              c_void_p]),
    "nf_host_choice_mt19937": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "nf_cconv_transform": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nf_cconv_gather": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                c_void_p]),
    "nf_trans_prepare_limits": (c_int, [ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "nf_trans_prepare": (c_int, [c_void_p, c_void_p, ctypes.POINTER(c_float), c_float, c_int, c_float, ctypes.POINTER(c_float), c_void_p,
                                c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nf_trans_front_max_pitch": (c_int, []),
    "nf_trans_all_pairs_max_points": (c_int, []),
    "nf_trans_front": (c_int, [c_void_p] * 5 + [c_int, c_float, c_float, c_int, c_int, c_int] + [c_void_p] * 13 + [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "nf_pinned_device_ptr": (c_void_p, [c_void_p]),
    "nf_hip_peek_error": (c_int, []),
    "nf_host_wait_word": (c_int, [c_void_p, c_int, ctypes.c_double]),
    "nf_gather_view_pixels": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nf_scale3": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nf_relu_bwd_add": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "nf_colsum": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "nf_cconv_split_db": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "nf_cconv_gf_packed_floats": (c_size_t, [c_int, c_int]),
    "nf_cconv_gf_pack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "nf_cconv_gf_plan": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                 ctypes.POINTER(c_size_t)]),
    "nf_cconv_gf_packed_split_bytes": (c_size_t, [c_int, c_int]),
    "nf_cconv_gf_pack_split": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "nf_cconv_gf_layer": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_float, c_void_p,
                                  c_void_p, c_void_p]),
    "nf_cconv3_workspace_floats": (c_size_t, [c_int]),
    "nf_cconv3_packed_floats": (c_size_t, []),
    "nf_cconv3_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "nf_cconv_gf_layer_g3": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "nf_cconv3_gather": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                 c_float, c_void_p, c_void_p, c_void_p]),
    "nf_cconv3_layer": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "nf_adam_step": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_double, ctypes.c_double,
                             c_float, c_float, c_void_p]),
    "nf_adam_step_dev": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_double, ctypes.c_double,
                                 c_float, c_float, c_void_p]),
    "nf_note_overflow": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "nf_note_overflow4": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nf_gather_view_pixels_tab": (c_int, [c_int, c_void_p, c_int, c_int, ctypes.c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nf_e2e_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_float,
                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nf_trans_step": (c_int, [ctypes.POINTER(TransStep), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                              c_void_p]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -m neurofluid_amd.build` (hipcc, gfx950). "
            "neurofluid_amd has no CPU/PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)   # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.nf_version() < 100:
        raise RuntimeError("libneurofluid_hip.so is too old")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().nf_last_error()
        raise RuntimeError(f"libneurofluid_hip {what} failed (rc={rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
