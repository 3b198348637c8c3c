"""ctypes binding of include/ojphgpu.h (libojphgpu.so).

This module only *binds*; it contains no codec logic and no CPU fallback.  If the shared library
is missing it raises ImportError-like RuntimeError loudly (build it with
``python -c "import __graft_entry__ as g; g.build()"``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libojphgpu.so")

OK, E_INVALID, E_NOMEM, E_HIP, E_CODESTREAM, E_OVERFLOW, E_BLOCK, E_AGAIN, E_UNCOLLECTED = 0, -1, -2, -3, -4, -5, -6, -7, -8
PROG_ORDERS = {"LRCP": 0, "RLCP": 1, "RPCL": 2, "PCRL": 3, "CPRL": 4}


class Coc(C.Structure):
    """ojphgpu_coc: a COC marker segment (per-component coding style)"""
    _fields_ = [
        ("rank", C.c_uint8), ("reversible", C.c_uint8), ("num_decomps", C.c_uint8),
        ("log_block_w", C.c_uint8), ("log_block_h", C.c_uint8), ("has_precincts", C.c_uint8),
        ("reserved", C.c_uint8 * 2), ("precinct_exps", C.c_uint8 * 36),
    ]


class LiftStep(C.Structure):
    _fields_ = [("a", C.c_int32), ("b", C.c_int32), ("e", C.c_int32), ("A", C.c_float)]


class Atk(C.Structure):
    """ojphgpu_atk: an ATK marker segment (a lifting kernel)"""
    _fields_ = [("index", C.c_uint8), ("reversible", C.c_uint8), ("num_steps", C.c_uint8), ("coeff_type", C.c_uint8),
                ("K", C.c_float), ("steps", LiftStep * 16)]


class Dfs(C.Structure):
    """ojphgpu_dfs: a DFS marker segment (which directions each decomposition level transforms)"""
    _fields_ = [("used", C.c_uint8), ("index", C.c_uint8), ("num_levels", C.c_uint8), ("reserved", C.c_uint8), ("types", C.c_uint8 * 32)]


class Lift(C.Structure):
    """ojphgpu_lift: a lifting kernel as the general DWT kernels take it"""
    _fields_ = [("num_steps", C.c_uint32), ("elem", C.c_uint32), ("horz", C.c_uint32), ("vert", C.c_uint32), ("K", C.c_float),
                ("steps", LiftStep * 16)]


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("num_comps", C.c_uint32),
        ("bit_depth", C.c_uint32), ("is_signed", C.c_uint32),
        ("reversible", C.c_uint32), ("num_decomps", C.c_uint32),
        ("block_w", C.c_uint32), ("block_h", C.c_uint32),
        ("color_transform", C.c_uint32), ("tile_w", C.c_uint32), ("tile_h", C.c_uint32),
        ("prog_order", C.c_uint32), ("qstep", C.c_float),
        ("precinct_w", C.c_uint32), ("precinct_h", C.c_uint32), ("tlm", C.c_uint32),
        ("reserved", C.c_uint32 * 4),
        ("precinct_exps", C.c_uint8 * 36),
        ("image_x0", C.c_uint32), ("image_y0", C.c_uint32), ("tile_x0", C.c_uint32), ("tile_y0", C.c_uint32),
        ("comp_dx", C.c_uint8 * 16), ("comp_dy", C.c_uint8 * 16),
        ("comp_depth", C.c_uint8 * 16), ("comp_sign", C.c_uint8 * 16),
        ("coc", Coc * 16),
        ("nlt_default", C.c_uint8), ("nlt_bd_default", C.c_uint8),
        ("nlt_comp", C.c_uint8 * 16), ("nlt_rank", C.c_uint8 * 16), ("nlt_bd", C.c_uint8 * 16),
        ("nlt_reserved", C.c_uint8 * 2),
        ("qcc_qfactor", C.c_uint8 * 16), ("qcc_ctype", C.c_uint8 * 16), ("qcc_rank", C.c_uint8 * 16),
        ("wavelet", C.c_uint8), ("part2_reserved", C.c_uint8 * 3), ("atk", Atk * 4), ("dfs", Dfs * 4),
    ]


class BandInfo(C.Structure):
    _fields_ = [
        ("tile", C.c_uint32), ("comp", C.c_uint32), ("res", C.c_uint32), ("band", C.c_uint32),
        ("x0", C.c_uint32), ("y0", C.c_uint32), ("w", C.c_uint32), ("h", C.c_uint32),
        ("K_max", C.c_uint32), ("delta", C.c_float), ("delta_inv", C.c_float),
        ("nbx", C.c_uint32), ("nby", C.c_uint32), ("first_block", C.c_uint32),
        ("plane_off", C.c_uint64), ("pitch", C.c_uint32), ("reserved", C.c_uint32),
    ]


class BlockInfo(C.Structure):
    _fields_ = [("band", C.c_uint32), ("x0", C.c_uint32), ("y0", C.c_uint32), ("w", C.c_uint32),
                ("h", C.c_uint32), ("K_max", C.c_uint32)]


class LevelInfo(C.Structure):
    _fields_ = [
        ("tile", C.c_uint32), ("comp", C.c_uint32), ("res", C.c_uint32),
        ("w", C.c_uint32), ("h", C.c_uint32), ("x_even", C.c_uint32), ("y_even", C.c_uint32),
        ("src_off", C.c_uint64), ("src_pitch", C.c_uint32),
        ("ll_off", C.c_uint64), ("ll_pitch", C.c_uint32),
        ("hl_off", C.c_uint64), ("hl_pitch", C.c_uint32),
        ("lh_off", C.c_uint64), ("lh_pitch", C.c_uint32),
        ("hh_off", C.c_uint64), ("hh_pitch", C.c_uint32), ("kind", C.c_uint32),
    ]


class CodedBlock(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("len1", C.c_uint32), ("len2", C.c_uint32),
                ("missing_msbs", C.c_uint32), ("num_passes", C.c_uint32)]


class PaddedBlock(C.Structure):            # ojphgpu_padded_block
    _fields_ = [("block", C.c_uint32), ("got", C.c_uint32), ("offset", C.c_uint64), ("len1", C.c_uint32), ("len2", C.c_uint32),
                ("missing_msbs", C.c_uint32), ("num_passes", C.c_uint32)]


class DwtDesc(C.Structure):
    _fields_ = [
        ("src_off", C.c_uint64), ("ll_off", C.c_uint64), ("hl_off", C.c_uint64),
        ("lh_off", C.c_uint64), ("hh_off", C.c_uint64),
        ("src_pitch", C.c_uint32), ("ll_pitch", C.c_uint32), ("hl_pitch", C.c_uint32),
        ("lh_pitch", C.c_uint32), ("hh_pitch", C.c_uint32),
        ("w", C.c_uint32), ("h", C.c_uint32), ("x_even", C.c_uint32), ("y_even", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class CbDesc(C.Structure):
    _fields_ = [
        ("coef_off", C.c_uint64), ("pitch", C.c_uint32), ("w", C.c_uint16), ("h", C.c_uint16),
        ("K_max", C.c_uint8), ("reversible", C.c_uint8), ("missing_msbs", C.c_uint8),
        ("num_passes", C.c_uint8), ("delta", C.c_float), ("len1", C.c_uint32), ("len2", C.c_uint32),
        ("data_off", C.c_uint64), ("scratch_cap", C.c_uint32), ("reserved", C.c_uint32),
    ]


class CbResult(C.Structure):
    _fields_ = [("offset", C.c_uint32), ("length", C.c_uint32)]


class ConvertDesc(C.Structure):
    _fields_ = [("plane_off", C.c_uint64), ("pitch", C.c_uint32), ("w", C.c_uint32),
                ("h", C.c_uint32), ("src_x0", C.c_uint32), ("src_y0", C.c_uint32),
                ("img_pitch", C.c_uint32), ("img_off", C.c_uint64), ("fmt", C.c_uint32), ("reserved", C.c_uint32)]


_lib = None

# name -> (restype, argtypes); also the list every test checks against include/ojphgpu.h
SIGNATURES = {
    "ojphgpu_plan_create": (C.c_int, [C.POINTER(Params), C.POINTER(C.c_void_p)]),
    "ojphgpu_plan_destroy": (None, [C.c_void_p]),
    "ojphgpu_plan_params": (C.c_int, [C.c_void_p, C.POINTER(Params)]),
    "ojphgpu_plan_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "ojphgpu_plan_comp_info": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "ojphgpu_plan_tile_parts": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "ojphgpu_dwt_forward_image16": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.c_void_p, C.c_void_p]),
    "ojphgpu_dwt_inverse_image16": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.c_void_p, C.c_void_p]),
    "ojphgpu_convert_forward16": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.c_void_p, C.c_void_p]),
    "ojphgpu_convert_inverse16": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.c_void_p, C.c_void_p]),
    "ojphgpu_encoder_run_device16": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ojphgpu_encoder_run_device8": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ojphgpu_decoder_run_device8": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ojphgpu_dwt_forward_image_ex": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                               C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "ojphgpu_dwt_inverse_image_ex": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                               C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "ojphgpu_convert_forward_ex": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_void_p, C.c_void_p, C.c_int]),
    "ojphgpu_convert_inverse_ex": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_void_p, C.c_void_p, C.c_int]),
    "ojphgpu_encode16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ojphgpu_decoder_run_device16": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ojphgpu_decode16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ojphgpu_ht_decode_layout": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ojphgpu_plan_comp_format": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "ojphgpu_plan_comp_style": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "ojphgpu_plan_comp_lift": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "ojphgpu_plan_set_comments": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_uint16), C.POINTER(C.c_uint16),
                                            C.c_uint32]),
    "ojphgpu_plan_restrict_resolution": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "ojphgpu_plan_bands": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "ojphgpu_plan_blocks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "ojphgpu_plan_levels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "ojphgpu_plan_comp_plane": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64),
                                          C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "ojphgpu_plan_coded_blocks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "ojphgpu_t2_parse_restricted": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p]),
    "ojphgpu_plan_padded_blocks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ojphgpu_t2_write": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                   C.POINTER(C.c_size_t)]),
    "ojphgpu_t2_write_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                         C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]),
    "ojphgpu_t2_write_main_header": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ojphgpu_t2_parse": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "ojphgpu_dwt_forward_general": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "ojphgpu_dwt_inverse_general": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "ojphgpu_dwt_forward_general_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]),
    "ojphgpu_dwt_inverse_general_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]),
    "ojphgpu_dwt_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32,
                                      C.c_uint32, C.c_void_p]),
    "ojphgpu_dwt_inverse": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32,
                                      C.c_uint32, C.c_void_p]),
    "ojphgpu_dwt_forward_image": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_uint32,
                                            C.c_uint32, C.c_void_p, C.c_void_p]),
    "ojphgpu_dwt_inverse_image": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_uint32,
                                            C.c_uint32, C.c_void_p, C.c_void_p]),
    "ojphgpu_ht_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ojphgpu_ht_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "ojphgpu_ht_decode_aux_words": (C.c_uint32, [C.c_uint32]),
    "ojphgpu_ht_decode_prep": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "ojphgpu_ht_decode_step1": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "ojphgpu_ht_decode_step2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "ojphgpu_ht_decode_refine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ojphgpu_decoder_ht_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "ojphgpu_convert_forward": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint32,
                                          C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "ojphgpu_convert_inverse": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint32,
                                          C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "ojphgpu_encoder_create": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "ojphgpu_encoder_create_tiles": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32,
                                               C.POINTER(C.c_void_p)]),
    "ojphgpu_encoder_finish_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]),
    "ojphgpu_encoder_finish_tiles_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]),
    "ojphgpu_decoder_create_tiles": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32,
                                               C.POINTER(C.c_void_p)]),
    "ojphgpu_encoder_create_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "ojphgpu_encoder_finish_frame": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ojphgpu_decoder_create_batch": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "ojphgpu_decoder_upload_frame": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]),
    "ojphgpu_encoder_destroy": (None, [C.c_void_p]),
    "ojphgpu_encoder_run_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ojphgpu_encoder_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ojphgpu_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ojphgpu_encoder_coded_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "ojphgpu_decoder_create": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "ojphgpu_decoder_destroy": (None, [C.c_void_p]),
    "ojphgpu_decoder_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "ojphgpu_decoder_run_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ojphgpu_decoder_failed_blocks": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "ojphgpu_decoder_fused_retries": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "ojphgpu_decoder_giveup_epoch": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "ojphgpu_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ojphgpu_encoder_ht_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "ojphgpu_encoder_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "ojphgpu_decoder_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "ojphgpu_encoder_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "ojphgpu_decoder_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "ojphgpu_encoder_level_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32)]),
    "ojphgpu_decoder_level_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32)]),
    "ojphgpu_enc_pipe_create": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]),
    "ojphgpu_enc_pipe_destroy": (None, [C.c_void_p]),
    "ojphgpu_enc_pipe_acquire": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "ojphgpu_enc_pipe_submit": (C.c_int, [C.c_void_p]),
    "ojphgpu_enc_pipe_collect": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "ojphgpu_enc_pipe_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "ojphgpu_dec_pipe_create": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_uint32,
                                          C.POINTER(C.c_void_p)]),
    "ojphgpu_dec_pipe_destroy": (None, [C.c_void_p]),
    "ojphgpu_dec_pipe_plan": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "ojphgpu_dec_pipe_acquire": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "ojphgpu_dec_pipe_submit": (C.c_int, [C.c_void_p]),
    "ojphgpu_dec_pipe_collect": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]),
    "ojphgpu_dec_pipe_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "ojphgpu_dec_pipe_fused_retries": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "ojphgpu_multi_encoder_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_uint32, C.POINTER(C.c_void_p)]),
    "ojphgpu_multi_encoder_destroy": (None, [C.c_void_p]),
    "ojphgpu_multi_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ojphgpu_multi_encode_container": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ojphgpu_multi_encoder_workers": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32]),
    "ojphgpu_multi_decoder_create": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_int), C.c_uint32,
                                              C.POINTER(C.c_void_p)]),
    "ojphgpu_multi_decoder_destroy": (None, [C.c_void_p]),
    "ojphgpu_multi_decoder_plan": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "ojphgpu_multi_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint32)]),
    "ojphgpu_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "ojphgpu_host_unregister": (C.c_int, [C.c_void_p]),
    "ojphgpu_multi_decode_container": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]),
    "ojphgpu_enc_pipe_set_pixels": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ojphgpu_enc_pipe_set_packed": (C.c_int, [C.c_void_p, C.c_int]),
    "ojphgpu_dec_pipe_set_packed": (C.c_int, [C.c_void_p, C.c_int]),
    "ojphgpu_unpack_bits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]),
    "ojphgpu_pack_bits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]),
    "ojphgpu_dec_pipe_set_pixels": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ojphgpu_unpack_pixels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int]),
    "ojphgpu_pack_pixels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_uint32]),
    "ojphgpu_version": (C.c_char_p, []),
}


def lib():
    """Loads libojphgpu.so; fails loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "openjph_amd: %s is missing -- the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU "
                "fallback." % LIB_PATH)
        # PyTorch-ROCm bundles its own libamdhip64.so (SONAME libamdhip64.so.7).  Import it first so
        # that libojphgpu.so binds to the SAME HIP runtime instance torch uses -- streams and
        # device pointers are exchanged between the two.  (Pure C/C++ users simply get /opt/rocm's.)
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is plumbing only
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


class OjphError(RuntimeError):
    """Mirrors the reference's OJPH_ERROR -> std::runtime_error("ojph error")."""

    def __init__(self, code, what=""):
        names = {E_INVALID: "invalid argument / unsupported parameters", E_NOMEM: "out of memory",
                 E_HIP: "HIP runtime error (no GPU?)", E_CODESTREAM: "malformed codestream",
                 E_OVERFLOW: "output buffer too small", E_BLOCK: "error decoding a codeblock",
                 E_AGAIN: "no free pipeline slot",
                 E_UNCOLLECTED: "the previous run of this decoder asked for a repeat and was never collected"}
        super().__init__("ojph error: %s (%d) %s" % (names.get(code, "?"), code, what))
        self.code = code


def check(rc, what=""):
    if rc != OK:
        raise OjphError(rc, what)
