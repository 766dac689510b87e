"""ctypes binding of libgossipnet_hip.so (the C ABI of include/gossipnet_hip.h).

There is NO CPU fallback: if the HIP library is missing, loading fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
from .build import LIB as LIB_PATH     # libgossipnet_hip.so (a GNET_TRACE / GNET_EXTRA_FLAGS measurement process: its own probe library)
GNET_MAX_BLOCKS = 64
ABI_VERSION = 8          # include/gossipnet_hip.h GNET_ABI_VERSION: the struct mirrors below belong to this version

OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_WORKSPACE, ERR_HIP = 0, -1, -2, -3, -4
_ERR = {ERR_INVALID: "invalid argument", ERR_UNSUPPORTED: "unsupported configuration",
        ERR_WORKSPACE: "workspace too small", ERR_HIP: "HIP runtime error"}


class GnetError(RuntimeError):
    pass


class InvalidArgumentError(GnetError, ValueError):
    """Mirrors tf.errors.InvalidArgumentError raised by the reference ops."""


class gnet_config(C.Structure):
    _fields_ = [("num_classes", C.c_int32), ("num_blocks", C.c_int32), ("neighbor_thresh", C.c_float),
                ("normalize_loss", C.c_int32), ("loss_multiplyer", C.c_float),
                ("shortcut_dim", C.c_int32), ("reduced_dim", C.c_int32), ("pairfeat_dim", C.c_int32),
                ("pwfeat_dim", C.c_int32), ("pwfeat_narrow_dim", C.c_int32),
                ("num_pwfeat_fc", C.c_int32), ("predict_fc_dim", C.c_int32), ("num_predict_fc", C.c_int32),
                ("num_block_pw_fc", C.c_int32), ("num_block_fc", C.c_int32), ("pw_feat_multiplyer", C.c_float),
                ("neighbor_feats", C.c_int32)]


class gnet_shape(C.Structure):
    _fields_ = [("n_img", C.c_int32), ("n_det", C.c_int32), ("n_gt", C.c_int32),
                ("n_edge", C.c_int64), ("n_anno", C.c_int64)]


class gnet_inputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dets", "det_scores", "det_classes", "det_off", "gt_boxes",
                                          "gt_crowd", "gt_classes", "gt_off", "anno_off")]


_PB = C.c_void_p * (GNET_MAX_BLOCKS + 1)


class gnet_buffers(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("row_ptr", "edge_c", "edge_n", "edge_iou", "edge_t", "edge_nz", "geo", "pw_tc", "pw_tn", "pw_h1", "pw_h2",
                                           "pw_feats")] +
                [(n, _PB) for n in ("block_feats", "blk_r", "blk_rc", "blk_rn", "blk_pm", "blk_q", "blk_rnb", "blk_h1", "blk_h2", "blk_parg")] +
                [(n, C.c_void_p) for n in ("head1", "head2", "prediction", "det_anno_iou", "labels", "weights",
                                           "det_gt_matching", "loss", "d_logits", "d_x", "d_pc", "d_rc", "d_rn",
                                           "d_pw", "d_h1", "d_g1", "ewin", "wprefix", "wlist", "xmask", "tflag", "apos", "tpos", "wrow", "spos", "rl_scratch", "pw_rows", "w1_s", "w1_t", "packed_t", "arena", "scratch_i", "match_ws")] +
                [("match_ws_bytes", C.c_size_t), ("arena_floats", C.c_size_t), ("profiler", C.c_void_p), ("start_feat", C.c_void_p)])


EXPORTS = ["gnet_param_count", "gnet_graph_count", "gnet_graph_fill", "gnet_graph_transpose", "gnet_workspace_bytes", "gnet_plan",
           "gnet_forward", "gnet_loss", "gnet_match_prepare", "gnet_backward", "gnet_backward_prepare", "det_matching_workspace_bytes", "det_matching_f32",
           "roi_pool_fwd_f32", "roi_pool_bwd_f32", "roi_pool_bwd_atomic_f32", "gnet_version", "gnet_profiler_create", "gnet_profiler_read",
           "gnet_profiler_destroy", "gnet_profiler_set_stride", "gnet_profiler_begin", "gnet_profiler_end", "gnet_adam_step", "gnet_momentum_step", "gnet_clip_by_norm",
           "gnet_fc_workspace_bytes", "gnet_fc_forward", "gnet_fc_backward", "gnet_box_iou", "gnet_debug_gemm", "gnet_abi_version", "gnet_abi_sizes"]

KCLASSES = ["graph", "pack", "pw_fwd", "node_fwd", "edge_fwd", "loss", "head_bwd", "winner_lists", "edge_bwd", "gather_winners",
            "node_bwd", "pw_bwd_main", "pw_w1_nodesums", "pw_w1_classrows", "reduce_partials", "edge_geometry"]

_lib = None


def abi_mirror():
    """What gnet_abi_sizes must report for the ctypes mirrors above (sizes, four field offsets of gnet_buffers)."""
    b = gnet_buffers
    return [C.sizeof(gnet_config), C.sizeof(gnet_shape), C.sizeof(gnet_inputs), C.sizeof(b),
            b.head1.offset, b.d_g1.offset, b.match_ws_bytes.offset, b.start_feat.offset]


def check_abi(lib):
    """Refuse a library whose struct layout differs from this binding's mirror -- whatever the source-hash record says
    (a shifted gnet_buffers hands the kernels wrong device pointers without any error)."""
    if not hasattr(lib, "gnet_abi_version") or not hasattr(lib, "gnet_abi_sizes"):
        raise GnetError("libgossipnet_hip.so predates the ABI guard (no gnet_abi_version): rebuild it with "
                        "`python -m gossipnet_amd.build`")
    lib.gnet_abi_version.restype = C.c_int
    lib.gnet_abi_version.argtypes = []
    lib.gnet_abi_sizes.restype = C.c_int
    lib.gnet_abi_sizes.argtypes = [C.POINTER(C.c_size_t)]
    got_v = lib.gnet_abi_version()
    sizes = (C.c_size_t * 8)()
    n_classes = lib.gnet_abi_sizes(sizes)
    if got_v != ABI_VERSION or list(sizes) != abi_mirror() or n_classes != len(KCLASSES):
        raise GnetError("libgossipnet_hip.so has ABI version %d, layout %s, %d kernel classes; this binding expects version "
                        "%d, layout %s, %d classes: rebuild the library (`python -m gossipnet_amd.build`) or update "
                        "gossipnet_amd/_lib.py" % (got_v, list(sizes), n_classes, ABI_VERSION, abi_mirror(), len(KCLASSES)))


def load():
    """Load the shared library (raises if it has not been built: no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    from . import build as _build
    global LIB_PATH
    if os.environ.get("GNET_LIB_AB"):
        # measurement only (A/B of two builds on one box, tools/): load exactly this file, no source-hash check, no rebuild -- the
        # ABI guard below still applies
        LIB_PATH = os.environ["GNET_LIB_AB"]
        import warnings
        warnings.warn("GNET_LIB_AB is set: loading %s WITHOUT the source-hash check (A/B measurement only -- results of this "
                      "process are not those of the library the tree builds)" % LIB_PATH)
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        check_abi(lib)
        return _bind(lib)
    # A library whose recorded source hash DIFFERS from the tree is rebuilt (never silently used); one without a record
    # (built by other tooling, deployed without the sources' toolchain) is loaded as it is, with a warning.
    stale = missing = not os.path.exists(LIB_PATH)
    if not missing:
        try:
            stale = open(LIB_PATH + ".srchash").read().strip() != _build.source_hash()
        except OSError:
            import warnings
            warnings.warn("libgossipnet_hip.so carries no source hash (%s.srchash): loading it as it is" % LIB_PATH)
    if stale:
        # not a fallback: the same HIP sources, compiled on the spot when hipcc is available.  build() serialises
        # concurrent callers (one rank per GPU under torchrun) with a file lock and re-checks the hash inside it.
        try:
            _build.build()
        except Exception as exc:      # noqa: BLE001
            raise GnetError("libgossipnet_hip.so is %s (%s) and could not be built (%s): run "
                            "`python -m gossipnet_amd.build` -- there is no CPU fallback"
                            % ("missing" if missing else "stale", LIB_PATH, exc))
    # torch bundles its own HIP runtime: it must be mapped first so that this library binds to the
    # same libamdhip64 as the streams/allocations it is handed (loading /opt/rocm's copy first breaks launches)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    check_abi(lib)
    return _bind(lib)


def _bind(lib):
    global _lib
    vp, i32, i64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
    P = C.POINTER
    lib.gnet_param_count.restype = i64
    lib.gnet_param_count.argtypes = [P(gnet_config)]
    lib.gnet_graph_count.restype = C.c_int
    lib.gnet_graph_count.argtypes = [vp, i32, vp, i32, f32, vp, vp, vp]
    lib.gnet_graph_fill.restype = C.c_int
    lib.gnet_graph_fill.argtypes = [vp, i32, vp, i32, f32, vp, vp, vp, vp, vp]
    lib.gnet_graph_transpose.restype = C.c_int
    lib.gnet_graph_transpose.argtypes = [vp, vp, vp, i64, vp, vp]
    lib.gnet_workspace_bytes.restype = sz
    lib.gnet_workspace_bytes.argtypes = [P(gnet_config), P(gnet_shape), C.c_int]
    lib.gnet_plan.restype = C.c_int
    lib.gnet_plan.argtypes = [P(gnet_config), P(gnet_shape), C.c_int, vp, sz, P(gnet_buffers)]
    lib.gnet_forward.restype = C.c_int
    lib.gnet_forward.argtypes = [P(gnet_config), P(gnet_shape), P(gnet_inputs), vp, P(gnet_buffers), C.c_int, vp]
    lib.gnet_loss.restype = C.c_int
    lib.gnet_loss.argtypes = [P(gnet_config), P(gnet_shape), P(gnet_inputs), vp, f32, P(gnet_buffers), C.c_int32, vp]
    lib.gnet_match_prepare.restype = C.c_int
    lib.gnet_match_prepare.argtypes = [P(gnet_config), P(gnet_shape), P(gnet_inputs), P(gnet_buffers), vp]
    lib.gnet_backward.restype = C.c_int
    lib.gnet_backward.argtypes = [P(gnet_config), P(gnet_shape), P(gnet_inputs), vp, P(gnet_buffers), vp, C.c_int32, vp, vp, vp]
    lib.gnet_backward_prepare.restype = C.c_int
    lib.gnet_backward_prepare.argtypes = [P(gnet_config), P(gnet_shape), P(gnet_inputs), vp, P(gnet_buffers), C.c_int32, vp]
    lib.det_matching_workspace_bytes.restype = sz
    lib.det_matching_workspace_bytes.argtypes = [i32, i32]
    lib.det_matching_f32.restype = C.c_int
    lib.det_matching_f32.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp, sz, vp]
    lib.roi_pool_fwd_f32.restype = C.c_int
    lib.roi_pool_fwd_f32.argtypes = [vp, i32, i32, i32, i32, vp, i32, i32, i32, f32, vp, vp, vp]
    lib.roi_pool_bwd_f32.restype = C.c_int
    lib.roi_pool_bwd_f32.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp]
    lib.roi_pool_bwd_atomic_f32.restype = C.c_int
    lib.roi_pool_bwd_atomic_f32.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp]
    lib.gnet_profiler_create.restype = C.c_int
    lib.gnet_profiler_create.argtypes = [i32, C.c_uint32, P(vp)]
    lib.gnet_profiler_read.restype = C.c_int
    lib.gnet_profiler_read.argtypes = [vp, P(C.c_double), P(i32)]
    lib.gnet_profiler_set_stride.restype = C.c_int
    lib.gnet_profiler_set_stride.argtypes = [vp, i32]
    lib.gnet_profiler_begin.restype = C.c_int
    lib.gnet_profiler_begin.argtypes = [vp, i32, vp]
    lib.gnet_profiler_end.restype = C.c_int
    lib.gnet_profiler_end.argtypes = [vp, i32, vp]
    lib.gnet_profiler_destroy.restype = C.c_int
    lib.gnet_profiler_destroy.argtypes = [vp]
    lib.gnet_adam_step.restype = C.c_int
    lib.gnet_adam_step.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, i64, f32, vp]
    lib.gnet_momentum_step.restype = C.c_int
    lib.gnet_momentum_step.argtypes = [vp, vp, vp, i64, f32, f32, f32, vp]
    lib.gnet_clip_by_norm.restype = C.c_int
    lib.gnet_clip_by_norm.argtypes = [vp, vp, i32, f32, vp]
    lib.gnet_fc_workspace_bytes.restype = sz
    lib.gnet_fc_workspace_bytes.argtypes = [i64, i64, i64]
    lib.gnet_fc_forward.restype = C.c_int
    lib.gnet_fc_forward.argtypes = [vp, vp, vp, i64, i64, i64, C.c_int, vp, vp, sz, vp]
    lib.gnet_fc_backward.restype = C.c_int
    lib.gnet_fc_backward.argtypes = [vp, vp, vp, vp, i64, i64, i64, C.c_int, vp, vp, vp, vp, sz, vp]
    lib.gnet_box_iou.restype = C.c_int
    lib.gnet_box_iou.argtypes = [vp, i32, vp, i32, vp, vp, vp, i32, vp, vp]
    lib.gnet_debug_gemm.restype = C.c_int
    lib.gnet_debug_gemm.argtypes = [vp, vp, i64, i64, i64, C.c_int, vp, vp, vp]
    lib.gnet_version.restype = C.c_char_p
    lib.gnet_version.argtypes = []
    _lib = lib
    return lib


def check(status, what):
    if status == OK:
        return
    msg = "%s failed: %s (%d)" % (what, _ERR.get(status, "unknown"), status)
    if status == ERR_INVALID:
        raise InvalidArgumentError(msg)
    raise GnetError(msg)
