"""ctypes binding of the C ABI declared in include/iago_hip.h.

There is no CPU fallback.  If libiago_hip.so has not been built, importing
anything that needs it raises; if it is loaded on a host without a HIP device,
every launch returns IAGO_ERR_HIP and `check` raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("IAGO_HIP_LIB") or os.path.join(HERE, "libiago_hip.so")

IAGO_OK = 0
IAGO_MAX_TURNS = 128
IAGO_ROLLOUT_TABLE_FLOATS = 3 * 2 * 256 * 8 + 64 + 4 + 2 * 512
ROLLOUT_MODE_INDEX = 3 * 2 * 256 * 8 + 64  # blob[mode] == 1.0: product form
TRACE_PASS = 0xFF

# every symbol include/iago_hip.h declares -- the drop-in boundary: what a binding of the reference's hot path calls
# (tests/test_abi.py checks the lists against the headers and the built library)
SYMBOLS = [
    "iago_abi_version", "iago_last_error", "iago_device_count",
    "iago_legal_moves", "iago_apply_moves", "iago_play_turn", "iago_encode_planes", "iago_encode_planes_indexed",
    "iago_judge", "iago_sample_moves", "iago_augment8",
    "iago_rollout_build_table", "iago_rollout",
    "iago_policy_forward_split3", "iago_value_forward_split", "iago_value_rollout",
    "iago_mcts_reset", "iago_leaf_values", "iago_mcts_best_move", "iago_mcts_advance_root", "iago_mcts_compact",
    "iago_mcts_mix_backup_lookahead", "iago_mcts_store_priors", "iago_mcts_descend",
    "iago_mcts_search_persistent", "iago_mcts_search_capacity", "iago_mcts_search_streams_create",
    "iago_mcts_search_streams_destroy", "iago_mcts_search_split", "iago_selfplay_policy",
    "iago_policy_grad_workspace_bytes", "iago_policy_reinforce_grad", "iago_adam_chainer",
]
# include/iago_hip_layers.h: single blocks, the ends of the nets, format conversions, a block's three gradient kernels
# (the modules' planes-fed forwards of small batches, the layer-by-layer cross-checks of the fused kernels)
LAYER_SYMBOLS = [
    "iago_bias_relu", "iago_conv3x3_split", "iago_conv3x3_split_trunk", "iago_split_nchw", "iago_merge_nchw",
    "iago_value_stem", "iago_value_stem_boards", "iago_value_head",
    "iago_conv3x3_f32", "iago_stem_f32", "iago_stem_f32_boards", "iago_policy_head",
    "iago_conv3x3_wgrad_split", "iago_conv3x3_bwd_data_split", "iago_split_scaled",
]
# include/iago_hip_experimental.h, outside the boundary: two schedules of the per-playout engine that measured slower
# (game-asynchronous steps, value look-ahead: engine.BatchedMCTS(async_steps=True / value_ahead=True)); the per-phase forms
# of a playout (arbitrary callables as nets, the IAGO_FUSED_* = 0 knobs)
EXPERIMENTAL_SYMBOLS = [
    "iago_value_rollout_async", "iago_value_forward_batch", "iago_mcts_value_ahead_rows", "iago_mcts_value_ahead_store",
    "iago_mcts_select", "iago_mcts_expand", "iago_mcts_pending", "iago_mcts_backup", "iago_mcts_mix_backup",
    "iago_mcts_expand_cached", "iago_mcts_fresh_leaves",
]


class IagoError(RuntimeError):
    pass


class RolloutArgs(C.Structure):
    _fields_ = [
        ("own", C.c_void_p), ("opp", C.c_void_p), ("n", C.c_int64),
        ("table", C.c_void_p), ("uniforms", C.c_void_p),
        ("seed", C.c_uint64), ("id_base", C.c_uint32), ("stream_id", C.c_uint32),
        ("stream_id_dev", C.c_void_p),
        ("z", C.c_void_p), ("final_own", C.c_void_p), ("final_opp", C.c_void_p),
        ("n_turns", C.c_void_p), ("trace", C.c_void_p), ("log_form", C.c_int),
        ("throughput_hint", C.c_int),
    ]


ADAM_MAX_TENSORS = 24


class AdamArgs(C.Structure):
    _fields_ = [
        ("p", C.c_void_p * ADAM_MAX_TENSORS), ("g", C.c_void_p * ADAM_MAX_TENSORS),
        ("m", C.c_void_p * ADAM_MAX_TENSORS), ("v", C.c_void_p * ADAM_MAX_TENSORS),
        ("step", C.c_void_p * ADAM_MAX_TENSORS),
        ("count", C.c_int64 * ADAM_MAX_TENSORS), ("n_tensors", C.c_int32),
        ("alpha_t", C.c_float), ("one_minus_beta1", C.c_float), ("one_minus_beta2", C.c_float),
        ("eps", C.c_float), ("weight_decay", C.c_float),
    ]


class PolicyGradArgs(C.Structure):
    _fields_ = [
        ("own", C.c_void_p), ("opp", C.c_void_p), ("action", C.c_void_p), ("reward", C.c_void_p),
        ("n", C.c_int64), ("n_mean", C.c_int64),
        ("w1", C.c_void_p), ("b1", C.c_void_p),
        ("w_hi", C.c_void_p * 7), ("w_lo", C.c_void_p * 7), ("wt_hi", C.c_void_p * 7), ("wt_lo", C.c_void_p * 7),
        ("bias", C.c_void_p * 7),
        ("w9", C.c_void_p), ("b10", C.c_void_p),
        ("g_w1", C.c_void_p), ("g_b1", C.c_void_p),
        ("g_w", C.c_void_p * 7), ("g_b", C.c_void_p * 7),
        ("g_w9", C.c_void_p), ("g_b10", C.c_void_p),
        ("loss", C.c_void_p), ("probs", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("overflow", C.c_void_p),
    ]


class ConvSplitLayer(C.Structure):
    """Mirror of iago_conv_split_layer (include/iago_hip.h)."""
    _fields_ = [
        ("x_hi", C.c_void_p), ("x_lo", C.c_void_p), ("w_hi", C.c_void_p), ("w_lo", C.c_void_p),
        ("bias", C.c_void_p), ("y_hi", C.c_void_p), ("y_lo", C.c_void_p),
        ("cin", C.c_int32), ("reserved", C.c_int32),
    ]


class PolicySplit3Args(C.Structure):
    """Mirror of iago_policy_split3_args (include/iago_hip.h)."""
    _fields_ = [
        ("own", C.c_void_p), ("opp", C.c_void_p), ("index", C.c_void_p), ("n_dev", C.c_void_p), ("n", C.c_int64),
        ("w1", C.c_void_p), ("b1", C.c_void_p),
        ("w_hi", C.c_void_p * 7), ("w_mid", C.c_void_p * 7), ("w_lo", C.c_void_p * 7), ("bias", C.c_void_p * 7),
        ("w9", C.c_void_p), ("b10", C.c_void_p), ("probs", C.c_void_p), ("overflow", C.c_void_p),
        ("parts", C.c_int32), ("scratch_rows", C.c_int32), ("scratch", C.c_void_p),
    ]


class SelfplayPolicyArgs(C.Structure):
    """Mirror of iago_selfplay_policy_args (include/iago_hip.h)."""
    _fields_ = [
        ("model1", C.c_void_p), ("model2", C.c_void_p), ("own", C.c_void_p), ("opp", C.c_void_p), ("n", C.c_int64),
        ("seed", C.c_uint64), ("id_base", C.c_uint32), ("max_turns", C.c_int32),
        ("rec_own", C.c_void_p), ("rec_opp", C.c_void_p), ("rec_act", C.c_void_p), ("n_turns", C.c_void_p),
        ("bad_probs", C.c_void_p),
    ]


class MctsAsync(C.Structure):
    """Mirror of iago_mcts_async (include/iago_hip.h)."""
    _fields_ = [
        ("parts", C.c_int32), ("reserved", C.c_int32), ("wait", C.c_void_p), ("done", C.c_void_p),
        ("roll", C.c_void_p), ("fq_index", C.c_void_p), ("fq_count", C.c_void_p), ("step", C.c_void_p),
        ("n_sims", C.c_void_p), ("scratch", C.c_void_p),
    ]


VALUE_IMAGE_BYTES = 34816   # IAGO_VALUE_IMAGE_BYTES


class MctsLookahead(C.Structure):
    """Mirror of iago_mcts_lookahead (include/iago_hip.h)."""
    _fields_ = [
        ("trigger", C.c_int32), ("slots", C.c_int32), ("next_seq", C.c_void_p), ("cache_seq", C.c_void_p),
        ("cache", C.c_void_p), ("q_count", C.c_void_p), ("q_capacity", C.c_int32), ("path_stride", C.c_int32),
        ("q_own", C.c_void_p), ("q_opp", C.c_void_p), ("q_game", C.c_void_p), ("q_seq", C.c_void_p),
        ("error", C.c_void_p), ("clear_word", C.c_void_p), ("path", C.c_void_p), ("path_len", C.c_void_p),
        ("z_log", C.c_void_p), ("z_log_n", C.c_void_p), ("z_log_rows", C.c_int32), ("reserved", C.c_int32),
        ("async_", C.c_void_p), ("value_ahead", C.c_void_p),
    ]


class MctsValueAhead(C.Structure):
    """Mirror of iago_mcts_value_ahead (include/iago_hip.h)."""
    _fields_ = [
        ("x_capacity", C.c_int32), ("row_capacity", C.c_int32), ("x_count", C.c_void_p),
        ("x_game", C.c_void_p), ("x_node", C.c_void_p), ("x_own", C.c_void_p), ("x_opp", C.c_void_p),
        ("row_count", C.c_void_p), ("row_own", C.c_void_p), ("row_opp", C.c_void_p), ("row_node", C.c_void_p),
        ("row_v", C.c_void_p), ("total", C.c_void_p),
    ]


SEARCH_QUEUE_ENTRIES = 4096   # IAGO_SEARCH_QUEUE_ENTRIES
SEARCH_GAMES_PER_WORKGROUP = 32   # IAGO_SEARCH_GAMES_PER_WORKGROUP


class MctsSearchArgs(C.Structure):
    """Mirror of iago_mcts_search_args (include/iago_hip.h)."""
    _fields_ = [
        ("tree", C.c_void_p), ("root_own", C.c_void_p), ("root_opp", C.c_void_p), ("active", C.c_void_p),
        ("c_puct", C.c_float), ("lmbda", C.c_float), ("n_thr", C.c_int32), ("n_sims", C.c_int32),
        ("net_workgroups", C.c_int32), ("time_limit_ms", C.c_int32),
        ("value", C.c_void_p), ("policy", C.c_void_p), ("rollout", C.c_void_p),
        ("cur_node", C.c_void_p), ("cur_own", C.c_void_p), ("cur_opp", C.c_void_p), ("path", C.c_void_p),
        ("path_stride", C.c_int32), ("z_log_rows", C.c_int32), ("done", C.c_void_p), ("roll", C.c_void_p),
        ("leaf_value", C.c_void_p), ("z_log", C.c_void_p), ("z_log_n", C.c_void_p), ("q_slots", C.c_void_p),
        ("ctl", C.c_void_p), ("rep_v", C.c_void_p), ("rep_p", C.c_void_p), ("totals", C.c_void_p),
        ("stats", C.c_void_p), ("wg_own", C.c_void_p), ("wg_opp", C.c_void_p),
        ("max_turns", C.c_int32), ("games_per_workgroup", C.c_int32), ("game_own", C.c_void_p), ("game_opp", C.c_void_p),
        ("n_turns", C.c_void_p), ("rec_own", C.c_void_p), ("rec_opp", C.c_void_p), ("rec_valid", C.c_void_p),
        ("rec_move", C.c_void_p), ("rec_pi", C.c_void_p),
        ("vtable", C.c_void_p), ("vtable_slots", C.c_int64),
        ("trace", C.c_void_p), ("trace_rows", C.c_int32), ("pace_margin", C.c_int32),
        ("max_cus", C.c_int32), ("reserved", C.c_int32),
    ]


class ValueSplitArgs(C.Structure):
    """Mirror of iago_value_split_args (include/iago_hip.h)."""
    _fields_ = [
        ("planes", C.c_void_p), ("own", C.c_void_p), ("opp", C.c_void_p), ("n", C.c_int64),
        ("w1", C.c_void_p), ("b1", C.c_void_p),
        ("w_hi", C.c_void_p * 7), ("w_lo", C.c_void_p * 7), ("bias", C.c_void_p * 7),
        ("w9_hi", C.c_void_p), ("w9_lo", C.c_void_p),
        ("b9", C.c_void_p), ("w10", C.c_void_p), ("w11", C.c_void_p),
        ("out", C.c_void_p), ("overflow", C.c_void_p), ("index", C.c_void_p), ("n_dev", C.c_void_p),
    ]


class MctsTree(C.Structure):
    """Mirror of iago_mcts_tree (include/iago_hip.h)."""
    _fields_ = [
        ("n_games", C.c_int64), ("capacity", C.c_int32), ("has_v", C.c_int32),
        ("nodes", C.c_void_p), ("n_nodes", C.c_void_p), ("root", C.c_void_p), ("overflow", C.c_void_p),
    ]


NODE_WORDS = 8   # sizeof(iago_mcts_node) / 4: n_visits, q, p, v, first_child, parent, action | n_children << 8, reserved


_lib = None
ABI_VERSION = 13   # iago_abi_version() of the include/iago_hip.h these bindings mirror


def lib():
    """The loaded library.  Raises IagoError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise IagoError(
            "iago_amd/libiago_hip.so is missing: run `python -m iago_amd.build` "
            "(or __graft_entry__.build()).  There is no CPU fallback.")
    # One HIP runtime per process: PyTorch bundles its own libamdhip64 (same
    # SONAME as /opt/rocm's).  Load torch's copy first so that this library's
    # NEEDED libamdhip64.so.7 binds to it instead of pulling in a second runtime
    # (two runtimes in one process: "no ROCm-capable device is detected").
    try:
        import torch
        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(bundled):
            C.CDLL(bundled, mode=C.RTLD_GLOBAL)
    except ImportError:  # C-only hosts use the system runtime through RUNPATH
        pass
    L = C.CDLL(SO_PATH)
    vp, i64 = C.c_void_p, C.c_int64
    L.iago_abi_version.restype = C.c_int
    if L.iago_abi_version() != ABI_VERSION:
        # the ctypes structures below mirror ONE version of include/iago_hip.h
        raise IagoError("%s has ABI version %d, these bindings are for %d: rebuild it "
                        "(`python -m iago_amd.build`)" % (SO_PATH, L.iago_abi_version(), ABI_VERSION))
    L.iago_last_error.restype = C.c_char_p
    L.iago_device_count.restype = C.c_int
    L.iago_legal_moves.argtypes = [vp, vp, vp, i64, vp]
    L.iago_apply_moves.argtypes = [vp, vp, vp, i64, vp]
    L.iago_play_turn.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, i64, vp]
    L.iago_encode_planes.argtypes = [vp, vp, vp, i64, vp]
    L.iago_encode_planes_indexed.argtypes = [vp, vp, vp, vp, i64, vp, vp]
    L.iago_judge.argtypes = [vp, vp, vp, i64, vp]
    L.iago_sample_moves.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, vp,
                                    i64, vp]
    L.iago_augment8.argtypes = [vp, vp, vp, vp, vp, vp, i64, vp]
    L.iago_bias_relu.argtypes = [vp, vp, i64, C.c_int32, vp]
    i32 = C.c_int32
    L.iago_conv3x3_split.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp, vp]
    L.iago_conv3x3_split_trunk.argtypes = [C.POINTER(ConvSplitLayer), i32, i64, vp, vp]
    L.iago_value_forward_split.argtypes = [C.POINTER(ValueSplitArgs), vp]
    L.iago_policy_forward_split3.argtypes = [C.POINTER(PolicySplit3Args), vp]
    L.iago_value_rollout.argtypes = [C.POINTER(ValueSplitArgs), C.POINTER(RolloutArgs), vp]
    L.iago_conv3x3_wgrad_split.argtypes = [vp, vp, vp, vp, i64, i32, vp, i32, vp, vp, vp]
    L.iago_conv3x3_bwd_data_split.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i64, vp]
    L.iago_split_scaled.argtypes = [vp, vp, vp, vp, vp, i64, i32, vp, vp, vp]
    L.iago_policy_grad_workspace_bytes.argtypes = [i64]
    L.iago_policy_grad_workspace_bytes.restype = i64
    L.iago_policy_reinforce_grad.argtypes = [C.POINTER(PolicyGradArgs), vp]
    L.iago_adam_chainer.argtypes = [C.POINTER(AdamArgs), vp]
    L.iago_split_nchw.argtypes = [vp, vp, vp, i64, i32, vp, vp]
    L.iago_merge_nchw.argtypes = [vp, vp, vp, i64, i32, vp]
    L.iago_value_stem.argtypes = [vp, vp, vp, vp, vp, i64, vp, vp]
    L.iago_value_stem_boards.argtypes = [vp, vp, vp, vp, vp, vp, i64, vp, vp]
    L.iago_value_head.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, vp]
    L.iago_conv3x3_f32.argtypes = [vp, vp, vp, vp, i64, i32, i32, vp, vp]
    L.iago_stem_f32.argtypes = [vp, vp, vp, vp, i64, vp, vp]
    L.iago_stem_f32_boards.argtypes = [vp, vp, vp, vp, vp, vp, i64, vp, vp]
    L.iago_policy_head.argtypes = [vp, vp, vp, vp, i64, vp, vp]
    L.iago_rollout_build_table.argtypes = [vp, vp, vp]
    L.iago_rollout.argtypes = [C.POINTER(RolloutArgs), vp]
    tp = C.POINTER(MctsTree)
    L.iago_mcts_reset.argtypes = [tp, vp, vp]
    L.iago_mcts_select.argtypes = [tp, vp, vp, vp, C.c_float, C.c_int32, C.c_int, vp, vp, vp, vp,
                                   vp, vp, vp]
    L.iago_mcts_expand.argtypes = [tp, vp, i64, vp, vp, vp, vp, vp]
    L.iago_leaf_values.argtypes = [vp, vp, C.c_float, vp, i64, vp]
    L.iago_mcts_backup.argtypes = [tp, vp, vp, vp, vp]
    L.iago_mcts_mix_backup.argtypes = [tp, vp, vp, vp, vp, C.c_float, vp, vp, vp]
    L.iago_mcts_pending.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp, vp]
    L.iago_mcts_best_move.argtypes = [tp, vp, vp, vp, vp]
    L.iago_mcts_advance_root.argtypes = [tp, vp, vp, vp]
    L.iago_mcts_compact.argtypes = [tp, tp, vp, vp, vp]
    lp = C.POINTER(MctsLookahead)
    L.iago_mcts_mix_backup_lookahead.argtypes = [tp, vp, vp, vp, vp, vp, vp, C.c_float, vp, vp, lp, vp]
    L.iago_mcts_store_priors.argtypes = [lp, vp, vp, vp]
    L.iago_mcts_expand_cached.argtypes = [tp, vp, vp, vp, vp, lp, vp, vp]
    L.iago_mcts_fresh_leaves.argtypes = [tp, vp, vp, vp, vp, vp, vp]
    L.iago_mcts_descend.argtypes = [tp, vp, vp, vp, C.c_float, i32, vp, vp, vp, vp, vp, lp, vp, vp, vp, vp]
    L.iago_value_forward_batch.argtypes = [C.POINTER(ValueSplitArgs), i32, i32, vp]
    vap = C.POINTER(MctsValueAhead)
    L.iago_mcts_value_ahead_rows.argtypes = [tp, vap, vp]
    L.iago_mcts_value_ahead_store.argtypes = [tp, vap, vp]
    L.iago_mcts_search_persistent.argtypes = [C.POINTER(MctsSearchArgs), vp]
    L.iago_mcts_search_capacity.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.iago_mcts_search_streams_create.argtypes = [i32, C.POINTER(vp)]
    L.iago_mcts_search_streams_destroy.argtypes = [vp]
    L.iago_mcts_search_split.argtypes = [C.POINTER(MctsSearchArgs), vp, vp]
    L.iago_selfplay_policy.argtypes = [C.POINTER(SelfplayPolicyArgs), vp]
    for name in SYMBOLS[3:] + LAYER_SYMBOLS + EXPERIMENTAL_SYMBOLS:
        getattr(L, name).restype = C.c_int
    L.iago_policy_grad_workspace_bytes.restype = i64   # (bytes: beyond 2^31 from ~7,000 rows on)
    _lib = L
    return L


def check(rc, what=""):
    if rc != IAGO_OK:
        msg = lib().iago_last_error().decode("utf-8", "replace")
        raise IagoError("%s failed (%d): %s" % (what or "iago call", rc, msg))


def device_count():
    return lib().iago_device_count()
