"""ctypes binding of libscopa_hip.so (the C ABI declared in include/scopa.h).

The product path: every solver entry point goes through this library's HIP kernels.  There is no
Python or CPU fallback -- if the library is missing or no GPU is present the call raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCOPA_HIP_LIBRARY: another build of the same sources (the host-only AddressSanitizer/UBSan build of `make -C tests/tools asan`);
# never a fallback -- a path that does not exist fails exactly like the default one
LIB_PATH = os.environ.get("SCOPA_HIP_LIBRARY") or os.path.join(_HERE, "libscopa_hip.so")

SCOPA_OK, SCOPA_EINVAL, SCOPA_ENODEV, SCOPA_EHIP, SCOPA_ESTATE, SCOPA_ENOMEM, SCOPA_ELIMIT, SCOPA_ETIMEOUT = 0, -1, -2, -3, -4, -5, -6, -7
N_NODES, N_DECISION, N_TERMINAL = 2229, 1653, 576

# every symbol include/scopa.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "scopa_abi_version", "scopa_strerror", "scopa_last_error", "scopa_ctx_create", "scopa_ctx_destroy",
    "scopa_ctx_synchronize", "scopa_deal_py_seed", "scopa_state_init", "scopa_state_step", "scopa_state_clone", "scopa_state_is_terminal",
    "scopa_state_current_player", "scopa_state_legal", "scopa_state_rewards_x2", "scopa_state_infoset_key",
    "scopa_key_to_string", "scopa_state_infoset_string", "scopa_step_batch", "scopa_step_batch_host", "scopa_set_deal",
    "scopa_tree_counts", "scopa_tree_export", "scopa_tables_reset", "scopa_tables_get", "scopa_tables_set",
    "scopa_visited_get", "scopa_cfr_exact_iterate", "scopa_cfr_exact_traverse", "scopa_cfr_exact_mode", "scopa_cfr_exact_traverse_from", "scopa_mccfr_replay", "scopa_mccfr_seed",
    "scopa_mccfr_iterate", "scopa_mccfr_traverse", "scopa_mccfr_delta_buffer", "scopa_mccfr_bind_delta", "scopa_mccfr_delta_get", "scopa_mccfr_delta_set", "scopa_mccfr_apply",
    "scopa_mccfr_iteration_counter", "scopa_mccfr_graph_mode", "scopa_debug_lds_limit", "scopa_sdcfr_frontier_width", "scopa_sdcfr_features", "scopa_sdcfr_expand",
    "scopa_sdcfr_terminal_values", "scopa_sdcfr_backward", "scopa_sdcfr_visits", "scopa_sdcfr_traverse_fused", "scopa_sdcfr_image_floats", "scopa_sdcfr_pack_weights", "scopa_sdcfr_tuning", "scopa_sdcfr_mode", "scopa_sdcfr_train_params", "scopa_sdcfr_train_step", "scopa_sdcfr_train_steps", "scopa_features_from_states",
    "scopa_eval_init_states", "scopa_eval_step", "scopa_eval_tabular_step", "scopa_eval_tabular_prepare", "scopa_eval_tabular_match", "scopa_cfr_sync_iterate", "scopa_multi_create", "scopa_multi_destroy",
    "scopa_multi_deal_py_seeds", "scopa_multi_set_perms", "scopa_multi_perms_get", "scopa_multi_build", "scopa_multi_cfr_exact_iterate",
    "scopa_multi_cfr_exact_iterate_lanes", "scopa_multi_cfr_sync_iterate", "scopa_multi_mccfr_iterate", "scopa_multi_exploitability", "scopa_multi_tables_get", "scopa_multi_counters", "scopa_full_deal_py_seed",
    "scopa_full_state_init", "scopa_full_state_step", "scopa_full_state_legal", "scopa_full_state_infoset_string",
    "scopa_full_step_batch", "scopa_full_step_batch_host", "scopa_full_random_playouts",
    "scopa_team_state_init", "scopa_team_state_step", "scopa_team_state_legal", "scopa_team_state_rewards_x2", "scopa_team_state_infoset_string",
    "scopa_team_step_batch", "scopa_team_step_batch_host", "scopa_team_random_playouts",
    "scopa_mccfr_iterate_sharded", "scopa_p2p_create", "scopa_p2p_connect", "scopa_p2p_allreduce_delta", "scopa_p2p_set_form", "scopa_p2p_set_budget", "scopa_p2p_status", "scopa_p2p_destroy", "scopa_exploitability", "scopa_counters", "scopa_prof_enable", "scopa_prof_read", "scopa_prof_device", "scopa_prof_phases", "scopa_prof_spread",
]


class ScopaError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        msg = f"{where}: status {status}"
        try:
            msg += f" ({lib().scopa_strerror(status).decode()})"
        except Exception:
            pass
        if detail:
            msg += f": {detail}"
        super().__init__(msg)


class State16(C.Structure):
    """scopa_state, 16 bytes (include/scopa.h)."""
    _fields_ = [("hand", C.c_uint16 * 2), ("table", C.c_uint32), ("nh", C.c_uint8 * 2), ("nt", C.c_uint8),
                ("step", C.c_uint8), ("ncap", C.c_uint8 * 2), ("scopas", C.c_uint8 * 2)]


STEP_CLONED, STEP_COUNT_MASK = 0x80, 0x7F   # include/scopa.h: SCOPA_STEP_CLONED, SCOPA_STEP_COUNT_MASK
STATE_DTYPE = np.dtype([("hand", "<u2", (2,)), ("table", "<u4"), ("nh", "u1", (2,)), ("nt", "u1"), ("step", "u1"),
                        ("ncap", "u1", (2,)), ("scopas", "u1", (2,))])
assert STATE_DTYPE.itemsize == 16 and C.sizeof(State16) == 16

_lib = None


def lib():
    """Load the library (raises OSError loudly if it has not been built: python -m scopa_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # PyTorch-ROCm bundles its own HIP/HSA runtime.  If libscopa_hip.so pulled in /opt/rocm's copy first, torch would
        # later start a SECOND runtime in the process and find no GPU; loading torch first makes both share one.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` or `python scopa_amd/build.py` "
                      "(the MiniScopa solver path has no fallback)")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u32, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64
    sig = {
        "scopa_abi_version": (i32, []),
        "scopa_strerror": (C.c_char_p, [i32]),
        "scopa_last_error": (C.c_char_p, [vp]),
        "scopa_ctx_create": (i32, [i32, vp, C.POINTER(vp)]),
        "scopa_ctx_destroy": (i32, [vp]),
        "scopa_ctx_synchronize": (i32, [vp]),
        "scopa_deal_py_seed": (i32, [i64, vp]),
        "scopa_state_init": (i32, [vp, C.POINTER(State16)]),
        "scopa_state_step": (i32, [C.POINTER(State16), i32]),
        "scopa_state_clone": (i32, [C.POINTER(State16), C.POINTER(State16)]),
        "scopa_state_is_terminal": (i32, [C.POINTER(State16)]),
        "scopa_state_current_player": (i32, [C.POINTER(State16)]),
        "scopa_state_legal": (i32, [C.POINTER(State16), i32, C.POINTER(i32 * 4), C.POINTER(i32)]),
        "scopa_state_rewards_x2": (i32, [C.POINTER(State16), C.POINTER(i32 * 2)]),
        "scopa_state_infoset_key": (i32, [C.POINTER(State16), i32, C.POINTER(u64)]),
        "scopa_key_to_string": (i32, [u64, C.c_char_p, i32]),
        "scopa_state_infoset_string": (i32, [C.POINTER(State16), i32, C.c_char_p, i32]),
        "scopa_step_batch": (i32, [vp, vp, vp, i64]),
        "scopa_step_batch_host": (i32, [vp, vp, vp, i64]),
        "scopa_set_deal": (i32, [vp, vp]),
        "scopa_tree_counts": (i32, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "scopa_tree_export": (i32, [vp, vp, vp, vp, vp, vp, vp]),
        "scopa_tables_reset": (i32, [vp]),
        "scopa_tables_get": (i32, [vp, vp, vp, vp]),
        "scopa_tables_set": (i32, [vp, vp, vp, vp]),
        "scopa_cfr_exact_iterate": (i32, [vp, i32, vp]),
        "scopa_cfr_exact_traverse": (i32, [vp, i32, C.POINTER(C.c_double)]),
        "scopa_cfr_exact_mode": (i32, [vp, i32]),
        "scopa_cfr_exact_traverse_from": (i32, [vp, i32, i32, vp, C.c_double, C.c_double, C.POINTER(C.c_double)]),
        "scopa_visited_get": (i32, [vp, vp]),
        "scopa_mccfr_replay": (i32, [vp, i32, vp, i64, C.POINTER(i64)]),
        "scopa_mccfr_seed": (i32, [vp, u64]),
        "scopa_mccfr_iterate": (i32, [vp, u32, u32]),
        "scopa_mccfr_traverse": (i32, [vp, u32, u32, u32]),
        "scopa_mccfr_delta_buffer": (i32, [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]),
        "scopa_mccfr_bind_delta": (i32, [vp, vp, C.c_size_t]),
        "scopa_mccfr_delta_get": (i32, [vp, vp]),
        "scopa_mccfr_delta_set": (i32, [vp, vp]),
        "scopa_mccfr_apply": (i32, [vp]),
        "scopa_mccfr_iteration_counter": (i32, [vp, C.POINTER(u32)]),
        "scopa_mccfr_graph_mode": (i32, [vp, i32]),
        "scopa_debug_lds_limit": (i32, [vp, i32]),
        "scopa_exploitability": (i32, [vp, vp, vp, vp]),
        "scopa_sdcfr_frontier_width": (i32, [i32, i32]),
        "scopa_sdcfr_features": (i32, [vp, i32, i64, vp, vp, vp]),
        "scopa_sdcfr_expand": (i32, [vp, i32, i32, i64, vp, vp, vp, vp, vp, u32, u32]),
        "scopa_sdcfr_terminal_values": (i32, [vp, i32, i64, vp, vp]),
        "scopa_sdcfr_backward": (i32, [vp, i32, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64]),
        "scopa_sdcfr_visits": (i32, [vp, C.POINTER(u64)]),
        "scopa_sdcfr_traverse_fused": (i32, [vp, i32, i32, vp, vp, vp, vp, i64, i64, vp, vp, u32, u32]),
        "scopa_sdcfr_image_floats": (i32, []),
        "scopa_sdcfr_pack_weights": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp]),
        "scopa_sdcfr_tuning": (i32, [vp, i32, i32]),
        "scopa_sdcfr_train_params": (i32, []),
        "scopa_sdcfr_train_step": (i32, [vp, vp, i32, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, i32, C.c_float, vp]),
        "scopa_sdcfr_train_steps": (i32, [vp, vp, i32, i32, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, i32, C.c_float, vp]),
        "scopa_sdcfr_mode": (i32, [vp, i32]),
        "scopa_features_from_states": (i32, [vp, vp, i64, vp, vp]),
        "scopa_eval_init_states": (i32, [vp, vp, i64]),
        "scopa_eval_step": (i32, [vp, vp, i64, vp, vp, u32, u32]),
        "scopa_eval_tabular_step": (i32, [vp, vp, vp, i64, i32, vp, vp, u32]),
        "scopa_eval_tabular_prepare": (i32, [vp, vp]),
        "scopa_eval_tabular_match": (i32, [vp, i64, i64, u32, vp, vp, vp]),
        "scopa_cfr_sync_iterate": (i32, [vp, i32]),
        "scopa_multi_create": (i32, [vp, i32, C.POINTER(vp)]),
        "scopa_multi_destroy": (i32, [vp]),
        "scopa_multi_deal_py_seeds": (i32, [vp, vp]),
        "scopa_multi_set_perms": (i32, [vp, vp]),
        "scopa_multi_perms_get": (i32, [vp, vp]),
        "scopa_multi_build": (i32, [vp, vp]),
        "scopa_multi_cfr_exact_iterate": (i32, [vp, i32]),
        "scopa_multi_cfr_sync_iterate": (i32, [vp, i32]),
        "scopa_multi_cfr_exact_iterate_lanes": (i32, [vp, i32]),
        "scopa_multi_exploitability": (i32, [vp, vp]),
        "scopa_multi_mccfr_iterate": (i32, [vp, u32, u32, u64]),
        "scopa_multi_tables_get": (i32, [vp, i32, vp, vp, vp, vp]),
        "scopa_multi_counters": (i32, [vp, C.POINTER(u64), C.POINTER(u64)]),
        "scopa_full_deal_py_seed": (i32, [i64, vp]),
        "scopa_full_state_init": (i32, [vp, u32, vp]),
        "scopa_full_state_step": (i32, [vp, vp, i32]),
        "scopa_full_state_legal": (i32, [vp, i32, C.POINTER(i32 * 3), C.POINTER(i32)]),
        "scopa_full_state_infoset_string": (i32, [vp, i32, C.c_char_p, i32]),
        "scopa_full_step_batch": (i32, [vp, vp, vp, vp, i64]),
        "scopa_full_step_batch_host": (i32, [vp, vp, vp, vp, i64, i64]),
        "scopa_full_random_playouts": (i32, [vp, vp, i64, vp, vp]),
        "scopa_mccfr_iterate_sharded": (i32, [vp, u32, u32, u32]),
        "scopa_p2p_create": (i32, [vp, i32, i32, vp]),
        "scopa_p2p_connect": (i32, [vp, vp]),
        "scopa_p2p_allreduce_delta": (i32, [vp]),
        "scopa_p2p_set_form": (i32, [vp, i32]),
        "scopa_p2p_set_budget": (i32, [vp, C.c_double]),
        "scopa_p2p_status": (i32, [vp, C.POINTER(i32), C.POINTER(u64)]),
        "scopa_p2p_destroy": (i32, [vp]),
        "scopa_team_state_init": (i32, [vp, vp]),
        "scopa_team_state_step": (i32, [vp, i32]),
        "scopa_team_state_legal": (i32, [vp, C.POINTER(i32 * 4), C.POINTER(i32)]),
        "scopa_team_state_rewards_x2": (i32, [vp, C.POINTER(i32 * 4)]),
        "scopa_team_state_infoset_string": (i32, [vp, i32, C.c_char_p, i32]),
        "scopa_team_step_batch": (i32, [vp, vp, vp, i64]),
        "scopa_team_step_batch_host": (i32, [vp, vp, vp, i64]),
        "scopa_team_random_playouts": (i32, [vp, vp, i64, vp, vp]),
        "scopa_counters": (i32, [vp, C.POINTER(u64), C.POINTER(u64)]),
        "scopa_prof_enable": (i32, [vp, i32]),
        "scopa_prof_read": (i32, [vp, C.POINTER(i64), C.POINTER(C.c_double)]),
        "scopa_prof_device": (i32, [vp, C.POINTER(i64), C.POINTER(C.c_double)]),
        "scopa_prof_phases": (i32, [vp, C.POINTER(C.c_double * 3)]),
        "scopa_prof_spread": (i32, [vp, C.POINTER(C.c_double * 3)]),
    }
    # A/B tooling only (tests/tools/ab_time.sh): a variant library built from an OLDER revision, named through SCOPA_HIP_LIBRARY, may lack entry points
    # added since -- with SCOPA_AB_OLD_LIBRARY=1 those are left unbound (calling one raises AttributeError); the product's own library must export them all
    lenient = bool(os.environ.get("SCOPA_HIP_LIBRARY")) and os.environ.get("SCOPA_AB_OLD_LIBRARY") == "1"
    for name, (res, args) in sig.items():
        if lenient and not hasattr(L, name):
            continue
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    if L.scopa_abi_version() != 1:
        raise OSError("libscopa_hip.so ABI version mismatch")
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def deal_py_seed(seed):
    """16-card permutation of random.seed(seed); random.shuffle(deck) (MiniDeck, mini_scopa_game.py:25-28)."""
    perm = np.zeros(16, np.uint8)
    rc = lib().scopa_deal_py_seed(int(seed), _ptr(perm))
    if rc:
        raise ScopaError(rc, "scopa_deal_py_seed")
    return perm


def key_to_string(key):
    buf = C.create_string_buffer(96)
    rc = lib().scopa_key_to_string(int(key), buf, 96)
    if rc < 0:
        raise ScopaError(rc, "scopa_key_to_string")
    return buf.value.decode()


class Context:
    """One scopa_ctx: a HIP device + stream + (after set_deal) one game tree and its tables."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        self._L = lib()
        rc = self._L.scopa_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(self._h))
        if rc:
            self._h = None
            raise ScopaError(rc, "scopa_ctx_create", "a GPU is required: the solver path has no CPU fallback")
        self.device = device
        self.n_infosets = 0
        self._children = []  # weakrefs of objects that hold device memory through this context (MultiDeal)

    def close(self):
        if getattr(self, "_h", None):
            for ref in getattr(self, "_children", []):
                child = ref()
                if child is not None:
                    child.close()
            self._L.scopa_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, where):
        if rc:
            raise ScopaError(rc, where, self._L.scopa_last_error(self._h).decode())

    def synchronize(self):
        self._ck(self._L.scopa_ctx_synchronize(self._h), "scopa_ctx_synchronize")

    # ---- batched step ---------------------------------------------------------------------
    def step_batch_host(self, states, actions):
        """states: np array of STATE_DTYPE (modified in place), actions: uint8."""
        assert states.dtype == STATE_DTYPE and states.flags.c_contiguous
        actions = np.ascontiguousarray(actions, np.uint8)
        assert actions.size == states.size
        self._ck(self._L.scopa_step_batch_host(self._h, _ptr(states), _ptr(actions), states.size), "scopa_step_batch_host")
        return states

    def step_batch(self, d_states_ptr, d_actions_ptr, n):
        self._ck(self._L.scopa_step_batch(self._h, C.c_void_p(d_states_ptr), C.c_void_p(d_actions_ptr), int(n)), "scopa_step_batch")

    # ---- deal / tree ------------------------------------------------------------------------
    def set_deal(self, perm16):
        perm = np.ascontiguousarray(perm16, np.uint8)
        assert perm.size == 16
        self._ck(self._L.scopa_set_deal(self._h, _ptr(perm)), "scopa_set_deal")
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._ck(self._L.scopa_tree_counts(self._h, C.byref(a), C.byref(b), C.byref(c)), "scopa_tree_counts")
        self.n_infosets = c.value
        self.perm = perm
        return self.n_infosets

    def tree_export(self):
        I = self.n_infosets
        out = dict(states=np.zeros(N_NODES, STATE_DTYPE), infoset=np.zeros(N_NODES, np.int32), r2=np.zeros((N_NODES, 2), np.int8),
                   infoset_key=np.zeros(I, np.uint64), infoset_nlegal=np.zeros(I, np.int8), infoset_legal=np.zeros((I, 4), np.int8))
        self._ck(self._L.scopa_tree_export(self._h, _ptr(out["states"]), _ptr(out["infoset"]), _ptr(out["r2"]),
                                           _ptr(out["infoset_key"]), _ptr(out["infoset_nlegal"]), _ptr(out["infoset_legal"])),
                 "scopa_tree_export")
        return out

    # ---- tables ------------------------------------------------------------------------------
    def tables_reset(self):
        self._ck(self._L.scopa_tables_reset(self._h), "scopa_tables_reset")

    def tables_get(self, regret=True, strategy=True, local=True):
        I = self.n_infosets
        R = np.zeros((I, 4)) if regret else None
        S = np.zeros((I, 4)) if strategy else None
        Lc = np.zeros((I, 4)) if local else None
        self._ck(self._L.scopa_tables_get(self._h, _ptr(R), _ptr(S), _ptr(Lc)), "scopa_tables_get")
        return R, S, Lc

    def tables_set(self, regret=None, strategy=None, local=None):
        arrs = []
        for a in (regret, strategy, local):
            if a is not None:
                a = np.ascontiguousarray(a, np.float64)
                assert a.shape == (self.n_infosets, 4)
            arrs.append(a)
        self._ck(self._L.scopa_tables_set(self._h, _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2])), "scopa_tables_set")

    # ---- solvers ------------------------------------------------------------------------------
    def cfr_exact_iterate(self, n_iters):
        rv = np.zeros((max(int(n_iters), 0), 2))
        self._ck(self._L.scopa_cfr_exact_iterate(self._h, int(n_iters), _ptr(rv)), "scopa_cfr_exact_iterate")
        return rv

    def cfr_exact_mode(self, sequential):
        """False (default): whole-tree traversals as a parallel schedule; True: the one-lane sequential walk.  Same tables bit for bit."""
        self._ck(self._L.scopa_cfr_exact_mode(self._h, 1 if sequential else 0), "scopa_cfr_exact_mode")

    def cfr_exact_traverse(self, traverser):
        v = C.c_double()
        self._ck(self._L.scopa_cfr_exact_traverse(self._h, int(traverser), C.byref(v)), "scopa_cfr_exact_traverse")
        return v.value

    def cfr_exact_traverse_from(self, traverser, path, reach_p0=1.0, reach_p1=1.0):
        pa = np.ascontiguousarray(path, np.int32)
        v = C.c_double()
        self._ck(self._L.scopa_cfr_exact_traverse_from(self._h, int(traverser), pa.size, _ptr(pa), float(reach_p0), float(reach_p1),
                                                       C.byref(v)), "scopa_cfr_exact_traverse_from")
        return v.value

    def visited_get(self):
        seq = np.zeros(self.n_infosets, np.uint32)
        self._ck(self._L.scopa_visited_get(self._h, _ptr(seq)), "scopa_visited_get")
        return seq

    def mccfr_replay(self, n_iters, uniforms):
        u = np.ascontiguousarray(uniforms, np.float64)
        used = C.c_int64()
        self._ck(self._L.scopa_mccfr_replay(self._h, int(n_iters), _ptr(u), u.size, C.byref(used)), "scopa_mccfr_replay")
        return used.value

    def mccfr_seed(self, seed):
        self._ck(self._L.scopa_mccfr_seed(self._h, int(seed)), "scopa_mccfr_seed")

    def mccfr_iterate(self, batch, n_iters):
        self._ck(self._L.scopa_mccfr_iterate(self._h, int(batch), int(n_iters)), "scopa_mccfr_iterate")

    def mccfr_traverse(self, iteration, b0, nb):
        self._ck(self._L.scopa_mccfr_traverse(self._h, int(iteration), int(b0), int(nb)), "scopa_mccfr_traverse")

    def mccfr_delta_buffer(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(self._L.scopa_mccfr_delta_buffer(self._h, C.byref(p), C.byref(n)), "scopa_mccfr_delta_buffer")
        return p.value, n.value

    def mccfr_bind_delta(self, d_ptr, nbytes):
        """Use a caller-owned device buffer (e.g. torch tensor .data_ptr()) as the all-reduce payload."""
        self._ck(self._L.scopa_mccfr_bind_delta(self._h, C.c_void_p(d_ptr) if d_ptr else None, int(nbytes)), "scopa_mccfr_bind_delta")

    def mccfr_delta_get(self):
        d = np.zeros((self.n_infosets, 5))
        self._ck(self._L.scopa_mccfr_delta_get(self._h, _ptr(d)), "scopa_mccfr_delta_get")
        return d

    def mccfr_delta_set(self, d):
        d = np.ascontiguousarray(d, np.float64)
        assert d.shape == (self.n_infosets, 5)
        self._ck(self._L.scopa_mccfr_delta_set(self._h, _ptr(d)), "scopa_mccfr_delta_set")

    def mccfr_apply(self):
        self._ck(self._L.scopa_mccfr_apply(self._h), "scopa_mccfr_apply")

    def mccfr_graph_mode(self, on):
        """replay scopa_mccfr_iterate's (traverse, apply) launches as captured HIP graphs of up to 64 iterations (same results)"""
        self._ck(self._L.scopa_mccfr_graph_mode(self._h, 1 if on else 0), "scopa_mccfr_graph_mode")

    def debug_lds_limit(self, nbytes):
        self._ck(self._L.scopa_debug_lds_limit(self._h, int(nbytes)), "scopa_debug_lds_limit")

    def mccfr_iteration(self):
        v = C.c_uint32()
        self._ck(self._L.scopa_mccfr_iteration_counter(self._h, C.byref(v)), "scopa_mccfr_iteration_counter")
        return v.value

    def exploitability(self, policy=None, return_policy=False):
        """-> dict(exploitability, br0, br1, value_p0[, policy]); policy=None evaluates the average policy."""
        pin = None if policy is None else np.ascontiguousarray(policy, np.float64)
        out = np.zeros(4)
        pout = np.zeros((self.n_infosets, 4)) if return_policy else None
        self._ck(self._L.scopa_exploitability(self._h, _ptr(pin), _ptr(out), _ptr(pout)), "scopa_exploitability")
        res = dict(exploitability=out[0], br0=out[1], br1=out[2], value_p0=out[3])
        if return_policy:
            res["policy"] = pout
        return res

    # ---- SDCFR / evaluation building blocks (device pointers: torch tensors' data_ptr()) --------------
    def sdcfr_features(self, ply, n, idx_ptr, feats_ptr, mask_ptr):
        self._ck(self._L.scopa_sdcfr_features(self._h, ply, n, C.c_void_p(idx_ptr), C.c_void_p(feats_ptr), C.c_void_p(mask_ptr)), "scopa_sdcfr_features")

    def sdcfr_expand(self, ply, traverser, n, idx_ptr, adv_ptr, child_ptr, pol_ptr, uniforms_ptr, iteration, b0):
        self._ck(self._L.scopa_sdcfr_expand(self._h, ply, traverser, n, C.c_void_p(idx_ptr), C.c_void_p(adv_ptr), C.c_void_p(child_ptr),
                                            C.c_void_p(pol_ptr), C.c_void_p(uniforms_ptr) if uniforms_ptr else None, iteration, b0), "scopa_sdcfr_expand")

    def sdcfr_terminal_values(self, traverser, n, idx_ptr, val_ptr):
        self._ck(self._L.scopa_sdcfr_terminal_values(self._h, traverser, n, C.c_void_p(idx_ptr), C.c_void_p(val_ptr)), "scopa_sdcfr_terminal_values")

    def sdcfr_backward(self, ply, traverser, n, idx_ptr, pol_ptr, child_val_ptr, val_ptr, feats_ptr, mask_ptr, mem_feat_ptr, mem_regret_ptr,
                       mem_mask_ptr, capacity, write_base):
        vp = lambda x: C.c_void_p(x) if x else None
        self._ck(self._L.scopa_sdcfr_backward(self._h, ply, traverser, n, vp(idx_ptr), vp(pol_ptr), vp(child_val_ptr), vp(val_ptr), vp(feats_ptr),
                                              vp(mask_ptr), vp(mem_feat_ptr), vp(mem_regret_ptr), vp(mem_mask_ptr), capacity, write_base),
                 "scopa_sdcfr_backward")

    def sdcfr_pack_weights(self, player, w1_ptr, b1_ptr, w2_ptr, b2_ptr, w3_ptr, b3_ptr, image_ptr):
        """torch tensors of one advantage net (W[out][in], float32) -> `player`'s half of the fused kernel's weight image."""
        self._ck(self._L.scopa_sdcfr_pack_weights(self._h, player, *(C.c_void_p(x) for x in (w1_ptr, b1_ptr, w2_ptr, b2_ptr, w3_ptr, b3_ptr, image_ptr))),
                 "scopa_sdcfr_pack_weights")

    def sdcfr_mode(self, forward_per_visit):
        """0 = policy table per launch + walks (default), 1 = a forward pass per visit inside the traversal kernel; same results"""
        self._ck(self._L.scopa_sdcfr_mode(self._h, 1 if forward_per_visit else 0), "scopa_sdcfr_mode")

    def sdcfr_tuning(self, traversals_per_task=0, wavefronts_per_task=0):
        """experiments: task shape of the fused traversal kernel (0 = the library's choice); results do not depend on it"""
        self._ck(self._L.scopa_sdcfr_tuning(self._h, traversals_per_task, wavefronts_per_task), "scopa_sdcfr_tuning")

    def sdcfr_train_steps(self, rows_ptr, n_rows, n_steps, feat_ptr, regret_ptr, mask_ptr, capacity, param_ptrs, state_ptr, first_step, lr, loss_ptr):
        """n_steps consecutive optimiser steps (the epochs of one train() call) on rows [n_steps][n_rows]"""
        self._ck(self._L.scopa_sdcfr_train_steps(self._h, C.c_void_p(rows_ptr), int(n_rows), int(n_steps), C.c_void_p(feat_ptr), C.c_void_p(regret_ptr), C.c_void_p(mask_ptr),
                                                 int(capacity), *(C.c_void_p(p) for p in param_ptrs), C.c_void_p(state_ptr), int(first_step), float(lr),
                                                 C.c_void_p(loss_ptr)), "scopa_sdcfr_train_steps")

    def sdcfr_train_step(self, rows_ptr, n_rows, feat_ptr, regret_ptr, mask_ptr, capacity, param_ptrs, state_ptr, step, lr, loss_ptr):
        """one optimiser step of an advantage net in two HIP launches (include/scopa.h); param_ptrs = (w1, b1, w2, b2, w3, b3) device pointers"""
        self._ck(self._L.scopa_sdcfr_train_step(self._h, C.c_void_p(rows_ptr), int(n_rows), C.c_void_p(feat_ptr), C.c_void_p(regret_ptr), C.c_void_p(mask_ptr),
                                                int(capacity), *(C.c_void_p(p) for p in param_ptrs), C.c_void_p(state_ptr), int(step), float(lr),
                                                C.c_void_p(loss_ptr)), "scopa_sdcfr_train_step")

    def sdcfr_traverse_fused(self, traverser, batch, weights_ptr, mem_feat_ptr, mem_regret_ptr, mem_mask_ptr, capacity, write_base,
                             root_values_ptr, uniforms_ptr, iteration, b0):
        self._ck(self._L.scopa_sdcfr_traverse_fused(self._h, traverser, batch, C.c_void_p(weights_ptr), C.c_void_p(mem_feat_ptr),
                                                    C.c_void_p(mem_regret_ptr), C.c_void_p(mem_mask_ptr), capacity, write_base,
                                                    C.c_void_p(root_values_ptr), C.c_void_p(uniforms_ptr) if uniforms_ptr else None,
                                                    iteration, b0), "scopa_sdcfr_traverse_fused")

    def sdcfr_visits(self):
        v = C.c_uint64()
        self._ck(self._L.scopa_sdcfr_visits(self._h, C.byref(v)), "scopa_sdcfr_visits")
        return v.value

    def features_from_states(self, states_ptr, n, feats_ptr, mask_ptr):
        self._ck(self._L.scopa_features_from_states(self._h, C.c_void_p(states_ptr), n, C.c_void_p(feats_ptr), C.c_void_p(mask_ptr)), "scopa_features_from_states")

    def eval_init_states(self, states_ptr, n):
        self._ck(self._L.scopa_eval_init_states(self._h, C.c_void_p(states_ptr), n), "scopa_eval_init_states")

    def eval_step(self, states_ptr, n, probs_ptr, seat_ptr, stream_id, ply_tag):
        self._ck(self._L.scopa_eval_step(self._h, C.c_void_p(states_ptr), n, C.c_void_p(probs_ptr) if probs_ptr else None, C.c_void_p(seat_ptr),
                                         stream_id, ply_tag), "scopa_eval_step")

    def eval_tabular_prepare(self, policy_ptr):
        """sampling thresholds of a tabular policy, once per evaluation; eval_tabular_step(..., policy_ptr=0, ...) then uses them"""
        self._ck(self._L.scopa_eval_tabular_prepare(self._h, C.c_void_p(policy_ptr)), "scopa_eval_tabular_prepare")

    def eval_tabular_step(self, states_ptr, idx_ptr, n, ply, policy_ptr, seat_ptr, stream_id):
        self._ck(self._L.scopa_eval_tabular_step(self._h, C.c_void_p(states_ptr), C.c_void_p(idx_ptr), n, ply, C.c_void_p(policy_ptr) if policy_ptr else None,
                                                 C.c_void_p(seat_ptr), stream_id), "scopa_eval_tabular_step")

    def eval_tabular_match(self, n, n_seat0, stream_id, states_ptr=0, idx_ptr=0):
        """the whole match of the prepared policy vs uniform random in one launch -> int64 [2][5]: per seat half (episodes, sum r x2, sum (r x2)^2,
        sum trained scopas, sum opponent scopas); optionally the final states / terminal indices into device buffers"""
        st = np.zeros((2, 5), np.int64)
        self._ck(self._L.scopa_eval_tabular_match(self._h, int(n), int(n_seat0), stream_id, C.c_void_p(states_ptr) if states_ptr else None,
                                                  C.c_void_p(idx_ptr) if idx_ptr else None, _ptr(st)), "scopa_eval_tabular_match")
        return st

    def cfr_sync_iterate(self, n_iters):
        self._ck(self._L.scopa_cfr_sync_iterate(self._h, int(n_iters)), "scopa_cfr_sync_iterate")

    def full_step_batch_host(self, states, actions, decks):
        assert states.dtype == FULL_STATE_DTYPE
        actions = np.ascontiguousarray(actions, np.uint8)
        decks = np.ascontiguousarray(decks, np.uint8).reshape(-1, 40)
        self._ck(self._L.scopa_full_step_batch_host(self._h, _ptr(states), _ptr(actions), _ptr(decks), decks.shape[0], states.size),
                 "scopa_full_step_batch_host")
        return states

    def full_step_batch(self, states_ptr, actions_ptr, decks_ptr, n):
        """device pointers: states[n] (64 B each) <- step(states[i], decks[states[i].game], actions[i])"""
        self._ck(self._L.scopa_full_step_batch(self._h, C.c_void_p(states_ptr), C.c_void_p(actions_ptr), C.c_void_p(decks_ptr), int(n)), "scopa_full_step_batch")

    def full_random_playouts(self, seeds):
        seeds = np.ascontiguousarray(seeds, np.int64)
        r2, plies = np.zeros(seeds.size, np.int8), np.zeros(seeds.size, np.int16)
        self._ck(self._L.scopa_full_random_playouts(self._h, _ptr(seeds), seeds.size, _ptr(r2), _ptr(plies)), "scopa_full_random_playouts")
        return r2, plies

    def p2p_create(self, rank, world):
        """-> this rank's 64-byte IPC handle (uint8[64]) of its peer-exchange inbox"""
        h = np.zeros(64, np.uint8)
        self._ck(self._L.scopa_p2p_create(self._h, int(rank), int(world), _ptr(h)), "scopa_p2p_create")
        return h

    def p2p_connect(self, handles):
        handles = np.ascontiguousarray(handles, np.uint8).reshape(-1, 64)
        self._ck(self._L.scopa_p2p_connect(self._h, _ptr(handles)), "scopa_p2p_connect")

    def p2p_allreduce_delta(self):
        self._ck(self._L.scopa_p2p_allreduce_delta(self._h), "scopa_p2p_allreduce_delta")

    def mccfr_iterate_sharded(self, b0, nb, n_iters=1):
        self._ck(self._L.scopa_mccfr_iterate_sharded(self._h, int(b0), int(nb), int(n_iters)), "scopa_mccfr_iterate_sharded")

    def p2p_status(self):
        """-> (waits that timed out, exchanges issued); synchronises the stream"""
        t, n = C.c_int32(), C.c_uint64()
        self._ck(self._L.scopa_p2p_status(self._h, C.byref(t), C.byref(n)), "scopa_p2p_status")
        return t.value, n.value

    def p2p_set_form(self, light):
        """False (default): plain accesses + system-scope fences; True: sc0 sc1 accesses ordered by s_waitcnt (validate first)"""
        self._ck(self._L.scopa_p2p_set_form(self._h, 1 if light else 0), "scopa_p2p_set_form")

    def p2p_set_budget(self, seconds):
        self._ck(self._L.scopa_p2p_set_budget(self._h, float(seconds)), "scopa_p2p_set_budget")

    def p2p_destroy(self):
        self._ck(self._L.scopa_p2p_destroy(self._h), "scopa_p2p_destroy")

    def team_step_batch_host(self, states, actions):
        assert states.dtype == TEAM_STATE_DTYPE
        actions = np.ascontiguousarray(actions, np.uint8)
        self._ck(self._L.scopa_team_step_batch_host(self._h, _ptr(states), _ptr(actions), states.size), "scopa_team_step_batch_host")
        return states

    def team_step_batch(self, states_ptr, actions_ptr, n):
        """device pointers: states[n] (40 B each) <- step(states[i], actions[i])"""
        self._ck(self._L.scopa_team_step_batch(self._h, C.c_void_p(states_ptr), C.c_void_p(actions_ptr), int(n)), "scopa_team_step_batch")

    def team_random_playouts(self, seeds):
        """-> (reward x2 of team 0 per game, scopas[n][4] per seat)"""
        seeds = np.ascontiguousarray(seeds, np.int64)
        r2, sc = np.zeros(seeds.size, np.int8), np.zeros((seeds.size, 4), np.uint8)
        self._ck(self._L.scopa_team_random_playouts(self._h, _ptr(seeds), seeds.size, _ptr(r2), _ptr(sc)), "scopa_team_random_playouts")
        return r2, sc

    def counters(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._ck(self._L.scopa_counters(self._h, C.byref(a), C.byref(b)), "scopa_counters")
        return a.value, b.value

    def prof_enable(self, stride=1):
        """stride: bracket every stride-th traversal launch with HIP events (True = 1, False/0 = off)."""
        self._ck(self._L.scopa_prof_enable(self._h, int(stride)), "scopa_prof_enable")

    def prof_read(self):
        n, ms = C.c_int64(), C.c_double()
        self._ck(self._L.scopa_prof_read(self._h, C.byref(n), C.byref(ms)), "scopa_prof_read")
        return n.value, ms.value

    def prof_phases(self):
        """after prof_device(): mean us per workgroup of the sampled launches in (prologue, walks, epilogue)"""
        out = (C.c_double * 3)()
        self._ck(self._L.scopa_prof_phases(self._h, C.byref(out)), "scopa_prof_phases")
        return tuple(out)

    def prof_spread(self):
        """(mean workgroup start behind the launch's first, last workgroup's start behind the first, longest workgroup) in us; after prof_device()."""
        out = (C.c_double * 3)()
        self._ck(self._L.scopa_prof_spread(self._h, C.byref(out)), "scopa_prof_spread")
        return [out[0], out[1], out[2]]

    def prof_device(self):
        """-> (traversal launches, their summed milliseconds by the kernel's own 100 MHz clock) since the context was created"""
        n, ms = C.c_int64(), C.c_double()
        self._ck(self._L.scopa_prof_device(self._h, C.byref(n), C.byref(ms)), "scopa_prof_device")
        return n.value, ms.value


FULL_STATE_DTYPE = np.dtype([("table", "<u8", (2,)), ("cap", "<u8", (2,)), ("hand", "<u4", (2,)), ("game", "<u4"), ("nh", "u1", (2,)),
                             ("nt", "u1"), ("deck_pos", "u1"), ("round", "u1"), ("last_capture", "u1"), ("scopas", "u1", (2,)),
                             ("step", "<u2"), ("terminal", "u1"), ("r2_p0", "i1"), ("flags", "u1"), ("pad", "u1", (7,))])
assert FULL_STATE_DTYPE.itemsize == 64


def full_deal_py_seed(seed):
    """40-card FullDeck(seed).cards as card ids (full_scopa_game.py:29-32)."""
    perm = np.zeros(40, np.uint8)
    rc = lib().scopa_full_deal_py_seed(int(seed), _ptr(perm))
    if rc:
        raise ScopaError(rc, "scopa_full_deal_py_seed")
    return perm


class FullState:
    """One packed FullScopa state driven through the host-side protocol (scopa_full_state_*)."""

    def __init__(self, seed=42, deck=None, game=0):
        self.deck = np.ascontiguousarray(full_deal_py_seed(seed) if deck is None else deck, np.uint8)
        self.s = np.zeros(1, FULL_STATE_DTYPE)
        rc = lib().scopa_full_state_init(_ptr(self.deck), int(game), _ptr(self.s))
        if rc:
            raise ScopaError(rc, "scopa_full_state_init")

    def step(self, action):
        rc = lib().scopa_full_state_step(_ptr(self.s), _ptr(self.deck), int(action))
        if rc:
            raise ScopaError(rc, "scopa_full_state_step")

    def legal(self, player=-1):
        out, n = (C.c_int32 * 3)(), C.c_int32()
        lib().scopa_full_state_legal(_ptr(self.s), int(player), C.byref(out), C.byref(n))
        return [out[i] for i in range(n.value)]

    def is_terminal(self):
        return bool(self.s[0]["terminal"])

    def current_player(self):
        return -4 if self.is_terminal() else int(self.s[0]["step"]) & 1

    def rewards(self):
        r = int(self.s[0]["r2_p0"])
        return [r / 2.0, -r / 2.0] if self.is_terminal() else [0, 0]

    def infoset_string(self, player):
        buf = C.create_string_buffer(256)
        lib().scopa_full_state_infoset_string(_ptr(self.s), int(player), buf, 256)
        return buf.value.decode()

    def snapshot(self):
        return unpack_full_state(self.s[0])


def unpack_full_state(s):
    tab = [int((int(s["table"][i // 10]) >> (6 * (i % 10))) & 63) for i in range(int(s["nt"]))]
    hands = [[int((int(s["hand"][p]) >> (6 * i)) & 63) for i in range(int(s["nh"][p]))] for p in range(2)]
    caps = [[c for c in range(40) if (int(s["cap"][p]) >> c) & 1] for p in range(2)]
    last = int(s["last_capture"])
    return dict(hands=hands, table=tab, caps=caps, scopas=[int(s["scopas"][0]), int(s["scopas"][1])], round=int(s["round"]),
                step=int(s["step"]), deck_remaining=40 - int(s["deck_pos"]), last=-1 if last == 255 else last)


TEAM_STATE_DTYPE = np.dtype([("history", "<u8"), ("table", "<u4"), ("hand", "<u2", (4,)), ("cap", "<u2", (4,)), ("nh", "u1", (4,)),
                             ("scopas", "u1", (4,)), ("nt", "u1"), ("step", "u1"), ("last_capture_team", "u1"), ("flags", "u1")])
assert TEAM_STATE_DTYPE.itemsize == 40


class TeamState:
    """One packed Team MiniScopa TPI state driven through the host-side protocol (scopa_team_state_*)."""

    def __init__(self, seed=42, perm=None):
        self.perm = np.ascontiguousarray(deal_py_seed(seed) if perm is None else perm, np.uint8)
        self.s = np.zeros(1, TEAM_STATE_DTYPE)
        rc = lib().scopa_team_state_init(_ptr(self.perm), _ptr(self.s))
        if rc:
            raise ScopaError(rc, "scopa_team_state_init")

    def copy(self):
        c = TeamState.__new__(TeamState)
        c.perm, c.s = self.perm, self.s.copy()
        return c

    def step(self, action):
        rc = lib().scopa_team_state_step(_ptr(self.s), int(action))
        if rc:
            raise ScopaError(rc, "scopa_team_state_step")

    def legal(self):
        out, n = (C.c_int32 * 4)(), C.c_int32()
        lib().scopa_team_state_legal(_ptr(self.s), C.byref(out), C.byref(n))
        return [out[i] for i in range(n.value)]

    def is_terminal(self):
        return bool(int(self.s[0]["flags"]) & 1)

    def seat(self):
        return int(self.s[0]["step"]) & 3

    def current_player(self):
        return -4 if self.is_terminal() else self.seat() >> 1

    def player_rewards(self):
        r = (C.c_int32 * 4)()
        lib().scopa_team_state_rewards_x2(_ptr(self.s), C.byref(r))
        return [r[i] / 2.0 for i in range(4)]

    def rewards(self):
        """per TEAM (TPIMiniScopaState.rewards, openspiel_team_mini_scopa.py:106-116)"""
        if not self.is_terminal():
            return [0, 0]
        r = self.player_rewards()
        return [(r[0] + r[1]) / 2, (r[2] + r[3]) / 2]

    def infoset_string(self, team):
        buf = C.create_string_buffer(256)
        lib().scopa_team_state_infoset_string(_ptr(self.s), int(team), buf, 256)
        return buf.value.decode()

    def history(self):
        h = int(self.s[0]["history"])
        return [(h >> (4 * i)) & 15 for i in range(int(self.s[0]["step"]))]

    def history_str(self):
        h = "-".join(map(str, self.history()))
        if self.is_terminal():
            return f"TERMINAL:{h}:" + ",".join(f"{r:.2f}" for r in self.rewards())
        return f"H:{h}:T{self.current_player()}"

    def snapshot(self):
        d = unpack_team_state(self.s[0])
        d.update(cur=self.current_player(), legal=self.legal(), info0=self.infoset_string(0), info1=self.infoset_string(1),
                 hist=self.history_str(), rewards=[float(r) for r in self.rewards()])
        return d


def unpack_team_state(s):
    hands = [[(int(s["hand"][p]) >> (4 * i)) & 15 for i in range(int(s["nh"][p]))] for p in range(4)]
    last = int(s["last_capture_team"])
    return dict(hands=hands, table=[(int(s["table"]) >> (4 * i)) & 15 for i in range(int(s["nt"]))],
                caps=[[c for c in range(16) if (int(s["cap"][p]) >> c) & 1] for p in range(4)], scopas=[int(x) for x in s["scopas"]],
                last=-1 if last == 255 else last, step=int(s["step"]), seat=int(s["step"]) & 3, term=bool(int(s["flags"]) & 1))


class MultiDeal:
    """n independent deals resident on one device, one workgroup per deal (scopa_multi_* in include/scopa.h)."""

    def __init__(self, ctx, n_deals):
        self.ctx, self.n = ctx, int(n_deals)
        self._L = lib()
        self._h = C.c_void_p()
        ctx._ck(self._L.scopa_multi_create(ctx._h, self.n, C.byref(self._h)), "scopa_multi_create")
        self.n_infosets = None
        import weakref
        ctx._children.append(weakref.ref(self))

    def close(self):
        if getattr(self, "_h", None):
            self._L.scopa_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def deal_py_seeds(self, seeds):
        s = np.ascontiguousarray(seeds, np.int64)
        assert s.size == self.n
        self.ctx._ck(self._L.scopa_multi_deal_py_seeds(self._h, _ptr(s)), "scopa_multi_deal_py_seeds")

    def set_perms(self, perms):
        p = np.ascontiguousarray(perms, np.uint8)
        assert p.shape == (self.n, 16)
        self.ctx._ck(self._L.scopa_multi_set_perms(self._h, _ptr(p)), "scopa_multi_set_perms")

    def perms(self):
        p = np.zeros((self.n, 16), np.uint8)
        self.ctx._ck(self._L.scopa_multi_perms_get(self._h, _ptr(p)), "scopa_multi_perms_get")
        return p

    def build(self):
        ninf = np.zeros(self.n, np.int32)
        self.ctx._ck(self._L.scopa_multi_build(self._h, _ptr(ninf)), "scopa_multi_build")
        self.n_infosets = ninf
        return ninf

    def cfr_exact_iterate(self, n_iters):
        self.ctx._ck(self._L.scopa_multi_cfr_exact_iterate(self._h, int(n_iters)), "scopa_multi_cfr_exact_iterate")

    def cfr_exact_iterate_lanes(self, n_iters):
        self.ctx._ck(self._L.scopa_multi_cfr_exact_iterate_lanes(self._h, int(n_iters)), "scopa_multi_cfr_exact_iterate_lanes")

    def cfr_sync_iterate(self, n_iters):
        self.ctx._ck(self._L.scopa_multi_cfr_sync_iterate(self._h, int(n_iters)), "scopa_multi_cfr_sync_iterate")

    def mccfr_iterate(self, batch, n_iters, seed=0x5C09A):
        self.ctx._ck(self._L.scopa_multi_mccfr_iterate(self._h, int(batch), int(n_iters), int(seed)), "scopa_multi_mccfr_iterate")

    def exploitability(self):
        out = np.zeros((self.n, 4))
        self.ctx._ck(self._L.scopa_multi_exploitability(self._h, _ptr(out)), "scopa_multi_exploitability")
        return out

    def tables_get(self, deal):
        I = int(self.n_infosets[deal])
        R, S, Lc, K = np.zeros((I, 4)), np.zeros((I, 4)), np.zeros((I, 4)), np.zeros(I, np.uint64)
        self.ctx._ck(self._L.scopa_multi_tables_get(self._h, int(deal), _ptr(R), _ptr(S), _ptr(Lc), _ptr(K)), "scopa_multi_tables_get")
        return R, S, Lc, K

    def counters(self):
        a, b = C.c_uint64(), C.c_uint64()
        self.ctx._ck(self._L.scopa_multi_counters(self._h, C.byref(a), C.byref(b)), "scopa_multi_counters")
        return a.value, b.value
