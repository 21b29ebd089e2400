"""ctypes binding of the CPU oracle (oracle/scopa_oracle.c).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg.  The product package `scopa_amd` never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libscopa_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("scopa_oracle.c", "scopa_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libscopa_oracle.so"])
    return _SO


class _State(C.Structure):
    _fields_ = [("hand", (C.c_int8 * 4) * 2), ("nh", C.c_int8 * 2), ("table", C.c_int8 * 8), ("nt", C.c_int8),
                ("ncap", C.c_int8 * 2), ("scopas", C.c_int8 * 2), ("step", C.c_int8), ("max_steps", C.c_int8)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.og_tree_build.restype = C.c_void_p
        L.og_tree_build.argtypes = [C.c_void_p]
        L.og_tree_free.argtypes = [C.c_void_p]
        L.og_philox_uniform.restype = C.c_double
        L.og_philox_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.og_policy_value.restype = C.c_double
        L.og_exploitability.restype = C.c_double
        L.og_mccfr_replay.restype = C.c_int64
        L.og_cfr_exact_from.restype = C.c_double
        L.og_deal_py_seed.argtypes = [C.c_int64, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def deal_py_seed(seed):
    perm = np.zeros(16, np.uint8)
    lib().og_deal_py_seed(int(seed), _p(perm))
    return perm


def philox4x32_10(ctr, key):
    c = np.array(ctr, np.uint32); k = np.array(key, np.uint32); o = np.zeros(4, np.uint32)
    lib().og_philox4x32_10(_p(c), _p(k), _p(o))
    return o


def philox_uniform(seed, c0, c1, c2, c3):
    return lib().og_philox_uniform(int(seed), int(c0), int(c1), int(c2), int(c3))


class State:
    """Single game state driven through the oracle's C functions."""

    def __init__(self, perm=None, seed=42):
        self.s = _State()
        if perm is None:
            perm = deal_py_seed(seed)
        self.perm = np.ascontiguousarray(perm, np.uint8)
        lib().og_reset(C.byref(self.s), _p(self.perm))

    def clone(self):
        """MiniScopaState.clone() (openspiel_mini_scopa.py:97-115): the copy's env has max_steps = 16 (:108)."""
        o = State.__new__(State)
        o.s = _State()
        lib().og_clone(C.byref(self.s), C.byref(o.s))
        o.perm = self.perm
        return o

    def legal(self, player=-1):
        out = (C.c_int * 4)()
        n = lib().og_legal(C.byref(self.s), int(player), out)
        return [out[i] for i in range(n)]

    def step(self, a):
        lib().og_step(C.byref(self.s), int(a))

    def is_terminal(self):
        return bool(lib().og_is_terminal(C.byref(self.s)))

    def current_player(self):
        return lib().og_current_player(C.byref(self.s))

    def rewards(self):
        r2 = (C.c_int * 2)()
        lib().og_rewards_x2(C.byref(self.s), r2)
        return [r2[0] / 2.0, r2[1] / 2.0]

    def infoset_string(self, player=-1):
        buf = C.create_string_buffer(96)
        lib().og_infoset_string(C.byref(self.s), int(player), buf)
        return buf.value.decode()

    def capture(self, card):
        idx = (C.c_int * 8)()
        n = lib().og_capture(C.byref(self.s), int(card), idx)
        return [idx[i] for i in range(n)]

    def snapshot(self):
        s = self.s
        hands = [[int(s.hand[p][i]) for i in range(s.nh[p])] for p in range(2)]
        table = [int(s.table[i]) for i in range(s.nt)]
        return dict(hands=hands, table=table, ncap=[int(s.ncap[0]), int(s.ncap[1])],
                    scopas=[int(s.scopas[0]), int(s.scopas[1])], step=int(s.step))

    @property
    def max_steps(self):
        return int(self.s.max_steps)


class Tree:
    """Flat game tree of one deal in reference DFS order + the oracle solvers over it."""

    def __init__(self, perm=None, seed=42):
        if perm is None:
            perm = deal_py_seed(seed)
        self.perm = np.ascontiguousarray(perm, np.uint8)
        L = lib()
        self.h = C.c_void_p(L.og_tree_build(_p(self.perm)))
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        L.og_tree_counts(self.h, C.byref(a), C.byref(b), C.byref(c))
        self.n_nodes, self.n_decision, self.n_infosets = a.value, b.value, c.value
        n = self.n_nodes
        self.term = np.zeros(n, np.int8); self.player = np.zeros(n, np.int8); self.nlegal = np.zeros(n, np.int8)
        self.depth = np.zeros(n, np.int8); self.infoset = np.zeros(n, np.int16); self.legal = np.zeros((n, 4), np.int8)
        self.child = np.zeros((n, 4), np.int32); self.r2 = np.zeros((n, 2), np.int8)
        L.og_tree_export(self.h, _p(self.term), _p(self.player), _p(self.nlegal), _p(self.depth), _p(self.infoset),
                         _p(self.legal), _p(self.child), _p(self.r2))
        I = self.n_infosets
        self.infoset_nlegal = np.zeros(I, np.int8); self.infoset_legal = np.zeros((I, 4), np.int8)
        self.infoset_player = np.zeros(I, np.int8)
        L.og_tree_infoset_meta(self.h, _p(self.infoset_nlegal), _p(self.infoset_legal), _p(self.infoset_player))
        buf = C.create_string_buffer(96)
        self.infoset_strings = []
        for i in range(I):
            L.og_tree_infoset_string(self.h, i, buf)
            self.infoset_strings.append(buf.value.decode())

    def __del__(self):
        try:
            lib().og_tree_free(self.h)
        except Exception:
            pass

    def states(self):
        n = self.n_nodes
        out = dict(hands=np.zeros((n, 2, 4), np.int8), nh=np.zeros((n, 2), np.int8), table=np.zeros((n, 8), np.int8),
                   nt=np.zeros(n, np.int8), ncap=np.zeros((n, 2), np.int8), scopas=np.zeros((n, 2), np.int8),
                   step=np.zeros(n, np.int8))
        lib().og_tree_states(self.h, _p(out["hands"]), _p(out["nh"]), _p(out["table"]), _p(out["nt"]), _p(out["ncap"]),
                             _p(out["scopas"]), _p(out["step"]))
        return out

    # ---- tables -------------------------------------------------------------------
    def tables(self):
        I = self.n_infosets
        R, S, Lc = np.zeros((I, 4)), np.zeros((I, 4)), np.zeros((I, 4))
        lib().og_tables_init(self.h, _p(R), _p(S), _p(Lc))
        return R, S, Lc

    def cfr_exact(self, R, S, Lc, n_iters):
        rv = np.zeros((n_iters, 2))
        lib().og_cfr_exact(self.h, _p(R), _p(S), _p(Lc), int(n_iters), _p(rv))
        return rv

    def cfr_exact_from(self, R, S, Lc, path, trav, r0, r1):
        pa = np.ascontiguousarray(path, np.int32)
        return lib().og_cfr_exact_from(self.h, _p(R), _p(S), _p(Lc), _p(pa), int(pa.size), int(trav), C.c_double(r0), C.c_double(r1))

    def cfr_sync(self, R, S, n_iters):
        lib().og_cfr_sync(self.h, _p(R), _p(S), int(n_iters))

    def mccfr_replay(self, R, S, n_iters, uniforms):
        u = np.ascontiguousarray(uniforms, np.float64)
        return lib().og_mccfr_replay(self.h, _p(R), _p(S), int(n_iters), _p(u), C.c_int64(u.size))

    def mccfr_batched(self, R, S, seed, iter0, n_iters, batch):
        vis = C.c_uint64(0)
        lib().og_mccfr_batched(self.h, _p(R), _p(S), C.c_uint64(seed), C.c_uint32(iter0), C.c_uint32(n_iters),
                               C.c_uint32(batch), C.byref(vis))
        return vis.value

    def mccfr_batched_delta(self, R, seed, iteration, b0, nb):
        I = self.n_infosets
        dR, dS = np.zeros((I, 4)), np.zeros((I, 4))
        dv, tv = C.c_uint64(0), C.c_uint64(0)
        lib().og_mccfr_batched_delta(self.h, _p(R), _p(dR), _p(dS), C.c_uint64(seed), C.c_uint32(iteration),
                                     C.c_uint32(b0), C.c_uint32(nb), C.byref(dv), C.byref(tv))
        return dR, dS, dv.value, tv.value

    def mccfr_batched_trace(self, R, seed, iteration, b, traverser):
        nodes = np.zeros(512, np.int32); acts = np.zeros(512, np.int8)
        n = lib().og_mccfr_batched_trace(self.h, _p(R), C.c_uint64(seed), C.c_uint32(iteration), C.c_uint32(b),
                                         int(traverser), _p(nodes), _p(acts), 512)
        return nodes[:n].copy(), acts[:n].copy()

    def sdcfr_traverse(self, nets, traverser, seed=0, iteration=0, b0=0, nb=1, uniforms=None):
        """nets: float32 [2][13776] (two torch state dicts flattened in their own order) -> (feat, regret, mask rows, values, visits)"""
        w = np.ascontiguousarray(nets, np.float32).reshape(2, -1)
        assert w.shape[1] == 13776
        feat, reg, mask = np.zeros((nb * 41, 34), np.float32), np.zeros((nb * 41, 16), np.float32), np.zeros((nb * 41, 16), np.float32)
        vals, vis = np.zeros(nb, np.float32), C.c_uint64(0)
        u = None if uniforms is None else np.ascontiguousarray(uniforms, np.float64)
        L = lib()
        L.og_sdcfr_traverse.restype = C.c_int64
        rows = L.og_sdcfr_traverse(self.h, _p(w), int(traverser), C.c_uint64(seed), C.c_uint32(iteration), C.c_uint32(b0), C.c_uint32(nb),
                                   _p(u), C.c_int64(0 if u is None else u.size), _p(feat), _p(reg), _p(mask), _p(vals), C.byref(vis))
        assert rows == nb * 41
        return feat, reg, mask, vals, vis.value

    def average_policy(self, S):
        P = np.zeros_like(S)
        lib().og_average_policy(self.h, _p(S), _p(P))
        return P

    def policy_value(self, P):
        return lib().og_policy_value(self.h, _p(np.ascontiguousarray(P)))

    def exploitability(self, P):
        br = np.zeros(2)
        e = lib().og_exploitability(self.h, _p(np.ascontiguousarray(P)), _p(br))
        return e, br


# ---- FullScopa (40 cards) ---------------------------------------------------------------------------------------------
class _FullState(C.Structure):
    _fields_ = [("deck", C.c_uint8 * 40), ("deck_pos", C.c_int8), ("hand", (C.c_int8 * 3) * 2), ("nh", C.c_int8 * 2),
                ("table", C.c_int8 * 40), ("nt", C.c_int8), ("cap", (C.c_uint8 * 40) * 2), ("ncap", C.c_int8 * 2),
                ("scopas", C.c_int8 * 2), ("round_number", C.c_int8), ("last_capture", C.c_int8), ("step", C.c_int16),
                ("terminal", C.c_int8), ("r2", C.c_int * 2)]


def full_deal_py_seed(seed):
    perm = np.zeros(40, np.uint8)
    lib().og_deal_py_seed  # ensure loaded
    lib().ogf_deal_py_seed(C.c_int64(int(seed)), _p(perm))
    return perm


class FullState:
    def __init__(self, perm=None, seed=42):
        self.s = _FullState()
        self.perm = np.ascontiguousarray(full_deal_py_seed(seed) if perm is None else perm, np.uint8)
        lib().ogf_reset(C.byref(self.s), _p(self.perm))

    def legal(self, player=-1):
        out = (C.c_int * 3)()
        n = lib().ogf_legal(C.byref(self.s), int(player), out)
        return [out[i] for i in range(n)]

    def step(self, a):
        lib().ogf_step(C.byref(self.s), int(a))

    def is_terminal(self):
        return bool(self.s.terminal)

    def current_player(self):
        return -4 if self.s.terminal else (self.s.step & 1)

    def rewards(self):
        return [self.s.r2[0] / 2.0, self.s.r2[1] / 2.0] if self.s.terminal else [0, 0]

    def infoset_string(self, player):
        buf = C.create_string_buffer(256)
        lib().ogf_infoset_string(C.byref(self.s), int(player), buf)
        return buf.value.decode()

    def snapshot(self):
        s = self.s
        return dict(hands=[[int(s.hand[p][i]) for i in range(s.nh[p])] for p in range(2)], table=[int(s.table[i]) for i in range(s.nt)],
                    caps=[sorted(int(s.cap[p][i]) for i in range(s.ncap[p])) for p in range(2)], scopas=[int(s.scopas[0]), int(s.scopas[1])],
                    round=int(s.round_number), step=int(s.step), deck_remaining=40 - int(s.deck_pos), last=int(s.last_capture))


# ---- Team MiniScopa TPI (4 seats, 16 plies) ---------------------------------------------------------------------------
class _TeamState(C.Structure):
    _fields_ = [("hand", (C.c_int8 * 4) * 4), ("nh", C.c_int8 * 4), ("table", C.c_int8 * 16), ("nt", C.c_int8),
                ("cap", (C.c_int8 * 16) * 4), ("ncap", C.c_int8 * 4), ("scopas", C.c_int8 * 4), ("last_capture_team", C.c_int8),
                ("step", C.c_int8), ("terminal", C.c_int8), ("history", C.c_int8 * 16), ("nhist", C.c_int8), ("r2", C.c_int * 4)]


class TeamState:
    def __init__(self, perm=None, seed=42):
        self.s = _TeamState()
        self.perm = np.ascontiguousarray(deal_py_seed(seed) if perm is None else perm, np.uint8)
        lib().ogt_reset(C.byref(self.s), _p(self.perm))

    def legal(self):
        out = (C.c_int * 4)()
        n = lib().ogt_legal(C.byref(self.s), out)
        return [out[i] for i in range(n)]

    def step(self, a):
        lib().ogt_step(C.byref(self.s), int(a))

    def is_terminal(self):
        return bool(self.s.terminal)

    def current_player(self):
        return lib().ogt_current_player(C.byref(self.s))

    def rewards(self):
        """TPIMiniScopaState.rewards (openspiel_team_mini_scopa.py:106-116): per TEAM"""
        return [self.s.r2[0] / 2.0, self.s.r2[2] / 2.0] if self.s.terminal else [0, 0]

    def player_rewards(self):
        return [self.s.r2[i] / 2.0 for i in range(4)]

    def infoset_string(self, team):
        buf = C.create_string_buffer(256)
        lib().ogt_infoset_string(C.byref(self.s), int(team), buf)
        return buf.value.decode()

    def history_str(self):
        h = "-".join(str(int(self.s.history[i])) for i in range(self.s.nhist))
        if self.s.terminal:
            return f"TERMINAL:{h}:" + ",".join(f"{r:.2f}" for r in self.rewards())
        return f"H:{h}:T{self.current_player()}"

    def snapshot(self):
        s = self.s
        return dict(hands=[[int(s.hand[p][i]) for i in range(s.nh[p])] for p in range(4)], table=[int(s.table[i]) for i in range(s.nt)],
                    caps=[sorted(int(s.cap[p][i]) for i in range(s.ncap[p])) for p in range(4)], scopas=[int(x) for x in s.scopas],
                    last=int(s.last_capture_team), step=int(s.step), seat=int(s.step) % 4, term=bool(s.terminal),
                    cur=self.current_player(), legal=self.legal(), info0=self.infoset_string(0), info1=self.infoset_string(1),
                    hist=self.history_str(), rewards=[float(r) for r in self.rewards()])

