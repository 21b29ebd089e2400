"""CPU-side checks of the product library: it loads, exports every symbol include/scopa.h declares, its
host-side single-state glue follows the reference (golden playouts), and it refuses to run solvers without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, unpack_state


def test_library_exports_every_declared_symbol(sl):
    hdr = open(os.path.join(ROOT, "include", "scopa.h")).read()
    declared = sorted(set(re.findall(r"\b(scopa_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 35
    L = sl.lib()
    for name in declared:
        assert hasattr(L, name), f"libscopa_hip.so does not export {name}"
    assert sorted(sl.SYMBOLS) == declared
    assert L.scopa_abi_version() == 1


def test_state_struct_is_16_bytes(sl):
    assert C.sizeof(sl.State16) == 16 and sl.STATE_DTYPE.itemsize == 16


def test_deal_matches_cpython_shuffle(sl, golden):
    for seed, perm in golden.json("deals.json").items():
        assert list(sl.deal_py_seed(int(seed))) == perm


def test_host_state_protocol_on_golden_playouts(sl, golden):
    L = sl.lib()
    buf = C.create_string_buffer(96)
    for c in golden.json("playouts.json"):
        s = sl.State16()
        perm = sl.deal_py_seed(c["seed"])
        assert L.scopa_state_init(perm.ctypes.data_as(C.c_void_p), C.byref(s)) == 0
        for a, tr in zip(c["actions"], c["trail"]):
            assert L.scopa_state_step(C.byref(s), a) == 0
            sn = unpack_state(s)
            for k in sn:
                assert sn[k] == tr[k], (c["seed"], c["actions"], k)
            assert bool(L.scopa_state_is_terminal(C.byref(s))) == tr["term"]
            assert L.scopa_state_current_player(C.byref(s)) == tr["cur"]
            for pl, k in ((0, "info0"), (1, "info1")):
                L.scopa_state_infoset_string(C.byref(s), pl, buf, 96)
                assert buf.value.decode() == tr[k]
        r2 = (C.c_int32 * 2)()
        L.scopa_state_rewards_x2(C.byref(s), C.byref(r2))
        assert [r2[0] / 2, r2[1] / 2] == c["rewards"]


def test_host_state_protocol_on_cloned_playouts(sl, golden):
    """scopa_state_clone = MiniScopaState.clone (openspiel_mini_scopa.py:97-115): the copy's step limit is 16 (:108), carried in the packed
    state (SCOPA_STEP_CLONED); every ply of the reference's cloned playouts through the C ABI."""
    L = sl.lib()
    buf = C.create_string_buffer(96)
    for c in golden.json("playouts_cloned.json"):
        s = sl.State16()
        perm = sl.deal_py_seed(c["seed"])
        assert L.scopa_state_init(perm.ctypes.data_as(C.c_void_p), C.byref(s)) == 0
        for a, cb, tr in zip(c["actions"], c["clone_before"], c["trail"]):
            if cb:
                t = sl.State16()
                assert L.scopa_state_clone(C.byref(s), C.byref(t)) == 0
                s = t
            assert L.scopa_state_step(C.byref(s), a) == 0
            assert bool(s.step & sl.STEP_CLONED) == tr["cloned"]
            sn = unpack_state(s)
            sn["step"] &= sl.STEP_COUNT_MASK
            for k in sn:
                assert sn[k] == tr[k], (c["seed"], c["actions"], k)
            assert bool(L.scopa_state_is_terminal(C.byref(s))) == tr["term"]
            assert L.scopa_state_current_player(C.byref(s)) == tr["cur"]
            out, n = (C.c_int32 * 4)(), C.c_int32()
            for pl, k in ((0, "legal0"), (1, "legal1")):
                assert L.scopa_state_legal(C.byref(s), pl, C.byref(out), C.byref(n)) == 0
                assert [out[i] for i in range(n.value)] == tr[k]
            for pl, k in ((0, "info0"), (1, "info1")):
                L.scopa_state_infoset_string(C.byref(s), pl, buf, 96)
                assert buf.value.decode() == tr[k]
            r2 = (C.c_int32 * 2)()
            L.scopa_state_rewards_x2(C.byref(s), C.byref(r2))
            assert [r2[0] / 2, r2[1] / 2] == tr["rewards"]
        t, tc = sl.State16(), c["terminal_clone"]   # a terminal state's clone stays terminal; a further action is a dead step
        assert L.scopa_state_clone(C.byref(s), C.byref(t)) == 0 and L.scopa_state_step(C.byref(t), tc["action"]) == 0
        sn = unpack_state(t)
        sn["step"] &= sl.STEP_COUNT_MASK
        assert all(sn[k] == tc[k] for k in sn) and L.scopa_state_is_terminal(C.byref(t)) == 1
        L.scopa_state_rewards_x2(C.byref(t), C.byref(r2))
        assert [r2[0] / 2, r2[1] / 2] == tc["rewards"]


def test_host_rules_match_oracle_on_random_states(sl, oracle):
    """capture rule / step on 20k random (deal, action string) pairs, incl. illegal actions, vs the oracle."""
    L = sl.lib()
    rng = np.random.RandomState(5)
    for trial in range(2500):
        perm = rng.permutation(16).astype(np.uint8)
        s = sl.State16()
        L.scopa_state_init(perm.ctypes.data_as(C.c_void_p), C.byref(s))
        o = oracle.State(perm=perm)
        for ply in range(8):
            legal = o.legal()
            a = int(legal[rng.randint(len(legal))]) if rng.rand() < 0.8 else int(rng.randint(16))
            L.scopa_state_step(C.byref(s), a)
            o.step(a)
            assert unpack_state(s) == o.snapshot(), (list(perm), ply, a)
            out, n = (C.c_int32 * 4)(), C.c_int32()
            L.scopa_state_legal(C.byref(s), -1, C.byref(out), C.byref(n))
            assert [out[i] for i in range(n.value)] == o.legal()


def test_capture_rule_on_arbitrary_tables(sl, oracle):
    """States no game can reach -- tables with several cards of one rank, full 8-card tables, any mover -- so that the bitmask
    subset-sum DP is compared with the oracle's literal list DP on everything the packed state can express, not only on what
    play produces: 30k random (hands, table, card) triples, the whole successor state compared."""
    L = sl.lib()
    rng = np.random.RandomState(77)
    for trial in range(30000):
        cards = rng.permutation(16)
        nt = int(rng.randint(0, 9))                       # 0..8 table cards (the packed state holds 8)
        nh0, nh1 = int(rng.randint(1, 5)), int(rng.randint(1, 5))
        if nt + nh0 + nh1 > 16:
            nt = 16 - nh0 - nh1
        table, h0, h1 = cards[:nt], cards[nt:nt + nh0], cards[nt + nh0:nt + nh0 + nh1]
        step = int(rng.randint(0, 7))
        mover = step & 1
        hand = h0 if mover == 0 else h1
        card = int(hand[rng.randint(len(hand))])
        s = sl.State16()
        s.hand[0] = sum(int(c) << (4 * i) for i, c in enumerate(h0)); s.hand[1] = sum(int(c) << (4 * i) for i, c in enumerate(h1))
        s.table = sum(int(c) << (4 * i) for i, c in enumerate(table))
        s.nh[0], s.nh[1], s.nt, s.step = nh0, nh1, nt, step
        o = oracle.State(perm=np.arange(16, dtype=np.uint8))
        for i in range(4):
            o.s.hand[0][i] = int(h0[i]) if i < nh0 else 0
            o.s.hand[1][i] = int(h1[i]) if i < nh1 else 0
        for i in range(8):
            o.s.table[i] = int(table[i]) if i < nt else 0
        o.s.nh[0], o.s.nh[1], o.s.nt, o.s.step = nh0, nh1, nt, step
        o.s.ncap[0] = o.s.ncap[1] = o.s.scopas[0] = o.s.scopas[1] = 0
        if nt == 8 and not o.capture(card):
            continue                                      # a ninth table card: not representable (and not reachable) -- see scopa_rules.h
        L.scopa_state_step(C.byref(s), card)
        o.step(card)
        assert unpack_state(s) == o.snapshot(), (list(table), list(hand), card)


def test_bad_arguments_are_rejected(sl):
    L = sl.lib()
    s = sl.State16()
    bad = np.zeros(16, np.uint8)  # not a permutation
    assert L.scopa_state_init(bad.ctypes.data_as(C.c_void_p), C.byref(s)) == sl.SCOPA_EINVAL
    assert L.scopa_state_step(C.byref(s), 99) == sl.SCOPA_EINVAL
    h = C.c_void_p()
    assert L.scopa_ctx_create(-1, None, C.byref(h)) == sl.SCOPA_EINVAL


def test_no_gpu_means_no_solver(sl):
    """The solver path has no CPU fallback: without a device, context creation fails with SCOPA_ENODEV."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(sl.ScopaError) as e:
        sl.Context(0)
    assert e.value.status == sl.SCOPA_ENODEV
