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
