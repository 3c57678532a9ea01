"""The NTT tile planner (rust-kzg_amd/csrc/ntt_plan.h) without a GPU: the C++ tables equal the Python prototype's
(tools/ntt_plan_sim.py), every plan is a valid schedule of the butterfly network (each round's thread elements pair up
along the round's stage bits, every element is owned by exactly one thread, waves keep their elements inside a phase),
and the prototype's simulation of the planned passes equals a direct transform."""
import ctypes as C
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ntt_plan_sim as S


def dump(lib, kind, T):
    rounds = (C.c_int * 24)()
    tab = (C.c_uint16 * (6 * 1024 * 4))()
    lib.kzgamd_ntt_plan_dump.restype = C.c_int
    n = lib.kzgamd_ntt_plan_dump(kind, T, rounds, tab)
    return n, list(rounds), tab


PLANS = [(k, T) for k in (0, 1, 2) for T in range(0, 13 if k == 0 else 11)]


@pytest.mark.parametrize("kind,T", PLANS)
def test_cpp_planner_equals_prototype(kzg, kind, T):
    lib = kzg.lib()
    n, rounds, tab = dump(lib, kind, T)
    pl = S.plan_pass(kind, T)
    assert n == len(pl["rounds"])
    for r, R in enumerate(pl["rounds"]):
        bit = R["pos"] if R["M"] else R["pair_bits"][0]
        assert rounds[4 * r: 4 * r + 4] == [R["pos"], R["M"], int(R["barrier_after"]), bit], (kind, T, r)
        for u in range(1024):
            e = pl["tab"][r][u]
            got = tab[(r * 1024 + u) * 4: (r * 1024 + u) * 4 + 4]
            assert got == [e[0], e[2], S.swz(e[0]), S.swz(e[2])], (kind, T, r, u)
            assert e[1] == e[0] | 1 << bit and e[3] == e[2] | 1 << bit


@pytest.mark.parametrize("kind,T", PLANS)
def test_plan_is_a_schedule(kind, T):
    pl = S.plan_pass(kind, T)
    stages = []
    owner_prev = None
    for r, R in enumerate(pl["rounds"]):
        seen = {}
        for u in range(1024):
            for i in pl["tab"][r][u]:
                assert i not in seen
                seen[i] = u >> 6
        assert len(seen) == 4096
        if r and not pl["rounds"][r - 1]["barrier_after"]:
            assert seen == owner_prev  # same wave owns the same elements: exchange without a barrier
        owner_prev = seen
        stages += [R["pos"] + k for k in range(R["M"])]
        assert R["conflicts"] <= 1
    assert stages == list(range(T))
    assert sum(R["barrier_after"] for R in pl["rounds"]) <= 1
    assert S.swz(0) == 0 and sorted(S.swz(i) for i in range(4096)) == list(range(4096))
    for a, b in ((5, 77), (4095, 1234), (1 << 11, 3)):
        assert S.swz(a ^ b) == S.swz(a) ^ S.swz(b)


@pytest.mark.parametrize("L", [0, 1, 3, 7, 8, 9, 11, 12, 13, 15])
def test_simulated_passes_equal_a_direct_transform(L):
    import random

    rnd = random.Random(L)
    x = [rnd.randrange(S.P) for _ in range(1 << L)]
    assert S.ntt_sim(x, L) == S.ntt_ref(x, L)


@pytest.mark.parametrize("T", range(0, 13))
def test_cpp_das_planner_equals_prototype(kzg, T):
    lib = kzg.lib()
    rounds = (C.c_int * 72)()
    tab = (C.c_uint16 * (12 * 1024 * 4))()
    lib.kzgamd_ntt_das_plan_dump.restype = C.c_int
    n = lib.kzgamd_ntt_das_plan_dump(T, rounds, tab)
    pl = S.plan_das(T)
    assert n == len(pl["rounds"]) and 4 <= n <= 12
    bmap = S.brev_map(T)
    for r, R in enumerate(pl["rounds"]):
        bit = R["pos"] if R["M"] else R["pair_bits"][0]
        pm = bmap if R["part"] else (lambda i: i)
        flags = R["part"] | int(R["unit"]) << 1 | int(R["twist"]) << 2
        assert list(rounds[6 * r: 6 * r + 6]) == [R["pos"], R["M"], int(R["barrier_after"]), bit, flags,
                                                  S.swz(pm(1 << bit))], (T, r)
        for u in range(1024):
            e = pl["tab"][r][u]
            got = tab[(r * 1024 + u) * 4: (r * 1024 + u) * 4 + 4]
            assert got == [e[0], e[2], S.swz(pm(e[0])), S.swz(pm(e[2]))], (T, r, u)
            # the position map is XOR-linear: the other two elements sit at lds ^ lds_bit
            assert S.swz(pm(e[1])) == S.swz(pm(e[0])) ^ S.swz(pm(1 << bit))


@pytest.mark.parametrize("T", range(0, 13))
def test_das_plan_is_a_schedule(T):
    pl = S.plan_das(T)
    owner_prev, stages = None, [[], []]
    for r, R in enumerate(pl["rounds"]):
        pm = S.brev_map(T) if R["part"] else (lambda i: i)
        seen = {}
        for u in range(1024):
            for i in pl["tab"][r][u]:
                assert pm(i) not in seen
                seen[pm(i)] = u >> 6
        assert len(seen) == 4096
        if r and not pl["rounds"][r - 1]["barrier_after"]:
            assert seen == owner_prev
        owner_prev = seen
        stages[R["part"]] += [R["pos"] + k for k in range(R["M"])]
        assert R["conflicts"] <= 1
        assert not R["unit"] or R["pos"] == 0
    assert stages == [list(range(T)), list(range(T))]
    # one barrier inside each half at most, one between the halves
    assert sum(R["barrier_after"] for R in pl["rounds"]) <= 3
    assert sum(R["twist"] for R in pl["rounds"]) == 1


@pytest.mark.parametrize("L", [0, 1, 2, 5, 8, 11, 12])
def test_simulated_fused_das_equals_two_transforms(L):
    import random

    rnd = random.Random(100 + L)
    x = [rnd.randrange(S.P) for _ in range(1 << L)]
    assert S.das_sim(x, L) == S.das_ref(x, L)
