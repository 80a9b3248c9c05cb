"""The range bookkeeping of the DAG's device pool (chad_tsdf_amd/csrc/range_pool.hpp, the host half of alloc.hpp's DevPool) against a
model, built with g++ (and under AddressSanitizer / UBSan by `make asan`): ranges handed out never overlap and stay inside the
reservation; a release merges with its free neighbours; what the pool did not hand out -- a second release, a misaligned or foreign
offset, a length that reaches into a free neighbour -- is refused and changes nothing."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import pyoracle   # (only for the sanitizer switch of `make asan`)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "pool_check.cpp")
ALIGN = 4096
NONE = (1 << 64) - 1


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("poolcheck") / "libpoolcheck.so")
    san = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"] if pyoracle.SANITIZE else ["-O2"]
    subprocess.check_call([shutil.which("g++") or "g++", "-std=c++17", "-Wall", "-Werror", *san, "-shared", "-fPIC", SRC, "-o", out])
    lib = C.CDLL(out)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    lib.poolcheck_new.restype = vp
    lib.poolcheck_new.argtypes = [u64]
    lib.poolcheck_delete.argtypes = [vp]
    lib.poolcheck_take.restype = u64
    lib.poolcheck_take.argtypes = [vp, u64]
    lib.poolcheck_give.argtypes = [vp, u64, u64]
    lib.poolcheck_reset.argtypes = [vp]
    lib.poolcheck_state.argtypes = [vp, vp, u32]
    return lib


def _state(lib, pool, room=4096):
    out = np.zeros(3 + 2 * room, np.uint64)
    lib.poolcheck_state(pool, out.ctypes.data, room)
    n = int(out[2])
    assert n <= room
    return int(out[0]), int(out[1]), [(int(out[3 + 2 * k]), int(out[4 + 2 * k])) for k in range(n)]


def _rounded(n):
    return (max(n, 1) + ALIGN - 1) // ALIGN * ALIGN


def _check(lib, pool, cap, held):
    """held: offset -> rounded length of everything handed out and not given back"""
    used, peak, free = _state(lib, pool)
    assert used == sum(held.values()) and peak >= used
    spans = sorted(list(held.items()) + free)
    at = 0
    for off, length in spans:   # handed-out and free ranges tile the reservation: no overlap, no gap
        assert off == at and length > 0 and off % ALIGN == 0, (off, length, at)
        at = off + length
    assert at == cap
    for (a, la), (b, _) in zip(free, free[1:]):   # neighbours are merged
        assert a + la < b


@pytest.mark.parametrize("seed", range(6))
def test_random_takes_and_gives_against_a_model(lib, seed):
    rng = np.random.default_rng(seed)
    cap = int(rng.integers(64, 4096)) * ALIGN
    pool = lib.poolcheck_new(cap)
    held = {}
    peak = 0
    try:
        for step in range(1200):
            if held and rng.random() < 0.45:
                off = list(held)[int(rng.integers(len(held)))]
                length = held.pop(off)
                assert lib.poolcheck_give(pool, off, length - int(rng.integers(0, ALIGN))) == 1   # (any length that rounds to the same)
            else:
                want = int(rng.integers(1, cap // 8 if rng.random() < 0.9 else cap // 2))
                off = lib.poolcheck_take(pool, want)
                _, _, free = _state(lib, pool)
                if off == NONE:
                    assert all(length < _rounded(want) for _, length in free), "a request was refused although a range was long enough"
                else:
                    assert off not in held
                    held[off] = _rounded(want)
            peak = max(peak, sum(held.values()))
            if step % 50 == 0:
                _check(lib, pool, cap, held)
        _check(lib, pool, cap, held)
        assert _state(lib, pool)[1] == peak
        for off, length in list(held.items()):
            assert lib.poolcheck_give(pool, off, length) == 1
        assert _state(lib, pool) == (0, peak, [(0, cap)])
    finally:
        lib.poolcheck_delete(pool)


def test_what_the_pool_did_not_hand_out_is_refused(lib):
    cap = 64 * ALIGN
    pool = lib.poolcheck_new(cap)
    try:
        a = lib.poolcheck_take(pool, 3 * ALIGN)
        b = lib.poolcheck_take(pool, 5 * ALIGN)
        c = lib.poolcheck_take(pool, ALIGN)
        held = {a: 3 * ALIGN, b: 5 * ALIGN, c: ALIGN}
        _check(lib, pool, cap, held)
        before = _state(lib, pool)
        assert lib.poolcheck_give(pool, a + 1, ALIGN) == 0            # misaligned
        assert lib.poolcheck_give(pool, cap, ALIGN) == 0              # outside
        assert lib.poolcheck_give(pool, cap - ALIGN, 2 * ALIGN) == 0  # reaches past the end
        assert lib.poolcheck_give(pool, a, cap) == 0                  # longer than everything handed out
        assert _state(lib, pool) == before
        assert lib.poolcheck_give(pool, b, 5 * ALIGN) == 1
        del held[b]
        _check(lib, pool, cap, held)
        before = _state(lib, pool)
        assert lib.poolcheck_give(pool, b, 5 * ALIGN) == 0            # a second release
        assert lib.poolcheck_give(pool, b + ALIGN, ALIGN) == 0        # the inside of a free range
        assert lib.poolcheck_give(pool, a, 4 * ALIGN) == 0            # a length that reaches into the free neighbour
        assert _state(lib, pool) == before
        lib.poolcheck_reset(pool)
        assert _state(lib, pool)[0] == 0 and _state(lib, pool)[2] == [(0, cap)]
        assert lib.poolcheck_take(pool, cap) == 0 and lib.poolcheck_take(pool, 1) == NONE
    finally:
        lib.poolcheck_delete(pool)


def test_large_stores_come_from_the_top_small_ones_from_the_bottom(lib):
    cap = 1024 * ALIGN
    pool = lib.poolcheck_new(cap)
    try:
        small = lib.poolcheck_take(pool, ALIGN)
        large = lib.poolcheck_take(pool, cap // 16)
        assert small == 0 and large == cap - cap // 16
    finally:
        lib.poolcheck_delete(pool)
