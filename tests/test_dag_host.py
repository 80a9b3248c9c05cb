"""Host half of Submap::finalize (chad_tsdf_amd/csrc/dag.hpp) against the oracle's restatement of submap.hpp:10-106 and
levels.hpp: the same leaves must give the same level arrays, address for address.  Runs without a GPU: the cluster stream
the device would deliver is made here from the oracle's leaves with the arithmetic of cluster.hpp:13-26."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from chad_tsdf_amd import synth
from oracle import pyoracle
from tests.util import clusters_of_leaves

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "dag_check.cpp")


@pytest.fixture(scope="module")
def dagcheck(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("dagcheck") / "libdagcheck.so")
    san = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"] if pyoracle.SANITIZE else ["-O2"]   # `make asan`
    subprocess.check_call([shutil.which("g++") or "g++", "-std=c++17", *san, "-shared", "-fPIC", SRC, "-o", out])
    lib = C.CDLL(out)
    vp, sz, u32 = C.c_void_p, C.c_size_t, C.c_uint32
    lib.dagcheck_create.restype = vp
    lib.dagcheck_destroy.argtypes = [vp]
    lib.dagcheck_add_submap.argtypes = [vp, vp, vp, vp, sz, vp]
    lib.dagcheck_level_words.restype = sz
    lib.dagcheck_level_words.argtypes = [vp, u32]
    lib.dagcheck_level_copy.argtypes = [vp, u32, vp]
    lib.dagcheck_level_counts.argtypes = [vp, u32, vp]
    lib.dagcheck_walk.restype = sz
    lib.dagcheck_walk.argtypes = [vp, u32, vp, vp, sz]
    lib.dagcheck_cursor_script.restype = sz
    lib.dagcheck_cursor_script.argtypes = [vp, u32, u32, vp, sz, vp, vp, vp]
    lib.dagcheck_sd_of_bits.restype = C.c_float
    lib.dagcheck_sd_of_bits.argtypes = [C.c_uint8, C.c_float]
    return lib


class HostDag:
    def __init__(self, lib):
        self.lib, self.h = lib, lib.dagcheck_create()

    def __del__(self):
        self.lib.dagcheck_destroy(self.h)

    def add(self, keys, tsdf, wts):
        roots = np.zeros(2, dtype=np.uint32)
        self.lib.dagcheck_add_submap(self.h, keys.ctypes.data, tsdf.ctypes.data, wts.ctypes.data if wts is not None else None, len(keys), roots.ctypes.data)
        return int(roots[0]), int(roots[1])

    def level(self, l):
        out = np.empty(self.lib.dagcheck_level_words(self.h, l), dtype=np.uint64 if l == 20 else np.uint32)
        self.lib.dagcheck_level_copy(self.h, l, out.ctypes.data)
        return out

    def counts(self, l):
        out = np.zeros(2, dtype=np.uint32)
        self.lib.dagcheck_level_counts(self.h, l, out.ctypes.data)
        return int(out[0]), int(out[1])

    def walk(self, root, n):
        mc = np.empty(n, dtype=np.uint64)
        bits = np.empty(n, dtype=np.uint8)
        k = self.lib.dagcheck_walk(self.h, root, mc.ctypes.data, bits.ctypes.data, n)
        return k, mc, bits


def run_cursor(lib, host, roots, script):
    script = np.asarray(script, dtype=np.int32)
    mc = np.empty(len(script), dtype=np.uint64)
    sd = np.empty(len(script), dtype=np.uint8)
    w = np.empty(len(script), dtype=np.uint8)
    lib.dagcheck_cursor_script(host.h, roots[0], roots[1], script.ctypes.data, len(script), mc.ctypes.data, sd.ctypes.data, w.ctypes.data)
    return mc, sd, w


def assert_same_dag(host, oracle):
    for l in range(21):
        assert np.array_equal(host.level(l), oracle.dag_level(l)), f"level {l}"
        assert host.counts(l) == oracle.dag_level_counts(l), f"level {l} uniques/dupes"


@pytest.mark.parametrize("saturate", [False, True])
def test_two_submaps_share_one_dag_like_the_reference(dagcheck, saturate):
    o = pyoracle.OracleMap(0.05, 0.10)
    o.set_saturate_weights(saturate)
    host = HostDag(dagcheck)
    for k, (n, seed) in enumerate([(60_000, 3), (40_000, 4)]):
        pts, pos = synth.cfg2_frame(n, seed=seed)
        o.insert(pts, pos)
        mc, sd, w = o.export_leaves()
        o.close_submap()
        roots = host.add(*clusters_of_leaves(mc, sd, w, 0.10, saturate))
        assert roots == o.dag_roots(k)
        assert_same_dag(host, o)
    # the same submap again: nothing new anywhere, every node found
    before = [len(host.level(l)) for l in range(21)]
    roots = host.add(*clusters_of_leaves(mc, sd, w, 0.10, saturate))
    assert roots == o.dag_roots(1) and [len(host.level(l)) for l in range(21)] == before


def test_empty_submap_and_single_voxel(dagcheck):
    o = pyoracle.OracleMap(0.05, 0.10)
    host = HostDag(dagcheck)
    o.insert(np.zeros((0, 3), np.float32), (0, 0, 0))       # a pose, no points: finalize of an empty octree
    o.close_submap()
    e = np.zeros(0, dtype=np.uint64)
    assert host.add(e, e.reshape(0, 8), None) == o.dag_roots(0)
    assert_same_dag(host, o)
    mc = np.array([0x7000000000000007, 0x0000000000000000], dtype=np.uint64)   # the two ends of the code range, unsorted on purpose
    sd = np.array([0.1, -0.1], dtype=np.float32)
    w = np.array([1, 300], dtype=np.uint32)
    o.close_submap_from_leaves(mc, sd, w)
    order = np.argsort(mc)
    assert host.add(*clusters_of_leaves(mc[order], sd[order], w[order], 0.10)) == o.dag_roots(1)
    assert_same_dag(host, o)


def test_walk_returns_every_leaf_in_morton_order(dagcheck):
    o = pyoracle.OracleMap(0.05, 0.10)
    pts, pos = synth.cfg2_frame(50_000, seed=9)
    o.insert(pts, pos)
    mc, sd, w = o.export_leaves()
    host = HostDag(dagcheck)
    root, _ = host.add(*clusters_of_leaves(mc, sd, w, 0.10))
    k, wmc, bits = host.walk(root, len(mc))
    assert k == len(mc) and np.array_equal(wmc, mc)
    back = np.array([dagcheck.dagcheck_sd_of_bits(int(b), 0.10) for b in np.unique(bits)], dtype=np.float32)
    lut = dict(zip(np.unique(bits).tolist(), back.tolist()))
    got = np.array([lut[int(b)] for b in bits], dtype=np.float32)
    assert np.abs(got - sd).max() <= 0.10 / 127 + 1e-6       # one quantisation step (cluster.hpp:13-26: truncation, not rounding)


def test_g8_golden_levels(dagcheck):
    g = np.load(os.path.join(ROOT, "tests", "golden", "g8_finalize.npz"))
    host = HostDag(dagcheck)
    for k in range(2):
        roots = host.add(*clusters_of_leaves(g[f"s{k}_mc"], g[f"s{k}_sd"], g[f"s{k}_w"].astype(np.uint32), 0.10))
        assert roots == tuple(int(x) for x in g[f"s{k}_roots"])
        for l in range(21):
            assert np.array_equal(host.level(l), g[f"s{k}_level{l}"]), (k, l)
            assert host.counts(l) == tuple(int(x) for x in g[f"s{k}_counts"][l])


@pytest.mark.parametrize("saturate", [False, True])
def test_cursor_is_the_iterator_the_reference_sketches(dagcheck, saturate):
    """DagCursor (dag.hpp) = TSDFMap::iterator of tsdf.hpp:126-151: begin() lands on the first leaf like tsdf.cpp:104-146, ++ walks
    every leaf in ascending Morton order (the order of dag_walk / lvr2.cpp:30-90), -- walks back, + / - jump, past either end it
    turns invalid.  Two submaps share the DAG; each is walked from its own roots."""
    NEG = -(2 ** 31)
    host = HostDag(dagcheck)
    subs = []
    for n, seed in ((30_000, 5), (20_000, 6)):
        o = pyoracle.OracleMap(0.05, 0.10)
        pts, pos = synth.cfg2_frame(n, seed=seed)
        o.insert(pts, pos)
        mc, sd, w = o.export_leaves()
        subs.append((mc, sd, w, host.add(*clusters_of_leaves(mc, sd, w, 0.10, saturate))))
    for mc, sd, w, roots in subs:
        n = len(mc)
        k, wmc, wbits = host.walk(roots[0], n)
        assert k == n
        fwd = run_cursor(dagcheck, host, roots, [0] + [1] * n)                  # first, then n times ++: the last one runs off the end
        assert np.array_equal(fwd[0][:n], mc) and np.array_equal(fwd[1][:n], wbits) and fwd[0][n] == np.uint64(2**64 - 1)
        want_w = np.minimum(w, 254).astype(np.uint8) if saturate else np.full(n, 255, np.uint8)   # Q6: the reference stores 0xff (submap.hpp:93)
        assert np.array_equal(fwd[2][:n], want_w)
        back = run_cursor(dagcheck, host, roots, [NEG] + [-1] * n)              # last, then n times --
        assert np.array_equal(back[0][:n], mc[::-1]) and back[0][n] == np.uint64(2**64 - 1)
        rng = np.random.default_rng(3)
        at, script, want = 0, [0], [0]
        for _ in range(400):                                                     # random jumps that stay inside
            step = int(rng.integers(-at, n - at))
            if step == 0:
                continue
            at += step
            script.append(step); want.append(at)
        got = run_cursor(dagcheck, host, roots, script)
        assert np.array_equal(got[0], mc[np.asarray(want)])
    # an empty submap: begin() is invalid at once
    e = np.zeros(0, dtype=np.uint64)
    roots = host.add(e, e.reshape(0, 8), None)
    assert run_cursor(dagcheck, host, roots, [0, NEG])[0].tolist() == [2**64 - 1] * 2
