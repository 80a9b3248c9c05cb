"""SURVEY.md section 8 rows f2 (read-back / iterator / query API the reference sketches at tsdf.hpp:120-161) and f3 (mesh extraction
in place of LVR2, lvr2.cpp:15-133).  f2 is checked against the oracle's own octree: the leaves it exports, looked up / interpolated
/ marched by a numpy restatement of the same arithmetic.  f3 is parity-unpinned (LVR2 is absent): checked by properties."""
import os

import numpy as np
import pytest

from chad_tsdf_amd import synth
from tests.util import SD_TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def chad():
    import chad_tsdf_amd

    chad_tsdf_amd._lib.load()
    return chad_tsdf_amd


@pytest.fixture(scope="module")
def oracle():
    from oracle import pyoracle

    pyoracle.lib()
    return pyoracle


class LeafTable:
    """Leaves (mc ascending, sd, w) as a look-up by voxel: what an octree walk of the oracle's own tree answers."""

    def __init__(self, oracle, leaves):
        self.oracle = oracle
        self.mc, self.sd, self.w = leaves

    def at(self, voxels):
        """(sd f32, w u32, found) for int32 [n,3] voxel coordinates."""
        code = self.oracle.morton_encode_ivec(np.ascontiguousarray(voxels, dtype=np.int32))
        i = np.searchsorted(self.mc, code)
        i = np.minimum(i, len(self.mc) - 1)
        found = self.mc[i] == code
        return np.where(found, self.sd[i], np.float32(np.nan)).astype(np.float32), np.where(found, self.w[i], 0).astype(np.uint32), found

    def trilinear(self, p, res):
        """query_kernels.hpp sample_trilinear in numpy f32, same operation order."""
        p = np.asarray(p, dtype=np.float32)
        recip = np.float32(1.0 / np.float64(np.float32(res)))
        res = np.float32(res)
        v = np.floor(p * recip).astype(np.int32)
        f = (p - v.astype(np.float32) * res) * recip
        c, wmin, ok = [], np.full(len(p), 0xFFFFFFFF, np.uint32), np.ones(len(p), bool)
        for k in range(8):
            sd, w, found = self.at(v + np.array([k & 1, (k >> 1) & 1, k >> 2], np.int32))
            c.append(sd); wmin = np.minimum(wmin, w); ok &= found
        fx, fy, fz = f[:, 0], f[:, 1], f[:, 2]
        with np.errstate(invalid="ignore"):
            c00 = c[0] + (c[1] - c[0]) * fx; c10 = c[2] + (c[3] - c[2]) * fx; c01 = c[4] + (c[5] - c[4]) * fx; c11 = c[6] + (c[7] - c[6]) * fx
            c0 = c00 + (c10 - c00) * fy; c1 = c01 + (c11 - c01) * fy
            out = c0 + (c1 - c0) * fz
        return np.where(ok, out, np.float32(np.nan)).astype(np.float32), np.where(ok, wmin, 0).astype(np.uint32)

    def raycast(self, o, d, t_max, res):
        """query_kernels.hpp k_raycast, ray by ray (small cases only)."""
        res = np.float32(res)
        dt = res * np.float32(0.5)
        hits = np.full(len(o), np.nan, np.float32)
        steps = int(np.float32(t_max) / dt) + 1
        t = np.arange(steps, dtype=np.float32) * dt
        for i in range(len(o)):
            p = o[i][None, :] + d[i][None, :] * t[:, None]
            sd, _ = self.trilinear(p.astype(np.float32), res)
            ok = ~np.isnan(sd)
            for k in range(1, steps):
                if ok[k] and ok[k - 1] and sd[k - 1] > 0 and sd[k] <= 0:
                    hits[i] = t[k - 1] + (t[k] - t[k - 1]) * (sd[k - 1] / (sd[k - 1] - sd[k]))
                    break
        return hits


def sphere_scan(n=1_000_000, r=5.0, seed=420):
    """The reference's demo scene (main.cpp:9-29): points on a sphere around the sensor at the origin."""
    return (synth.ray_directions(seed, n) * r).astype(np.float32), np.zeros(3, np.float32)


def test_iterator_walks_a_closed_submap(chad, oracle):
    """TSDFMap::iterator (tsdf.hpp:126-151): position, signed distance, weight of every leaf of a closed submap, ascending Morton
    order, read through the DAG; ++ / -- / + / - / last as the reference sketches them."""
    pts, pos = synth.cfg2_frame(150_000, seed=17)
    m = chad.TSDFMap(0.05, 0.10)
    o = oracle.OracleMap(0.05, 0.10)
    m.insert(pts, pos); o.insert(pts, pos)
    m.close_submap(); o.close_submap()
    omc, osd, ow = o.export_leaves(0)
    got = np.concatenate(list(m.iter_leaves(0, chunk=50_000)))
    assert len(got) == len(omc)
    vox = oracle.morton_decode_ivec(omc)                       # MortonCode::decode (morton.hpp:29-37)
    want_pos = vox.astype(np.float32) * np.float32(0.05)       # octree.hpp:157 / lvr2.cpp:78-79: vec3(leaf_chunk) * voxelsize
    assert np.array_equal(np.stack([got["x"], got["y"], got["z"]], 1), want_pos)
    assert np.all(got["weight"] == 255)                        # Q6: what the reference's DAG holds
    assert np.abs(got["signed_distance"] - osd).max() <= 0.10 / 127 * 1.01 + 0.10 * SD_TOL    # 8-bit quantisation (cluster.hpp:13-26)
    mc_dag, sd_dag = m.dag_leaves(0)
    assert np.array_equal(got["signed_distance"], sd_dag)
    from chad_tsdf_amd.tsdf import LeafIterator

    it = LeafIterator(m, 0)
    n = len(omc)
    assert it.valid and it.get()[0] == tuple(want_pos[0])
    assert it.advance(1000) == 1000 and it.get()[0] == tuple(want_pos[1000])
    assert it.advance(-400) == -400 and it.get()[0] == tuple(want_pos[600])
    it.seek(last=True)
    assert it.get()[0] == tuple(want_pos[n - 1])
    assert it.advance(5) == 0 and not it.valid                  # one past the end: the reference's end()
    it.seek()
    assert it.advance(-1) == 0 and not it.valid
    it.close()
    with pytest.raises(chad._lib.ChadError):
        LeafIterator(m, 1)                                      # tsdf.cpp:106-110: a faulty root address throws
    keep = LeafIterator(m, 0)
    m.reset()                                                   # the DAG is gone: the iterator is dead, not dangling
    assert not keep.valid
    with pytest.raises(chad._lib.ChadError):
        keep.advance(1)
    m.insert(pts, pos)
    m.close_submap()
    orphan = LeafIterator(m, 0)
    assert orphan.valid
    m.close()                                                   # ... and an iterator that outlives its map is cut loose by the destroy
    assert not orphan.valid
    orphan.close()


def test_queries_of_the_active_submap_match_the_oracles_octree(chad, oracle):
    res, trunc = 0.05, 0.10
    m = chad.TSDFMap(res, trunc)
    o = oracle.OracleMap(res, trunc)
    for k in range(3):
        pts, pos = synth.cfg2_frame(200_000, seed=60 + k, sensor=(0.3 + 0.1 * k, -0.2, 0.1))
        m.insert(pts, pos); o.insert(pts, pos)
    table = LeafTable(oracle, o.export_leaves())
    mine = LeafTable(oracle, m.export_leaves())
    rng = np.random.default_rng(5)
    surface = pts[rng.integers(0, len(pts), 40_000)]
    q = (surface + rng.normal(0, 0.04, surface.shape)).astype(np.float32)          # around the surface: inside the truncation band
    q = np.concatenate([q, rng.uniform(-6, 6, (2000, 3)).astype(np.float32), np.array([[np.nan, 0, 0], [1e30, 0, 0]], np.float32)])
    # the leaf of the voxel containing the point
    sd, w = m.query_sd(q, trilinear=False)
    with np.errstate(invalid="ignore"):
        vox = np.floor(q * np.float32(1.0 / np.float64(np.float32(res)))).astype(np.int64).clip(-2**20 + 1, 2**20 - 1).astype(np.int32)
    osd, ow, found = table.at(vox)
    found[-2:] = False
    assert np.array_equal(w != 0, found) and np.array_equal(w[found], ow[found])
    assert np.isnan(sd[~found]).all()
    assert (np.abs(sd[found] - osd[found]) / np.maximum(np.abs(osd[found]), trunc)).max() <= SD_TOL
    msd, mw, _ = mine.at(vox)
    assert np.array_equal(sd[found].view(np.uint32), msd[found].view(np.uint32))   # the device's own export, bit for bit
    assert found[:40_000].mean() > 0.9
    # trilinear between the eight leaves around the point
    sd, w = m.query_sd(q[:-2], trilinear=True)
    wsd, ww = mine.trilinear(q[:-2], res)
    ok = ~np.isnan(wsd)
    assert np.array_equal(np.isnan(sd), ~ok) and np.array_equal(w, ww)
    assert np.abs(sd[ok] - wsd[ok]).max() <= 1e-6 * trunc                          # same f32 arithmetic on the same leaves
    osd2, _ = table.trilinear(q[:-2], res)
    assert np.array_equal(np.isnan(osd2), ~ok)
    assert (np.abs(sd[ok] - osd2[ok]) / trunc).max() <= 2 * SD_TOL                 # against the oracle's leaves: eight 1e-4 leaves combined
    assert ok.mean() > 0.5
    m.close()


def test_raycast_hits_the_surface(chad, oracle):
    """tsdf.hpp:157 "raycast ... + physics hit": rays from the sensor into the cfg2 room hit its walls where the box says, and
    the marching arithmetic equals its restatement over the oracle's leaves."""
    res, trunc = 0.05, 0.10
    pts, pos = synth.cfg2_frame(800_000, seed=71)
    m = chad.TSDFMap(res, trunc)
    m.insert(pts, pos)
    d = synth.ray_directions(99, 3000).astype(np.float32)
    o3 = np.repeat(pos[None, :], len(d), 0).astype(np.float32)
    t = m.raycast(o3, d, 9.0)
    c = synth.CFG2
    with np.errstate(divide="ignore"):
        tt = np.where(d > 0, (np.asarray(c["hi"]) - pos) / d, np.where(d < 0, (np.asarray(c["lo"]) - pos) / d, np.inf)).min(axis=1)
    hit = ~np.isnan(t)
    assert hit.mean() > 0.7     # (far, grazing parts of floor and ceiling are sparsely observed: incomplete cells give no sample)
    err = np.abs(t[hit] - tt[hit])
    assert err.max() < 3 * res and np.quantile(err, 0.99) < res    # the wall, to within the voxel size (worst where two walls meet: mixed normals)
    assert err.mean() < 0.2 * res
    none = m.raycast(o3[:10], d[:10], 0.5)                     # nothing within half a metre of the sensor
    assert np.isnan(none).all()
    o = oracle.OracleMap(res, trunc)
    o.insert(pts, pos)
    table = LeafTable(oracle, o.export_leaves())
    want = table.raycast(o3[:40], d[:40], 9.0, res)
    assert np.array_equal(np.isnan(want), np.isnan(t[:40]))
    both = ~np.isnan(want)
    assert np.abs(want[both] - t[:40][both]).max() <= 0.02 * res   # same march; leaves differ by <= 1e-4 * trunc
    m.close()


def edge_use(tri):
    e = np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]])
    und = np.sort(e, axis=1).astype(np.uint64)
    key = und[:, 0] << np.uint64(32) | und[:, 1]
    _, inv, count = np.unique(key, return_inverse=True, return_counts=True)
    return e, key, count, inv


@pytest.mark.parametrize("from_dag,tetrahedra", [(False, False), (True, False), (False, True), (True, True)])
def test_mesh_of_the_reference_demo_sphere(chad, tmp_path, from_dag, tetrahedra):
    """f3 by properties (LVR2 is absent: parity-unpinned).  The reference's demo (main.cpp:9-39): points on a 5 m sphere around
    the sensor, TSDFMap(0.05, 0.1), insert, save("mesh.ply").  The mesh -- marching cubes by default, marching tetrahedra on request:
    every vertex within one voxel of r = 5 m; a 2-manifold with boundary (every edge in one or two triangles, used once per direction
    when shared); consistently oriented towards the sensor; all of the sphere covered; a unit normal per vertex, towards the sensor;
    the PLY holds exactly that mesh, normals included (lvr2.cpp:296-313)."""
    res, trunc = 0.05, 0.10
    pts, pos = sphere_scan()
    m = chad.TSDFMap(res, trunc, mesh_from_dag=from_dag, mesh_tetrahedra=tetrahedra)
    m.insert(pts, pos)
    path = str(tmp_path / "mesh.ply")
    m.save(path)                                               # closes the submap, finalizes, reconstructs submap 0 (tsdf.cpp:76-86)
    v, tri = m.mesh(0)
    assert len(v) > 100_000 and len(tri) > (200_000 if tetrahedra else 150_000) and tri.max() < len(v)
    r = np.linalg.norm(v.astype(np.float64), axis=1)
    assert np.abs(r - 5.0).max() < res and np.abs(r - 5.0).mean() < 0.25 * res
    assert len(np.unique(v, axis=0)) == len(v)                 # welded: every vertex once
    e, key, count, inv = edge_use(tri)
    assert count.max() == 2                                    # no edge in three triangles
    assert (count == 2).mean() > 0.95                          # boundary edges only where cells were culled
    shared = count[inv] == 2                                   # a shared edge is walked once in each direction: consistent orientation
    fwd = (e[:, 0] < e[:, 1])
    per_edge_fwd = np.bincount(inv[shared], weights=fwd[shared].astype(np.float64), minlength=len(count))
    assert np.all(per_edge_fwd[count == 2] == 1)
    a, b, c = (v[tri[:, k]].astype(np.float64) for k in range(3))
    normal = np.cross(b - a, c - a)
    area = 0.5 * np.linalg.norm(normal, axis=1)
    inward = np.einsum("ij,ij->i", normal, -(a + b + c)) > 0           # towards the sensor at the centre (increasing sd)
    assert area[inward].sum() / area.sum() > 0.9999
    assert abs(area.sum() / (4 * np.pi * 25.0) - 1.0) < 0.05            # the whole sphere, once
    # a normal per vertex: the normalised sum of the unit normals of the faces around it (lvr2.cpp:296-297), pointing at the sensor
    vn = m.mesh_normals(0)
    assert vn.shape == v.shape
    unit = normal / np.maximum(np.linalg.norm(normal, axis=1), 1e-300)[:, None]
    summed = np.zeros_like(v, dtype=np.float64)
    for k in range(3):
        np.add.at(summed, tri[:, k], unit)
    summed /= np.maximum(np.linalg.norm(summed, axis=1), 1e-300)[:, None]
    assert np.abs(vn - summed).max() < 1e-4
    assert np.abs(np.linalg.norm(vn.astype(np.float64), axis=1) - 1.0).max() < 1e-5
    towards_sensor = -np.einsum("ij,ij->i", vn.astype(np.float64), v.astype(np.float64)) / np.linalg.norm(v.astype(np.float64), axis=1)
    assert towards_sensor.min() > 0.5 and towards_sensor.mean() > 0.98
    # the file: binary little-endian PLY with these vertices, their normals and the faces
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    assert b"element vertex %d" % len(v) in head and b"element face %d" % len(tri) in head and b"binary_little_endian" in head
    assert head.index(b"property float z") < head.index(b"property float nx") < head.index(b"property float nz") < head.index(b"element face")
    rows = np.frombuffer(body[: 24 * len(v)], "<f4").reshape(-1, 6)
    assert np.array_equal(rows[:, :3], v) and np.array_equal(rows[:, 3:], vn)
    faces = np.frombuffer(body[24 * len(v):], np.dtype([("n", "u1"), ("i", "<u4", 3)]))
    assert len(faces) == len(tri) and np.all(faces["n"] == 3) and np.array_equal(faces["i"], tri)
    m.close()


def test_mesh_is_marching_cubes_over_the_references_grid(chad, oracle):
    """f3: the default triangulation against the oracle's restatement of what LVR2's boxes do with ChadGrid's cells (lvr2.cpp:246-251) --
    the published marching-cubes algorithm: case bit i = corner i's distance positive (FastBox::getIndex), the 256-case table
    (scripts/mc_table_check.py checks every case of it), corners by linear interpolation along cell edges.  Grid AND triangulation from
    the oracle (ChadGrid over its own NodeLevels, then oracle_marching_cubes); mesh from the device, read through the DAG as the reference
    reads it: the two must be the same set of oriented triangles, coordinate bits included (triangles that lose a corner to the welding --
    a surface through a grid point -- are dropped on both sides).  Agreement with the published algorithm, not with LVR2's output:
    LVR2 is absent, and BilinearFastBox::optimizePlanarFaces (lvr2.cpp:262) is not modelled."""
    res, trunc = 0.05, 0.10
    for pts, pos in (sphere_scan(400_000), synth.cfg2_frame(300_000)):
        m = chad.TSDFMap(res, trunc, mesh_from_dag=True)
        m.insert(pts, pos)
        mc, sd, w = m.export_leaves()
        m.close_submap()
        o = oracle.OracleMap(res, trunc)
        o.close_submap_from_leaves(mc, sd, w)
        codes, corners, _ = o.chad_grid(0)
        want = o.marching_cubes(codes, corners, res) + np.float32(0.0)   # (-0 -> +0: the same place)
        v, tri = m.mesh(0)
        got = v[tri] + np.float32(0.0)
        assert len(got) > 50_000
        # every vertex as its index among all distinct vertices; a triangle as its three indices, rotated so that the smallest comes first
        both = np.concatenate([got.reshape(-1, 3), want.reshape(-1, 3)]).view(np.uint32)
        _, index = np.unique(both, axis=0, return_inverse=True)
        index = index.reshape(-1, 3)
        index = index[(index[:, 0] != index[:, 1]) & (index[:, 1] != index[:, 2]) & (index[:, 0] != index[:, 2])]   # (only `want` still has such)
        first = np.argmin(index, axis=1)
        rolled = np.stack([index[np.arange(len(index)), (first + k) % 3] for k in range(3)], axis=1)
        g, w_ = rolled[:len(got)], rolled[len(got):]
        order = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]
        assert len(g) == len(got) and len(w_) == len(g), (len(got), len(w_))
        assert np.array_equal(order(g), order(w_))
        m.close()


def test_mesh_vertices_on_cell_edges_are_the_marching_cubes_vertices_of_the_references_grid(chad, oracle):
    """f3 with CHAD_FLAG_MESH_TETRAHEDRA, what can be pinned of that triangulation: the VERTEX SET.  Whatever case table a marching-cubes box uses (LVR2's
    FastBox::getSurface, called at lvr2.cpp:247), its vertices are the points where sd = 0 crosses one of the twelve edges of a complete
    cell, placed by linear interpolation between the edge's two query points -- a definition that involves no table.  Computed here in
    f64 from the ORACLE's ChadGrid (lvr2.cpp:15-133 restated: cells, corner values in the vertex order of lvr2.cpp:92-102), that set
    must be exactly the set of the device mesh's vertices that lie on cell edges (the marching tetrahedra of mesh_kernels.hpp add
    vertices on face and body diagonals, which no marching-cubes mesh has; a cube's twelve edges are edges of the Kuhn split too).
    A restated definition, not upstream's output: "agreement with the oracle", like the grid itself."""
    from scipy.spatial import cKDTree

    res, trunc = 0.05, 0.10
    pts, pos = sphere_scan(400_000)
    m = chad.TSDFMap(res, trunc, mesh_from_dag=True, mesh_tetrahedra=True)
    m.insert(pts, pos)
    mc, sd, w = m.export_leaves()
    m.close_submap()
    o = oracle.OracleMap(res, trunc)
    o.close_submap_from_leaves(mc, sd, w)
    codes, corners, _ = o.chad_grid(0)
    low = chad.tsdf.decode_morton(codes).astype(np.float64)                       # lowest corner of every complete cell, in voxels
    off = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], np.float64)   # lvr2.cpp:92-102
    edges = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
    want = []
    d = corners.astype(np.float64)
    for a, b in edges:
        cross = (d[:, a] < 0) != (d[:, b] < 0)                                     # inside = negative signed distance (behind the surface)
        t = d[cross, a] / (d[cross, a] - d[cross, b])
        pa, pb = (low[cross] + off[a]) * res, (low[cross] + off[b]) * res
        want.append(pa + (pb - pa) * t[:, None])
    want = np.unique(np.round(np.concatenate(want), 7), axis=0)                    # an edge belongs to up to four cells
    v, tri = m.mesh(0)
    v = v.astype(np.float64)
    on_lattice = np.abs(v / res - np.round(v / res)) < 1e-4
    on_edge = on_lattice.sum(axis=1) >= 2                                          # two coordinates on the voxel lattice: a cell edge
    assert 10_000 < len(want) and 0.3 < on_edge.mean() < 0.9
    far_from_mesh = cKDTree(v[on_edge]).query(want)[0]
    far_from_want = cKDTree(want).query(v[on_edge])[0]
    assert far_from_mesh.max() < 5e-6, f"{(far_from_mesh >= 5e-6).sum()} marching-cubes vertices have no counterpart in the mesh"
    assert far_from_want.max() < 5e-6, f"{(far_from_want >= 5e-6).sum()} mesh vertices on cell edges are no marching-cubes vertices"
    m.close()


@pytest.mark.parametrize("scene", ["sphere", "cfg2"])
def test_mesh_grid_is_the_references_chad_grid(chad, oracle, scene):
    """f3, the half that IS reference code: ChadGrid's constructor (lvr2.cpp:15-133) walks the submap's DAG, makes a query point per
    leaf at its voxel corner with the de-quantised byte (cluster.hpp:33-52), puts it into the eight cells around it and culls the
    cells with a missing corner (lvr2.cpp:115-129).  The oracle restates exactly that over its own NodeLevels; the device hands out
    the grid its mesh kernel cuts (chad_map_mesh_cells).  With the mesh read through the DAG (CHAD_FLAG_MESH_FROM_DAG, what the
    reference does) the complete-cell sets must be identical and every corner value bit-equal; from the exact means the cell set
    is the same and a corner differs by at most the 8-bit step.  (The triangulation behind it is LVR2's: parity unpinned.)"""
    pts, pos = sphere_scan(400_000) if scene == "sphere" else synth.cfg2_frame(300_000)
    res, trunc = 0.05, 0.10
    m = chad.TSDFMap(res, trunc, mesh_from_dag=True)
    m.insert(pts, pos)
    mc, sd, w = m.export_leaves()
    m.close_submap()
    o = oracle.OracleMap(res, trunc)
    o.close_submap_from_leaves(mc, sd, w)     # the same leaves: the DAGs are word for word the same (test_gpu_finalize.py)
    assert m.dag_roots(0) == o.dag_roots(0)
    want_codes, want_corners, query_points = o.chad_grid(0)
    assert query_points == len(mc) and len(want_codes) > 10_000
    codes, corners = m.mesh_cells(0)
    assert np.array_equal(codes, want_codes), "complete-cell sets differ"
    assert np.array_equal(corners.view(np.uint32), want_corners.view(np.uint32)), "corner values differ from the reference's de-quantisation"
    m.close()
    exact = chad.TSDFMap(res, trunc)          # the library's default: the exact means of the dense archive
    exact.insert(pts, pos)
    exact.close_submap()
    codes2, corners2 = exact.mesh_cells(0)
    assert np.array_equal(codes2, want_codes)
    assert np.abs(corners2 - want_corners).max() <= trunc / 127 + 1e-6
    exact.close()
    # ... and against the oracle's OWN inserts: the same leaf set, hence the same cells; a byte may fall on the other side of a step
    # (the signed distances agree to 1e-4 * trunc, not bit for bit).  On the sphere that is rare; the box room's walls lie exactly on
    # voxel faces, so whole planes of voxels sit at signed distances of exactly -res / 2, 0, +res / 2 ..., i.e. on the half-way points
    # of the byte code (-0.05 m -> 63.5): there a difference in the last bit decides the byte, and a quarter of the corners flip
    own = oracle.OracleMap(res, trunc)
    own.insert(pts, pos)
    own.close_submap()
    own_codes, own_corners, _ = own.chad_grid(0)
    assert np.array_equal(codes, own_codes)
    assert np.abs(corners - own_corners).max() <= trunc / 127 + 1e-6
    if scene == "sphere":
        assert (corners != own_corners).mean() < 2e-3


def test_mesh_of_a_room_and_of_the_active_submap(chad):
    """A box room seen from inside: planar walls (every vertex within a voxel of a wall plane), and the active submap (-1) meshes
    like the closed one it becomes."""
    res, trunc = 0.05, 0.10
    pts, pos = synth.cfg2_frame(600_000, seed=23)
    # the room's walls lie exactly on voxel corners (5.0 = 100 voxels): the surface would pass exactly through grid points, where
    # welded vertices coincide and no triangulation is a manifold.  Move the room off the lattice.
    shift = np.array([0.0137, 0.0071, 0.0113], np.float32)
    pts, pos = pts + shift, pos + shift
    m = chad.TSDFMap(res, trunc)
    m.insert(pts, pos)
    v_act, t_act = m.mesh(-1)
    m.close_submap()
    v, tri = m.mesh(0)
    assert np.array_equal(v, v_act) and np.array_equal(tri, t_act)
    c = synth.CFG2
    lo, hi = np.asarray(c["lo"]) + shift, np.asarray(c["hi"]) + shift
    dist = np.minimum(np.abs(v - lo), np.abs(v - hi)).min(axis=1)
    # on the walls -- except where the scan is sparse and grazing (far floor and ceiling): neighbourhoods below 8 points get the
    # per-point fallback normal (normals.hpp:127-134), which tilts the field by up to the truncation distance there
    assert dist.max() <= 3 * res and np.quantile(dist, 0.95) < res and np.median(dist) < 0.1 * res and len(tri) > 100_000
    _, _, count, _ = edge_use(tri)
    assert count.max() == 2
    m.close()


def test_mesh_and_iterator_when_only_the_dag_is_kept(chad):
    """CHAD_FLAG_DAG_ONLY releases a closed submap's dense bricks once it is folded into the DAG (what the reference keeps): the
    mesh is then reconstructed from the leaves read back through the DAG, as the reference's own mesher does (lvr2.cpp:30-90), and
    equals the mesh a map with CHAD_FLAG_MESH_FROM_DAG makes from the same scan."""
    pts, pos = sphere_scan(300_000)
    a = chad.TSDFMap(0.05, 0.10, dag_only=True)
    b = chad.TSDFMap(0.05, 0.10, mesh_from_dag=True)
    for m in (a, b):
        m.insert(pts, pos)
        m.close_submap()
    va, ta = a.mesh(0)
    vb, tb = b.mesh(0)
    assert len(ta) > 50_000 and np.array_equal(va, vb) and np.array_equal(ta, tb)
    with pytest.raises(chad._lib.ChadError):
        a.export_leaves(0)            # the dense copy is gone ...
    assert len(np.concatenate(list(a.iter_leaves(0)))) == len(b.export_leaves(0)[0])   # ... the DAG is not
    a.close(); b.close()


def test_space_carving_extension_matches_its_oracle(chad, oracle):
    """SURVEY.md section 8 f4 (upstream roadmap, never built there: README.md:59-60).  The definition -- every ray gives the voxels
    it crosses before its truncation band one observation of +trunc where they already hold a leaf, nothing is created -- is stated
    twice, in the oracle (carve_frame) and in the HIP library (k_carve): leaf sets and weights must agree exactly, signed distances
    within 1e-4.  Scene: a room scanned from one pose, then from a pose from which many rays cross the first scan's band."""
    res, trunc = 0.05, 0.10
    m = chad.TSDFMap(res, trunc, space_carving=True)
    plain = chad.TSDFMap(res, trunc)
    o = oracle.OracleMap(res, trunc)
    o.set_carving(True)
    total = 0
    for k, sensor in enumerate([(0.3, -0.2, 0.1), (2.5, 1.0, -0.4), (-3.0, -2.0, 0.6)]):
        pts, pos = synth.cfg2_frame(120_000, seed=80 + k, sensor=sensor)
        if k == 2:   # an obstacle that was there in the first two scans' free space: rays of later frames pass through its band
            pts = np.concatenate([pts, synth.box_scan(30_000, (-0.5, -0.5, -0.5), (0.5, 0.5, 0.5), (0.0, 0.0, 0.0), 99) + np.float32(0.0)]).astype(np.float32)
        m.insert(pts, pos); plain.insert(pts, pos); o.insert(pts, pos)
        total += m.stats()["updates"]
    got, want, base = m.export_leaves(), o.export_leaves(), plain.export_leaves()
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
    assert np.array_equal(got[0], base[0])                       # carving creates nothing ...
    carved = got[2] > base[2]
    assert 1_000 < carved.sum() and np.all(got[2] >= base[2])    # ... it adds observations to leaves that rays pass through
    assert int(got[2].sum()) == total                             # (they count as updates)
    assert np.all(got[1][carved] >= base[1][carved] - 1e-6)      # towards +trunc: free space
    err = np.abs(got[1] - want[1]) / np.maximum(np.abs(want[1]), trunc)
    assert err.max() <= SD_TOL
    m.close(); plain.close()


def test_submap_positions_and_re_posing(chad):
    """Loop-closure support: the poses of a submap's frames come back (Submap::positions, submap.hpp:110),
    and a rigid correction of a closed submap moves its leaves and its mesh -- and nothing else."""
    m = chad.TSDFMap(0.05, 0.10)
    poses = []
    for k in range(0, 130, 10):      # 0.5 m per frame: the submap closes when frame 110 arrives
        pts, pos = synth.cfg3_frame(k, 30_000)
        m.insert(pts, pos)
        poses.append(pos)
    m.sync()
    assert m.closed_submaps == 1
    assert np.array_equal(m.submap_positions(0), np.stack(poses[:11])) and np.array_equal(m.submap_positions(-1), np.stack(poses[11:]))
    before = np.concatenate(list(m.iter_leaves(0)))
    v0, t0 = m.mesh(0)
    c, s_ = np.float32(np.cos(0.3)), np.float32(np.sin(0.3))
    R = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], np.float32)
    t = np.array([0.5, -1.25, 0.125], np.float32)
    with pytest.raises(chad._lib.ChadError):
        m.set_submap_pose(-1, R, t)              # only closed submaps
    m.set_submap_pose(0, R, t)
    after = np.concatenate(list(m.iter_leaves(0)))
    p0 = np.stack([before["x"], before["y"], before["z"]], 1)
    p1 = np.stack([after["x"], after["y"], after["z"]], 1)
    assert np.allclose(p1, p0 @ R.T + t, atol=1e-5) and np.array_equal(after["signed_distance"], before["signed_distance"])
    v1, t1 = m.mesh(0)
    assert np.array_equal(t1, t0) and np.allclose(v1, v0 @ R.T + t, atol=1e-5)
    assert np.array_equal(m.export_leaves(0)[0], m.export_leaves(0)[0])   # the stored voxels do not move
    m.reset()
    assert len(m.submap_positions(-1)) == 0
    m.close()
