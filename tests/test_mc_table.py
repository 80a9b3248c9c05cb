"""The marching-cubes case table behind chad_map_mesh: every one of its 256 cases checked for what a typing
error would break, the two generated include files (the product's and the oracle's own) up to date, and the oracle's restatement of
the algorithm on cells whose answer is known by hand."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script():
    spec = importlib.util.spec_from_file_location("mc_table_check", os.path.join(ROOT, "scripts", "mc_table_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_every_case_of_the_table_is_a_manifold_patch_with_one_winding():
    census = _script().check()
    assert sum(census.values()) == 256 and census[0] == 2 and census[1] == 16 and max(census) == 5


def test_generated_include_files_are_up_to_date():
    mc = _script()
    for path, (qualifier, name) in mc.FILES.items():
        assert open(os.path.join(ROOT, path)).read() == mc.emit(qualifier, name), f"{path}: run scripts/mc_table_check.py --emit"


def test_oracle_marching_cubes_on_cells_known_by_hand():
    from oracle import pyoracle

    res = np.float32(0.5)
    code0 = np.zeros(1, np.uint64)
    pyoracle.lib().oracle_morton_encode_ivec(np.array([[2, -3, 4]], np.int32).ctypes.data, 1, code0.ctypes.data)
    # all corners behind the surface, or all in front: nothing
    for sign in (-1.0, 1.0):
        assert len(pyoracle.OracleMap.marching_cubes(code0, np.full((1, 8), sign, np.float32), res)) == 0
    # corner 0 alone in front of the surface: one triangle cutting the three edges at corner 0, a quarter of the way along each,
    # facing corner 0 (increasing signed distance)
    corners = np.full((1, 8), -3.0, np.float32)
    corners[0, 0] = 1.0
    tri = pyoracle.OracleMap.marching_cubes(code0, corners, res)
    assert tri.shape == (1, 3, 3)
    base = np.array([2, -3, 4], np.float64) * 0.5
    want = {tuple(base + 0.5 * 0.25 * np.eye(3)[k]) for k in range(3)}
    assert {tuple(np.round(p.astype(np.float64), 6)) for p in tri[0]} == {tuple(np.round(w, 6)) for w in want}
    normal = np.cross(tri[0, 1] - tri[0, 0], tri[0, 2] - tri[0, 0]).astype(np.float64)
    assert np.dot(normal, base - tri[0].mean(axis=0)) > 0
    # a plane z = 0.25 cells up (bottom face in front, top face behind): two triangles, all four vertices on the vertical edges
    corners = np.array([[1, 1, 1, 1, -3, -3, -3, -3]], np.float32)
    tri = pyoracle.OracleMap.marching_cubes(code0, corners, res)
    assert tri.shape == (2, 3, 3) and np.allclose(tri[..., 2], (4 + 0.25) * 0.5)
    area = sum(0.5 * np.linalg.norm(np.cross(t[1] - t[0], t[2] - t[0])) for t in tri.astype(np.float64))
    assert abs(area - 0.25) < 1e-6
    assert all(np.cross(t[1] - t[0], t[2] - t[0])[2] < 0 for t in tri.astype(np.float64))   # facing the bottom corners: increasing signed distance
