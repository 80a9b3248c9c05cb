"""The drop-in C++ class (include/chad/tsdf.hpp over the C ABI): a program written against the reference's public header
compiles and links here without a GPU, and on a GPU its map equals the oracle's."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "api_check.cpp")
LIB_DIR = os.path.join(ROOT, "chad_tsdf_amd", "lib")


def _build(out_dir):
    if not os.path.exists(os.path.join(LIB_DIR, "libchad_hip.so")):
        subprocess.check_call(["make", "-C", ROOT, "-s"])
    cxx = shutil.which("g++") or "/opt/rocm/bin/hipcc"
    exe = os.path.join(out_dir, "api_check")
    subprocess.check_call([cxx, "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe, "-L", LIB_DIR, "-lchad_hip",
                           "-L/opt/rocm/lib", "-Wl,-rpath," + LIB_DIR, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"])
    return exe


def test_reference_style_program_compiles_and_links(tmp_path):
    assert os.path.exists(_build(str(tmp_path)))


def _build_overloads(out_dir):
    """The glm / Eigen overloads (reference tsdf.hpp:68-115) against test-only stand-ins for the two headers (tests/cpp/shims)."""
    if not os.path.exists(os.path.join(LIB_DIR, "libchad_hip.so")):
        subprocess.check_call(["make", "-C", ROOT, "-s"])
    cxx = shutil.which("g++") or "/opt/rocm/bin/hipcc"
    exe = os.path.join(out_dir, "overloads_check")
    subprocess.check_call([cxx, "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "cpp", "shims"),
                           os.path.join(ROOT, "tests", "cpp", "overloads_check.cpp"), "-o", exe, "-L", LIB_DIR, "-lchad_hip",
                           "-L/opt/rocm/lib", "-Wl,-rpath," + LIB_DIR, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"])
    return exe


def test_glm_and_eigen_overloads_compile(tmp_path):
    assert os.path.exists(_build_overloads(str(tmp_path)))


@pytest.mark.gpu
def test_glm_and_eigen_overloads_build_the_same_map(tmp_path):
    run = subprocess.run([_build_overloads(str(tmp_path)), "30000"], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.startswith("ok"), run.stdout + run.stderr


@pytest.mark.gpu
def test_cpp_class_builds_the_oracles_map(tmp_path):
    from chad_tsdf_amd import synth
    from oracle import pyoracle
    from tests.util import SD_TOL

    exe = _build(str(tmp_path))
    pts, pos = synth.cfg2_frame(90_000, seed=13)
    pts.tofile(tmp_path / "points.f32")
    out = tmp_path / "leaves.bin"
    run = subprocess.run([exe, str(tmp_path / "points.f32"), repr(float(pos[0])), repr(float(pos[1])), repr(float(pos[2])), str(out)],
                         capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    raw = out.read_bytes()
    n = int(np.frombuffer(raw[:8], np.uint64)[0])
    mc = np.frombuffer(raw[8:8 + 8 * n], np.uint64)
    sd = np.frombuffer(raw[8 + 8 * n:8 + 12 * n], np.float32)
    w = np.frombuffer(raw[8 + 12 * n:8 + 16 * n], np.uint32)
    o = pyoracle.OracleMap(0.05, 0.10)
    o.insert(pts, pos)
    omc, osd, ow = o.export_leaves()
    assert np.array_equal(mc, omc) and np.array_equal(w, ow)
    assert (np.abs(sd - osd) / np.maximum(np.abs(osd), 0.10)).max() <= SD_TOL


@pytest.mark.skipif(shutil.which("cmake") is None, reason="cmake not installed")
def test_cmake_consumer_links_chad_tsdf_target(tmp_path):
    """The reference's consumers write `target_link_libraries(app PRIVATE chad::tsdf)` (reference CMakeLists.txt:18-19,
    README.md:20-29): a consumer project that adds this repository as a subdirectory must configure, build (hipcc
    cross-compiles gfx950 without a GPU) and link against that target."""
    (tmp_path / "CMakeLists.txt").write_text(
        "cmake_minimum_required(VERSION 3.18)\nproject(consumer LANGUAGES CXX)\n"
        f"add_subdirectory({ROOT} chad_tsdf)\n"
        f"add_executable(app {SRC})\ntarget_link_libraries(app PRIVATE chad::tsdf)\n")
    build = tmp_path / "build"
    gen = ["-G", "Ninja"] if shutil.which("ninja") else []
    subprocess.check_call(["cmake", "-S", str(tmp_path), "-B", str(build), *gen], stdout=subprocess.DEVNULL)
    subprocess.check_call(["cmake", "--build", str(build)], stdout=subprocess.DEVNULL)
    assert (build / "app").exists() and (build / "chad_tsdf" / "libchad_hip.so").exists()
