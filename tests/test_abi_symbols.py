"""The C-ABI library loads without a GPU and exports exactly what include/chad_hip.h declares."""
import ctypes
import os
import re

import pytest

from chad_tsdf_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "chad_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(chad_[a-z_0-9]+)\s*\(", text)))


def test_library_is_built_in_tree():
    assert os.path.exists(_lib.LIB_PATH), "run `make` or __graft_entry__.build()"
    assert _lib.LIB_PATH.startswith(ROOT)


def test_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in chad_hip.h but not exported"
    # and the Python binding covers the same set
    assert sorted(_lib.SYMBOLS) == declared


def test_struct_layouts_match_header(tmp_path):
    """chad_hip.h is plain C; the ctypes structures have the sizes and offsets a C compiler gives them."""
    import subprocess

    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "chad_hip.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(chad_options), offsetof(chad_options, stream),'
        ' offsetof(chad_options, update_mode), sizeof(chad_stats), offsetof(chad_stats, sort_passes), offsetof(chad_stats, ms_total),'
        ' offsetof(chad_options, dag_pool_bytes), sizeof(chad_pipeline_times), offsetof(chad_pipeline_times, alloc_s),'
        ' offsetof(chad_pipeline_times, comm_transport), sizeof(chad_dag_stats), offsetof(chad_pipeline_times, comm_p2p_calls),'
        ' offsetof(chad_dag_stats, submaps), offsetof(chad_dag_stats, device_bytes));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(_lib.Options), _lib.Options.stream.offset, _lib.Options.update_mode.offset,
            ctypes.sizeof(_lib.Stats), _lib.Stats.sort_passes.offset, _lib.Stats.ms_total.offset,
            _lib.Options.dag_pool_bytes.offset, ctypes.sizeof(_lib.PipelineTimes), _lib.PipelineTimes.alloc_s.offset,
            _lib.PipelineTimes.comm_transport.offset, ctypes.sizeof(_lib.DagStats), _lib.PipelineTimes.comm_p2p_calls.offset,
            _lib.DagStats.submaps.offset, _lib.DagStats.device_bytes.offset]
    assert got == want


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("this check is about GPU-less hosts")
    from chad_tsdf_amd import TSDFMap

    with pytest.raises(_lib.ChadError) as err:
        TSDFMap(0.05, 0.1)
    assert err.value.status == _lib.ERR_NO_DEVICE


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "chad_tsdf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in text and "libchad_oracle" not in text and "oracle/" not in text.replace("the oracle", ""), f
