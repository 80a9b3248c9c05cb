"""chad_tsdf_amd -- MI355X-native implementation of chad::TSDFMap::insert().

The package holds the HIP kernels + C ABI (csrc/, built into lib/libchad_hip.so), the host-side mirror of the
reference's TSDFMap class (tsdf.py), the frame-sharded multi-GPU map (sharded.py) and the synthetic scan
generators of the BASELINE configurations (synth.py).
"""
from .tsdf import TSDFMap, morton_codes, sort_pairs, walk_rays  # noqa: F401
