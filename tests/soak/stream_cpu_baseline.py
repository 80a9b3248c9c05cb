"""The oracle (CPU restatement of the reference, one core) on the first N frames of the cfg3 stream: the CPU figure quoted
beside scripts/stream_bench.py in DESIGN.md.   python tests/soak/stream_cpu_baseline.py [frames]"""
import json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from chad_tsdf_amd import synth
from oracle import pyoracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
frames = [synth.cfg3_frame(k) for k in range(n)]
o = pyoracle.OracleMap(0.05, 0.10)
t0 = time.perf_counter()
for pts, pos in frames:
    o.insert(pts, pos)
print(json.dumps({"workload": "cfg3, first %d frames" % n, "cpu_ms_per_frame": (time.perf_counter() - t0) / n * 1e3, "cores": 1}))
