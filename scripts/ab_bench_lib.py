"""The default bench with another build of the library (same-box A/B of two builds): python scripts/ab_bench_lib.py LIB.so [bench args]"""
import io, json, os, sys, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from padne_amd import _hip
lib = os.path.abspath(sys.argv[1])
_hip.LIB_PATH = lib
sys.argv = ["bench.py"] + sys.argv[2:] + ["--no-cpu-baseline", "--no-seam", "--no-small", "--no-dist-one-rank"]
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
print(os.path.basename(lib), round(d["value"], 2), round(d["ms_per_step"], 2), round(d["preconditioner"]["setup_ms_per_step"], 2),
      round(d["us_per_iteration"], 1), d["iterations"])
