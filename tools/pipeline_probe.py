"""bench.py's `pipeline` extra alone: sustained end-to-end throughput over distinct cfg2-sized batches, host work overlapped.
    python tools/pipeline_probe.py [workers=3] [players=4] [batches=8] [pageable] [nohost]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (first: its HIP runtime is the one the engine library binds to)
import bench
kw = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
pinned = "pageable" not in sys.argv
w, p, nb = int(kw.get("workers", 3)), int(kw.get("players", 4)), int(kw.get("batches", 8))
out = {"hbm": bench.pipeline_extra(0, 0, -1, n_batches=nb, workers=w, players=p, pinned=pinned)}
if "nohost" not in sys.argv:
    out["to_host"] = bench.pipeline_extra(0, 0, -1, n_batches=max(6, nb // 2), workers=w, players=p, copy_out=True, pinned=pinned)
print(json.dumps(out, indent=1))
