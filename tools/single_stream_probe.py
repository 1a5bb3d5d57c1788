"""One live handle pulled 8192 samples at a time (bench.py's single_stream extra alone), with the engine's own trace."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench.single_stream_extra(pulls=int(sys.argv[1]) if len(sys.argv) > 1 else 50)))
