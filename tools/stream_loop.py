import sys, os, traceback
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import importlib
from colord_amd.device import Context
m = importlib.import_module("test_gpu_stream")
ctx = Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for i in range(n):
    try:
        m.test_chunked_equals_one_call_200_mbases(ctx)
    except Exception as e:
        bad += 1
        print("iteration", i, "FAILED:", repr(e)[:1500], flush=True)
        traceback.print_exc()
        break
print("iterations done:", i + 1, "failures:", bad, flush=True)
