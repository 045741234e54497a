"""Development aid: with the environment bench.py sets up for N > 1, nothing but our own line reaches stdout."""
import os, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
if os.environ.get("NCCL_DEBUG", "").upper() in ("", "VERSION"):
    os.environ["NCCL_DEBUG"] = "WARN"
os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
from regularizepsf_amd import _native
c = _native.Comm(0, 0, 1, _native.Comm.unique_id())
c.barrier(); print("STDOUT-ONLY-LINE", flush=True); c.close()
