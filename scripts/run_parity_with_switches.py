"""Run the full-size parity tests (batch 32 cases) with library debug switches applied first, e.g. to compare kernel policies:
    python scripts/run_parity_with_switches.py 72 75 77      (the GPU context is created before the switches are set)."""
import os, sys as _s; _s.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, torch
torch.cuda.init(); torch.zeros(1, device="cuda:0")
from depth_image_captioning_pub_amd import _lib
l = _lib.load()
for c in sys.argv[1:]:
    l.dic_debug_force_staged_gemm(int(c))
import pytest
sys.exit(pytest.main(["tests/test_fullsize_parity_gpu.py", "-q", "-m", "gpu", "-s", "-k", "32"]))
