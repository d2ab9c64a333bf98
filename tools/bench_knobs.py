"""bench.py with tuning knobs set first (tuning build only):
DISN_AMD_LIB=disn_amd/csrc/libdisn_amd_tuning.so KNOBS=l4_ranges=2,conv_wide_min=4 python tools/bench_knobs.py --steps 240 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
torch.cuda.init(); torch.zeros(1, device="cuda:0")   # the HIP runtime comes up through torch first (as in bench.py)
import _tuning
for kv in filter(None, os.environ.get("KNOBS", "").split(",")):
    k, v = kv.split("=")
    _tuning.set_knob(k, int(v))
import bench
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
bench.main()
