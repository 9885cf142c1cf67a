"""Per-tensor error of the 24 x 24 virtual-view training step's gradients against the reference run in double: the reference's own
fp32 run beside the HIP path (the numbers behind tests/test_gpu_render.py::test_virtual_view_gradients_against_the_reference_in_double)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_gpu_render as T
rows, (l64, l32, lhip) = T.virt24_errors()
print("loss: float64 %.9f   reference fp32 %.9f (%.1e)   HIP %.9f (%.1e)" % (l64, l32, abs(l32 - l64) / l64, lhip, abs(lhip - l64) / l64))
print("%-40s %10s %14s %14s %8s" % ("tensor", "|grad|", "ref fp32 err", "HIP err", "ratio"))
for k, n, er, eh in sorted(rows, key=lambda r: -r[3]):
    print("%-40s %10.3e %14.2e %14.2e %8.2f" % (k, n, er, eh, eh / max(er, 1e-30)))
