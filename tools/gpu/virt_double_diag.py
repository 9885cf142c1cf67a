"""Per-tensor error of a virtual-view training step's gradients against the reference run in double: the reference's own fp32 run
beside the HIP path (the numbers behind tests/test_gpu_render.py::test_virtual_view_gradients_[72_]against_the_reference_in_double).
    python tools/gpu/virt_double_diag.py [24|72]        (arithmetic mode: MORPHEUS_MLP=b3|f32, MORPHEUS_FIELD_BWD)
DIAG_WARP_MODE / DIAG_FIELD_MODE = f32 | b3 force the arithmetic of ONE net group (warp nets / field nets) against the process mode:
which group carries an error."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_gpu_render as T
from morpheus_amd import ops
for var, name in (("DIAG_WARP_MODE", "prepare_warp_operands"), ("DIAG_FIELD_MODE", "prepare_field_operands")):
    if os.environ.get(var):
        def patched(*a, _f=getattr(ops, name), _m=os.environ[var], **kw):
            kw["mode"] = _m
            return _f(*a, **kw)
        setattr(ops, name, patched)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 24
args = (72, 32, "round6.npz", "virt72d") if size == 72 else (24, 24, "round5.npz", "virt24")
rows, (l64, l32, lhip) = T.virt24_errors(*args)
print("%d x %d, MORPHEUS_MLP=%s MORPHEUS_FIELD_BWD=%s DIAG_WARP_MODE=%s DIAG_FIELD_MODE=%s" % (size, size, os.environ.get("MORPHEUS_MLP", "b3"),
      os.environ.get("MORPHEUS_FIELD_BWD", "b3"), os.environ.get("DIAG_WARP_MODE", "-"), os.environ.get("DIAG_FIELD_MODE", "-")))
print("loss: float64 %.9f   reference fp32 %.9f (%.1e)   HIP %.9f (%.1e)" % (l64, l32, abs(l32 - l64) / l64, lhip, abs(lhip - l64) / l64))
print("%-40s %10s %14s %14s %8s %16s" % ("tensor", "|grad|", "ref fp32 err", "HIP err", "ratio", "HIP 2nd worst"))
for k, n, er, eh, eh2 in sorted(rows, key=lambda r: -r[3]):
    print("%-40s %10.3e %14.2e %14.2e %8.2f %16.2e" % (k, n, er, eh, eh / max(er, 1e-30), eh2))
