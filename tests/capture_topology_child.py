"""Child process of tests/test_discriminator_gpu.py::test_capture_refuses_second_level_join: a helper stream forked from the SIDE
branch of an open capture and joined back into that branch is the construct hipStreamEndCapture faults on (ROCm 7.2,
tools/capture_probe.py); the engine must refuse it with an error before the edge is recorded, and the process must live on."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "srgan-st_amd")]
import torch  # noqa: E402
from srganst import ops  # noqa: E402
from srganst.engine import _GraphedStep  # noqa: E402

x, y = torch.ones(1024, device="cuda"), torch.ones(1024, device="cuda")
side = torch.cuda.Stream()


def fn():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        x.mul_(2.0)
        with ops.SideStream(y):          # a helper stream forked from the side branch ...
            y.add_(1.0)
        ops.join_side()                  # ... joined into the side branch
    main.wait_stream(side)


ops.OVERLAP = True
st = _GraphedStep(fn, warmup_calls=1)
st()                                     # eager: legal
torch.cuda.synchronize()
try:
    st()                                 # capture: refused
    print("NOT-REFUSED")
except ops.CaptureTopologyError as e:
    print("REFUSED:", str(e)[:60])
ops.OVERLAP = False
x.mul_(2.0)                              # the process and the device are fine
torch.cuda.synchronize()
print("ALIVE", float(x[0]))
