"""CPU oracle for the SRGAN-ST training hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-torch CPU restatement of the
reference algorithm (SebastianBitsch/SRGAN-ST: model.py, loss.py:380-413,
utils.py:194-280, train.py:116-164, warmup.py:74-96).  It exists to *check* the
HIP path and to be timed as the CPU baseline.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package (``srgan-st_amd/srganst``) never does and has no CPU
fallback.

Parity pin: every function here is validated against outputs of the reference
itself (imported on CPU in the build container by ``tests/golden/make_golden.py``)
- the committed ``tests/golden/*.npz`` fixtures hold those reference outputs and
``tests/test_oracle_golden.py`` re-checks the oracle against them without the
reference being present.  The reference has no tests / golden vectors of its own
(SURVEY.md section 4).  One step is pinned by documentation rather than by
execution: torchvision's ``Grayscale`` (absent from the image) is restated from
its published ITU-R 601 weights (0.2989, 0.587, 0.114).
"""
