"""Shared by the GPU tests that run the SAME optimisation twice through two implementations / schedules and compare the parameters after a
few Adam steps.  The first step is compared strictly by the callers (loss values at 2e-6, every gradient at 1e-5).  From the second step on
two runs are two samples of a slightly chaotic system, even two runs of ONE implementation on ONE stream: the float atomics sum in a different
order every launch, the parameters then differ in their last bits, and now and then that moves a borderline fragment in or out of a pixel's
list (tests/test_gpu_configs.py::_fragment_flips) -- the gradient of the texels / vertices under that pixel changes by one pixel's worth and
Adam (which divides by sqrt(v)) turns it into a step of a different length for those elements.  Measured on the scenes of these tests
(tools/diag/race_hunt.py: 30 single-stream runs of 7 steps against a single-stream reference): 2 to 10 runs of 30 take the other branch of
such a flip, with 1.9 % of the parameters further apart than 1e-4 (largest difference 8e-4) at epoch 0, 0.1 % (largest 0.0144) at epoch 800.
What a wait that does not hold, a missing term or a wrong optimiser step looks like is different in kind: most elements of the affected
tensors are off, by a step's length.  So: almost all elements within 1e-4, hardly any off by a visible fraction of a step, none by more
than two steps' lengths."""


def assert_same_trajectory(pa, pb, lr_max=5e-2):
    diff = (pa - pb).abs()
    far = float((diff > 1e-4).float().mean())
    very = float((diff > 0.05 * lr_max).float().mean())
    worst = float(diff.max())
    assert far < 5e-2 and very < 5e-3 and worst < 2 * lr_max, (far, very, worst)
