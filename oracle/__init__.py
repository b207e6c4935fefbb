"""oracle/ -- CPU restatement of the UST-RUN hot path.  TEST INFRASTRUCTURE ONLY.

This package is the parity checker for the HIP path in ``ust-run_amd/``.  It is
imported only by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py``.  Nothing under ``ust-run_amd/`` may import it: the product
path fails loudly when the HIP library is missing, it never falls back here.

What it restates (reference file:line in each function's docstring):
  * networks/unet_parts.py:8-76, networks/unet_model.py:6-39   -> unet_ref
  * utils/losses.py:194-268, torch CE/BCE call sites train.py:516-519 -> losses_ref
  * utils/ramps.py:19-26, utils/metrics.py:114-231            -> host_ref
  * train.py:82-93,158-251,577-870 (SSL step, EMA, LR, CutMix, FFT mix) -> step_ref

Arithmetic runs on torch-CPU fp32 tensors (conv2d/conv_transpose2d as the
dense contraction primitive, everything else written out explicitly) and numpy.

Parity pinning: the reference owns no tests or golden vectors, so
the oracle is pinned against outputs of the reference itself, imported from
/root/reference in the build container by ``tools/gen_goldens.py`` and committed
as small fixtures under ``tests/golden/`` (torch version recorded in each file).
``tests/test_oracle_golden.py`` checks every oracle function against them.
"""
