"""CPU oracle for the U-Net / ASPP tile pipeline.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this package.  The product path (`satellite_computervision_amd`) never
does: it fails loudly when the HIP extension is missing.

What this restates
------------------
* `oracle.keras_ops` / `oracle.unet` / `oracle.losses`: the arithmetic of the hot
  path of `/root/reference/utils/model_tools.py` (U-Net blocks :174-415, ASPP
  :533-574, losses :25-166).  That arithmetic lives in a THIRD-PARTY dependency
  that is absent from /root/reference and is not pinned by it (TensorFlow /
  tf.keras, imported at `utils/model_tools.py:8-15`; no requirements file of any
  kind).  The restatement follows the documented Keras layer semantics listed
  in SURVEY.md Appendix A.  PARITY UNPINNED: the reference holds no tests,
  golden vectors or fixtures for these ops and TensorFlow cannot be imported
  in the build container, so the floating-point oracle is cross-checked only
  by an independent implementation (PyTorch-CPU functional ops) in
  `tests/test_oracle_cpu.py`.
* `oracle.tiling`: `generate_chip_indices` / `extract_chips` / `predict_chips`
  of `/root/reference/utils/prediction_tools.py:87-156`.  PINNED: checked
  against outputs of the reference's own function bodies executed in the build
  container (`tests/golden/make_reference_fixtures.py` -> `tests/golden/
  tiling_reference.npz`) and the known answers of SURVEY.md Appendix D.
"""
