"""CPU oracle for the SVD / LKGD denoising hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch fp32 restatement of the reference algorithm for
the one path this repo accelerates.  It exists so that the
HIP path in ``lkgd_amd`` can be checked against something that follows the
reference line by line.  It is NOT the product:

* only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
  ``bench.py`` may import it;
* nothing under ``lkgd_amd/`` imports it, and ``lkgd_amd`` has no CPU fallback
  (it raises when the HIP library is missing).

Parity status ("pinned" = checked against output of the reference's own code
executed in the build container, fixtures under ``tests/golden/``):

* scheduler (``oracle/scheduler.py``)        pinned - reference file run through name-only stubs
* top-level UNet wiring, both signatures      pinned - reference ``forward`` run over these blocks
* LK fuse (``oracle/unet.py::lk_fuse``)        pinned - same run (LK variant)
* joint-attention hooks (``oracle/patch_hooks.py``) pinned - reference ``patch/patch.py`` ToMeBlock
* denoising loop body                          pinned - reference loop lines re-executed by the golden script
* diffusers 0.27.2 block interiors (``oracle/blocks.py``)  **parity unpinned**: diffusers,
  peft and core_qnn are not vendored in /root/reference and are not installable here
  (no network); the blocks restate the pinned version's published source and are
  checked op-by-op against torch.nn.functional plus a structural parameter-count gate
  (1 524 623 082 parameters for the SVD config).
"""
