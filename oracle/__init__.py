"""
oracle/ — CPU restatement of the NeuroFluid hot path (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker / the reported CPU baseline.  Nothing under ``neurofluid_amd/``
imports it; the product path fails loudly when the HIP library is missing.

Pinning status:
  * renderer stages A0-A10 (ray_utils / nerf / renderer):  PINNED against golden vectors produced by
    importing /root/reference in the authoring container (tests/golden/gen_golden.py).
  * pytorch3d ``ball_query`` (first-K-by-index):  parity UNPINNED (library not vendored / importable);
    restated from its documented semantics in oracle/csrc/nf_oracle.c.
  * Open3D ``ContinuousConv`` / ``FixedRadiusSearch``:  parity UNPINNED (same reason); restated from
    Ummenhofer et al. ICLR 2020 + the Open3D 0.15.2 layer contract in oracle/trans_oracle.py.
"""
