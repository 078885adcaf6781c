"""CPU oracle for the TowerUNet hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product. Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / the timed CPU baseline. The product path
(``cultionet_amd``) never imports this package and fails loudly when the HIP
extension is missing.

Contents
--------
``na2d_ref``          two independent pure-PyTorch restatements of natten 0.17.1's
                      NeighborhoodAttention2D (third-party dependency of the
                      reference, absent here; semantics restated from the
                      published algorithm -- *parity for this one op is pinned
                      only by cross-checking the two restatements and by
                      property tests*).
``towerunet_oracle``  plain ``torch.nn`` fp32 restatement of the reference's
                      TowerUNet / CultioNet / calc_loss, state-dict compatible
                      with the reference (442 keys at the default config).
``refimport``         stub-imports the real reference from /root/reference
                      (this container only) to validate the restatement and to
                      generate ``tests/golden`` fixtures.
``make_golden``       the script that wrote ``tests/golden/*``.
"""
