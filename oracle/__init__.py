"""oracle/ -- TEST INFRASTRUCTURE ONLY: CPU restatement of the reference algorithm for the hop-aggregation path.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this
package, and only as the checker / the reported baseline.  The product (``h2gcn_amd/``) never does.

Pinning status (see DESIGN.md "Oracle"):
* operand construction (``operands.py``: removeEye / nhoodSplit / normalize / hop-group glue) -- PINNED against
  the reference's own code imported in the build container, through the fixtures in ``tests/golden/``;
* SpMM arithmetic (``spmm_oracle.c`` / ``gcn_layer.py``) -- the arithmetic lives in TensorFlow (third party,
  not vendored, not installable here; the reference has no tests or golden vectors for it): **parity unpinned**
  against TensorFlow itself; pinned against scipy ``csr @ dense`` (fp32 loop-order-identical, fp64 bound).
"""
