"""CPU oracle for the CDAN/CBAM restoration path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(`multi-degradation-image-enhancement_amd/`, `models/`) may import this
package.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg use it, and there only as the checker / the reported
host-CPU baseline -- never as the thing being measured or shipped.

Parity status: PINNED.  `oracle.cdan_oracle` is checked against outputs of
the reference itself (`/root/reference/models/{cdan,cbam}.py`, imported on
CPU in the build container by `tests/golden/make_golden.py`); those outputs
are committed under `tests/golden/*.npz` and re-checked by
`tests/test_oracle_golden.py` on every run.
"""
