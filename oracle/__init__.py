"""CPU oracle for graphcast-lite's hot path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package; the product (`graphcast-lite_amd/`) never does and has no CPU fallback.
Parity status: graph layout, loss and threshold schedule are pinned by fixtures generated from
the reference's own code (`tests/golden/make_golden.py`); the PyG-backed layers are
"parity unpinned" (see `oracle/pyg_ops.py`).
"""
