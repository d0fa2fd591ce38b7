"""oracle/ -- TEST INFRASTRUCTURE ONLY.

A CPU restatement (plain PyTorch-CPU fp32 functional ops + numpy) of the
`FlowHighSR.generate()` inference path of resemble-ai/flowhigh.  It is the
checker the HIP path is compared with; it is never the product.

Rules (enforced by tests/test_layout.py):
  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
    leg may import anything from this package;
  * nothing under `flowhigh_amd/` imports it, and the product path raises if
    the HIP extension is missing instead of falling back to this code.

Parity status: the reference ships NO tests, golden vectors or known-answer
values for this path and its trained weights are a
Hugging Face artefact that is not reachable from here.  The oracle is therefore
pinned against outputs of the reference itself, imported in the build
container under import shims (`oracle/ref_shim.py`) with seeded synthetic
weights; the generating script is `oracle/make_golden.py` and the vectors live
in `tests/golden/`.  Trained-weight parity: unpinned.
"""
