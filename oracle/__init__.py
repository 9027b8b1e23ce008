"""ORACLE — test infrastructure, not product code.

`oracle/ref_infer.py` is a CPU restatement of the reference's
`SynthesizerTrn.infer` path (PyTorch-CPU fp32 + NumPy), pinned against golden
vectors captured from the real reference (`tests/golden/`).  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` import it.
"""
