"""TEST INFRASTRUCTURE ONLY.

CPU restatement of the VQACL / VL-T5 forward-backward path.  Nothing in the
product package (`vqacl_amd/`) may import from here; only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg do, and there only
as the checker / the timed CPU baseline.
"""
