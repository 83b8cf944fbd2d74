"""TEST INFRASTRUCTURE ONLY -- CPU restatements of the reference hot path.

Nothing under `oracle/` may be imported by the product (`bde2vid_amd/`).  Allowed
importers: `tests/`, `__graft_entry__.smoke()`, and `bench.py`'s `cpu_baseline` leg.
"""
