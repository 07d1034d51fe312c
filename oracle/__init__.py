"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU (torch fp32 / numpy) restatement of the reference's Fast-Pose-Distillation
hot path, used as the parity checker.  Nothing in the product package
(`fast-human-pose-estimation.pytorch_amd/`, `tools/`) may import from here; only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg do.

Parity pinning: the reference ships no tests or golden vectors, so
the restatement is pinned against the reference's OWN modules
(/root/reference/lib/models/hourglass.py, lib/core/loss.py) executed on CPU in
the build container: `tests/golden/make_golden.py` imports them, runs seeded
synthetic inputs and commits the outputs as `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every function here against those files.
"""
