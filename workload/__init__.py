"""Synthetic workload definition shared by bench.py, the tests and the oracle: the reference's configs/base.yaml values
and hard-coded constants (config.py), seeded random-init checkpoints in the reference's key layout (weights.py -- no
pretrained weights are available offline) and seeded synthetic inputs of the BASELINE.json shapes (inputs.py).
Pure data generation: no model arithmetic lives here (the CPU restatement of the path is oracle/, test-only)."""
