"""Parts of bench.py that are importable on their own: workload tables, byte models (unit-tested on CPU), PMC child passes,
rank plumbing."""
