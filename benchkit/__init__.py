"""bench.py in modules: common (constants), record (budget, compact line, watchdog), launcher (self-started ranks), roofline,
workloads, sharded.  bench.py at the repo root holds the argument parser, main() and the cpu_baseline leg (the only code outside
tests/ that touches oracle/)."""
