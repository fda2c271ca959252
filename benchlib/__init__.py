"""The parts of bench.py (repo root): common.py - workloads, index, ranks, the timed loops; legs.py - the legs behind the timed
region (oracle parity + CPU baseline, the CPU model of the plan-guided stage, the variants, the MS-emitting entry points, host to
host); call.py - `bench.py --call` / `--config C5`.  bench.py itself keeps main(): the headline line."""
