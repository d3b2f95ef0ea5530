"""Parts of bench.py (the driver's contract lives in bench.py itself: arguments, the timed region, the ONE JSON line):
workloads.py -- the synthetic workloads and their placement; cpu.py -- the CPU baselines (the only place that touches oracle/);
hostpath.py -- the PCIe-inclusive host-path figures and `--mode host`; gather.py -- the all-gather leg of N > 1;
traffic.py -- HBM traffic by rocprofv3 --pmc child passes; frows.py -- the SURVEY 8(f) rows."""
