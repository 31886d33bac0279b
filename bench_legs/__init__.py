"""The workload legs of bench.py (one module per family); bench.py itself keeps the CLI, the timed headline, the JSON line and the watchdog."""
