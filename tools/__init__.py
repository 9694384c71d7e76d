"""measurement, generator and profiling scripts (run as `python tools/<name>.py` from the repo root); `hostcpus` is imported by bench.py and tests/conftest.py"""
