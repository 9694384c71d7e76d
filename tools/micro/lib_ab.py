#!/usr/bin/env python3
"""Time a few GEMM shapes with several builds of the library (tools/micro/lib*.so given on the command line), one child
process per build, forced variant from VARIANT (default 4)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for lib in sys.argv[1:]:
    print(os.path.basename(lib), flush=True)
    env = dict(os.environ, LKGD_HIP_LIB=os.path.abspath(lib))
    subprocess.run([sys.executable, os.path.join(HERE, "wide_knobs.py"), "--single"], env=env, check=False)
