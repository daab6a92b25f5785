#!/usr/bin/env python3
"""In-process A/B of plan options on a long-filter configuration (BASELINE config 3 shape by default).
usage: tools/ab_c3.py [--wavelet sym8 --rows 64 --n 777777 --levels 8 --boundary 0] name:opt=val,... ..."""
import subprocess, sys, os
args = sys.argv[1:]
base = ["--rows", "64", "--n", "777777", "--levels", "8", "--wavelet", "sym8", "--rounds", "5", "--reps", "5"]
sys.exit(subprocess.call([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab_plan.py")] + base + args))
