"""bench.py against another build of the library: python tools/ab_bench.py <lib.so | default> [bench.py arguments]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                      # (puts fdn-tip2025_amd on sys.path)
import fdn_hip
if sys.argv[1] != "default":
    fdn_hip._LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
bench.main()
