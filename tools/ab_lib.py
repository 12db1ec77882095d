"""A/B two builds of the library on the same box: python tools/ab_lib.py <lib.so> <bench args...>"""
import os, runpy, sys
sys.path.insert(0, "/root/repo")
from stochopy_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [sys.argv[2]] + sys.argv[3:]
runpy.run_path(sys.argv[0], run_name="__main__")
