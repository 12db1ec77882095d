"""Run pytest against another build of the library: python tools/pytest_with_lib.py <lib.so> <pytest args...>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stochopy_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import pytest
sys.exit(pytest.main(sys.argv[2:]))
