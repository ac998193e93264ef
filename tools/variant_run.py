"""Runs a tools/ script against a variant library built by tools/variant_build.sh (experiments only).
usage: python tools/variant_run.py <name> <script.py> [args...]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liodom_amd.api as api
api._LIB = os.path.join(ROOT, "build", "variants", "lib%s.so" % sys.argv[1])
api.is_stale = lambda: False
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
