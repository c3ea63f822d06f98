"""Diagnostics under tools/diag/ run with the fixtures and hooks of tests/conftest.py."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
_spec = importlib.util.spec_from_file_location("_tests_conftest", os.path.join(ROOT, "tests", "conftest.py"))
_m = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_m)
pytest_configure = _m.pytest_configure
hip_ops_factory = _m.hip_ops_factory
