import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tools'),
          os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests must never silently pass without a GPU: on a box without
    one they are reported as skipped, and only when they were not asked for."""
    if _gpu_available():
        return
    marker_expr = config.getoption('-m') or ''
    if 'gpu' in marker_expr and 'not gpu' not in marker_expr:
        return          # explicitly requested: let them fail loudly
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
