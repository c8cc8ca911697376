import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


os.environ.setdefault('EKAID_DP_ASSERT_ALL_FIRED', '1')      # FlatAdam.step checks that every live parameter delivered a gradient


if os.environ.get('EKAID_POISON', '0') not in ('0', '', 'off', 'false'):
    # debug run: NaN-filled torch.empty / 0xFF-filled workspaces (ekaid_amd.hip.enable_poison) -- an uninitialised read fails a test
    from ekaid_amd import hip as _hip
    _hip.enable_poison()


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
