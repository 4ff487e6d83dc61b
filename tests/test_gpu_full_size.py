"""The benchmark sweep at FULL size against the oracle on every node
(tests/full_parity.py): 256^3 x 64 x 32 fp64, J bit for bit, policy index
exact.  The oracle side is ~32 s of OpenMP C on the 256-thread GPU box; on a
host with few cores it would take many minutes, so the test needs >= 64
threads (or SDP_FULL_PARITY=1)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.timeout(1500)
def test_full_256cubed_sweep_matches_the_oracle_on_every_node(gpu):
    if (os.cpu_count() or 1) < 64 and not os.environ.get('SDP_FULL_PARITY'):
        pytest.skip('needs >= 64 host threads for the full-size oracle sweep (or SDP_FULL_PARITY=1)')
    out = subprocess.run([sys.executable, os.path.join(HERE, 'full_parity.py')],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1400)
    text = out.stdout.decode()
    assert out.returncode == 0, text
    assert 'nodes: 16777216   J bit-identical: True' in text and 'index mismatches: 0' in text, text
