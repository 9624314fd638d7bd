"""GPU: a slice of the differential fuzz campaign (profiles/fuzz_campaign.py): random inputs, parameters and push
counts, device table against the oracle's, field by field."""
import os
import subprocess
import sys
import json

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("first_seed,env", [(20_000, {}), (21_000, {"IPX_SHORT_MAX_AVG": "16"}),
                                            (22_000, {"IPX_SORT_TILE": "4096"}),
                                            (23_000, {"IPX_LONG": "1", "IPX_COMBINE": "0"})])
def test_fuzz_slice(first_seed, env, oracle_lib):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "fuzz_campaign.py"), "60", str(first_seed)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env), cwd=ROOT)
    lines = r.stdout.decode().strip().splitlines()
    assert lines, r.stderr.decode()[-2000:]
    summary = json.loads(lines[-1])
    assert r.returncode == 0 and summary["iterations"] == 60 and summary["failures"] == 0, "\n".join(lines[-5:])
