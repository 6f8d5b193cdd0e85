"""The example scripts (examples/, the reference's examples after the import change) run end to end on the
device, each in its own interpreter as a user would start it, at a reduced ensemble size."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [("example_mcmc_1d.py", []), ("example_mcmc_2d_singleLevel.py", ["256", "1200"]),
         ("example_mcmc_2d_twoLevel.py", ["256", "800"]), ("example_mcmc_2d_hierarchical.py", ["256", "600"]),
         ("example_inference_linearModel_twoLevel.py", ["128", "600"]),
         ("example_inference_lotkaVolterra_singleLevel.py", ["256", "400"]),
         ("example_inference_lotkaVolterra_singleLevel.py", ["256", "600", "adaptive"]),   # from the reference's own start
         ("example_inference_lotkaVolterra_twoLevel.py", ["256", "300"]),
         ("example_inference_lotkaVolterra_twoLevel.py", ["256", "400", "adaptive"])]


@pytest.mark.parametrize("script,args", CASES, ids=["-".join([c[0]] + c[1][2:]) for c in CASES])
def test_example_runs(script, args):
    env = dict(os.environ, MPLBACKEND="Agg", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script), *args], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = r.stdout.lower()
    assert "acceptance rate" in out and "nan" not in out, r.stdout[-1500:]
    if "adaptive" in args:
        assert "chains flagged non-finite: 0" in out, r.stdout[-1500:]
