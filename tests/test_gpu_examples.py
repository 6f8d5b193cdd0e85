"""The example scripts (examples/, the reference's examples after the import change) run end to end on the
device, each in its own interpreter as a user would start it, at a reduced ensemble size."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [("example_mcmc_1d.py", []), ("example_mcmc_2d_singleLevel.py", ["256", "1200"]),
         ("example_mcmc_2d_twoLevel.py", ["256", "800"]), ("example_mcmc_2d_hierarchical.py", ["256", "600"]),
         ("example_inference_linearModel_twoLevel.py", ["128", "1500"]),
         ("example_inference_lotkaVolterra_singleLevel.py", ["256", "400"]),
         ("example_inference_lotkaVolterra_singleLevel.py", ["256", "600", "adaptive"]),   # from the reference's own start
         ("example_inference_lotkaVolterra_twoLevel.py", ["256", "300"]),
         ("example_inference_lotkaVolterra_twoLevel.py", ["256", "400", "adaptive"]),
         ("example_inference_logisticGrowth_singleLevel.py", ["256", "600"])]    # the extension point's worked example


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
    if script == "example_inference_logisticGrowth_singleLevel.py":
        import re
        m = re.search(r"posterior mean: \[\s*([-0-9.e]+)\s+([-0-9.e]+)\s*\]", out)
        truth = np.log(np.array([0.9, 2.5]))
        assert m and abs(float(m.group(1)) - truth[0]) < 0.3 and abs(float(m.group(2)) - truth[1]) < 0.15, r.stdout[-1500:]
        assert "chains flagged non-finite: 0" in out
    if script == "example_inference_linearModel_twoLevel.py":
        # the reference's flow (example_inference_linearModel_twoLevel.py:157-236): four reported stages, each with its
        # acceptance rate, its thinning stride from the device IAT and its posterior mean near the true parameter
        import re
        stages = re.findall(r"results for (target mrw|surrogate mrw|vanilla mlda|aem mlda)\n-+\n+acceptance rate: ([0-9.]+)\n"
                            r"iat estimate: (\d+)\nposterior mean: \[\s*([-0-9.e]+)\s+([-0-9.e]+)\s*\]", out)
        assert [s[0] for s in stages] == ["target mrw", "surrogate mrw", "vanilla mlda", "aem mlda"], r.stdout[-2500:]
        for name, rate, iat, m0, m1 in stages:
            assert 0.05 < float(rate) < 0.95 and int(iat) >= 1, (name, rate, iat)
            if name != "surrogate mrw":     # the surrogate's posterior is another one
                assert abs(float(m0) - 1.5) < 0.6 and abs(float(m1) - 0.5) < 0.8, (name, m0, m1)
        assert "burning-in mlda with" in out and "chains past mindatasize" in out
