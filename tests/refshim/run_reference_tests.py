#!/usr/bin/env python
"""Run the REFERENCE'S OWN unit tests for the hot path on the stand-in for JAX (tests/refshim): the check that the stand-in
is faithful enough for the reference's tests of its HMC / NUTS / adaptation code to pass on it.

    python tests/refshim/run_reference_tests.py [--all-in-scope]      # from the repo root, where /root/reference exists

Default: the quick selection tests/test_reference_tests_on_shim.py runs (about a minute).  ``--all-in-scope`` adds the slow
statistical tests (tests/adaptation/test_adaptation.py: ChEES / MEADS end to end; test_sampling.py's univariate-normal tests
for hmc / nuts / ghmc and its window-adaptation regression test, 18 parameter sets: about two hours of Python loops).  Nothing is written into /root/reference (no bytecode, no pytest cache); chex / absl are tests/refshim's small
restatements.  Out of scope and therefore not selected: lbfgs, pareto-k, divergence concentration, isokinetic / implicit
integrators, low-rank metrics, float64 variants (the stand-in is float32 like JAX's default), thinning, random-walk samplers."""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

QUICK = [
    "tests/mcmc/test_uturn.py",
    "tests/mcmc/test_trajectory.py",
    "tests/mcmc/test_integrators.py::IntegratorTest::test_euclidean_integrator",
    "tests/mcmc/test_metrics.py::CovarianceFormattingTest",
    "tests/mcmc/test_metrics.py::GaussianEuclideanMetricsTest",
    "tests/adaptation/test_mass_matrix.py",
    "tests/optimizers/test_optimizers.py::OptimizerTest::test_dual_averaging",
    "tests/test_util.py::RunInferenceAlgorithmTest",
    "tests/test_diagnostics.py::DiagnosticsTest",
    "tests/test_diagnostics.py::RhatTest",
    "tests/test_diagnostics.py::EssBulkTest",
    "tests/test_diagnostics.py::EssTailTest",
    "tests/adaptation/test_adaptation.py::test_adaptation_schedule",
]
SLOW = [  # measured once (NOTEBOOK.md section 16.10): 18 min, 5 min, 103 min
    "tests/adaptation/test_adaptation.py",
    "tests/mcmc/test_sampling.py::UnivariateNormalTest::test_hmc",
    "tests/mcmc/test_sampling.py::UnivariateNormalTest::test_nuts",
    "tests/mcmc/test_sampling.py::UnivariateNormalTest::test_ghmc",
    "tests/mcmc/test_sampling.py::LinearRegressionTest::test_window_adaptation",
]
DESELECT = ["f64", "float64"]


def run(selection, timeout=10800, extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        ini = os.path.join(tmp, "pytest.ini")
        with open(ini, "w") as fh:
            fh.write("[pytest]\nfilterwarnings =\n    ignore\n")
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1",
                   PYTHONPATH=os.pathsep.join([HERE, REF, ROOT]))
        cmd = [sys.executable, "-m", "pytest", "-c", ini, "--rootdir", tmp, "-p", "no:cacheprovider", "-p", "refshim_boot",
               "-q", "-k", " and ".join(f"not {d}" for d in DESELECT), *extra, *[os.path.join(REF, s) for s in selection]]
        return subprocess.run(cmd, env=env, cwd=tmp, timeout=timeout, capture_output=True, text=True)


if __name__ == "__main__":
    sel = QUICK + (SLOW if "--all-in-scope" in sys.argv else [])
    r = run(sel, extra=("--durations=10",))
    print(r.stdout[-4000:])
    sys.exit(r.returncode)
