"""Drop-in check of the API surface against the reference's SOURCE: for every entry point of
the hot path, the reference's positional parameters are a prefix of ours in the same order, every keyword-only
parameter of the reference is accepted by name, literal defaults are equal, and the state / info tuples have the same
fields in the same order.  The reference is parsed with ``ast`` (it cannot be imported here: no JAX), so this runs
only where ``/root/reference`` exists -- this container -- and skips on the GPU box (CPU test, no GPU needed).

Engine-only extras (``chain_block``, ``use_graph``, ``chain_offset``, ``fuse_target`` ...) must be keyword-only or come
after the reference's positional parameters, so a call written for the reference binds the same way here."""
import ast
import importlib
import inspect
import os

import pytest

REF = "/root/reference/blackjax/"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference is not on this box")


def _ref_module(path):
    with open(REF + path) as fh:
        return ast.parse(fh.read())


def _find(tree, name, inside=None):
    """Top-level definition ``name``; ``inside="f"``: the function ``name`` nested in top-level function ``f``."""
    scope = tree.body
    if inside is not None:
        outer = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == inside)
        scope = [n for n in ast.walk(outer) if n is not outer]
    return next(n for n in scope if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name == name)


def _ref_signature(node):
    a = node.args
    pos = [x.arg for x in a.posonlyargs + a.args]
    defaults = {pos[len(pos) - len(a.defaults) + i]: d for i, d in enumerate(a.defaults)}
    defaults.update({x.arg: d for x, d in zip(a.kwonlyargs, a.kw_defaults) if d is not None})
    literal = {}
    for k, d in defaults.items():
        try:
            literal[k] = ast.literal_eval(d)
        except (ValueError, SyntaxError):
            pass  # a name or a lambda: compared by role in the GPU parity tests, not here
    return pos, [x.arg for x in a.kwonlyargs], literal


def _ref_fields(node):
    return [s.target.id for s in node.body if isinstance(s, ast.AnnAssign)]


# (reference file, definition[, enclosing function]) -> (our module, attribute)
_FUNCTIONS = [
    ("mcmc/hmc.py", "init", None, "hmc", "init"),
    ("mcmc/hmc.py", "flip_momentum", None, "hmc", "flip_momentum"),
    ("mcmc/hmc.py", "build_kernel", None, "hmc", "build_kernel"),
    ("mcmc/hmc.py", "as_top_level_api", None, "hmc", "as_top_level_api"),
    ("mcmc/nuts.py", "build_kernel", None, "nuts", "build_kernel"),
    ("mcmc/nuts.py", "as_top_level_api", None, "nuts", "as_top_level_api"),
    ("mcmc/dynamic_hmc.py", "init", None, "dynamic_hmc", "init"),
    ("mcmc/dynamic_hmc.py", "build_kernel", None, "dynamic_hmc", "build_kernel"),
    ("mcmc/dynamic_hmc.py", "as_top_level_api", None, "dynamic_hmc", "as_top_level_api"),
    ("mcmc/dynamic_hmc.py", "halton_sequence", None, "dynamic_hmc", "halton_sequence"),
    ("mcmc/dynamic_hmc.py", "halton_trajectory_length", None, "dynamic_hmc", "halton_trajectory_length"),
    ("mcmc/adjusted_mclmc.py", "rescale", None, "dynamic_hmc", "rescale"),
    ("mcmc/ghmc.py", "init", None, "ghmc", "init"),
    ("mcmc/ghmc.py", "build_kernel", None, "ghmc", "build_kernel"),
    ("mcmc/ghmc.py", "as_top_level_api", None, "ghmc", "as_top_level_api"),
    ("adaptation/staged_adaptation.py", "build_schedule", None, "adaptation", "build_schedule"),
    ("adaptation/staged_adaptation.py", "staged_adaptation", None, "adaptation", "staged_adaptation"),
    ("adaptation/window_adaptation.py", "window_adaptation", None, "adaptation", "window_adaptation"),
    ("adaptation/chees_adaptation.py", "base", None, "chees", "base"),
    ("adaptation/chees_adaptation.py", "chees_adaptation", None, "chees", "chees_adaptation"),
    ("adaptation/meads_adaptation.py", "base", None, "meads", "base"),
    ("adaptation/meads_adaptation.py", "meads_adaptation", None, "meads", "meads_adaptation"),
    ("diagnostics.py", "potential_scale_reduction", None, "diagnostics", "potential_scale_reduction"),
    ("diagnostics.py", "rhat", None, "diagnostics", "rhat"),
    ("diagnostics.py", "effective_sample_size", None, "diagnostics", "effective_sample_size"),
    ("diagnostics.py", "ess_bulk", None, "diagnostics", "ess_bulk"),
    ("diagnostics.py", "ess_tail", None, "diagnostics", "ess_tail"),
    ("util.py", "run_inference_algorithm", None, "util", "run_inference_algorithm"),
    ("adaptation/base.py", "return_all_adapt_info", None, "adaptation", "return_all_adapt_info"),
    ("adaptation/base.py", "get_filter_adapt_info_fn", None, "adaptation", "get_filter_adapt_info_fn"),
]

_TUPLES = [
    ("mcmc/hmc.py", "HMCState", "hmc", "HMCState"),
    ("mcmc/hmc.py", "HMCInfo", "hmc", "HMCInfo"),
    ("mcmc/nuts.py", "NUTSInfo", "nuts", "NUTSInfo"),
    ("mcmc/integrators.py", "IntegratorState", "hmc", "IntegratorState"),
    ("mcmc/integrators.py", "IntegratorState", "integrators", "IntegratorState"),
    ("mcmc/dynamic_hmc.py", "DynamicHMCState", "dynamic_hmc", "DynamicHMCState"),
    ("mcmc/ghmc.py", "GHMCState", "ghmc", "GHMCState"),
    ("adaptation/staged_adaptation.py", "StagedAdaptationState", "adaptation", "StagedAdaptationState"),
    ("adaptation/step_size.py", "DualAveragingAdaptationState", "adaptation", "DualAveragingAdaptationState"),
    ("adaptation/mass_matrix.py", "WelfordAlgorithmState", "adaptation", "WelfordAlgorithmState"),
    ("adaptation/mass_matrix.py", "MassMatrixAdaptationState", "adaptation", "MassMatrixAdaptationState"),
    ("adaptation/chees_adaptation.py", "ChEESAdaptationState", "chees", "ChEESAdaptationState"),
    ("adaptation/meads_adaptation.py", "MEADSAdaptationState", "meads", "MEADSAdaptationState"),
    ("adaptation/base.py", "AdaptationResults", "adaptation", "AdaptationResults"),
    ("adaptation/base.py", "AdaptationInfo", "adaptation", "AdaptationInfo"),
    ("base.py", "SamplingAlgorithm", "base", "SamplingAlgorithm"),
    ("base.py", "AdaptationAlgorithm", "base", "AdaptationAlgorithm"),
]


def _ours(mod, attr):
    return getattr(importlib.import_module("blackjax_amd." + mod), attr)


@pytest.mark.parametrize("path,name,inside,mod,attr", _FUNCTIONS, ids=[f"{m}.{a}" for _, _, _, m, a in _FUNCTIONS])
def test_entry_point_binds_like_the_reference(path, name, inside, mod, attr):
    pos, kwonly, literal = _ref_signature(_find(_ref_module(path), name, inside))
    params = inspect.signature(_ours(mod, attr)).parameters
    ours_pos = [k for k, v in params.items() if v.kind in (v.POSITIONAL_ONLY, v.POSITIONAL_OR_KEYWORD)]
    assert ours_pos[: len(pos)] == pos, f"{path}:{name}: positional parameters {pos} vs ours {ours_pos}"
    for k in kwonly:
        assert k in params, f"{path}:{name}: keyword {k!r} of the reference is not accepted"
        assert params[k].kind in (params[k].KEYWORD_ONLY, params[k].POSITIONAL_OR_KEYWORD)
    for k in pos + kwonly:  # required there <=> required here (a default added here would hide a missing argument)
        ref_required = k not in literal and not _has_default(path, name, inside, k)
        assert (params[k].default is inspect.Parameter.empty) == ref_required, f"{path}:{name}: {k!r} required-ness"
    for k, v in literal.items():
        assert params[k].default == v, f"{path}:{name}: default of {k!r} is {params[k].default!r}, reference {v!r}"
    # engine-only extras never sit in front of a reference parameter and are optional
    for k in list(params)[len(pos):]:
        if k not in kwonly:
            assert params[k].default is not inspect.Parameter.empty or params[k].kind == params[k].VAR_KEYWORD, (
                f"{path}:{name}: engine-only parameter {k!r} must be optional")


def _has_default(path, name, inside, arg):
    a = _find(_ref_module(path), name, inside).args
    pos = [x.arg for x in a.posonlyargs + a.args]
    with_default = set(pos[len(pos) - len(a.defaults):]) | {x.arg for x, d in zip(a.kwonlyargs, a.kw_defaults) if d is not None}
    return arg in with_default


@pytest.mark.parametrize("path,name,mod,attr", _TUPLES, ids=[f"{m}.{a}" for _, _, m, a in _TUPLES])
def test_state_and_info_tuples_have_the_reference_fields(path, name, mod, attr):
    ref = _ref_fields(_find(_ref_module(path), name))
    ours = list(_ours(mod, attr)._fields)
    assert ours[: len(ref)] == ref, f"{path}:{name}: fields {ref} vs ours {ours}"


def _std_normal(q):
    return -0.5 * (q * q).sum(-1)


def _check_run(ref_path, ref_outer, run):
    pos, kwonly, literal = _ref_signature(_find(_ref_module(ref_path), "run", inside=ref_outer))
    params = inspect.signature(run).parameters
    assert list(params)[: len(pos)] == pos, (pos, list(params))
    for k in kwonly:
        assert k in params, k
    for k in pos + kwonly:
        ref_required = not _has_default(ref_path, "run", ref_outer, k)
        assert (params[k].default is inspect.Parameter.empty) == ref_required, k
    for k, v in literal.items():
        assert params[k].default == v, (k, params[k].default, v)


def test_warmup_run_signatures():
    """``run`` of the four warm-ups: window / staged adaptation (staged_adaptation.py:756, ``num_steps=None`` = "not
    given" = 1 000), ChEES (chees_adaptation.py:737), MEADS (meads_adaptation.py:710)."""
    import blackjax_amd as bjx

    w = bjx.window_adaptation(bjx.hmc, _std_normal, num_integration_steps=3)
    assert isinstance(w, bjx.AdaptationAlgorithm)
    _check_run("adaptation/staged_adaptation.py", "staged_adaptation", w.run)
    _check_run("adaptation/staged_adaptation.py", "staged_adaptation",
               bjx.staged_adaptation(bjx.nuts, _std_normal, "welford_dense").run)
    _check_run("adaptation/chees_adaptation.py", "chees_adaptation", bjx.chees_adaptation(_std_normal, 8).run)
    _check_run("adaptation/meads_adaptation.py", "meads_adaptation", bjx.meads_adaptation(_std_normal, 8).run)


def test_integrator_coefficients_equal_the_reference():
    """integrators.py:321-369: the module-level coefficient lists, evaluated from the reference's own assignments (plain
    float arithmetic on names assigned above them), equal ours to the last bit -- and so do the fp32 values the kernels get."""
    import numpy as np

    from blackjax_amd import integrators as ours

    env = {}
    for node in _ref_module("mcmc/integrators.py").body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            try:
                env[node.targets[0].id] = eval(compile(ast.Expression(node.value), "<ref>", "eval"), {"__builtins__": {}}, dict(env))
            except Exception:
                pass  # generate_euclidean_integrator(...) and friends: not arithmetic
    for name in ("velocity_verlet", "mclachlan", "yoshida", "omelyan"):
        ref = env[name + "_coefficients"]
        got = getattr(ours, name).coefficients
        assert list(got) == [float(c) for c in ref], name
        assert np.array_equal(np.asarray(got, np.float32), np.asarray(ref, np.float32))
        assert getattr(ours, name).num_gradients_per_step == (len(ref) - 1) // 2


def test_dual_averaging_constants_equal_the_reference():
    """dual_averaging.py:53-55 / step_size.py:65-67 (t0, gamma, kappa) and chees_adaptation.py's target acceptance rate."""
    from blackjax_amd import adaptation, chees

    for path, name in (("optimizers/dual_averaging.py", "dual_averaging"), ("adaptation/step_size.py", "dual_averaging_adaptation")):
        _, _, literal = _ref_signature(_find(_ref_module(path), name))
        assert (literal["t0"], literal["gamma"], literal["kappa"]) == (adaptation._DA_T0, adaptation._DA_GAMMA, adaptation._DA_KAPPA)
        ours = inspect.signature(chees._da_update).parameters
        assert (ours["t0"].default, ours["gamma"].default, ours["kappa"].default) == (literal["t0"], literal["gamma"], literal["kappa"])
    opt = next(n for n in _ref_module("adaptation/chees_adaptation.py").body if isinstance(n, ast.Assign)
               and isinstance(n.targets[0], ast.Name) and n.targets[0].id == "OPTIMAL_TARGET_ACCEPTANCE_RATE")
    assert inspect.signature(chees.chees_adaptation).parameters["target_acceptance_rate"].default == ast.literal_eval(opt.value)


def test_top_level_names_of_the_hot_path_exist():
    """blackjax/__init__.py: the names a user of the hot path imports."""
    tree = _ref_module("__init__.py")
    assigned = {t.id for n in tree.body if isinstance(n, ast.Assign) for t in n.targets if isinstance(t, ast.Name)}
    import blackjax_amd as bjx

    for name in ("hmc", "nuts", "mhmc", "multinomial_hmc", "dynamic_hmc", "dhmc", "dmhmc", "ghmc", "hmc_family",
                 "window_adaptation", "staged_adaptation", "chees_adaptation", "meads_adaptation"):
        assert name in assigned or name in _all_of(tree), f"the reference has no top-level {name!r} (stale list?)"
        assert hasattr(bjx, name), name


def _all_of(tree):
    for n in tree.body:
        if isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id == "__all__" for t in n.targets):
            return set(ast.literal_eval(n.value))
    imported = set()
    for n in tree.body:
        if isinstance(n, ast.ImportFrom):
            imported |= {a.asname or a.name for a in n.names}
    return imported
