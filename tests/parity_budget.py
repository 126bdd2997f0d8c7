"""Regression guard on top of SURVEY.md 8c's tolerances (VERDICT r05 weak point 2: every loop / UNet parity test asserted rel-L2 < 3e-2
where the build measures 2-7e-3, so a kernel that lost three bits of precision would still pass).

tests/golden/parity_budget.json holds, per check, the relative L2 the build MEASURED when the budget was recorded (round 6, MI355X;
values are bitwise reproducible run to run, and move only by fp16 accumulation-order noise between kernel builds).  `check` asserts
SURVEY's hard bound AND `value <= FACTOR x recorded`.  A deliberately degraded build (P of the d = 40 attention rounded to 8 mantissa
bits; the GEGLU GELU replaced by its tanh approximation) fails these checks while the real build passes with >= 2x margin: the record of
that experiment is profiles/r06_parity_budget_degraded.log.

Re-recording (only after a deliberate numerical change, with the reason in the commit message):
    MD_PARITY_RECORD=tests/golden/parity_budget.json python -m pytest tests -m gpu -q        (GPU box)
"""
import json
import os

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity_budget.json")
FACTOR = 2.0            # loop / UNet level: whole-path fp16 noise, moves by a few percent between kernel builds
KERNEL_FACTOR = 1.25    # kernel level (check_kernel): one operator against its fp32 restatement on seeded inputs -- the value is the
                        # operator's own rounding (output rounding ~2.8e-4 relative L2 for an fp16 result) and is bitwise reproducible
_seen = {}


def _load(path=PATH):
    try:
        with open(path) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return {}


def check(key, value, hard_bound=3e-2, factor=FACTOR):
    """value: measured relative L2 of check `key`.  Fails on value > hard_bound or value > factor x the recorded value."""
    value = float(value)
    print(f"\nPARITY_MEASURE {key} {value:.6e}")
    rec = os.environ.get("MD_PARITY_RECORD")
    if rec:
        d = _load(rec)
        e = d.setdefault("checks", {}).setdefault(key, {})
        e["rel_l2"] = max(value, e.get("rel_l2", 0.0))               # parametrised repeats of one key: the larger one is the budget
        e["hard_bound"] = hard_bound
        with open(rec, "w") as fh:
            json.dump(d, fh, indent=1, sort_keys=True)
    assert value <= hard_bound, (key, value, hard_bound)
    if os.environ.get("MD_PARITY_NO_BUDGET") == "1" or rec:
        return
    budget = _load().get("checks", {}).get(key)
    assert budget is not None, f"no recorded parity budget for '{key}' in {PATH}: record it on a GPU box (see the module docstring)"
    assert value <= factor * budget["rel_l2"], (f"{key}: relative L2 {value:.3e} exceeds {factor} x the recorded {budget['rel_l2']:.3e} "
                                                f"-- a precision regression (SURVEY's bound {hard_bound:.0e} alone would not have caught it)")


def check_kernel(what, got=None, ref=None, factor=KERNEL_FACTOR, value=None):
    """Kernel-level guard, called from the `close` helpers of the operator tests: relative L2 of `got` against the fp32 restatement `ref`,
    keyed by the running test (PYTEST_CURRENT_TEST, parameters included), the helper's label and the call's ordinal inside the test.
    The whole-UNet budgets cannot see a kernel that loses a few bits (measured: P of the d = 40 attention cut to 8 significant bits moves
    G8 / G9 / the 20-step loop by < 1 %: profiles/r06_parity_budget_degraded.log); one operator against its own restatement can."""
    import torch
    node = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" (")[0].split("/")[-1]            # "test_file.py::test_name[params]"
    n = _seen[(node, what)] = _seen.get((node, what), 0) + 1
    key = f"kernel:{node}|{what}|{n}"
    if value is None:                                   # value=: the caller computed the relative L2 itself (operands resident on the GPU)
        g, r = got.detach().double().flatten().cpu(), ref.detach().double().flatten().cpu()
        den = float(r.norm())
        value = float((g - r).norm()) / den if den > 0 else float((g - r).norm())
    value = float(value)
    rec = os.environ.get("MD_PARITY_RECORD")
    if rec:
        d = _load(rec)
        d.setdefault("kernel_checks", {})[key] = value
        with open(rec, "w") as fh:
            json.dump(d, fh, indent=1, sort_keys=True)
        return value
    if os.environ.get("MD_PARITY_NO_BUDGET") == "1":
        return value
    budget = _load().get("kernel_checks", {}).get(key)
    assert budget is not None, f"no recorded kernel budget for '{key}' in {PATH}: record it on a GPU box (see the module docstring)"
    assert value <= factor * budget + 1e-7, (f"{key}: relative L2 {value:.3e} exceeds {factor} x the recorded {budget:.3e} -- this operator lost precision "
                                             f"(its elementwise bound |err| <= 1e-2 max|ref| + 1e-3 alone would not have caught it)")
    return value
