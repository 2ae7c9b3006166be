"""Drop-in surface: every class / method / function the reference's callers address has the
reference's parameter names, order and defaults (fixture: tests/golden/api_signatures.json, parsed from the reference's
AST by tests/golden/make_api_signatures.py).  Extras are allowed only as further parameters WITH defaults after the
reference's own (or keyword-only)."""
import importlib
import inspect
import json
import os

import pytest

from conftest import GOLDEN_DIR

with open(os.path.join(GOLDEN_DIR, 'api_signatures.json')) as _f:
    SIGS = json.load(_f)


def _resolve(key):
    mod, _, name = key.partition(':')
    obj = importlib.import_module(mod)
    for part in name.split('.'):
        obj = getattr(obj, part)
    return obj


def _same_default(ours, ref_src):
    if ref_src is None:
        return True                                    # required in the reference; a default here only adds valid calls
    if ours is inspect.Parameter.empty:
        return False
    try:
        want = eval(ref_src, {})                       # literals only: numbers, strings, lists, None, booleans
    except Exception:
        return True                                    # an expression over module globals: names are checked, value is not
    return ours == want and type(ours) is type(want)


@pytest.mark.parametrize('key', sorted(SIGS))
def test_signature_matches_reference(key):
    ref = SIGS[key]
    obj = _resolve(key)
    sig = inspect.signature(obj)
    params = [p for p in sig.parameters.values() if p.name != 'self']
    positional = [p for p in params if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
    names = [p.name for p in positional]
    want = [p['name'] for p in ref['params']]
    assert names[:len(want)] == want, (ref['source'], names, want)
    for p, r in zip(positional, ref['params']):
        assert _same_default(p.default, r['default']), (ref['source'], p.name, p.default, r['default'])
    for extra in positional[len(want):]:               # additions must not change what a reference call means
        assert extra.default is not inspect.Parameter.empty, (ref['source'], extra.name)
    has_var = lambda kind: any(p.kind == kind for p in params)
    if ref['varargs']:
        assert has_var(inspect.Parameter.VAR_POSITIONAL), ref['source']
    if ref['kwargs']:
        assert has_var(inspect.Parameter.VAR_KEYWORD), ref['source']
