"""The run-time expression evaluator behind ParsedFunction on the CPU box: every expression the
reference's applications ship (001_Poisson/input/*.json, 1-D/3-D variants) plus the operator/function set of the parser
library, against Python's own evaluation of the same formula."""
import math

import numpy as np
import pytest

from femus_amd import capi

PTS = np.array([[0.3, 0.8, -0.2, 0.5], [1.7, -0.4, 0.9, 0.0], [0.0, 0.0, 0.0, 0.0], [-1.25, 2.5, 0.75, 3.0]])

CASES = [
    ("0.", lambda x, y, z, t: 0.0),
    ("1.", lambda x, y, z, t: 1.0),
    ("0.2", lambda x, y, z, t: 0.2),
    ("0.5+1./pi*atan(1000.*(y-0.8))", lambda x, y, z, t: 0.5 + 1. / math.pi * math.atan(1000. * (y - 0.8))),   # input.json, "left"
    ("10.*exp(-5.*x) - 4.*exp(-x)", lambda x, y, z, t: 10. * math.exp(-5. * x) - 4. * math.exp(-x)),           # input1D.json source
    ("-x^2", lambda x, y, z, t: -(x ** 2)),
    ("2^-1^2", lambda x, y, z, t: 2 ** -(1 ** 2)),
    ("2^3^2", lambda x, y, z, t: 2.0 ** 9),
    ("(x+y)*(z-t)/ (1+x*x)", lambda x, y, z, t: (x + y) * (z - t) / (1 + x * x)),
    ("sin(pi*x)*cos(pi*y)+e", lambda x, y, z, t: math.sin(math.pi * x) * math.cos(math.pi * y) + math.e),
    ("if(x<0.5 & y>=0.1, 3., -2.)", lambda x, y, z, t: 3. if (x < 0.5 and y >= 0.1) else -2.),
    ("max(x,y)+min(z,t)*abs(x)", lambda x, y, z, t: max(x, y) + min(z, t) * abs(x)),
    ("!(x>1) | (y=2.5)", lambda x, y, z, t: 1.0 if ((not x > 1) or y == 2.5) else 0.0),
    ("sqrt(x*x+y*y)-hypot(x,y)+pow(2,z)+atan2(y,x)+log(exp(t))+tanh(x)+floor(y)+ceil(z)+int(2.4)+x%0.7",
     lambda x, y, z, t: pow(2, z) + math.atan2(y, x) + t + math.tanh(x) + math.floor(y) + math.ceil(z) + 2 + math.fmod(x, 0.7)),
    ("1e-3*x + .5e1*y + 3.E+0", lambda x, y, z, t: 1e-3 * x + 5.0 * y + 3.0),
]


@pytest.mark.parametrize("text,fn", CASES)
def test_expression_matches_python(text, fn):
    e = capi.Expr(text, "x,y,z,t")
    ref = np.array([fn(*p) for p in PTS])
    got = np.array([e(p) for p in PTS])
    assert np.allclose(got, ref, rtol=1e-15, atol=1e-15)
    assert np.array_equal(e(PTS), got)                    # batch entry point: identical
    e.destroy()


@pytest.mark.parametrize("bad", ["1+", "foo(x)", "x+q", "(x", "1 2", "if(x,1)", "x,y"])
def test_syntax_errors_are_reported(bad):
    with pytest.raises(capi.FemusHipError):
        capi.Expr(bad, "x,y,z,t")


def test_parser_refuses_unbounded_nesting_instead_of_overflowing_the_stack():
    """a long run of '(' or '-' from a JSON input (advisor, round 1): an error, not a crash"""
    from femus_amd import capi
    for text in ("(" * 100000 + "x" + ")" * 100000, "-" * 100000 + "x", "2" + "^-2" * 50000):
        with pytest.raises(capi.FemusHipError, match="nested deeper"):
            capi.Expr(text)
    e = capi.Expr("(" * 100 + "x+1" + ")" * 100)
    assert e([2.0, 0, 0, 0]) == 3.0


def test_numbers_do_not_depend_on_the_process_locale():
    import locale
    from femus_amd import capi
    old = locale.setlocale(locale.LC_NUMERIC)
    try:
        for name in ("de_DE.UTF-8", "fr_FR.UTF-8", "it_IT.UTF-8"):
            try:
                locale.setlocale(locale.LC_NUMERIC, name)
                break
            except locale.Error:
                continue
        assert capi.Expr("0.5*x + 1.25e1 + .5 + 3.")([2.0, 0, 0, 0]) == 0.5 * 2 + 12.5 + 0.5 + 3.0
    finally:
        locale.setlocale(locale.LC_NUMERIC, old)


def test_program_and_variable_count_are_exposed():
    """fh_expr_program / fh_expr_nvars: what the device evaluator is handed (two-call size query), fh_version"""
    f = capi.Expr("2*x+sin(y)", "x,y")
    assert f.n_variables() == 2
    code, consts = f.program()
    assert code.size >= 5 and 2.0 in consts.tolist()
    g = capi.Expr("2*x+sin(y)", "x,y,z,t")
    assert g.n_variables() == 4 and np.array_equal(g.program()[1], consts)
    assert capi.version().startswith("femus_hip")
    f.destroy(), g.destroy()


def test_oversized_requests_are_errors_not_crashes():
    """no C++ exception and no 32-bit wrap-around leaves the host-side entry points: a box whose node count passes 2^31 is refused,
    and running out of host memory inside the mesh generator comes back as an error code (child process with an address-space limit)"""
    import subprocess
    import sys
    with pytest.raises(capi.FemusHipError, match="32-bit"):
        capi.Mesh.box(1200, 1200, 1200)
    code = ("import resource, sys\n"
            "resource.setrlimit(resource.RLIMIT_AS, (6 << 30, 6 << 30))\n"
            "from femus_amd import capi\n"
            "try:\n    capi.Mesh.box(380, 380, 380)\n    print('BUILT')\n"
            "except capi.FemusHipError as e:\n    print('ERROR', e)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=str(__import__("pathlib").Path(__file__).resolve().parents[1]))
    assert r.returncode == 0 and "ERROR" in r.stdout and "out of host memory" in r.stdout, (r.stdout[-500:], r.stderr[-500:])
