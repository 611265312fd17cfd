"""Host-only logic of the mirrored `least_squares` namespace (no GPU, no library call): the formula parser, the kwargs
dataclasses' validation and defaults, expression plumbing.  Reference: polars_ols/least_squares.py:66-160, utils.py:61-108,
__init__.py:35-295."""
import numpy as np
import pytest


def test_formula_parser_matches_the_reference_front_end():                  # utils.py:61-108 (patsy is absent here)
    from polars_ols_amd.least_squares import Frame, _parse_formula

    exprs, icpt = _parse_formula("y ~ x1 + x2", include_dependent_variable=True)
    assert [e.output_name for e in exprs] == ["y", "x1", "x2"] and icpt
    # the intercept rule is the reference's literal substring test `"-1" not in formula` (utils.py:94)
    exprs, icpt = _parse_formula("y ~ x1 + x2 -1", include_dependent_variable=True)
    assert [e.output_name for e in exprs] == ["y", "x1", "x2"] and not icpt
    for f in ("x1 + x2 - 1", "x1 + x2 + 0", "x1 + x2"):                  # "- 1" with a space and "+ 0" KEEP it there too
        exprs, icpt = _parse_formula(f, include_dependent_variable=False)
        assert [e.output_name for e in exprs] == ["x1", "x2"] and icpt, f
    exprs, icpt = _parse_formula("x1", include_dependent_variable=False)
    assert [e.output_name for e in exprs] == ["x1"] and icpt
    # interactions (utils.py:104-106: product column named "a:b"), the docstring example of the reference (:73-78)
    exprs, icpt = _parse_formula("y ~ x1 + x2 + x3:x4", include_dependent_variable=True)
    assert [e.output_name for e in exprs] == ["y", "x1", "x2", "x3:x4"]
    fr = Frame(x3=np.array([1.0, 2.0, 3.0]), x4=np.array([2.0, 0.5, -1.0]))
    assert np.array_equal(exprs[-1]._column(fr), np.array([2.0, 1.0, -3.0]))
    exprs, _ = _parse_formula("y ~ a*b + c:a:c - b", include_dependent_variable=True)      # a*b = a + b + a:b; "- b" removes a term
    assert [e.output_name for e in exprs] == ["y", "a", "a:b", "c:a"]
    for bad in ("y ~ log(x1)", "y ~ C(group)", "y ~ x1 ** 2", "y ~ x1 / x2", "y ~ (x1 + x2):x3", "y ~ x1 + 2"):
        with pytest.raises(NotImplementedError):
            _parse_formula(bad, include_dependent_variable=True)
    with pytest.raises(AssertionError):                                     # "must provide exactly one LHS variable"
        _parse_formula("x1 + x2", include_dependent_variable=True)
    with pytest.raises(AssertionError):                                     # "can not provide LHS variables in this context"
        _parse_formula("y ~ x1 + x2", include_dependent_variable=False)


def test_kwargs_validation_and_defaults():                                 # least_squares.py:73-77, 101-160
    from polars_ols_amd import OLSKwargs, RLSKwargs, RollingKwargs

    o = OLSKwargs()
    assert (o.alpha, o.l1_ratio, o.max_iter, o.tol, o.positive, o.solve_method, o.rcond, o.null_policy) == \
        (0.0, None, 1000, 1e-5, False, None, None, "ignore")
    r = RLSKwargs()
    assert (r.half_life, r.initial_state_covariance, r.initial_state_mean, r.null_policy) == (None, 10.0, None, "drop")
    w = RollingKwargs()
    assert (w.window_size, w.min_periods, w.use_woodbury, w.alpha, w.null_policy) == (1_000_000, None, None, None, "drop_window")
    for bad in (dict(null_policy="nope"), dict(solve_method="cholesky")):
        with pytest.raises(AssertionError):
            OLSKwargs(**bad)
    with pytest.raises(AssertionError):
        OLSKwargs(null_policy="drop_window")                               # only the rolling models take it (:109-118)
    assert set(OLSKwargs().to_dict()) == {"alpha", "l1_ratio", "max_iter", "tol", "positive", "solve_method", "rcond", "null_policy"}


def test_expression_plumbing_without_a_device():
    from polars_ols_amd import Frame, col
    from polars_ols_amd.least_squares import parse_into_expr, struct

    f = Frame(a=np.arange(4.0), b=np.ones(4))
    assert np.array_equal(f.select(col("a"), (-col("b")).alias("nb"))["nb"], -np.ones(4))
    assert np.array_equal(f.select((col("a") * 2.0).alias("a2"))["a2"], 2.0 * np.arange(4.0))
    assert parse_into_expr("a").output_name == "a" and col("a").alias("z").output_name == "z"
    with pytest.raises(TypeError):
        parse_into_expr(3.0)
    e = col("y").least_squares.ols("a", "b", mode="coefficients").over("g").alias("c")
    assert e._over == "g" and e.output_name == "c" and e._fn is not None       # deferred: nothing ran yet
    assert [x.output_name for x in struct("y1", "y2")._fields] == ["y1", "y2"]
    with pytest.raises(AssertionError):
        col("y").least_squares.ols("a", mode="preds")                          # least_squares.py:266
    with pytest.raises(AssertionError):
        col("c").least_squares.predict("a", null_policy="nope")                # least_squares.py:470


def test_namespace_methods_mirror_the_reference():                          # __init__.py:35-295
    from polars_ols_amd.least_squares import LeastSquares

    expected = {"least_squares", "ols", "multi_target_ols", "wls", "ridge", "lasso", "elastic_net", "rls", "rolling_ols",
                "expanding_ols", "from_formula", "predict", "predict_from_formula"}
    assert expected <= {m for m in dir(LeastSquares) if not m.startswith("_")}
