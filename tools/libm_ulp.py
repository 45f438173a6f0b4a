#!/usr/bin/env python3
"""Development aid: the largest distance (in ULP) between the device libm and the host libm (through the
oracle, i.e. glibc) per function of the libm family, on 200 k random arguments each."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import supersonic_amd as ss  # noqa: E402
from oracle import oracle  # noqa: E402
from helpers import ulp_distance  # noqa: E402

n = 200000
rng = np.random.default_rng(3)
NA = ss.NamedAttribute
schema = ss.TupleSchema([ss.Attribute("w", ss.DOUBLE), ss.Attribute("u", ss.DOUBLE), ss.Attribute("p", ss.DOUBLE), ss.Attribute("q", ss.DOUBLE)])
view = ss.View(schema, [rng.standard_normal(n) * 200.0, rng.random(n) * 2.0 - 1.0, np.abs(rng.standard_normal(n)) * 1000.0 + 1e-9, rng.standard_normal(n) * 8.0])
W, U, P, Q = NA("w"), NA("u"), NA("p"), NA("q")
fns = {"exp(q*8)": ss.Exp(ss.Multiply(Q, ss.ConstDouble(8.0))), "ln(p)": ss.LnQuiet(P), "log10(p)": ss.Log10Quiet(P), "log2(p)": ss.Log2Quiet(P),
       "sin(w)": ss.Sin(W), "cos(w)": ss.Cos(W), "tan(w)": ss.Tan(W), "asin(u)": ss.Asin(U), "acos(u)": ss.Acos(U), "atan(w)": ss.Atan(W),
       "sinh(q)": ss.Sinh(Q), "cosh(q)": ss.Cosh(Q), "tanh(q)": ss.Tanh(Q), "asinh(w)": ss.Asinh(W), "acosh(p+1)": ss.Acosh(ss.Plus(P, ss.ConstDouble(1.0))),
       "atanh(u)": ss.Atanh(U), "pow(p,q)": ss.PowerQuiet(P, Q), "atan2(w,u)": ss.Atan2(W, U)}
ctx = ss.Context(0)
for name, e in fns.items():
    op = ss.Compute(ss.CompoundExpression().AddAs("y", e), ss.ScanView(view))
    got = ss.drain(op.CreateCursor(ctx), 1 << 20).column(0).data
    _s, cols = oracle.run(op)
    d = ulp_distance(got, cols[0][0])
    print("%-12s max %4.0f ULP, %.3f %% of values differ" % (name, d.max(), 100.0 * (d > 0).mean()))
