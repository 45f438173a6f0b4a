#!/usr/bin/env python3
"""Golden vectors transcribed (as DATA: inputs and expected outputs) from the reference's own
tests for the Filter -> Project/Compute -> Aggregate (+Sort) path.  Running this script rewrites
tests/golden/reference_tests.json.  Every case cites the reference test it restates.

Conventions
  * None is the NULL literal (`__` in supersonic/testing/block_builder.h).
  * Expression tests of the reference are `BlockBuilder<In..., Out>` tables whose LAST column is
    the expected result of `Factory(AttributeAt(0), AttributeAt(1), ...)`
    (supersonic/testing/expression_test_helper.h:89-95): here "plan" = Compute(expr, Scan).
  * Operation tests use TestDataBuilder: columns are named col0, col1, ... and are NULLABLE
    (supersonic/testing/block_builder.h); here likewise.
  * STRING payload columns of the reference's operation tests are outside the device hot path
    (SURVEY 8f.2); they are substituted by INT64 codes with the same ordering ("a" < "b" < ...),
    which keeps which-rows-pass / which-group / sort-order semantics intact.  Cases whose POINT
    is a STRING feature are not transcribed.
  * FakePredicate(values, nulls) of filter_test.cc:52-135 is restated as an input BOOL column
    "p" that the filter reads with NamedAttribute("p") and then projects away.
"""
import json
import os

I32, I64, U32, U64, F32, F64, BOOL, DATE, STR = "INT32", "INT64", "UINT32", "UINT64", "FLOAT", "DOUBLE", "BOOL", "DATE", "STRING"
DATETIME, BINARY = "DATETIME", "BINARY"
INF, NAN = "inf", "nan"
CASES = []


def expr_case(name, source, types, rows, factory, nullable=True, expect_name=None, expect_type=None,
              expect_nullable=None, expect_error=None):
    """types: input types + output type (last); rows: input values + expected (last)."""
    n_in = len(types) - 1
    CASES.append({
        "name": name, "source": source, "kind": "expression",
        "input": {"schema": [["col%d" % i, types[i], nullable] for i in range(n_in)],
                  "rows": [r[:n_in] for r in rows]},
        "plan": ["Compute", [factory] + [["AttributeAt", i] for i in range(n_in)], "INPUT"],
        "expected": {"types": [types[-1]], "rows": [[r[-1]] for r in rows],
                     "names": [expect_name] if expect_name else None,
                     "nullable": [expect_nullable] if expect_nullable is not None else None},
        "ordered": True, "expect_error": expect_error})


def bind_case(name, source, factory, in_types, in_nullable, out_name, out_type, out_nullable, expect_error=None):
    n_in = len(in_types)
    CASES.append({
        "name": name, "source": source, "kind": "binding",
        "input": {"schema": [["$%d" % i, in_types[i], in_nullable[i]] for i in range(n_in)], "rows": []},
        "plan": ["Compute", [factory] + [["AttributeAt", i] for i in range(n_in)], "INPUT"],
        "expected": {"types": [out_type] if out_type else None, "rows": [], "names": [out_name] if out_name else None,
                     "nullable": [out_nullable] if out_nullable is not None else None},
        "ordered": True, "expect_error": expect_error})


def op_case(name, source, schema, rows, plan, exp_types, exp_rows, ordered=True, exp_names=None, exp_nullable=None,
            expect_error=None):
    CASES.append({
        "name": name, "source": source, "kind": "operation",
        "input": {"schema": schema, "rows": rows}, "plan": plan,
        "expected": {"types": exp_types, "rows": exp_rows, "names": exp_names, "nullable": exp_nullable},
        "ordered": ordered, "expect_error": expect_error})


def cols(types, nullable=True):
    return [["col%d" % i, t, nullable] for i, t in enumerate(types)]


def expr_plan_case(name, source, in_types, expr, out_type, rows, nullable=True):
    """An expression over AttributeAt(i) inputs given as a DSL tree; rows: inputs + expected (last)."""
    n_in = len(in_types)
    CASES.append({
        "name": name, "source": source, "kind": "expression",
        "input": {"schema": [["col%d" % i, in_types[i], nullable] for i in range(n_in)], "rows": [r[:n_in] for r in rows]},
        "plan": ["Compute", expr, "INPUT"],
        "expected": {"types": [out_type], "rows": [[r[-1]] for r in rows], "names": None, "nullable": None},
        "ordered": True, "expect_error": None})


A = "supersonic/expression/core/arithmetic_expressions_test.cc"
E = "supersonic/expression/core/elementary_expressions_test.cc"

# ---- bitwise and shift operators (elementary_expressions_test.cc:304-319,537-617) ---------------
EB = "supersonic/expression/core/elementary_expressions_test.cc"
bind_case("BitwiseNotBinding", EB + ":305", "BitwiseNot", [I32], [False], "(~$0)", I32, False)
bind_case("BitwiseNot_bool_fails", EB + ":306", "BitwiseNot", [BOOL], [False], None, None, None, expect_error=402)
bind_case("BitwiseNot_date_fails", EB + ":307", "BitwiseNot", [DATE], [False], None, None, None, expect_error=402)
expr_case("BitwiseNot_int32", EB + ":309-313", [I32, I32], [[None, None], [0, -1], [1234567, -1234568]], "BitwiseNot")
expr_case("BitwiseNot_uint64", EB + ":315-318", [U64, U64], [[0, 18446744073709551615], [123456789, 18446744073586094826]], "BitwiseNot")
bind_case("BitwiseAndBinding", EB + ":538-540", "BitwiseAnd", [I32, U64], [False, False],
          "(CAST_INT32_TO_INT64($0) & CAST_UINT64_TO_INT64($1))", I64, False)
bind_case("BitwiseAnd_bool_int_fails", EB + ":541", "BitwiseAnd", [BOOL, I32], [False, False], None, None, None, expect_error=402)
bind_case("BitwiseAnd_bool_bool_fails", EB + ":542", "BitwiseAnd", [BOOL, BOOL], [False, False], None, None, None, expect_error=402)
expr_case("BitwiseAnd", EB + ":544-548", [I32, I64, I64], [[12, 12, 12], [12, 1000000000000, 0], [-1, 12345, 12345]], "BitwiseAnd")
bind_case("BitwiseAndNotBinding", EB + ":552-553", "BitwiseAndNot", [I32, I32], [False, False], "(~$0 & $1)", I32, False)
expr_case("BitwiseAndNot", EB + ":555-560", [I32, I32, I32], [[1, 1, 0], [3, 7, 4], [10, 7, 5], [None, 8, None]], "BitwiseAndNot")
bind_case("BitwiseOrBinding", EB + ":564-566", "BitwiseOr", [I32, U64], [False, False],
          "(CAST_INT32_TO_INT64($0) | CAST_UINT64_TO_INT64($1))", I64, False)
expr_case("BitwiseOr", EB + ":568-572", [I64, I64, I64], [[1, 0, 1], [1099511627776, 549755813888, 1649267441664], [3, 5, 7]], "BitwiseOr")
bind_case("BitwiseXorBinding", EB + ":576-577", "BitwiseXor", [U32, U64], [False, False], "(CAST_UINT32_TO_UINT64($0) ^ $1)", U64, False)
expr_case("BitwiseXor", EB + ":579-588", [U32, U32, U32],
          [[1, 1, 0], [2, 1, 3], [3, 1, 2], [5, 10, 15], [19, 39, 52], [None, 1, None], [1, None, None], [None, None, None]], "BitwiseXor")
bind_case("ShiftLeftBinding_uint32_int64", EB + ":592-593", "ShiftLeft", [U32, I64], [False, False], "($0 << $1)", U32, False)
bind_case("ShiftLeftBinding_int64_uint32", EB + ":594", "ShiftLeft", [I64, U32], [False, False], "($0 << $1)", I64, False)
bind_case("ShiftLeft_uint32_bool_fails", EB + ":595", "ShiftLeft", [U32, BOOL], [False, False], None, None, None, expect_error=402)
bind_case("ShiftLeft_bool_int64_fails", EB + ":596", "ShiftLeft", [BOOL, I64], [False, False], None, None, None, expect_error=402)
bind_case("ShiftLeft_date_int32_fails", EB + ":597", "ShiftLeft", [DATE, I32], [False, False], None, None, None, expect_error=402)
expr_case("ShiftLeft", EB + ":600-604", [U32, I32, U32], [[1, 4, 16], [3, 2, 12], [5, 31, 2147483648]], "ShiftLeft")
bind_case("ShiftRightBinding_int32_uint64", EB + ":608", "ShiftRight", [I32, U64], [False, False], "($0 >> $1)", I32, False)
expr_case("ShiftRight", EB + ":610-616", [I32, I32, I32], [[1, 1, 0], [2, 1, 1], [-1, 1, -1], [-4, 1, -2], [None, 1, None]], "ShiftRight")

# ---- casts (templated/cast_expression_test.cc) ------------------------------------------------
expr_plan_case("DateToDatetimeCast", "supersonic/expression/templated/cast_expression_test.cc:335-341", [DATE],
               ["CastToType", "DATETIME", ["AttributeAt", 0]], "DATETIME",
               [[1, 86400000000], [100, 8640000000000], [14600, 1261440000000000]], nullable=False)
expr_plan_case("StandardCast_int32_to_int64", "supersonic/expression/templated/cast_expression_test.cc:318-327", [I32],
               ["CastToType", "INT64", ["AttributeAt", 0]], "INT64",
               [[476, 476], [1054, 1054], [1453, 1453], [1517, 1517], [1914, 1914], [1939, 1939], [2011, 2011]], nullable=False)
expr_plan_case("StandardCast_uint64_to_uint32", "supersonic/expression/templated/cast_expression_test.cc:329-333", [U64],
               ["CastToType", "UINT32", ["AttributeAt", 0]], "UINT32",
               [[1234, 1234], [18446744073709551615, 4294967295], [0, 0]], nullable=False)

expr_plan_case("ProjectingCast_int32_to_uint32", "supersonic/expression/templated/cast_expression_test.cc:289-294", [I32],
               ["CastToType", "UINT32", ["AttributeAt", 0]], "UINT32", [[1, 1], [13, 13], [-1, 4294967295]], nullable=False)
expr_plan_case("ProjectingCast_uint64_to_int64", "supersonic/expression/templated/cast_expression_test.cc:296-300", [U64],
               ["CastToType", "INT64", ["AttributeAt", 0]], "INT64", [[1, 1], [18446744073709551597, -19], [1234567, 1234567]], nullable=False)
expr_plan_case("NoOpCast_double", "supersonic/expression/templated/cast_expression_test.cc:303-309", [F64],
               ["CastToType", "DOUBLE", ["AttributeAt", 0]], "DOUBLE", [[3.14, 3.14], [1.41, 1.41], [9.81, 9.81], [2.71, 2.71]], nullable=False)
# ---- projecting_expressions_test.cc:60-183 over the fixture block (STRING, INT32, DOUBLE, INT32) -----------------
PE = "supersonic/expression/core/projecting_expressions_test.cc"
PROWS = [["1", 12, 5.1, 22], ["2", 13, 6.2, 23], ["3", 14, 7.3, 24], ["4", None, 8.4, 25], [None, 16, None, 26]]
PTYPES = [STR, I32, F64, I32]
for i in range(4):
    CASES.append({"name": "Projecting_AttributeAtSelects_%d_string" % i, "source": PE + ":75-83", "kind": "expression",
                  "input": {"schema": cols(PTYPES), "rows": PROWS}, "plan": ["Compute", ["AttributeAt", i], "INPUT"],
                  "expected": {"types": [PTYPES[i]], "rows": [[r[i]] for r in PROWS], "names": ["col%d" % i], "nullable": [True]},
                  "ordered": True, "expect_error": None})
    CASES.append({"name": "Projecting_NamedAttributeSelects_%d_string" % i, "source": PE + ":85-93", "kind": "expression",
                  "input": {"schema": cols(PTYPES), "rows": PROWS}, "plan": ["Compute", ["NamedAttribute", "col%d" % i], "INPUT"],
                  "expected": {"types": [PTYPES[i]], "rows": [[r[i]] for r in PROWS], "names": ["col%d" % i], "nullable": [True]},
                  "ordered": True, "expect_error": None})
CASES.append({"name": "Projecting_Flat_string", "source": PE + ":156-165", "kind": "expression",
              "input": {"schema": cols(PTYPES), "rows": PROWS},
              "plan": ["Compute", ["CompoundExpression", ["Add", ["AttributeAt", 0]], ["Add", ["AttributeAt", 3]]], "INPUT"],
              "expected": {"types": [STR, I32], "rows": [[r[0], r[3]] for r in PROWS], "names": ["col0", "col3"], "nullable": [True, True]},
              "ordered": True, "expect_error": None})
CASES.append({"name": "Projecting_Alias_string", "source": PE + ":174-183", "kind": "expression",
              "input": {"schema": cols(PTYPES), "rows": PROWS}, "plan": ["Compute", ["Alias", "Some alias", ["AttributeAt", 0]], "INPUT"],
              "expected": {"types": [STR], "rows": [[r[0]] for r in PROWS], "names": ["Some alias"], "nullable": [True]},
              "ordered": True, "expect_error": None})
CASES.append({"name": "Projecting_AliasFailsOnTooManyColumns_string", "source": PE + ":185-193", "kind": "binding",
              "input": {"schema": cols(PTYPES), "rows": []},
              "plan": ["Compute", ["Alias", "Some other alias", ["CompoundExpression", ["Add", ["AttributeAt", 0]], ["Add", ["AttributeAt", 1]]]], "INPUT"],
              "expected": {"types": None, "rows": [], "names": None, "nullable": None}, "ordered": True, "expect_error": 401})

# ---- the libm family (math_expressions_test.cc:32-95,123-135,362-672).  Where the reference's expectation is a libm
# call (log(1000.), pow(3, -0.3) ...) the transcription makes the same call through Python's math module; the
# device libm is held to max_ulp units in the last place of these values (exact for the oracle, which calls libm).
import math  # noqa: E402
MX = "supersonic/expression/core/math_expressions_test.cc"
LIBM_ULP = 4


def libm_case(name, lines, factory, types, rows):
    expr_case(name, MX + lines, types, rows, factory, nullable=False)
    CASES[-1]["max_ulp"] = LIBM_ULP


for nm, fac, out_nullable in [("EXP", "Exp", False), ("LN", "LnNulling", True), ("LN", "LnQuiet", False), ("LOG10", "Log10Nulling", True),
                              ("LOG10", "Log10Quiet", False), ("LOG2", "Log2Nulling", True), ("LOG2", "Log2Quiet", False), ("SIN", "Sin", False),
                              ("COS", "Cos", False), ("TAN", "Tan", False), ("ASIN", "Asin", False), ("ACOS", "Acos", False), ("ATAN", "Atan", False),
                              ("SINH", "Sinh", False), ("COSH", "Cosh", False), ("TANH", "Tanh", False), ("ASINH", "Asinh", False),
                              ("ACOSH", "Acosh", False), ("ATANH", "Atanh", False)]:
    bind_case("MathBinding_" + fac, MX + ":32-77", fac, [F64], [False], nm + "($0)", F64, out_nullable)
bind_case("MathBinding_Cot", MX + ":63", "Cot", [F64], [False], "(CONST_DOUBLE /. TAN($0))", F64, False)
bind_case("MathBinding_ToDegrees", MX + ":73", "ToDegrees", [F64], [False], "($0 * CONST_DOUBLE)", F64, False)
bind_case("MathBinding_ToRadians", MX + ":74", "ToRadians", [F64], [False], "($0 * CONST_DOUBLE)", F64, False)
bind_case("MathBinding_Atan2", MX + ":80-81", "Atan2", [F64, F64], [False, False], "ATAN2($0, $1)", F64, False)
bind_case("MathBinding_LogQuiet", MX + ":84-85", "LogQuiet", [F64, F64], [False, False], "(LN($1) /. LN($0))", F64, False)
bind_case("MathBinding_PowerSignaling", MX + ":86-87", "PowerSignaling", [F64, F64], [False, False], "POW($0, $1)", F64, False)
bind_case("MathBinding_PowerQuiet", MX + ":88-89", "PowerQuiet", [F64, F64], [False, False], "POW($0, $1)", F64, False)
bind_case("MathBinding_PowerNulling", MX + ":91-92", "PowerNulling", [F64, F64], [False, False], "POW($0, $1)", F64, True)
bind_case("MathBinding_LogNulling", MX + ":93-94", "LogNulling", [F64, F64], [False, False], "(LN($1) /. LN($0))", F64, True)
bind_case("MathBindingWithCast_Exp_float", MX + ":124-125", "Exp", [F32], [False], "EXP(CAST_FLOAT_TO_DOUBLE($0))", F64, False)
bind_case("MathBindingWithCast_Sin_int32", MX + ":126-127", "Sin", [I32], [False], "SIN(CAST_INT32_TO_DOUBLE($0))", F64, False)
bind_case("MathBindingWithCast_Cos_uint32", MX + ":128-129", "Cos", [U32], [False], "COS(CAST_UINT32_TO_DOUBLE($0))", F64, False)
bind_case("MathBindingWithCast_Tan_uint64", MX + ":130-131", "Tan", [U64], [False], "TAN(CAST_UINT64_TO_DOUBLE($0))", F64, False)
bind_case("MathBindingWithCast_Sin_int64", MX + ":132-133", "Sin", [I64], [False], "SIN(CAST_INT64_TO_DOUBLE($0))", F64, False)
bind_case("RoundWithPrecisionBinding_float_int32", MX + ":97-103", "RoundWithPrecision", [F32, I32], [False, False],
          "ROUND_WITH_MULTIPLIER(CAST_FLOAT_TO_DOUBLE($0), POW(CONST_DOUBLE, CAST_INT32_TO_DOUBLE($1)))", F64, False)
bind_case("RoundWithPrecisionBinding_uint32_uint64", MX + ":105-110", "RoundWithPrecision", [U32, U64], [False, False],
          "ROUND_WITH_MULTIPLIER(CAST_UINT32_TO_DOUBLE($0), POW(CONST_DOUBLE, CAST_UINT64_TO_DOUBLE($1)))", F64, False)
bind_case("RoundWithPrecision_double_precision_fails", MX + ":112", "RoundWithPrecision", [F64, F64], [False, False], None, None, None, expect_error=402)
libm_case("RoundWithPrecision", ":195-209", "RoundWithPrecision", [F64, I32, F64],
          [[4., 0, 4.], [4., 2, 4.], [4., -1, 0.], [1024., -3, 1000.], [3.14, 1, 3.1], [3.141592, 4, 3.1416], [3.141592, 5, 3.14159], [-0.4, 0, -0.],
           [-0.6, 0, -1.], [0.5, 0, 1.], [0.1, 20, 0.1]])
libm_case("LnNulling", ":362-370", "LnNulling", [F64, F64], [[1., 0.], [1000., math.log(1000.)], [math.exp(1), 1.], [0., None], [-1., None]])
libm_case("LnQuiet", ":372-381", "LnQuiet", [F64, F64], [[1., 0.], [1000., math.log(1000.)], [0., "-inf"], [-1., NAN]])
libm_case("Log10Nulling", ":383-390", "Log10Nulling", [F64, F64], [[1., 0.], [1000., 3.], [0., None], [-1., None]])
libm_case("Log10Quiet", ":392-401", "Log10Quiet", [F64, F64], [[1., 0.], [1000., 3.], [0., "-inf"], [-1., NAN]])
libm_case("Log2Nulling", ":403-411", "Log2Nulling", [F64, F64], [[1., 0.], [1000., math.log2(1000.)], [1024., 10.], [0., None], [-1., None]])
libm_case("Log2Quiet", ":413-423", "Log2Quiet", [F64, F64], [[1., 0.], [1000., math.log2(1000.)], [1024., 10.], [0., "-inf"], [-1., NAN]])
libm_case("Exp", ":425-431", "Exp", [F64, F64], [[0., 1.], [1000., INF], [-1., math.exp(-1.)]])
libm_case("ExpWithIntInputType", ":433-438", "Exp", [I32, F64], [[-4, math.exp(-4.)], [4, math.exp(4.)]])
libm_case("LogNulling", ":440-451", "LogNulling", [F64, I32, F64],
          [[2., 4, 2.], [4., 4, 1.], [10., 2, math.log(2.) / math.log(10.)], [0., 2, None], [2., 0, None], [-1., 5, None], [5., -3, None], [-8., -8, None]])
POW_ROWS = [[1., 0., 1.], [2., 2., 4.], [2.5, 2., 6.25], [4., 0.5, 2.], [-1., 2., 1.], [0., 0., 1.], [0., 0.5, 0.], [6.25, 0.5, 2.5], [0.5, -1., 2.],
            [-1., -1., -1.], [3., -0.3, math.pow(3, -0.3)]]
libm_case("PowerSignaling", ":475-489", "PowerSignaling", [F64, F64, F64], POW_ROWS)
libm_case("PowerNulling", ":496-512", "PowerNulling", [F64, F64, F64], POW_ROWS + [[-1., 0.5, None], [-1., -0.5, None]])
libm_case("PowerQuiet", ":514-532", "PowerQuiet", [F64, F64, F64], POW_ROWS + [[-1., 0.5, NAN], [-1., -0.5, NAN]])
expr_case("PowerSignaling_fails", MX + ":491-494", [F64, F64, F64], [[-1., 0.5, None], [-1., -0.5, None]], "PowerSignaling", nullable=False, expect_error=104)
libm_case("Sin", ":539-545", "Sin", [F64, F64], [[0., 0.], [math.acos(0), 1.], [1., math.sin(1)]])
libm_case("Cos", ":547-553", "Cos", [F64, F64], [[0., 1.], [math.acos(-1), -1.], [1., math.cos(1)]])
libm_case("Tan", ":555-561", "Tan", [F64, F64], [[0., 0.], [123., math.tan(123)], [1., math.tan(1)]])
libm_case("Cot", ":563-569", "Cot", [F64, F64], [[1., 1. / math.tan(1.)], [2., 1. / math.tan(2.)], [3.14, 1. / math.tan(3.14)]])
libm_case("Asin", ":571-577", "Asin", [F64, F64], [[0.5, math.asin(0.5)], [-0.5, math.asin(-0.5)], [0.14, math.asin(0.14)]])
libm_case("Acos", ":579-585", "Acos", [F64, F64], [[0.5, math.acos(0.5)], [-0.5, math.acos(-0.5)], [0.14, math.acos(0.14)]])
libm_case("Atan", ":587-593", "Atan", [F64, F64], [[1., math.atan(1.)], [2., math.atan(2.)], [3.14, math.atan(3.14)]])
libm_case("Atan2", ":595-601", "Atan2", [F64, F64, F64], [[1., 1., math.atan2(1., 1.)], [2., 0., math.atan2(2., 0.)], [3.14, 0., math.atan2(3.14, 0.)]])
libm_case("Sinh", ":603-609", "Sinh", [F64, F64], [[0., 0.], [1.3, math.sinh(1.3)], [2.1, math.sinh(2.1)]])
libm_case("Cosh", ":611-617", "Cosh", [F64, F64], [[0., 1.], [1., math.cosh(1.)], [2., math.cosh(2.)]])
libm_case("Tanh", ":619-625", "Tanh", [F64, F64], [[0., 0.], [123., math.tanh(123)], [1., math.tanh(1)]])
libm_case("Asinh", ":627-633", "Asinh", [F64, F64], [[0.5, math.asinh(0.5)], [-0.5, math.asinh(-0.5)], [0.14, math.asinh(0.14)]])
libm_case("Acosh", ":635-641", "Acosh", [F64, F64], [[0.5, NAN], [-0.5, NAN], [0.14, NAN]])
libm_case("Atanh", ":643-649", "Atanh", [F64, F64], [[1., INF], [2., NAN], [3.14, NAN]])
libm_case("ToDegrees", ":651-657", "ToDegrees", [F64, F64], [[0., 0.], [math.pi, 180.], [math.pi / 2., 90.]])
libm_case("ToRadians", ":659-665", "ToRadians", [F64, F64], [[0., 0.], [180., math.pi], [90., math.pi / 2.]])

# ---- vector_logic_test.cc:46-83: left[i] = (i % 3 == 0), right[i] = (i % 5 == 0) ----------------------------
VL = "supersonic/expression/vector/vector_logic_test.cc"
expr_case("VectorLogic_Or", VL + ":46-54", [BOOL, BOOL, BOOL], [[i % 3 == 0, i % 5 == 0, i % 3 == 0 or i % 5 == 0] for i in range(200)], "Or", nullable=False)
expr_case("VectorLogic_And", VL + ":56-63", [BOOL, BOOL, BOOL], [[i % 3 == 0, i % 5 == 0, i % 15 == 0] for i in range(150)], "And", nullable=False)
expr_case("VectorLogic_AndNot", VL + ":65-73", [BOOL, BOOL, BOOL], [[i % 3 == 0, i % 5 == 0, i % 3 != 0 and i % 5 == 0] for i in range(170)], "AndNot",
          nullable=False)
expr_case("VectorLogic_Not", VL + ":75-83", [BOOL, BOOL], [[i % 3 == 0, i % 3 != 0] for i in range(120)], "Not", nullable=False)

# The TestCastBinding<from, to, is_implicit>(success) matrix (cast_expression_test.cc:63-287), the entries with
# is_implicit == false (CastTo is the explicit cast; implicit ones only arise inside promotions).  Targets of
# type BINARY / DATA_TYPE are outside the path and left out.
CX = "supersonic/expression/templated/cast_expression_test.cc"
EXPLICIT_CASTS = [
    (U32, I32, True, ":80-95"), (U32, U32, True, ":80-95"), (U32, I64, True, ":80-95"), (U32, U64, True, ":80-95"), (U32, F32, True, ":80-95"),
    (U32, F64, True, ":80-95"), (U32, DATE, False, ":80-95"), (U32, DATETIME, False, ":80-95"), (U32, STR, False, ":80-95"), (U32, BOOL, False, ":80-95"),
    (I64, I32, True, ":97-121"), (I64, U32, True, ":97-121"), (I64, F32, True, ":97-121"), (I64, U64, True, ":97-121"), (I64, DATE, False, ":97-121"),
    (I64, DATETIME, False, ":97-121"), (I64, STR, False, ":97-121"), (I64, BOOL, False, ":97-121"),
    (U64, I32, True, ":123-147"), (U64, U32, True, ":123-147"), (U64, F32, True, ":123-147"), (U64, U64, True, ":123-147"), (U64, I64, True, ":123-147"),
    (F32, I32, False, ":149-172"), (F32, U32, False, ":149-172"), (F32, I64, False, ":149-172"), (F32, U64, False, ":149-172"), (F32, F64, True, ":149-172"),
    (F32, F32, True, ":149-172"),
    (F64, U32, False, ":174-193"), (F64, U64, False, ":174-193"), (F64, F32, True, ":174-193"),
    (DATE, DATE, True, ":195-213"), (DATE, DATETIME, True, ":195-213"), (DATE, I32, False, ":195-213"), (DATE, U32, False, ":195-213"),
    (DATE, I64, False, ":195-213"), (DATE, U64, False, ":195-213"), (DATE, F32, False, ":195-213"), (DATE, F64, False, ":195-213"),
    (DATE, STR, False, ":195-213"), (DATE, BOOL, False, ":195-213"),
    (DATETIME, DATETIME, True, ":215-231"), (BOOL, BOOL, True, ":233-249"), (STR, STR, True, ":251-269"),
    (STR, I32, False, ":251-269"), (STR, U32, False, ":251-269"), (STR, I64, False, ":251-269"), (STR, U64, False, ":251-269"), (STR, F32, False, ":251-269"),
    (STR, F64, False, ":251-269"), (STR, DATE, False, ":251-269"), (STR, DATETIME, False, ":251-269"), (STR, BOOL, False, ":251-269"),
]
for (frm, to, ok, lines) in EXPLICIT_CASTS:
    CASES.append({
        "name": "CastBinding_%s_to_%s" % (frm, to), "source": CX + lines, "kind": "binding",
        "input": {"schema": [["$0", frm, False]], "rows": []},
        "plan": ["Compute", ["CastToType", to, ["AttributeAt", 0]], "INPUT"],
        "expected": {"types": [to] if ok else None, "rows": [],
                     "names": [("$0" if frm == to else "CAST_%s_TO_%s($0)" % (frm, to))] if ok else None, "nullable": [False] if ok else None},
        "ordered": True, "expect_error": None if ok else 402})

# ---- arithmetic (arithmetic_expressions_test.cc) ---------------------------------------------
bind_case("NegateBinding_double", A + ":25-27", "Negate", [F64], [False], "(-$0)", F64, False)
bind_case("NegateBinding_int32", A + ":25-28", "Negate", [I32], [False], "(-$0)", I32, False)
bind_case("NegateBinding_uint32", A + ":25-29", "Negate", [U32], [False], "(-$0)", I32, False)
expr_case("Negate_float", A + ":34-40", [F32, F32], [[3., -3.], [0., -0.], [-3., 3.], [11.2, -11.2]], "Negate")
expr_case("Negate_uint64", A + ":42-46", [U64, I64], [[0, 0], [4, -4], [12314, -12314]], "Negate")
expr_case("Negate_uint32_null", A + ":48-52", [U32, I32], [[13, -13], [None, None], [0, 0]], "Negate")
bind_case("PlusBinding_int64", A + ":60-61", "Plus", [I64, I64], [False, False], "($0 + $1)", I64, False)
bind_case("PlusBinding_nullable", A + ":62-63", "Plus", [I64, I64], [False, True], "($0 + $1)", I64, True)
bind_case("PlusBinding_uint64_float", A + ":64-65", "Plus", [U64, F32], [False, False],
          "(CAST_UINT64_TO_DOUBLE($0) + CAST_FLOAT_TO_DOUBLE($1))", F64, False)
expr_case("Plus", A + ":68-75", [I64, I64, I64], [[-1, 1, 0], [-2, 2, 0], [2, 2, 4], [13, 1, 14]], "Plus")
expr_case("PlusNullable", A + ":77-83", [I64, I64, I64], [[1, None, None], [-1, 2, 1], [None, 2, None]], "Plus")
expr_case("PlusLeftColumnNullable", A + ":85-91", [I64, I64, I64], [[-1, 2, 1], [-1, -2, -3], [None, 2, None]], "Plus")
expr_case("PlusDifferentTypes", A + ":93-100", [I64, I32, I64], [[-1, 1, 0], [-2, 2, 0], [2, 2, 4], [3, 1, 4]], "Plus")
bind_case("Binding_Minus", A + ":103-104", "Minus", [I64, I64], [False, False], "($0 - $1)", I64, False)
bind_case("Binding_Multiply", A + ":105-106", "Multiply", [I64, I64], [False, False], "($0 * $1)", I64, False)
bind_case("Binding_DivideSignaling", A + ":107-108", "DivideSignaling", [F64, F64], [False, False], "($0 /. $1)", F64, False)
bind_case("Binding_DivideQuiet", A + ":109-110", "DivideQuiet", [F64, F64], [False, False], "($0 /. $1)", F64, False)
bind_case("Binding_CppDivideSignaling", A + ":111-112", "CppDivideSignaling", [I32, I32], [False, False], "($0 / $1)", I32, False)
bind_case("Binding_ModulusSignaling", A + ":113-114", "ModulusSignaling", [U32, U32], [False, False], "($0 % $1)", U32, False)
bind_case("Binding_CppDivideNulling", A + ":116-117", "CppDivideNulling", [I64, I64], [False, False], "($0 / $1)", I64, True)
bind_case("Binding_ModulusNulling", A + ":118-119", "ModulusNulling", [I32, I32], [False, False], "($0 % $1)", I32, True)
bind_case("Binding_DivideNulling", A + ":120-121", "DivideNulling", [F64, F64], [False, False], "($0 /. $1)", F64, True)
bind_case("Binding_ModulusNulling_cast", A + ":123-124", "ModulusNulling", [I64, I32], [False, False],
          "($0 % CAST_INT32_TO_INT64($1))", I64, True)
expr_case("Minus", A + ":126-133", [I64, I32, I64], [[1, None, None], [2, -1, 3], [-1, 2, -3], [None, 2, None]], "Minus")
expr_case("Multiply", A + ":135-143", [I64, I32, I64], [[1, None, None], [2, 2, 4], [2, 0, 0], [20, 20, 400], [None, 2, None]], "Multiply")
expr_case("DivideQuiet", A + ":145-153", [I64, I32, F64], [[2, 2, 1.], [3, 1, 3.], [1, 2, 0.5], [1, 0, INF], [0, 0, NAN]], "DivideQuiet")
expr_case("DivideNulling", A + ":155-163", [I64, F32, F64], [[2, 2, 1.], [3, 1, 3.], [1, 2, 0.5], [1, 0, None], [0, 0, None]], "DivideNulling")
expr_case("DivideSignaling_fails", A + ":172-175", [F64, I32, F64], [[1, 0, None], [0, 0, None]], "DivideSignaling", expect_error=104)
expr_case("CppDivideNulling", A + ":178-186", [I64, I32, I64], [[5, 2, 2], [2, 2, 1], [-3, 1, -3], [3, 0, None], [0, 3, 0]], "CppDivideNulling")
expr_case("CppDivideSignaling", A + ":188-197", [I32, I32, I32],
          [[5, 2, 2], [2, 2, 1], [-3, 1, -3], [0, 3, 0], [None, 0, None], [0, None, None]], "CppDivideSignaling")
expr_case("CppDivideSignaling_fails", A + ":199-203", [I32, I32, I32], [[3, 0, None], [0, 0, None], [-1, 0, None]],
          "CppDivideSignaling", expect_error=104)
expr_case("ModulusNulling", A + ":206-216", [I32, I32, I32],
          [[5, 2, 1], [-1, 5, -1], [0, 3, 0], [7, 5, 2], [None, 4, None], [4, 0, None], [4, -3, 1]], "ModulusNulling")
expr_case("ModulusSignaling", A + ":218-227", [I32, I32, I32],
          [[5, 2, 1], [-1, 5, -1], [0, 3, 0], [7, 5, 2], [None, 4, None], [-4, -3, -1]], "ModulusSignaling")
expr_case("ModulusSignaling_fails", A + ":229-232", [I64, I64, I64], [[1, 0, None], [0, 0, None]], "ModulusSignaling", expect_error=104)

# ---- logic / IS NULL / IF (elementary_expressions_test.cc) -----------------------------------
bind_case("NotBinding", E + ":256-257", "Not", [BOOL], [False], "(NOT $0)", BOOL, False)
bind_case("NotBinding_int32_fails", E + ":261", "Not", [I32], [False], None, None, None, expect_error=402)
expr_case("Not", E + ":264-270", [BOOL, BOOL], [[False, True], [True, False], [None, None]], "Not")
expr_case("IsNull", E + ":272-281", [I32, BOOL], [[1, False], [None, True], [4, False], [1, False], [0, False], [None, True]], "IsNull")
bind_case("Binding_And", E + ":323-324", "And", [BOOL, BOOL], [False, False], "($0 AND $1)", BOOL, False)
bind_case("Binding_AndNot", E + ":325", "AndNot", [BOOL, BOOL], [False, False], "($0 !&& $1)", BOOL, False)
bind_case("Binding_Or", E + ":326", "Or", [BOOL, BOOL], [False, False], "($0 OR $1)", BOOL, False)
bind_case("Binding_Xor", E + ":327", "Xor", [BOOL, BOOL], [False, False], "($0 XOR $1)", BOOL, False)
bind_case("BindingIfNull_notnull_left", E + ":331", "IfNull", [I32, I32], [False, False], "$0", I32, False)
bind_case("BindingIfNull_nullable_left", E + ":332-333", "IfNull", [I32, I32], [True, False], "IFNULL($0, $1)", I32, False)
bind_case("BindingIfNull_both_nullable", E + ":335-336", "IfNull", [I32, I32], [True, True], "IFNULL($0, $1)", I32, True)
bind_case("BindingIfNull_cast_left", E + ":342-343", "IfNull", [I32, I64], [False, False], "CAST_INT32_TO_INT64($0)", I64, False)
bind_case("BindingIfNull_bool_int_fails", E + ":347", "IfNull", [BOOL, I32], [False, False], None, None, None, expect_error=402)
bind_case("BindingIfNullWithCast", E + ":350-353", "IfNull", [I32, F32], [True, False], "IFNULL(CAST_INT32_TO_FLOAT($0), $1)", F32, False)
expr_case("IfNull", E + ":355-364", [I64, I64, I64],
          [[1, 20, 1], [10, 20, 10], [None, 20, 20], [None, 15, 15], [7, None, 7], [None, None, None]], "IfNull")
expr_case("IfNullWithCast", E + ":366-375", [I64, I32, I64],
          [[1, 20, 1], [10, 20, 10], [None, 20, 20], [None, 15, 15], [7, None, 7], [None, None, None]], "IfNull")
T, F, N = True, False, None
expr_case("Xor", E + ":396-410", [BOOL, BOOL, BOOL],
          [[T, T, F], [T, F, T], [T, N, N], [F, T, T], [F, F, F], [F, N, N], [N, T, N], [N, F, N], [N, N, N]], "Xor")
bind_case("Xor_bool_int_fails", E + ":398", "Xor", [BOOL, I32], [False, False], None, None, None, expect_error=402)
expr_case("And", E + ":421-433", [BOOL, BOOL, BOOL],
          [[F, F, F], [F, T, F], [F, N, F], [T, F, F], [T, T, T], [T, N, N], [N, F, F], [N, T, N], [N, N, N]], "And")
expr_case("AndWithoutNulls", E + ":435-442", [BOOL, BOOL, BOOL], [[F, F, F], [F, T, F], [T, F, F], [T, T, T]], "And", nullable=False)
expr_case("Or", E + ":462-474", [BOOL, BOOL, BOOL],
          [[F, F, F], [F, T, T], [F, N, N], [T, F, T], [T, T, T], [T, N, T], [N, F, N], [N, T, T], [N, N, N]], "Or")
expr_case("OrNotNullable", E + ":495-502", [BOOL, BOOL, BOOL], [[F, F, F], [T, F, T], [F, T, T], [T, T, T]], "Or", nullable=False)
expr_case("AndNot", E + ":504-516", [BOOL, BOOL, BOOL],
          [[F, F, F], [F, T, T], [F, N, N], [T, F, F], [T, T, F], [T, N, F], [N, F, F], [N, T, N], [N, N, N]], "AndNot")
bind_case("BasicIf_binding", E + ":622-623", "If", [BOOL, I32, I32], [False, False, False], "IF $0 THEN $1 ELSE $2", I32, False)
expr_case("BasicIf", E + ":625-629", [BOOL, I32, I32, I32], [[T, 1, 2, 1], [F, 3, 4, 4], [T, 1, 1, 1]], "If")
expr_case("IfWithNullCondition", E + ":645-649", [BOOL, I32, I32, I32], [[F, 1, 2, 2], [N, 1, 2, 2], [T, 1, 2, 1]], "If")
expr_case("IfWithNullThen", E + ":659-663", [BOOL, DATE, DATE, DATE], [[T, 1, 2, 1], [T, N, 2, N], [F, N, 2, 2]], "If")
expr_case("IfWithNullOtherwise", E + ":673-677", [BOOL, BOOL, BOOL, BOOL], [[T, T, F, T], [T, T, N, T], [F, T, N, N]], "If")
expr_case("IfWithAllNullable", E + ":687-701", [BOOL, I32, I32, I32],
          [[T, 1, 2, 1], [F, 1, 2, 2], [N, 1, 2, 2], [T, N, 2, N], [F, N, 2, 2], [N, N, 2, 2], [T, 1, N, 1], [F, 1, N, N],
           [N, 1, N, N], [T, N, N, N], [F, N, N, N], [N, N, N, N]], "If")

bind_case("BasicNullingIf_binding", E + ":632-635", "NullingIf", [BOOL, I32, I64], [False, False, False],
          "IF $0 THEN CAST_INT32_TO_INT64($1) ELSE $2", I64, False)
expr_case("BasicNullingIf", E + ":637-641", [BOOL, I32, I64, I64], [[T, 1, 2, 1], [F, 3, 4, 4], [T, 1, 1, 1]], "NullingIf")
expr_case("NullingIfWithNullCondition", E + ":651-655", [BOOL, I32, I32, I32], [[F, 1, 2, 2], [N, 1, 2, N], [T, 1, 2, 1]], "NullingIf")
expr_case("NullingIfWithNullThen", E + ":665-669", [BOOL, DATE, DATE, DATE], [[T, 1, 2, 1], [T, N, 2, N], [F, N, 2, 2]], "NullingIf")
expr_case("NullingIfWithNullOtherwise", E + ":679-683", [BOOL, BOOL, BOOL, BOOL], [[T, T, F, T], [T, T, N, T], [F, T, N, N]], "NullingIf")
expr_case("NullingIfWithAllNullable", E + ":703-716", [BOOL, I32, I32, I32],
          [[T, 1, 2, 1], [F, 1, 2, 2], [N, 1, 2, N], [T, N, 2, N], [F, N, 2, 2], [N, N, 2, N], [T, 1, N, 1], [F, 1, N, N],
           [N, 1, N, N], [T, N, N, N], [F, N, N, N], [N, N, N, N]], "NullingIf")
expr_case("XorWithoutNulls", E + ":412-419", [BOOL, BOOL, BOOL], [[T, T, F], [T, F, T], [F, T, T], [F, F, F]], "Xor", nullable=False)
expr_case("IsNullNotNull_string", E + ":283-293", [STR, BOOL],
          [["I am", F], ["You are", F], ["He/she/it is", F], ["We are", F], ["You are", F], ["They are", F], ["", F]], "IsNull", nullable=False)
CASES.append({"name": "Cast_int32_to_double", "source": E + ":96-97", "kind": "binding", "input": {"schema": [["$0", I32, False]], "rows": []},
              "plan": ["Compute", ["CastToType", "DOUBLE", ["AttributeAt", 0]], "INPUT"],
              "expected": {"types": [F64], "rows": [], "names": ["CAST_INT32_TO_DOUBLE($0)"], "nullable": [False]}, "ordered": True, "expect_error": None})
expr_plan_case("Cast_int32_to_double_values", E + ":99-103", [I32], ["CastToType", "DOUBLE", ["AttributeAt", 0]], "DOUBLE",
               [[1, 1.0], [2, 2.0], [-1, -1.0]], nullable=False)
# CalculateCommonType (elementary_expressions_test.cc:40-86), observed through IFNULL, which binds both arguments to it
for (a_, b_, r_, ln) in [(F64, I32, F64, "40-46"), (F64, I64, F64, "40-46"), (F64, U32, F64, "40-46"), (F64, U64, F64, "40-46"), (F64, F32, F64, "40-46"),
                         (F32, I32, F32, "48-54"), (F32, I64, F64, "48-54"), (F32, U32, F32, "48-54"), (F32, U64, F64, "48-54"), (F32, F64, F64, "48-54"),
                         (I64, I32, I64, "56-62"), (I64, F32, F64, "56-62"), (I64, U32, I64, "56-62"), (I64, U64, I64, "56-62"), (I64, F64, F64, "56-62"),
                         (I32, I64, I64, "64-70"), (I32, F32, F32, "64-70"), (I32, U32, I64, "64-70"), (I32, U64, I64, "64-70"), (I32, F64, F64, "64-70"),
                         (U32, I64, I64, "72-78"), (U32, F32, F32, "72-78"), (U32, I32, I64, "72-78"), (U32, U64, U64, "72-78"), (U32, F64, F64, "72-78"),
                         (U64, I64, I64, "80-86"), (U64, F32, F64, "80-86"), (U64, I32, I64, "80-86"), (U64, U32, U64, "80-86"), (U64, F64, F64, "80-86")]:
    bind_case("CommonType_%s_%s" % (a_, b_), E + ":" + ln, "IfNull", [a_, b_], [True, False], None, r_, False)

# ---- the Primer's column add (test/guide/primer.cc:205-221) -----------------------------------
expr_case("Primer_ColumnAdd", "test/guide/primer.cc:205-221", [I32, I32, I32],
          [[a, b, c] for a, b, c in zip(range(8), [3, 4, 6, 8, 1, 2, 2, 9], [3, 5, 8, 11, 5, 7, 8, 16])], "Plus", nullable=False)

# ---- Compute (supersonic/cursor/core/compute_test.cc:30-81; STRING col0 dropped) ---------------
CT = "supersonic/cursor/core/compute_test.cc"
compute_schema = [["col1", I32, True], ["col2", F64, True], ["col3", I64, True]]
compute_rows = [[12, 5.0, 5], [13, 6.0, 6], [None, None, None]]
op_case("Compute_NamedAttribute", CT + ":51-62", compute_schema, compute_rows,
        ["Compute", ["CompoundExpression", ["AddAs", "col0", ["NamedAttribute", "col1"]]], "INPUT"],
        [I32], [[12], [13], [None]], exp_names=["col0"])
op_case("Compute_CompoundWithArithmetics", CT + ":64-81", compute_schema, compute_rows,
        ["Compute", ["CompoundExpression", ["AddAs", "col0", ["NamedAttribute", "col1"]],
                     ["AddAs", "col1", ["Plus", ["NamedAttribute", "col1"], ["NamedAttribute", "col3"]]]], "INPUT"],
        [I32, I64], [[12, 17], [13, 19], [None, None]], exp_names=["col0", "col1"])

# ---- Filter (supersonic/cursor/core/filter_test.cc:151-351) ------------------------------------
FT = "supersonic/cursor/core/filter_test.cc"


def filter_case(name, src, rows, pvals, pnulls, expected, n_payload=2):
    schema = [["col0", I32, True], ["col1", I64, True], ["p", BOOL, pnulls is not None]]
    data = [[r[0], r[1], (None if (pnulls and pnulls[i]) else pvals[i])] for i, r in enumerate(rows)]
    op_case(name, src, schema, data,
            ["Filter", ["NamedAttribute", "p"], ["ProjectNamedAttributes", ["col0", "col1"]], "INPUT"],
            [I32, I64], expected)


two = [[1, 65], [3, 66]]   # (1,"A"), (3,"B")
filter_case("Filter_AllPassing", FT + ":151-163", two, [True, True], None, [[1, 65], [3, 66]])
filter_case("Filter_NonePassing", FT + ":165-175", two, [False, False], None, [])
filter_case("Filter_OnePassing", FT + ":177-191", two, [False, True], None, [[3, 66]])
K2 = 2048  # Cursor::kDefaultRowCount * 2
filter_case("Filter_ResultsAcrossBlocks", FT + ":206-226", [[i, 65] for i in range(K2)], [bool(i % 2) for i in range(K2)], None,
            [[i, 65] for i in range(K2) if i % 2])
filter_case("Filter_NoResultsFromFirstBlock", FT + ":228-247", [[i, 65] for i in range(K2)], [i >= 1024 for i in range(K2)], None,
            [[i, 65] for i in range(1024, K2)])
filter_case("Filter_NoResultsFromSecondBlock", FT + ":249-267", [[i, 65] for i in range(K2)], [i < 1024 for i in range(K2)], None,
            [[i, 65] for i in range(1024)])
filter_case("Filter_NullableAllPassing", FT + ":269-283", two, [True, True], [False, False], [[1, 65], [3, 66]])
filter_case("Filter_NullableOnePassing", FT + ":285-300", two, [True, True], [True, False], [[3, 66]])
op_case("Filter_FilterOnProjected", FT + ":302-316", cols([I32, I64]), two,
        ["Filter", ["Equal", ["NamedAttribute", "col0"], ["ConstInt32", 1]], ["ProjectNamedAttribute", "col0"], "INPUT"],
        [I32], [[1]], exp_names=["col0"])
op_case("Filter_FilterOnDropped", FT + ":318-332", cols([I32, I64]), two,
        ["Filter", ["Equal", ["NamedAttribute", "col1"], ["ConstInt64", 65]], ["ProjectNamedAttribute", "col0"], "INPUT"],
        [I32], [[1]], exp_names=["col0"])

# ---- ScalarAggregate (supersonic/cursor/core/aggregate_scalar_test.cc:53-90) ---------------------
# An aggregation named X_DISTINCT is AddDistinctAggregation(X, ...).
ST = "supersonic/cursor/core/aggregate_scalar_test.cc"
spec5 = [["MAX", "col0", "max"], ["SUM", "col0", "sum"], ["COUNT", "", "count(*)"], ["COUNT", "col0", "count"],
         ["COUNT_DISTINCT", "col0", "count distinct"]]
op_case("ScalarAggregate_Integers", ST + ":53-70", cols([I32]), [[13], [3], [3], [None], [7]],
        ["ScalarAggregate", spec5, "INPUT"], [I32, I32, U64, U64, U64], [[13, 26, 5, 4, 3]],
        exp_names=["max", "sum", "count(*)", "count", "count distinct"], exp_nullable=[True, True, False, False, False])
op_case("ScalarAggregate_EmptyInput", ST + ":72-86", cols([I32]), [],
        ["ScalarAggregate", spec5, "INPUT"], [I32, I32, U64, U64, U64], [[None, None, 0, 0, 0]])

# ---- GroupAggregate (supersonic/cursor/core/aggregate_groups_test.cc) --------------------------
GT = "supersonic/cursor/core/aggregate_groups_test.cc"
op_case("Group_SimpleAggregation", GT + ":102-117", cols([I32]), [[1], [3]],
        ["GroupAggregate", ["CompoundSingleSourceProjector"], [["SUM", "col0", "sum"]], "INPUT"],
        [I32], [[4]], exp_names=["sum"], exp_nullable=[True])
op_case("Group_DistinctAggregation", GT + ":220-236", cols([I32]), [[3], [4], [4], [3]],
        ["GroupAggregate", ["CompoundSingleSourceProjector"], [["SUM_DISTINCT", "col0", "sum"]], "INPUT"], [I32], [[7]])
op_case("Group_DistinctCountAggregation", GT + ":238-255", cols([I32]), [[3], [4], [4], [3]],
        ["GroupAggregate", ["CompoundSingleSourceProjector"], [["COUNT_DISTINCT", "col0", "count", I32]], "INPUT"], [I32], [[2]])
op_case("Group_DistinctCountAggregationNeedsInputColumn", GT + ":257-270", cols([I32]), [[3]],
        ["GroupAggregate", ["CompoundSingleSourceProjector"], [["COUNT_DISTINCT", "", "count", I32]], "INPUT"], None, [], expect_error=403)
op_case("Group_CountWithInputColumn", GT + ":119-133", cols([I32]), [[1], [3]],
        ["GroupAggregate", ["CompoundSingleSourceProjector"], [["COUNT", "col0", "count"]], "INPUT"],
        [U64], [[2]], exp_nullable=[False])
op_case("Group_AggregationWithGroupBy", GT + ":272-295", cols([I32, I32], nullable=False), [[1, 3], [3, -3], [1, 4], [3, -5]],
        ["GroupAggregate", ["ProjectNamedAttribute", "col0"], [["SUM", "col1", "sum"]], "INPUT"],
        [I32, I32], [[1, 7], [3, -8]], ordered=False, exp_names=["col0", "sum"], exp_nullable=[False, True])
# GroupAggregateOptions::max_unique_keys_in_result = 2 (CreateGroupAggregate's last argument): the first limit + 1 keys in
# first-seen order keep a row, rows with any other key (5 here) aggregate into the last kept row (key 4)
op_case("Group_AggregationWithGroupBy_UniqueRowLimit", GT + ":296-328", cols([I32, I32], nullable=False),
        [[1, 3], [3, -3], [1, 4], [3, -5], [4, 5], [3, -1], [5, 1], [4, 3], [1, -2]],
        ["GroupAggregate", ["ProjectNamedAttribute", "col0"], [["SUM", "col1", "sum"]], "INPUT", {"max_unique_keys_in_result": 2}],
        [I32, I32], [[1, 5], [3, -9], [4, 9]], ordered=False, exp_names=["col0", "sum"], exp_nullable=[False, True])
op_case("Group_GroupByNullableColumn", GT + ":330-354", cols([I32, I32]), [[3, -3], [None, 4], [3, -5], [None, 1]],
        ["GroupAggregate", ["ProjectNamedAttribute", "col0"], [["SUM", "col1", "sum"]], "INPUT"],
        [I32, I32], [[3, -8], [None, 5]], ordered=False, exp_names=["col0", "sum"], exp_nullable=[True, True])
# (a NULLABLE 64-bit key needs 65 packed key bits, beyond the device table's 64: INT32 codes here)
op_case("Group_GroupBySecondColumn", GT + ":356-377", cols([I32, I32]), [[-3, 2], [2, 1], [3, 1], [-2, 2]],   # foo=2, bar=1
        ["GroupAggregate", ["ProjectNamedAttribute", "col1"], [["SUM", "col0", "sum"]], "INPUT"],
        [I32, I32], [[2, -5], [1, 5]], ordered=False, exp_names=["col1", "sum"])
two_key_rows = [[2, 1, 3], [1, 2, -3], [2, 1, 4], [1, 3, -5]]                                                  # foo=2, bar=1
op_case("Group_GroupByTwoColumns", GT + ":379-401", cols([I32, I32, I32], nullable=False), two_key_rows,
        ["GroupAggregate", ["ProjectNamedAttributes", ["col0", "col1"]], [["SUM", "col2", "sum"]], "INPUT"],
        [I32, I32, I32], [[2, 1, 7], [1, 2, -3], [1, 3, -5]], ordered=False)
op_case("Group_TwoColumnsMultipleAggregations", GT + ":403-429", cols([I32, I32, I32], nullable=False), two_key_rows,
        ["GroupAggregate", ["ProjectNamedAttributes", ["col0", "col1"]],
         [["SUM", "col2", "sum"], ["MIN", "col2", "min"], ["COUNT", "", "count"]], "INPUT"],
        [I32, I32, I32, I32, U64], [[2, 1, 7, 3, 2], [1, 2, -3, -3, 1], [1, 3, -5, -5, 1]], ordered=False)

EMPTYP = ["CompoundSingleSourceProjector"]
op_case("Group_CountWithNullableInputColumn", GT + ":135-149", cols([I32]), [[1], [None]],
        ["GroupAggregate", EMPTYP, [["COUNT", "col0", "count"]], "INPUT"], [U64], [[1]], exp_nullable=[False])
op_case("Group_CountAll", GT + ":151-165", cols([I32]), [[1], [None]],
        ["GroupAggregate", EMPTYP, [["COUNT", "", "count"]], "INPUT"], [U64], [[2]], exp_nullable=[False])
op_case("Group_AggregationWithOnlyNullInputs", GT + ":167-180", cols([I32]), [[None], [None]],
        ["GroupAggregate", EMPTYP, [["SUM", "col0", "sum"]], "INPUT"], [I32], [[None]])
op_case("Group_OutputTypeDifferentFromInputType", GT + ":182-197", cols([I32]), [[1], [3]],
        ["GroupAggregate", EMPTYP, [["SUM", "col0", "sum", I64]], "INPUT"], [I64], [[4]], exp_names=["sum"], exp_nullable=[True])
op_case("Group_MultipleAggregations", GT + ":199-218", cols([I32]), [[1], [2], [3]],
        ["GroupAggregate", EMPTYP, [["SUM", "col0", "sum"], ["MAX", "col0", "max"], ["MIN", "col0", "min"]], "INPUT"],
        [I32, I32, I32], [[6, 3, 1]], exp_names=["sum", "max", "min"])
op_case("Group_GroupByWithoutAggregateFunctions_string", GT + ":430-447", cols([STR]), [["foo"], ["bar"], ["foo"], ["bar"]],
        ["GroupAggregate", ["ProjectNamedAttribute", "col0"], [], "INPUT"], [STR], [["foo"], ["bar"]], ordered=False)
# a GroupAggregate without key columns returns NO row for an empty input (a ScalarAggregate returns one)
op_case("Group_AggregationOnEmptyInput", GT + ":449-460", cols([DATETIME]), [],
        ["GroupAggregate", EMPTYP, [["MIN", "col0", "min"]], "INPUT"], [DATETIME], [])
op_case("Group_AggregationOnEmptyInputWithGroupByColumn_string", GT + ":462-475", cols([STR, DATETIME]), [],
        ["GroupAggregate", ["ProjectNamedAttribute", "col0"], [["MIN", "col1", "min"]], "INPUT"], [STR, DATETIME], [])
op_case("Group_CountOnEmptyInput", GT + ":477-488", cols([DATETIME]), [],
        ["GroupAggregate", EMPTYP, [["COUNT", "col0", "count"]], "INPUT"], [U64], [])
op_case("Group_CountOnEmptyInputWithGroupByColumn_string", GT + ":490-503", cols([STR, DATETIME]), [],
        ["GroupAggregate", ["ProjectNamedAttribute", "col0"], [["COUNT", "col1", "count"]], "INPUT"], [STR, U64], [])
op_case("Group_AggregationInputColumnMissingError", GT + ":505-513", cols([I32]), [],
        ["GroupAggregate", EMPTYP, [["SUM", "NotExistingCol", "sum"]], "INPUT"], None, [], expect_error=403)
op_case("Group_AggregationResultColumnExistsError", GT + ":515-525", cols([I32, I32]), [],
        ["GroupAggregate", EMPTYP, [["SUM", "col0", "result_col"], ["MIN", "col1", "result_col"]], "INPUT"], None, [], expect_error=404)
# BestEffortGroupAggregate (aggregate.h:230-250): "At 20 bytes quota, the buffer is filled after processing 3 rows" -- the block holds
# 20 / (INT32 + is_null + INT32 + is_null) = 2 groups, key 2 (row 3) finds no room: {1: 3 + 4, 3: -3} is returned, the aggregation starts
# anew at row 3.  The reference compares with SetIgnoreRowOrder(true)
op_case("Group_BestEffortGroupAggregate", GT + ":601-626", cols([I32, I32]), [[1, 3], [1, 4], [3, -3], [2, 4], [3, -5]],
        ["BestEffortGroupAggregate", ["ProjectNamedAttribute", "col0"], [["SUM", "col1", "sum"]], "INPUT",
         {"memory_quota": 20, "estimated_result_row_count": 2}],
        [I32, I32], [[1, 7], [3, -3], [2, 4], [3, -5]], ordered=False, exp_names=["col0", "sum"], exp_nullable=[True, True])
op_case("Group_BestEffortGroupAggregateWithoutGroupByColumns", GT + ":628-647", cols([I32]), [[1], [2], [3]],
        ["BestEffortGroupAggregate", EMPTYP, [["SUM", "col0", "sum"]], "INPUT", {"memory_quota": 100, "estimated_result_row_count": 2}],
        [I32], [[6]], exp_names=["sum"])
op_case("Group_NoGroupByColumns", GT + ":702-727", cols([I32]), [[1], [1], [3], [3], [2], [3], [1]],
        ["GroupAggregate", EMPTYP, [["SUM", "col0", "sum"], ["COUNT", "col0", "cnt"]], "INPUT"], [I32, U64], [[14, 7]])

# ---- ColumnAggregator unit tests (supersonic/cursor/core/column_aggregator_test.cc), restated as GroupAggregate:
# the test's result_index[] (which result row every input row updates) becomes the group key column "g"
CA = "supersonic/cursor/core/column_aggregator_test.cc"


def by_index(index_lists, value_lists):
    return [[g, v] for idx, vals in zip(index_lists, value_lists) for g, v in zip(idx, vals)]


IDX = [0, 1, 2, 3]
op_case("ColumnAggregator_ComputeSimpleAggregation", CA + ":60-81", [["g", I32, False], ["v", I64, False]],
        by_index([IDX, IDX], [[-5, 0, 4, 4], [-2, 3, 1, -1]]),
        ["GroupAggregate", ["ProjectNamedAttribute", "g"], [["MIN", "v", "r"]], "INPUT"], [I32, I64],
        [[0, -5], [1, 0], [2, 1], [3, -1]], ordered=False)
op_case("ColumnAggregator_SumInt32StoredAsInt64", CA + ":83-117", [["g", I32, False], ["v", I32, True]],
        by_index([IDX, IDX], [[-5, 0, 4, 4], [-2, 3, 1, -1]]),
        ["GroupAggregate", ["ProjectNamedAttribute", "g"], [["SUM", "v", "r", I64]], "INPUT"], [I32, I64],
        [[0, -7], [1, 3], [2, 5], [3, 3]], ordered=False)
op_case("ColumnAggregator_SumUint32StoredAsInt64", CA + ":119-145", [["g", I32, False], ["v", U32, True]],
        by_index([IDX, IDX], [[2, 3, 1, 4294967295], [5, 0, 4, 4]]),
        ["GroupAggregate", ["ProjectNamedAttribute", "g"], [["SUM", "v", "r", I64]], "INPUT"], [I32, I64],
        [[0, 7], [1, 3], [2, 5], [3, 4294967299]], ordered=False)
op_case("ColumnAggregator_ComputeAggregationOfValuesWithNulls", CA + ":147-186", [["g", I32, False], ["v", I32, True]],
        by_index([IDX, IDX], [[-2, None, 1, None], [None, None, 4, 4]]),
        ["GroupAggregate", ["ProjectNamedAttribute", "g"], [["SUM", "v", "r"]], "INPUT"], [I32, I32],
        [[0, -2], [1, None], [2, 5], [3, 4]], ordered=False)
op_case("ColumnAggregator_ResultIndexRespected", CA + ":216-234", [["g", I32, False], ["v", I32, True]],
        by_index([[2, 2, 2, 2]], [[1, 1, 1, 1]]),
        ["GroupAggregate", ["ProjectNamedAttribute", "g"], [["SUM", "v", "r"]], "INPUT"], [I32, I32], [[2, 4]], ordered=False)
op_case("ColumnAggregator_ComputeCount", CA + ":236-252", cols([I64]), [[-5], [0], [4], [4]],
        ["ScalarAggregate", [["COUNT", "col0", "r", I64]], "INPUT"], [I64], [[4]], exp_nullable=[False])
op_case("ColumnAggregator_ComputeCountWithoutInputColumn", CA + ":254-267", cols([I64]), [[-5], [0], [4], [4]],
        ["ScalarAggregate", [["COUNT", "", "r", I64]], "INPUT"], [I64], [[4]], exp_nullable=[False])
op_case("ColumnAggregator_ComputeCountOfValuesWithNulls", CA + ":269-290", cols([I32]), [[None], [0], [None], [4]],
        ["ScalarAggregate", [["COUNT", "col0", "r", I32]], "INPUT"], [I32], [[2]], exp_nullable=[False])
# CONCAT: the values of a group, printed, joined with ',' in input order (two UpdateAggregation calls = two input blocks)
op_case("ColumnAggregator_ComputeConcatOfInts", CA + ":365-388", [["g", I32, False], ["v", I32, True]],
        [[0, -5], [0, 0], [0, 345], [0, 2], [0, -2], [0, 3], [0, 1]],
        ["GroupAggregate", ["ProjectNamedAttribute", "g"], [["CONCAT", "v", "r"]], "INPUT"], [I32, STR], [[0, "-5,0,345,2,-2,3,1"]], ordered=False)
op_case("ColumnAggregator_ComputeConcatOfStrings_string", CA + ":390-412", [["g", I32, False], ["v", STR, True]],
        [[0, "baba"], [0, "baba"], [0, "dada"], [0, "aba"], [0, "wada"]],
        ["GroupAggregate", ["ProjectNamedAttribute", "g"], [["CONCAT", "v", "r"]], "INPUT"], [I32, STR], [[0, "baba,baba,dada,aba,wada"]], ordered=False)
# aggregation_operators_test.cc:247-272 (ConcatStrings / ConcatInts): the operator appends ',' + value to what it has; the
# first value of a group is assigned, not aggregated (column_aggregator.cc:108-124) -- together: "G,az,elle, is a kind of an antelope"
op_case("AggregationOperators_ConcatStrings_string", "supersonic/base/infrastructure/aggregation_operators_test.cc:247-260", cols([STR]),
        [["G"], ["az"], ["elle"], [" is a kind of an antelope"]],
        ["ScalarAggregate", [["CONCAT", "col0", "r"]], "INPUT"], [STR], [["G,az,elle, is a kind of an antelope"]])
op_case("AggregationOperators_ConcatInts", "supersonic/base/infrastructure/aggregation_operators_test.cc:262-272", cols([I32]),
        [[-7], [None], [0]],
        ["ScalarAggregate", [["CONCAT", "col0", "r"]], "INPUT"], [STR], [["-7,0"]])
# PrintTyped (types_infrastructure_test.cc:53-90, PrinterTest.ShouldPrintNormalValues): every TestPrinter<type>(value, text) of the types CONCAT
# accepts (column_aggregator.cc:496-515), reached through CONCAT, whose values PrintTyped prints; ',' between the values of one type
TI = "supersonic/base/infrastructure/types_infrastructure_test.cc"
for _t, _vals, _text, _lines in (
        (I32, [316, -5], "316,-5", ":56-57"), (U32, [2316, 4294967291], "2316,4294967291", ":59-60"),
        (I64, [334153124625418816, -334153124625418816], "334153124625418816,-334153124625418816", ":62-63"),
        (U64, [334153124625418816, 18112590949084132800], "334153124625418816,18112590949084132800", ":65-66"),
        (F32, [1.18, -1.18, 0.0, -1.244e24, -1.43e5], "1.18,-1.18,0,-1.244e+24,-143000", ":68-72"),
        (F64, [1.18, -1.18, 0.0, -1.244e24, -1.43e5], "1.18,-1.18,0,-1.244e+24,-143000", ":74-78"),
        (BOOL, [True, False], "TRUE,FALSE", ":80-81"),
        (DATETIME, [1260189023 * 1000000, -24 * 60 * 60 * 1000000, 24 * 60 * 60 * 1000000], "2009/12/07-12:30:23,1969/12/31-00:00:00,1970/01/02-00:00:00", ":83-85"),
        (DATE, [14585, -1, 1], "2009/12/07,1969/12/31,1970/01/02", ":87-89")):
    op_case("TypesInfrastructure_Print_%s_ThroughConcat" % _t, TI + _lines, cols([_t]), [[v] for v in _vals] + [[None]],
            ["ScalarAggregate", [["CONCAT", "col0", "r"]], "INPUT"], [STR], [[_text]])
op_case("ColumnAggregator_NotSupportedAggregationDetected_string", CA + ":518-526", cols([STR]), [],
        ["ScalarAggregate", [["SUM", "col0", "r"]], "INPUT"], None, [], expect_error=405)
op_case("ColumnAggregator_NotSupportedCountOutputTypeDetected", CA + ":528-534", cols([I32]), [],
        ["ScalarAggregate", [["COUNT", "col0", "r", DATETIME]], "INPUT"], None, [], expect_error=405)

# ---- AggregateClusters (supersonic/cursor/core/aggregate_clusters_test.cc:28-150) ---------------
CL = "supersonic/cursor/core/aggregate_clusters_test.cc"
cl_rows = [[0, 13], [2, 4], [2, 5], [2, -4], [2, -6], [1, 3], [1, 4], [1, -3]]
op_case("Clusters_AggregateClusters", CL + ":33-45,70-82", cols([I32, I32]), cl_rows,
        ["AggregateClusters", ["ProjectNamedAttribute", "col0"], [["SUM", "col1", "sum"]], "INPUT"],
        [I32, I32], [[0, 13], [2, -1], [1, 4]])
op_case("Clusters_WithoutClusteredColumn", CL + ":105-121", cols([I32]), [[13], [3], [7]],
        ["AggregateClusters", ["CompoundSingleSourceProjector"], [["SUM", "col0", "sum"]], "INPUT"], [I32], [[23]])
op_case("Clusters_EmptyInputWithClusteredColumn", CL + ":123-134", cols([I64, I32]), [],
        ["AggregateClusters", ["ProjectNamedAttribute", "col0"], [["SUM", "col1", "sum"]], "INPUT"], [I64, I32], [])

op_case("Clusters_EmptyInputWithoutClusteredColumn_string", CL + ":139-150", cols([STR]), [],
        ["AggregateClusters", ["CompoundSingleSourceProjector"], [["MAX", "col0", "max"]], "INPUT"], [STR], [])
op_case("Clusters_MultiColumnAggregateClusters_string", CL + ":152-183", cols([STR, I32, STR, I32]),
        [["a", 0, "a", 13], ["a", 2, "a", 4], ["a", 2, "a", 5], ["a", 2, "b", -4], ["a", 2, "b", -6], ["a", 1, "b", 3], ["a", 1, "b", 4],
         ["a", 1, "bbbbbbbb", -3]],
        ["AggregateClusters", ["Compound", ["ProjectNamedAttributeAs", "col0", "A"], ["ProjectNamedAttributeAs", "col1", "B"],
                               ["ProjectNamedAttributeAs", "col2", "C"]],
         [["SUM", "col1", "sum1"], ["SUM", "col3", "sum3"]], "INPUT"],
        [STR, I32, STR, I32, I32], [["a", 0, "a", 0, 13], ["a", 2, "a", 4, 9], ["a", 2, "b", 4, -10], ["a", 1, "b", 2, 7], ["a", 1, "bbbbbbbb", 1, -3]],
        exp_names=["A", "B", "C", "sum1", "sum3"])
op_case("Clusters_BadGroupBy", CL + ":185-199", cols([I32]), [[13], [3], [7]],
        ["AggregateClusters", ["ProjectNamedAttributeAs", "col1", "B"], [["MIN", "col0", "min"]], "INPUT"], None, [], expect_error=403)
op_case("Clusters_ResultingColumnsNamesConflict", CL + ":221-232", cols([I32, I32]), [],
        ["AggregateClusters", ["ProjectNamedAttributeAs", "col0", "A"], [["SUM", "col1", "A"]], "INPUT"], None, [], expect_error=404)

# ---- ScalarAggregate on strings (aggregate_scalar_test.cc:37-51,91-104) ----
op_case("ScalarAggregate_AggregateStrings_string", ST + ":33-45,91-104", cols([STR]),
        [["f"], ["c"], ["a"], ["b"], ["g"], ["a"], ["d"], ["a"], [None], ["e"]],
        ["ScalarAggregate", [["MAX", "col0", "max"], ["COUNT", "", "count(*)"], ["COUNT", "col0", "count"], ["COUNT_DISTINCT", "col0", "count distinct"]], "INPUT"],
        [STR, U64, U64, U64], [["g", 10, 9, 7]])

# ---- Project (supersonic/cursor/core/project_test.cc) ----------------------------------------------
PT = "supersonic/cursor/core/project_test.cc"
op_case("Project_FirstColumnFromInput_string", PT + ":34-45", cols([I32, STR]), [[1, "foo"], [3, "bar"]],
        ["Project", ["ProjectNamedAttribute", "col0"], "INPUT"], [I32], [[1], [3]])
op_case("Project_SecondColumnFromInput_string", PT + ":47-58", cols([I32, STR]), [[1, "foo"], [3, "bar"]],
        ["Project", ["ProjectNamedAttribute", "col1"], "INPUT"], [STR], [["foo"], ["bar"]])
op_case("Project_EmptyInput_string", PT + ":60-65", cols([I32, STR]), [],
        ["Project", ["ProjectNamedAttribute", "col1"], "INPUT"], [STR], [])
op_case("Project_InvalidProjectorSpecification_string", PT + ":79-88", cols([I32, STR]), [[1, "foo"], [3, "bar"]],
        ["Project", ["ProjectNamedAttribute", "incorrect_name"], "INPUT"], None, [], expect_error=403)

# ---- Sort (supersonic/cursor/core/sort_test.cc:121-330; letters -> INT64 codes a=1, b=2, ...) ---
SO = "supersonic/cursor/core/sort_test.cc"


def L(ch):
    return None if ch is None else ord(ch) - ord("a") + 1


def srows(pairs):
    return [[k, L(v)] for k, v in pairs]


op_case("Sort_OneIntegerColumnNoDuplicatesNoNulls", SO + ":121-145", cols([I32, I64]),
        srows([(2, "b"), (3, "c"), (1, "a"), (7, "g"), (4, "d"), (6, "f"), (5, "e")]),
        ["Sort", [["col0", "ASCENDING"]], None, "INPUT"], [I32, I64],
        srows([(1, "a"), (2, "b"), (3, "c"), (4, "d"), (5, "e"), (6, "f"), (7, "g")]))
op_case("Sort_OneIntegerColumnNoDuplicatesWithNulls", SO + ":76-89,147-156", cols([I32, I64]),
        srows([(None, "a"), (3, "c"), (None, "a"), (7, "g"), (4, "d"), (6, "f"), (5, "e")]),
        ["Sort", [["col0", "ASCENDING"]], None, "INPUT"], [I32, I64],
        srows([(None, "a"), (None, "a"), (3, "c"), (4, "d"), (5, "e"), (6, "f"), (7, "g")]))
op_case("Sort_OneColumnWithDuplicatesAndNulls", SO + ":189-215", cols([I64]),
        [[L("a")], [L("c")], [L("a")], [None], [L("d")], [None], [L("e")]],
        ["Sort", [["col0", "ASCENDING"]], None, "INPUT"], [I64],
        [[None], [None], [L("a")], [L("a")], [L("c")], [L("d")], [L("e")]])
op_case("Sort_OneIntegerColumnMostlyNullsDescending", SO + ":217-241", cols([I32, I64]),
        srows([(None, "a"), (None, "a"), (None, "a"), (7, "g"), (None, "a"), (None, "a"), (None, "a")]),
        ["Sort", [["col0", "DESCENDING"]], None, "INPUT"], [I32, I64],
        srows([(7, "g")] + [(None, "a")] * 6))
op_case("Sort_TwoColumnsFirstUnique", SO + ":243-269", cols([I32, I64]),
        srows([(2, "x"), (3, "v"), (1, "z"), (7, "x"), (4, "w"), (6, "x"), (5, "y")]),
        ["Sort", [["col0", "ASCENDING"], ["col1", "DESCENDING"]], None, "INPUT"], [I32, I64],
        srows([(1, "z"), (2, "x"), (3, "v"), (4, "w"), (5, "y"), (6, "x"), (7, "x")]))
op_case("Sort_TwoColumnsFirstConst", SO + ":271-297", cols([I32, I64]),
        srows([(1, "x"), (1, "v"), (1, "z"), (1, "x"), (1, "w"), (1, "x"), (1, "y")]),
        ["Sort", [["col0", "ASCENDING"], ["col1", "ASCENDING"]], None, "INPUT"], [I32, I64],
        srows([(1, "v"), (1, "w"), (1, "x"), (1, "x"), (1, "x"), (1, "y"), (1, "z")]))
op_case("Sort_TwoColumnsFirstMixed", SO + ":299-328", cols([I32, I64]),
        srows([(3, "z"), (None, "v"), (2, "z"), (3, None), (None, "w"), (3, "x"), (1, "x"), (None, "y")]),
        ["Sort", [["col0", "ASCENDING"], ["col1", "ASCENDING"]], None, "INPUT"], [I32, I64],
        srows([(None, "v"), (None, "w"), (None, "y"), (1, "x"), (2, "z"), (3, None), (3, "x"), (3, "z")]))

mixed = [(3, "z"), (None, "v"), (2, "z"), (3, None), (None, "w"), (3, "x"), (1, "x"), (None, "y")]
op_case("Sort_TwoColumnsFirstMixedDescending", SO + ":331-360", cols([I32, I64]), srows(mixed),
        ["Sort", [["col0", "DESCENDING"], ["col1", "DESCENDING"]], None, "INPUT"], [I32, I64],
        srows([(3, "z"), (3, "x"), (3, None), (2, "z"), (1, "x"), (None, "y"), (None, "w"), (None, "v")]))
op_case("Sort_TwoColumnsFirstMixed_string", SO + ":299-328", cols([I32, STR]), [list(r) for r in mixed],
        ["Sort", [["col0", "ASCENDING"], ["col1", "ASCENDING"]], None, "INPUT"], [I32, STR],
        [[None, "v"], [None, "w"], [None, "y"], [1, "x"], [2, "z"], [3, None], [3, "x"], [3, "z"]])
op_case("Sort_TwoColumnsFirstMixedDescending_string", SO + ":331-360", cols([I32, STR]), [list(r) for r in mixed],
        ["Sort", [["col0", "DESCENDING"], ["col1", "DESCENDING"]], None, "INPUT"], [I32, STR],
        [[3, "z"], [3, "x"], [3, None], [2, "z"], [1, "x"], [None, "y"], [None, "w"], [None, "v"]])
# the sort key (position 2, DESCENDING) is not part of the result projection (position 1)
op_case("Sort_Projections", SO + ":362-381", cols([I32, I32, I32]), [[3, 105, 210], [6, 111, 201], [2, 102, 203], [3, 104, 205]],
        ["Sort", [["col2", "DESCENDING"]], ["ProjectAttributeAt", 1], "INPUT"], [I32], [[105], [104], [102], [111]])
op_case("Filter_FilterOnDropped_string", "supersonic/cursor/core/filter_test.cc:316-329", cols([I32, STR]), [[1, "A"], [3, "B"]],
        ["Filter", ["Equal", ["NamedAttribute", "col1"], ["ConstString", "A"]], ["ProjectNamedAttribute", "col0"], "INPUT"], [I32], [[1]])

def bind_plan_case(name, source, expr, in_types, in_nullable, out_name, out_type, out_nullable, expect_error=None):
    n_in = len(in_types)
    CASES.append({
        "name": name, "source": source, "kind": "binding",
        "input": {"schema": [["$%d" % i, in_types[i], in_nullable[i]] for i in range(n_in)], "rows": []},
        "plan": ["Compute", expr, "INPUT"],
        "expected": {"types": [out_type] if out_type else None, "rows": [], "names": [out_name] if out_name else None,
                     "nullable": [out_nullable] if out_nullable is not None else None},
        "ordered": True, "expect_error": expect_error})


# ---- IN (comparison_bound_expressions_test.cc: binding names; the reference has no evaluation
# ---- table for IN, its SQL NULL rule is restated from comparison_expressions.h:75-84) -----------
CB = "supersonic/expression/core/comparison_bound_expressions_test.cc"
IN4 = ["InList", ["AttributeAt", 0], ["AttributeAt", 1], ["AttributeAt", 2], ["AttributeAt", 3]]
bind_plan_case("InSet_int32", CB + ":120-121", IN4, [I32, I32, I32, I32], [False] * 4, "$0 IN ($1, $2, $3)", BOOL, False)
bind_plan_case("InSet_empty", CB + ":122", ["InList", ["AttributeAt", 0]], [I64], [False], "$0 IN ()", BOOL, False)
bind_plan_case("InSet_multi_1", CB + ":126-127", IN4, [I32, I64, I64, I64], [False] * 4, "CAST_INT32_TO_INT64($0) IN ($1, $2, $3)", BOOL, False)
bind_plan_case("InSet_multi_2", CB + ":128-129", IN4, [I64, I32, I64, I64], [False] * 4, "$0 IN (CAST_INT32_TO_INT64($1), $2, $3)", BOOL, False)
bind_plan_case("InSet_multi_3", CB + ":130-132", IN4, [F64, I32, U32, U64], [False] * 4,
               "$0 IN (CAST_INT32_TO_DOUBLE($1), CAST_UINT32_TO_DOUBLE($2), CAST_UINT64_TO_DOUBLE($3))", BOOL, False)
bind_plan_case("InSet_multi_4", CB + ":133-135", IN4, [I32, I32, F64, U64], [False] * 4,
               "CAST_INT32_TO_DOUBLE($0) IN (CAST_INT32_TO_DOUBLE($1), $2, CAST_UINT64_TO_DOUBLE($3))", BOOL, False)

# ---- CASE (case_expression_test.cc) ------------------------------------------------------------
# STRING THEN/OTHERWISE payloads are substituted by INT64 codes ("A" -> 65 ...), see the header.
C = "supersonic/expression/core/case_expression_test.cc"
AT = lambda i: ["AttributeAt", i]  # noqa: E731
expr_plan_case("Case_BasicInt32", C + ":33-50",
               [I32], ["CaseList", AT(0), ["ConstInt64", 0], ["ConstInt32", 1], ["ConstInt64", 1], ["ConstInt32", 2], ["ConstInt64", 2]], I64,
               [[1, 1], [2, 2], [3, 0], [4, 0], [5, 0], [None, 0]])
expr_plan_case("Case_NullThen", C + ":52-70 (selector INT32 instead of STRING)",
               [I32, U32], ["CaseList", AT(0), AT(1), ["ConstInt32", 1], ["ConstUint32", 1], ["ConstInt32", 5], ["NullOf", "UINT32"]], U32,
               [[0, 10, 10], [9, 9, 9], [1, 8, 1], [4, 7, 7], [5, 6, None], [None, None, None]])
expr_plan_case("Case_ForceCast", C + ":72-93",
               [I32, U32], ["CaseList", AT(0), AT(1), ["ConstInt64", 3], ["ConstUint64", 4], ["ConstFloat", 5.0], ["ConstDouble", 10.0]], F64,
               [[1, 10, 10.0], [2, 9, 9.0], [3, 8, 4.0], [4, 7, 7.0], [5, 6, 10.0], [None, None, None]])
IFCASE = ["CaseList", AT(0), AT(2), ["ConstBool", True], AT(1)]
expr_plan_case("Case_IfCaseWithNullCondition", C + ":112-118", [BOOL, I64, I64], IFCASE, I64,
               [[False, 65, 66, 66], [None, 67, 68, 68], [True, 69, 70, 69]])
expr_plan_case("Case_IfCaseAllNullable", C + ":178-195", [BOOL, I64, I64], IFCASE, I64,
               [[True, 65, 66, 65], [False, 67, 68, 68], [True, None, 70, None], [False, None, 72, 72], [True, 73, None, 73],
                [False, 75, None, None], [True, None, None, None], [False, None, None, None], [None, 77, 78, 78], [None, None, 80, 80],
                [None, 82, None, None], [None, None, None, None], [False, 87, 89, 89]])

# ---- exact math family (math_expressions_test.cc) -------------------------------------------------
M = "supersonic/expression/core/math_expressions_test.cc"
bind_case("Math_Ceil_double", M + ":34", "Ceil", [F64], [False], "CEIL($0)", F64, False)
bind_case("Math_Ceil_uint32", M + ":35", "Ceil", [U32], [False], "$0", U32, False)
bind_case("Math_CeilToInt_double", M + ":36", "CeilToInt", [F64], [False], "CEIL_TO_INT($0)", I64, False)
bind_case("Math_CeilToInt_uint64", M + ":37", "CeilToInt", [U64], [False], "$0", U64, False)
bind_case("Math_Floor_float", M + ":40", "Floor", [F32], [False], "FLOOR($0)", F32, False)
bind_case("Math_FloorToInt_float", M + ":41", "FloorToInt", [F32], [False], "FLOOR_TO_INT($0)", I64, False)
bind_case("Math_Round_float", M + ":50", "Round", [F32], [False], "ROUND($0)", F32, False)
bind_case("Math_Round_int32", M + ":51", "Round", [I32], [False], "$0", I32, False)
bind_case("Math_RoundToInt_double", M + ":53", "RoundToInt", [F64], [False], "CEIL_TO_INT(ROUND($0))", I64, False)
bind_case("Math_SqrtQuiet", M + ":55", "SqrtQuiet", [F64], [False], "SQRT($0)", F64, False)
bind_case("Math_SqrtNulling", M + ":56", "SqrtNulling", [F64], [False], "SQRT($0)", F64, True)
bind_case("Math_SqrtSignaling", M + ":57", "SqrtSignaling", [F64], [False], "SQRT($0)", F64, False)
bind_case("Math_Trunc", M + ":58", "Trunc", [F64], [False], "TRUNC($0)", F64, False)
bind_case("Math_Abs_int32", M + ":137", "Abs", [I32], [False], "ABS($0)", U32, False)
bind_case("Math_Abs_uint32", M + ":138", "Abs", [U32], [False], "$0", U32, False)
bind_case("Math_Abs_int64", M + ":139", "Abs", [I64], [False], "ABS($0)", U64, False)
bind_case("Math_Abs_float", M + ":141", "Abs", [F32], [False], "ABS($0)", F32, False)
expr_case("Math_Round", M + ":147-156", [F64, F64], [[4., 4.], [0.5, 1.], [-0.5, -1.], [0.49, 0.], [-0.49, -0.], [2345., 2345.]], "Round")
expr_case("Math_RoundFloat", M + ":158-167", [F32, F32], [[4., 4.], [0.5, 1.], [-0.5, -1.], [0.49, 0.], [-0.49, -0.], [2345., 2345.]], "Round")
expr_case("Math_RoundToIntOnDouble", M + ":169-180", [F64, I64],
          [[4., 4], [0.5, 1], [-0.5, -1], [0.49, 0], [None, None], [-0.49, 0], [2345., 2345], [-3.65309740835E17, -365309740835000000]], "RoundToInt")
expr_case("Math_RoundToIntOnFloat", M + ":182-193", [F32, I64],
          [[4., 4], [0.5, 1], [-0.5, -1], [0.49, 0], [None, None], [-0.49, 0], [2345., 2345], [-3.65309E5, -365309]], "RoundToInt")
expr_case("Math_Ceil", M + ":211-219", [F64, F64], [[3., 3.], [-3., -3.], [-2.9, -2.], [1.9, 2.], [-2.1, -2.], [1.001, 2.]], "Ceil")
expr_case("Math_Ceil_float", M + ":221-224", [F32, F32], [[0.1, 1.], [-0.9, -0.]], "Ceil")
expr_case("Math_Ceil_int32", M + ":226-229", [I32, I32], [[7, 7], [-1, -1]], "Ceil")
expr_case("Math_CeilToInt", M + ":232-240", [F64, I64], [[3., 3], [-3., -3], [-2.9, -2], [1.9, 2], [-2.1, -2], [1.001, 2]], "CeilToInt")
expr_case("Math_CeilToInt_float", M + ":242-245", [F32, I64], [[0.1, 1], [-0.9, 0]], "CeilToInt")
expr_case("Math_Floor", M + ":253-261", [F64, F64], [[3., 3.], [-3., -3.], [-2.9, -3.], [1.9, 1.], [-2.1, -3.], [1.001, 1.]], "Floor")
expr_case("Math_Floor_float", M + ":263-266", [F32, F32], [[0.1, 0.], [-0.9, -1.]], "Floor")
expr_case("Math_FloorToInt", M + ":274-282", [F64, I64], [[3., 3], [-3., -3], [-2.9, -3], [1.9, 1], [-2.1, -3], [1.001, 1]], "FloorToInt")
expr_case("Math_FloorToInt_float", M + ":284-287", [F32, I64], [[0.1, 0], [-0.9, -1]], "FloorToInt")
expr_case("Math_Trunc", M + ":295-303", [F64, F64], [[3., 3.], [-3., -3.], [-2.9, -2.], [1.9, 1.], [-2.1, -2.], [1.001, 1.]], "Trunc")
expr_case("Math_Trunc_float", M + ":305-308", [F32, F32], [[0.1, 0.], [None, None]], "Trunc")
expr_case("Math_Trunc_uint32", M + ":310-313", [U32, U32], [[7, 7], [0, 0]], "Trunc")
expr_case("Math_Abs", M + ":316-324", [I32, U32], [[0, 0], [1, 1], [-3, 3], [-2147483648, 2147483648]], "Abs")
expr_case("Math_SqrtNulling", M + ":326-334", [F64, F64], [[0.25, 0.5], [1., 1.], [40000., 200.], [0., 0.], [-1, None]], "SqrtNulling")
expr_case("Math_SqrtSignaling", M + ":336-343", [F64, F64], [[0.25, 0.5], [1., 1.], [None, None], [0., 0.]], "SqrtSignaling")
expr_case("Math_SqrtSignaling_failure", M + ":344-348", [F64, F64], [[-1., None], [-123321., None]], "SqrtSignaling", expect_error=104)
expr_case("Math_SqrtQuiet", M + ":350-360", [F64, F64], [[0.25, 0.5], [1., 1.], [-123., NAN], [40000., 200.], [0., 0.], [-1, NAN]], "SqrtQuiet")
expr_case("Math_IsFinite", M + ":712-722", [F32, BOOL], [[0., True], [1234., True], [INF, False], [NAN, False], [None, None]], "IsFinite")
expr_case("Math_IsInf", M + ":724-734", [F64, BOOL], [[0., False], [1234., False], [INF, True], [NAN, False], [None, None]], "IsInf")
expr_case("Math_IsNaN", M + ":736-746", [F32, BOOL], [[0., False], [1234., False], [INF, False], [NAN, True], [None, None]], "IsNaN")
expr_case("Math_IsNormal", M + ":748-758", [F64, BOOL], [[0., False], [1234., True], [INF, False], [NAN, False], [None, None]], "IsNormal")

# ---- real STRING columns (dictionary codes on the device, see include/ssgpu.h) ---------------------
expr_plan_case("Case_BasicInt32ToString", C + ":33-50",
               [I32], ["CaseList", AT(0), ["ConstString", "other"], ["ConstInt32", 1], ["ConstString", "one"], ["ConstInt32", 2], ["ConstString", "two"]], STR,
               [[1, "one"], [2, "two"], [3, "other"], [4, "other"], [5, "other"], [None, "other"]])
expr_plan_case("Case_StringToUInt32", C + ":52-70",
               [STR, U32], ["CaseList", AT(0), AT(1), ["ConstString", "one"], ["ConstUint32", 1], ["ConstString", "null"], ["NullOf", "UINT32"]], U32,
               [["zero", 10, 10], ["nine", 9, 9], ["one", 8, 1], ["NULL", 7, 7], ["null", 6, None], [None, None, None]])
expr_plan_case("Case_IfCaseAllNullable_strings", C + ":178-195", [BOOL, STR, STR], IFCASE, STR,
               [[True, "A", "B", "A"], [False, "C", "D", "D"], [True, None, "F", None], [False, None, "H", "H"], [True, "I", None, "I"],
                [False, "K", None, None], [True, None, None, None], [False, None, None, None], [None, "M", "N", "N"], [None, None, "P", "P"],
                [None, "R", None, None], [None, None, None, None], [False, "W", "Y", "Y"]])
bind_plan_case("InSet_string", CB + ":118-119", IN4, [STR, STR, STR, STR], [False] * 4, "$0 IN ($1, $2, $3)", BOOL, False)
bind_plan_case("InSet_string_int_mismatch", CB + ":136", ["InList", ["AttributeAt", 0], ["AttributeAt", 1], ["AttributeAt", 2]],
               [I32, STR, I32], [False] * 3, None, None, None, expect_error=402)

# ---- operation tests with their original STRING columns ------------------------------------------------
op_case("Group_GroupBySecondColumn_string", GT + ":356-377", cols([I32, STR]), [[-3, "foo"], [2, "bar"], [3, "bar"], [-2, "foo"]],
        ["GroupAggregate", ["ProjectNamedAttribute", "col1"], [["SUM", "col0", "sum"]], "INPUT"],
        [STR, I32], [["foo", -5], ["bar", 5]], ordered=False, exp_names=["col1", "sum"])
op_case("Group_GroupByWithoutAggregateFunctions_string", GT + ":430-447", cols([STR]), [["foo"], ["bar"], ["foo"], ["bar"]],
        ["GroupAggregate", ["ProjectNamedAttribute", "col0"], [], "INPUT"], [STR], [["foo"], ["bar"]], ordered=False)
op_case("Sort_OneIntegerColumnNoDuplicatesNoNulls_string", SO + ":121-145", cols([I32, STR]),
        [[2, "b"], [3, "c"], [1, "a"], [7, "g"], [4, "d"], [6, "f"], [5, "e"]],
        ["Sort", [["col0", "ASCENDING"]], None, "INPUT"], [I32, STR],
        [[1, "a"], [2, "b"], [3, "c"], [4, "d"], [5, "e"], [6, "f"], [7, "g"]])
op_case("Sort_OneStringColumnWithDuplicatesAndNulls_string", SO + ":190-215", cols([STR]),
        [["a"], ["c"], ["a"], [None], ["d"], [None], ["e"]],
        ["Sort", [["col0", "ASCENDING"]], None, "INPUT"], [STR], [[None], [None], ["a"], ["a"], ["c"], ["d"], ["e"]])
op_case("Sort_OneEmptyStringColumn_string", SO + ":180-188", cols([STR]), [], ["Sort", [["col0", "ASCENDING"]], None, "INPUT"], [STR], [])

# ---- HashJoin (hash_join_test.cc) with the tests' own STRING payloads ---------------------------------
HJ = "supersonic/cursor/core/hash_join_test.cc"
ALLP = [[0, ["ProjectAllAttributes", "L."]], [1, ["ProjectAllAttributes", "R."]]]
JT = [I64, STR, I64, STR]
JN = ["L.col0", "L.col1", "R.col0", "R.col1"]
R1, R2 = [[1, "a"]], [[2, "b"]]
R12345 = [[1, "a"], [2, "b"], [3, "c"], [4, "d"], [5, "e"]]
R654321 = [[6, "f"], [5, "e"], [4, "d"], [3, "c"], [2, "b"], [1, "a"]]


def join_case(name, src, jtype, lrows, rrows, expected, uniqueness="UNIQUE", keys=(0,)):
    CASES.append({
        "name": name, "source": src, "kind": "operation",
        "input": {"schema": cols([I64, STR]), "rows": lrows}, "input2": {"schema": cols([I64, STR]), "rows": rrows},
        "plan": ["HashJoin", jtype, list(keys), list(keys), ALLP, uniqueness, "INPUT", "INPUT2"],
        "expected": {"types": JT, "rows": expected, "names": JN, "nullable": None},
        "ordered": True, "expect_error": None})


join_case("HashJoin_1_InnerJoin_1", HJ + ":140-150", "INNER", R1, R1, [[1, "a", 1, "a"]])
join_case("HashJoin_1_LeftOuterJoin_1", HJ + ":152-163", "LEFT_OUTER", R1, R1, [[1, "a", 1, "a"]])
join_case("HashJoin_1_InnerJoin_2", HJ + ":165-174", "INNER", R1, R2, [])
join_case("HashJoin_1_LeftOuterJoin_2", HJ + ":176-187", "LEFT_OUTER", R1, R2, [[1, "a", None, None]])
join_case("HashJoin_12345_InnerJoin_654321", HJ + ":189-203", "INNER", R12345, R654321, [[k, v, k, v] for k, v in R12345])
join_case("HashJoin_654321_InnerJoin_12345", HJ + ":205-219", "INNER", R654321, R12345, [[k, v, k, v] for k, v in R654321[1:]])
join_case("HashJoin_654321_LeftOuterJoin_12345", HJ + ":222-238", "LEFT_OUTER", R654321, R12345,
          [[6, "f", None, None]] + [[k, v, k, v] for k, v in R654321[1:]])

# the same TEST_P cases with rhs_key_uniqueness() == NOT_UNIQUE (hash_join_test.cc:136-138), then the tests
# whose rhs keys really repeat: matches of one lhs row come out in rhs order
NU = "NOT_UNIQUE"
join_case("HashJoin_1_InnerJoin_1_not_unique", HJ + ":140-150", "INNER", R1, R1, [[1, "a", 1, "a"]], NU)
join_case("HashJoin_1_LeftOuterJoin_2_not_unique", HJ + ":176-187", "LEFT_OUTER", R1, R2, [[1, "a", None, None]], NU)
join_case("HashJoin_12345_InnerJoin_654321_not_unique", HJ + ":189-203", "INNER", R12345, R654321, [[k, v, k, v] for k, v in R12345], NU)

# two-column (INT64, STRING) keys -- 96 packed bits -- in both uniqueness modes (TEST_P); a NULL in any key column matches nothing
R1a1b2a2b = [[1, "a"], [1, "b"], [2, "a"], [2, "b"]]
R1a1NNaNN = [[1, "a"], [1, None], [None, "a"], [None, None]]
for U in ("UNIQUE", NU):
    sfx = "" if U == "UNIQUE" else "_not_unique"
    join_case("HashJoin_1a1b2a2b_InnerJoin_1a1b2a2b" + sfx, HJ + ":305-319", "INNER", R1a1b2a2b, R1a1b2a2b, [[k, v, k, v] for k, v in R1a1b2a2b], U, (0, 1))
    join_case("HashJoin_1a1NNaNN_InnerJoin_1a1NNaNN" + sfx, HJ + ":355-366", "INNER", R1a1NNaNN, R1a1NNaNN, [[1, "a", 1, "a"]], U, (0, 1))
    join_case("HashJoin_1a1NNaNN_LeftOuterJoin_1a1NNaNN" + sfx, HJ + ":368-382", "LEFT_OUTER", R1a1NNaNN, R1a1NNaNN,
              [[1, "a", 1, "a"], [1, None, None, None], [None, "a", None, None], [None, None, None, None]], U, (0, 1))

# ---- short circuit: which rows every child is evaluated on (skip vectors) ------------------------------------------
# The reference's short-circuit tests (supersonic/testing/short_circuit_tester.h:37-62) give, per input row, the
# skip vector each child of an operator must RECEIVE: a skipped child is not evaluated there, so a failing child
# raises no evaluation error on that row.  Restated as data: for every row whose incoming skip flag is false and
# every child k, the case wraps child k in an expression that fails (0 % 0, ModulusSignaling) on every row it is
# evaluated on, and expects error 104 exactly when the table says the child is NOT skipped, else the table's result.
# A BOOL child c becomes Or(Equal(0 % z, 1), c), an integer child Plus(0 % z, c): the failing part comes first, so it
# sees the skip vector the child itself was handed (z = extra NOT NULL INT32 column of zeros, the last input).
def short_circuit_cases(name, source, factory, child_types, out_type, table):
    """table rows: [skip_in, child0, skip0, child1, skip1, ..., result] as in the reference's BlockBuilder."""
    n = len(child_types)
    for r, row in enumerate(table):
        if row[0]:
            continue            # rows the parent itself skips cannot be driven from outside an expression
        vals = [row[1 + 2 * k] for k in range(n)]
        skips = [row[2 + 2 * k] for k in range(n)]
        for k in range(n):
            zcol = ["AttributeAt", n]
            boom = ["ModulusSignaling", ["ConstInt32", 0], zcol]
            args = [["AttributeAt", i] for i in range(n)]
            args[k] = (["Or", ["Equal", boom, ["ConstInt32", 1]], args[k]] if child_types[k] == BOOL else ["Plus", boom, args[k]])
            CASES.append({
                "name": "%s_row%d_child%d" % (name, r, k), "source": source, "kind": "expression",
                "input": {"schema": [["col%d" % i, child_types[i], True] for i in range(n)] + [["z", I32, False]], "rows": [vals + [0]]},
                "plan": ["Compute", [factory] + args, "INPUT"],
                "expected": {"types": [out_type], "rows": [[row[-1]]], "names": None, "nullable": None},
                "ordered": True, "expect_error": None if skips[k] else 104})


T, F_ = True, False
short_circuit_cases("ShortCircuit_And", E + ":444-461", "And", [BOOL, BOOL], BOOL, [
    [F_, F_, F_, F_, T, F_], [F_, F_, F_, None, T, F_], [F_, T, F_, F_, F_, F_], [F_, T, F_, None, F_, None], [F_, None, F_, T, F_, None],
    [F_, None, F_, F_, F_, F_], [F_, T, F_, T, F_, T], [T, F_, T, F_, T, None], [T, None, T, None, T, None], [T, T, T, T, T, None]])
short_circuit_cases("ShortCircuit_Or", E + ":476-494", "Or", [BOOL, BOOL], BOOL, [
    [F_, F_, F_, F_, F_, F_], [F_, F_, F_, None, F_, None], [F_, T, F_, F_, T, T], [F_, T, F_, None, T, T], [F_, None, F_, T, F_, T],
    [F_, None, F_, None, F_, None], [F_, None, F_, F_, F_, None], [T, F_, T, F_, T, None], [T, F_, T, None, T, None], [T, None, T, F_, T, None],
    [T, None, T, None, T, None]])
short_circuit_cases("ShortCircuit_AndNot", E + ":518-536", "AndNot", [BOOL, BOOL], BOOL, [
    [F_, F_, F_, T, F_, T], [F_, F_, F_, None, F_, None], [F_, T, F_, F_, T, F_], [F_, T, F_, None, T, F_], [F_, None, F_, F_, F_, F_],
    [F_, None, F_, None, F_, None], [F_, None, F_, T, F_, None], [T, F_, T, F_, T, None], [T, F_, T, None, T, None], [T, None, T, F_, T, None],
    [T, None, T, None, T, None]])
short_circuit_cases("ShortCircuit_If", E + ":718-735", "If", [BOOL, I32, I32], I32, [
    [F_, T, F_, 1, F_, 2, T, 1], [F_, T, F_, None, F_, 2, T, None], [F_, T, F_, 1, F_, None, T, 1], [F_, F_, F_, None, T, 2, F_, 2],
    [F_, F_, F_, 1, T, None, F_, None], [F_, None, F_, 1, T, 2, F_, 2], [T, T, T, 1, T, 2, T, None], [T, F_, T, 1, T, 2, T, None],
    [T, None, T, 1, T, 2, T, None]])
short_circuit_cases("ShortCircuit_NullingIf", E + ":737-754", "NullingIf", [BOOL, I32, I32], I32, [
    [F_, T, F_, 1, F_, 2, T, 1], [F_, T, F_, None, F_, 2, T, None], [F_, T, F_, 1, F_, None, T, 1], [F_, F_, F_, None, T, 2, F_, 2],
    [F_, F_, F_, 1, T, None, F_, None], [F_, None, F_, 1, T, 2, T, None], [T, T, T, 1, T, 2, T, None], [T, F_, T, 1, T, 2, T, None],
    [T, None, T, 1, T, 2, T, None]])
# IfNull: the reference's table holds STRINGs ("One Ring To Rule Them All"); INT32 codes here, same NULL pattern
short_circuit_cases("ShortCircuit_IfNull", E + ":377-394", "IfNull", [I32, I32], I32, [
    [F_, 1, F_, 2, T, 1], [F_, 3, F_, 4, T, 3], [F_, 5, F_, None, T, 5], [F_, None, F_, 6, F_, 6], [F_, None, F_, None, F_, None],
    [T, 1, T, 2, T, None], [T, 3, T, None, T, None], [T, None, T, 7, T, None], [T, None, T, None, T, None], [F_, None, F_, 5, F_, 5]])

# CaseShortCircuit (case_expression_test.cc:195-253): A + CASE B WHEN C THEN 0 % D ELSE 0 % E -- which of the two
# failing branches is evaluated, and that a NULL A skips the whole CASE
CASE_SC = ["Plus", AT(0), ["CaseList", AT(1), ["ModulusSignaling", ["ConstInt32", 0], AT(4)], AT(2), ["ModulusSignaling", ["ConstInt32", 0], AT(3)]]]
expr_plan_case("Case_ShortCircuit", C + ":229-245", [I32, I32, I32, I32, I32], CASE_SC, I32,
               [[None, 1, 1, 0, 0, None], [None, 1, 2, 0, 0, None], [None, None, 2, 0, 0, None], [None, 1, None, 0, 0, None], [None, None, None, 0, 0, None],
                [0, 1, 1, 1, 0, 0], [0, 1, 2, 0, 1, 0], [0, None, 1, 0, 1, 0], [0, None, None, 0, 1, 0], [0, 1, None, 0, 1, 0],
                [0, 1, 1, None, 0, None], [0, 1, 2, 0, None, None], [0, None, 1, 0, None, None]])
for _i, _row in enumerate([[0, 1, 1, 0, None], [0, 1, 2, None, 0], [0, None, None, None, 0]]):
    expr_plan_case("Case_ShortCircuit_fails_%d" % _i, C + ":247-252", [I32, I32, I32, I32, I32], CASE_SC, I32, [_row + [None]])
    CASES[-1]["expect_error"] = 104

join_case("HashJoin_654321_LeftOuterJoin_12345_not_unique", HJ + ":222-238", "LEFT_OUTER", R654321, R12345,
          [[6, "f", None, None]] + [[k, v, k, v] for k, v in R654321[1:]], NU)
R2b2b2c = [[2, "b"], [2, "b"], [2, "c"]]
join_case("HashJoin_12345_InnerJoin_2b2b2c", HJ + ":240-252", "INNER", R12345, R2b2b2c,
          [[2, "b", 2, "b"], [2, "b", 2, "b"], [2, "b", 2, "c"]], NU)
join_case("HashJoin_12345_LeftOuterJoin_2b2b2c", HJ + ":254-271", "LEFT_OUTER", R12345, R2b2b2c,
          [[1, "a", None, None], [2, "b", 2, "b"], [2, "b", 2, "b"], [2, "b", 2, "c"], [3, "c", None, None], [4, "d", None, None], [5, "e", None, None]], NU)
join_case("HashJoin_2b2b2c_InnerJoin_2b2b2c", HJ + ":112-121,273-281", "INNER", R2b2b2c, R2b2b2c,
          [[2, "b", 2, "b"], [2, "b", 2, "b"], [2, "b", 2, "c"], [2, "b", 2, "b"], [2, "b", 2, "b"], [2, "b", 2, "c"],
           [2, "c", 2, "b"], [2, "c", 2, "b"], [2, "c", 2, "c"]], NU)
# every lhs row matches 1100 rhs rows: more than one 1024-row result view per lhs row
R5k = [[k, v] for _ in range(1100) for k, v in R12345]
join_case("HashJoin_12345_InnerJoin_5k_Rows", HJ + ":321-353", "INNER", R12345, R5k,
          [[k, v, k, v] for k, v in R12345 for _ in range(1100)], NU)


# ---- the guide's GroupAggregate (test/guide/primer.cc:230-346: GroupedSums over key INT32, data DOUBLE) -----------
op_case("Primer_GroupAggregateTest", "test/guide/primer.cc:294-346", [["key", I32, False], ["data", F64, False]],
        [[k, d] for k, d in zip([1, 2, 3, 1, 2, 3, 1, 2], [1.5, 3.0, 3.0, 7.6, 5.5, 2.0, 1.6, 9.5])],
        ["GroupAggregate", ["ProjectNamedAttribute", "key"], [["SUM", "data", "data_sums"]], "INPUT"],
        [I32, F64], [[1, 1.5 + 7.6 + 1.6], [2, 3.0 + 5.5 + 9.5], [3, 3.0 + 2.0]], ordered=False, exp_names=["key", "data_sums"])

# ---- test/guide/group_sort.cc: two-key grouping (STRING + BOOL keys, MIN / MAX), single-column sort --------------
GS = "test/guide/group_sort.cc"
EMP_SCHEMA = [["name", STR, False], ["age", I32, False], ["salary", I32, False], ["department", STR, False], ["full_time", BOOL, False]]
EMP_PLAN = ["GroupAggregate", ["Compound", ["ProjectNamedAttributeAs", "full_time", "Works full time?"], ["ProjectAttributeAt", 3]],
            [["MIN", "salary", "min_salary"], ["MAX", "age", "max_age"]], "INPUT"]


def emp_expected(rows):
    """TestResults() of the fixture (group_sort.cc:186-241): min salary / max age per (full_time, department)."""
    acc = {}
    for _name, age, sal, dept, ft in rows:
        k = (ft, dept)
        acc[k] = (min(acc[k][0], sal), max(acc[k][1], age)) if k in acc else (sal, age)
    return [[k[0], k[1], v[0], v[1]] for k, v in acc.items()]


_small = [[n, a, s, d, f] for n, a, s, d, f in zip(["John", "Darrel", "Greg", "Amanda", "Stacy"], [20, 25, 32, 31, 33], [1800, 3300, 4800, 3500, 1900],
                                                   ["Accounting", "Sales", "Sales", "IT", "IT"], [False, True, False, True, False])]
op_case("GroupSort_SmallGroupingTest", GS + ":251-283", EMP_SCHEMA, _small, EMP_PLAN, [BOOL, STR, I32, I32], emp_expected(_small),
        ordered=False, exp_names=["Works full time?", "department", "min_salary", "max_age"])
# LargeRandomGroupingTest draws from rand(); the same pools and value ranges from a fixed generator here
_NAMES = ["John", "James", "Alan", "Judy", "Anne", "Ray", "Grace"]
_DEPTS = ["IT", "Sales", "Legal", "Services", "Advertising", "Research", "Operations", "Compliance", "Public Relations", "Human Resources",
          "Research", "Engineering", "Deployment", "Accounting", "Tech Support"]


def _lcg(seed):
    state = [seed]

    def nxt():
        state[0] = (state[0] * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        return state[0] >> 33
    return nxt


_r = _lcg(2012)
_large = [[_NAMES[_r() % 7], 0, 0, _DEPTS[_r() % 15], False] for _ in range(2000)]
for _row in _large:
    _row[1] = _r() % 60 + 20
    _row[2] = (_r() % 1900) * 10 + 1000
    _row[4] = bool(_r() % 2)
op_case("GroupSort_LargeRandomGroupingTest", GS + ":284-350", EMP_SCHEMA, _large, EMP_PLAN, [BOOL, STR, I32, I32], emp_expected(_large), ordered=False)

GRADES = [[i, g] for i, g in zip(range(1, 9), [4.5, 4.2, 3.5, 4.8, 4.2, 3.9, 3.2, 4.8])]
op_case("GroupSort_SmallSortingTest", GS + ":532-541", [["id", I32, False], ["grade", F64, False]], GRADES,
        ["Sort", [["grade", "ASCENDING"]], None, "INPUT"], [I32, F64], sorted(GRADES, key=lambda r: r[1]), ordered=False)
CASES[-1]["sorted_on"] = [1]     # the fixture checks order on the key and the multiset of rows (ties are unordered: sort.h:42)
_r = _lcg(77)
_grades = [[i + 1, 4.0 * (_r() % (1 << 31)) / float((1 << 31) - 1) + 1.0] for i in range(3000)]
op_case("GroupSort_LargeSortingTest", GS + ":543-556", [["id", I32, False], ["grade", F64, False]], _grades,
        ["Sort", [["grade", "ASCENDING"]], None, "INPUT"], [I32, F64], sorted(_grades, key=lambda r: r[1]), ordered=False)
CASES[-1]["sorted_on"] = [1]

# ---- test/smoke_test.cc:79-104: Plus over a two-row table, named attributes -----------------------------------------
op_case("Smoke_ExpressionTest", "test/smoke_test.cc:79-104", [["a", I32, True], ["b", I32, True]], [[1, 2], [3, 5]],
        ["Compute", ["Plus", ["NamedAttribute", "a"], ["NamedAttribute", "b"]], "INPUT"], [I32], [[3], [8]])

# ---- vector_primitives_test.cc: the column primitives under the expressions (wrap-around adds :33-71; the typed fixtures
# ---- :490-507,557-562 check every element against the C++ operator on random data -- restated with a fixed generator and
# ---- the same per-element operator) ------------------------------------------------------------------------------------
VP = "supersonic/expression/vector/vector_primitives_test.cc"
IMAX, IMIN = 2147483647, -2147483648
expr_case("VectorPrimitive_AddDirect", VP + ":33-51", [I32, I32, I32], [[1, 1, 2], [IMAX, IMIN, -1], [-5, 0, -5], [0, 0, 0], [IMAX, 1, IMIN]], "Plus", nullable=False)
expr_case("VectorPrimitive_AddIndirect", VP + ":53-68", [I32, I32, I32], [[1, 1, 2], [IMAX, 0, IMAX], [-5, 0, -5], [0, IMIN, IMIN], [IMAX, 1, IMIN]], "Plus", nullable=False)


def _wrap(v, t):
    if t == I32:
        return (v + (1 << 31)) % (1 << 32) - (1 << 31)
    if t == U32:
        return v % (1 << 32)
    if t == I64:
        return (v + (1 << 63)) % (1 << 64) - (1 << 63)
    return v


def _f32(x):
    import struct
    return struct.unpack("f", struct.pack("f", x))[0]


_r = _lcg(424242)
for _opn, _fac, _fn in (("ADD", "Plus", lambda a, b: a + b), ("SUBTRACT", "Minus", lambda a, b: a - b), ("MULTIPLY", "Multiply", lambda a, b: a * b)):
    for _t in (I32, U32, I64, F64, F32):
        _rows = []
        for _ in range(64):
            if _t in (F64, F32):
                a, b = (_r() % 2000001 - 1000000) / 64.0, (_r() % 2000001 - 1000000) / 128.0     # exact in FLOAT too
                if _t == F32:
                    a, b = _f32(a), _f32(b)
                    _rows.append([a, b, _f32(_fn(a, b))])
                else:
                    _rows.append([a, b, _fn(a, b)])
            else:
                bits = 64 if _t == I64 else 32
                a = _wrap(_r() | (_r() << 31) | (_r() << 62), _t) if bits == 64 else _wrap(_r() | (_r() << 31), _t)
                b = _wrap(_r() | (_r() << 31) | (_r() << 62), _t) if bits == 64 else _wrap(_r() | (_r() << 31), _t)
                _rows.append([a, b, _wrap(_fn(a, b), _t)])
        expr_case("VectorPrimitive_%s_%s" % (_opn, _t), VP + ":490-507", [_t, _t, _t], _rows, _fac, nullable=False)
_rows = [[(_r() % 2000001 - 1000000) / 64.0, float(_r() % 1000 + 1)] for _ in range(64)]
expr_case("VectorPrimitive_DIVIDE_SIGNALING_DOUBLE", VP + ":557", [F64, F64, F64], [[a, b, a / b] for a, b in _rows], "DivideSignaling", nullable=False)
for _n, _fac, _fn in (("OR", "Or", lambda a, b: a or b), ("AND", "And", lambda a, b: a and b), ("AND_NOT", "AndNot", lambda a, b: (not a) and b)):
    expr_case("VectorPrimitive_%s_BOOL" % _n, VP + ":560-562", [BOOL, BOOL, BOOL],
              [[a, b, _fn(a, b)] for a in (False, True) for b in (False, True)], _fac, nullable=False)
# integer division skips rows whose divisor is zero (RunDirectSkipRightZeroTest, :512-555): those rows are NULL in the nulling form
_rows = [[_wrap(_r() | (_r() << 31), I32), (_r() % 7) - 3] for _ in range(50)]
expr_case("VectorPrimitive_DIVIDE_INT32_SkipRightZero", VP + ":508-522", [I32, I32, I32],
          [[a, b, None if b == 0 else _wrap(int(abs(a) // abs(b)) * (1 if (a < 0) == (b < 0) else -1), I32)] for a, b in _rows], "CppDivideNulling", nullable=False)

# ---- binary_column_computers_test.cc:116-235: nullers and failers of the division family over a DOUBLE block; the skip vector of
# ---- the fixture = the incoming NULLs of the left argument ---------------------------------------------------------------
BC = "supersonic/expression/vector/binary_column_computers_test.cc"
expr_case("BinaryComputers_BinaryTrivialNuller", BC + ":116-129", [F64, F64, F64], [[None if i % 2 == 0 else 1.0, 1.0, None if i % 2 == 0 else 1.0] for i in range(5)], "DivideSignaling")
expr_case("BinaryComputers_BinaryNuller", BC + ":131-144", [F64, F64, F64],
          [[None if i % 2 == 0 else 1.0, 0.0 if i < 2 else 1.0, None if (i % 2 == 0 or i < 2) else 1.0] for i in range(5)], "DivideNulling")
expr_case("BinaryComputers_CheckAndNullFailureCount", BC + ":146-163", [F64, F64, F64], [[None if i % 2 == 0 else 1.0, 0.0, None] for i in range(5)], "DivideSignaling", expect_error=104)
expr_case("BinaryComputers_CheckAndNull", BC + ":165-181", [F64, F64, F64],
          [[None if i % 3 == 0 else 1.0, 0.0 if i % 2 == 0 else 1.0, None if (i % 2 == 0 or i % 3 == 0) else 1.0] for i in range(5)], "DivideNulling")
expr_case("BinaryComputers_EvaluationFailure", BC + ":183-193", [F64, F64, F64], [[1.0, 0.0 if i == 3 else 1.0, None] for i in range(5)], "DivideSignaling", expect_error=104)
expr_case("BinaryComputers_NonSelectiveCalculation", BC + ":195-213", [F64, F64, F64],
          [[None if i > 3 else 1.0, float(i), None if (i == 0 or i > 3) else 1.0 / i] for i in range(5)], "DivideNulling")
expr_case("BinaryComputers_SelectiveCalculation", BC + ":215-233", [F64, F64, F64], [[1.0, i + 1.0, 1.0 / (i + 1.0)] for i in range(5)], "CppDivideSignaling")

# ---- aggregation_operators_test.cc: the per-value operators under the aggregates, restated as ScalarAggregate over a column whose
# ---- first value is the operator's starting result -----------------------------------------------------------------------
AO = "supersonic/base/infrastructure/aggregation_operators_test.cc"
op_case("AggregationOperators_Sum_UINT32_to_INT64", AO + ":152-162", cols([U32]), [[1], [30]],
        ["ScalarAggregate", [["SUM", "col0", "s", I64]], "INPUT"], [I64], [[31]])
op_case("AggregationOperators_Max_FLOAT", AO + ":164-178", cols([F32]), [[0.0], [1.0], [0.5], ["-inf"]],
        ["ScalarAggregate", [["MAX", "col0", "m"]], "INPUT"], [F32], [[1.0]])
op_case("AggregationOperators_Max_FLOAT_inf", AO + ":164-178", cols([F32]), [[0.0], [1.0], [0.5], ["-inf"], ["inf"]],
        ["ScalarAggregate", [["MAX", "col0", "m"]], "INPUT"], [F32], [["inf"]])
op_case("AggregationOperators_MaxNoCastArguments", AO + ":200-210", cols([I32]), [[0], [1], [-1]],
        ["ScalarAggregate", [["MAX", "col0", "m", U32]], "INPUT"], [U32], [[1]])
op_case("AggregationOperators_MinForStrings", AO + ":212-227", cols([STR]), [["weasel"], ["zebra"], ["gnu"], ["hippopotamus"], ["antelope (a big one)"]],
        ["ScalarAggregate", [["MIN", "col0", "m"]], "INPUT"], [STR], [["antelope (a big one)"]])
op_case("AggregationOperators_MinForStrings_prefix", AO + ":212-224", cols([STR]), [["weasel"], ["zebra"], ["gnu"], ["hippopotamus"]],
        ["ScalarAggregate", [["MIN", "col0", "m"]], "INPUT"], [STR], [["gnu"]])
op_case("AggregationOperators_First", AO + ":275-285", cols([I32]), [[7], [1], [9]], ["ScalarAggregate", [["FIRST", "col0", "f"]], "INPUT"], [I32], [[7]])
op_case("AggregationOperators_Last", AO + ":287-297", cols([I32]), [[7], [1], [9]], ["ScalarAggregate", [["LAST", "col0", "l"]], "INPUT"], [I32], [[9]])
op_case("AggregationOperators_CrossTypeAssignment", AO + ":35-42", cols([I32]), [[1]],
        ["ScalarAggregate", [["FIRST", "col0", "f", I64]], "INPUT"], [I64], [[1]])


# ---- bound-expression factories: result names, promotions and bind failures (arithmetic_bound_expressions_test.cc,
# ---- elementary_bound_expressions_test.cc, math_bound_expressions_test.cc; inputs are the fixture's NOT NULL $0, $1, ...
# ---- unless TestBoundFactoryWithNulls says otherwise; expect_error -1 = TestBoundFactoryFailure: any bind failure) -------
AB = "supersonic/expression/core/arithmetic_bound_expressions_test.cc"
for _n, _f, _t, _name in (("Negate_INT32", "Negate", [I32], "(-$0)"), ("Negate_UINT32", "Negate", [U32], "(-$0)"),
                          ("Plus", "Plus", [I32, I64], "(CAST_INT32_TO_INT64($0) + $1)"), ("Multiply", "Multiply", [I32, I64], "(CAST_INT32_TO_INT64($0) * $1)"),
                          ("Minus", "Minus", [I32, I64], "(CAST_INT32_TO_INT64($0) - $1)"),
                          ("DivideSignaling", "DivideSignaling", [U32, U64], "(CAST_UINT32_TO_DOUBLE($0) /. CAST_UINT64_TO_DOUBLE($1))"),
                          ("DivideNulling", "DivideNulling", [U32, U64], "(CAST_UINT32_TO_DOUBLE($0) /. CAST_UINT64_TO_DOUBLE($1))"),
                          ("DivideQuiet", "DivideQuiet", [U32, U64], "(CAST_UINT32_TO_DOUBLE($0) /. CAST_UINT64_TO_DOUBLE($1))"),
                          ("CppDivideSignaling", "CppDivideSignaling", [U32, U64], "(CAST_UINT32_TO_UINT64($0) / $1)"),
                          ("CppDivideNulling", "CppDivideNulling", [U32, U64], "(CAST_UINT32_TO_UINT64($0) / $1)"),
                          ("ModulusSignaling", "ModulusSignaling", [U32, U64], "(CAST_UINT32_TO_UINT64($0) % $1)"),
                          ("ModulusNulling", "ModulusNulling", [U32, U64], "(CAST_UINT32_TO_UINT64($0) % $1)")):
    bind_case("ArithmeticBound_" + _n, AB + ":26-77", _f, _t, [False] * len(_t), _name, None, None)

EBT = "supersonic/expression/core/elementary_bound_expressions_test.cc"
bind_plan_case("ElementaryBound_CastTo_same", EBT + ":44-47", ["CastToType", I64, AT(0)], [I64], [False], "$0", I64, False)
bind_plan_case("ElementaryBound_CastTo_widen", EBT + ":44-47", ["CastToType", I64, AT(0)], [I32], [False], "CAST_INT32_TO_INT64($0)", I64, False)
bind_case("ElementaryBound_IfNull_not_nullable", EBT + ":82", "IfNull", [I32, I64], [False, False], "CAST_INT32_TO_INT64($0)", None, None)
bind_case("ElementaryBound_IfNull_nullable_left", EBT + ":83-84", "IfNull", [I32, I32], [True, False], "IFNULL($0, $1)", None, None)
bind_case("ElementaryBound_IfNull_nullable_right", EBT + ":85", "IfNull", [I32, I32], [False, True], "$0", None, None)
for _i, (_t, _name) in enumerate((([BOOL, I32, BOOL, I32], "CASE($0, $1, $2, $3)"), ([STR, STR, STR, STR], "CASE($0, $1, $2, $3)"),
                                  ([U32, STR, I32, STR, I64, STR, U64, STR],
                                   "CASE(CAST_UINT32_TO_INT64($0), $1, CAST_INT32_TO_INT64($2), $3, $4, $5, CAST_UINT64_TO_INT64($6), $7)"),
                                  ([U32, STR, I32, STR, I64, STR, U64, STR, F64, STR, F32, STR],
                                   "CASE(CAST_UINT32_TO_DOUBLE($0), $1, CAST_INT32_TO_DOUBLE($2), $3, CAST_INT64_TO_DOUBLE($4), $5, CAST_UINT64_TO_DOUBLE($6), $7,"
                                   " $8, $9, CAST_FLOAT_TO_DOUBLE($10), $11)"),
                                  ([STR, U32, STR, I32, STR, I64, STR, U64, STR, F64, STR, F32],
                                   "CASE($0, CAST_UINT32_TO_DOUBLE($1), $2, CAST_INT32_TO_DOUBLE($3), $4, CAST_INT64_TO_DOUBLE($5), $6, CAST_UINT64_TO_DOUBLE($7), $8, $9,"
                                   " $10, CAST_FLOAT_TO_DOUBLE($11))"))):
    bind_case("ElementaryBound_Case_%d" % _i, EBT + ":88-118", "CaseList", _t, [False] * len(_t), _name, None, None)
for _i, _t in enumerate(([], [BOOL, I32, I64], [BOOL, I32, BOOL, I32, BOOL, STR], [BOOL, STR, BOOL, I32, BOOL, I32], [BOOL, STR, U32, STR])):
    if _t:
        bind_case("ElementaryBound_Case_fails_%d" % _i, EBT + ":119-137", "CaseList", _t, [False] * len(_t), None, None, None, expect_error=-1)
for _n, _f in (("If", "If"), ("IfNulling", "NullingIf")):
    bind_case("ElementaryBound_%s_INT64" % _n, EBT + ":168-187", _f, [BOOL, I32, I64], [False] * 3, "IF $0 THEN CAST_INT32_TO_INT64($1) ELSE $2", None, None)
    bind_case("ElementaryBound_%s_FLOAT" % _n, EBT + ":168-187", _f, [BOOL, F32, U32], [False] * 3, "IF $0 THEN $1 ELSE CAST_UINT32_TO_FLOAT($2)", None, None)
    bind_case("ElementaryBound_%s_condition_fails" % _n, EBT + ":168-187", _f, [I32, STR, STR], [False] * 3, None, None, None, expect_error=-1)
    bind_case("ElementaryBound_%s_branches_fail" % _n, EBT + ":168-187", _f, [BOOL, U32, STR], [False] * 3, None, None, None, expect_error=-1)
for _n, _f, _name in (("Or", "Or", "($0 OR $1)"), ("And", "And", "($0 AND $1)"), ("AndNot", "AndNot", "($0 !&& $1)"), ("Xor", "Xor", "($0 XOR $1)")):
    bind_case("ElementaryBound_" + _n, EBT + ":190-210", _f, [BOOL, BOOL], [False, False], _name, None, None)
bind_case("ElementaryBound_Not", EBT + ":190-192", "Not", [BOOL], [False], "(NOT $0)", None, None)
bind_case("ElementaryBound_IsNull", EBT + ":214-216", "IsNull", [F32], [True], "ISNULL($0)", None, None)

MB = "supersonic/expression/core/math_bound_expressions_test.cc"
for _n, _f, _t, _name in (("Exp", "Exp", [F64], "EXP($0)"), ("Exp_INT32", "Exp", [I32], "EXP(CAST_INT32_TO_DOUBLE($0))"), ("LnNulling", "LnNulling", [F64], "LN($0)"),
                          ("LnQuiet", "LnQuiet", [F64], "LN($0)"), ("Log10Nulling", "Log10Nulling", [F64], "LOG10($0)"), ("Log2Quiet", "Log2Quiet", [F64], "LOG2($0)"),
                          ("LogNulling", "LogNulling", [F64, F64], "(LN($1) /. LN($0))"), ("PowerSignaling", "PowerSignaling", [F64, F64], "POW($0, $1)"),
                          ("PowerNulling", "PowerNulling", [F64, F64], "POW($0, $1)"), ("PowerQuiet", "PowerQuiet", [F64, F64], "POW($0, $1)"),
                          ("SqrtSignaling", "SqrtSignaling", [F64], "SQRT($0)"), ("SqrtNulling", "SqrtNulling", [F64], "SQRT($0)"), ("SqrtQuiet", "SqrtQuiet", [F64], "SQRT($0)"),
                          ("Sin", "Sin", [F64], "SIN($0)"), ("Cos", "Cos", [F64], "COS($0)"), ("TanQuiet", "Tan", [F64], "TAN($0)"),
                          ("Round", "Round", [F64], "ROUND($0)"), ("Round_INT32", "Round", [I32], "$0"),
                          ("RoundToInt", "RoundToInt", [F64], "CEIL_TO_INT(ROUND($0))"), ("RoundToInt_UINT32", "RoundToInt", [U32], "$0"),
                          ("Floor", "Floor", [F64], "FLOOR($0)"), ("Floor_INT32", "Floor", [I32], "$0"), ("FloorToInt", "FloorToInt", [F64], "FLOOR_TO_INT($0)"),
                          ("FloorToInt_INT32", "FloorToInt", [I32], "$0"), ("Ceil", "Ceil", [F64], "CEIL($0)"), ("Ceil_INT32", "Ceil", [I32], "$0"),
                          ("CeilToInt", "CeilToInt", [F64], "CEIL_TO_INT($0)"), ("CeilToInt_INT32", "CeilToInt", [I32], "$0"), ("Trunc", "Trunc", [F64], "TRUNC($0)"),
                          ("Trunc_UINT64", "Trunc", [U64], "$0"), ("IsFinite", "IsFinite", [F64], "IS_FINITE($0)"), ("IsNormal", "IsNormal", [F64], "IS_NORMAL($0)"),
                          ("IsNaN", "IsNaN", [F64], "IS_NAN($0)"), ("IsInf", "IsInf", [F64], "IS_INF($0)")):
    bind_case("MathBound_" + _n, MB + ":27-149", _f, _t, [False] * len(_t), _name, None, None)


# ---- comparison_bound_expressions_test.cc:35-65: names of the comparison factories (BINARY columns are outside the path) ----
for _n, _f, _t, _name in (("Equal", "Equal", [I32, I64], "($0 == $1)"), ("NotEqual_string", "NotEqual", [STR, STR], "($0 <> $1)"),
                          ("Greater_date", "Greater", [DATE, DATE], "($1 < $0)"), ("Greater_datetime", "Greater", [DATETIME, DATETIME], "($1 < $0)"),
                          ("GreaterOrEqual_bool", "GreaterOrEqual", [BOOL, BOOL], "($1 <= $0)"),
                          ("Less_int_double", "Less", [I32, F64], "(CAST_INT32_TO_DOUBLE($0) < $1)"), ("Less_double_int", "Less", [F64, I32], "($0 < CAST_INT32_TO_DOUBLE($1))"),
                          ("LessOrEqual_float", "LessOrEqual", [F32, F32], "($0 <= $1)"), ("IsOdd", "IsOdd", [I32], "IS_ODD($0)"), ("IsEven", "IsEven", [U64], "IS_EVEN($0)")):
    bind_case("ComparisonBound_" + _n, CB + ":35-65", _f, _t, [False] * len(_t), _name, BOOL, False)
bind_plan_case("InSet_string", CB + ":117-119", IN4, [STR, STR, STR, STR], [False] * 4, "$0 IN ($1, $2, $3)", BOOL, False)
IN3 = ["InList", ["AttributeAt", 0], ["AttributeAt", 1], ["AttributeAt", 2]]
bind_plan_case("InSet_int_string_fails", CB + ":136", IN3, [I32, STR, I32], [False] * 3, None, None, None, expect_error=-1)
bind_plan_case("InSet_string_int_fails", CB + ":137", IN3, [STR, STR, I32], [False] * 3, None, None, None, expect_error=-1)


# ---- terminal_expressions_test.cc:57-66,101-127: NULL and constant terminals over a 5-row input ----------------------------
TE = "supersonic/expression/infrastructure/terminal_expressions_test.cc"
_five = [[i] for i in range(5)]
op_case("Terminal_NullsAreNull", TE + ":57-66", [["x", I32, False]], _five, ["Compute", ["NullOf", F64], "INPUT"], [F64], [[None]] * 5,
        exp_names=["NULL"], exp_nullable=[True])
op_case("Terminal_ConstInt32", TE + ":101-113", [["x", I32, False]], _five, ["Compute", ["ConstInt32", 100], "INPUT"], [I32], [[100]] * 5,
        exp_names=["CONST_INT32"], exp_nullable=[False])
op_case("Terminal_ConstString", TE + ":115-127", [["x", I32, False]], _five, ["Compute", ["ConstString", "Supersonic"], "INPUT"], [STR], [["Supersonic"]] * 5,
        exp_names=["CONST_STRING"], exp_nullable=[False])


# ---- unary_column_computers_test.cc:107-177: NULL flow of unary computers (the fixture's skip vector = the input's NULLs) ----
UC = "supersonic/expression/vector/unary_column_computers_test.cc"
expr_case("UnaryComputers_EvaluationCopiesNulls", UC + ":107-118", [F64, F64], [[None if i % 3 != 0 else 2.0, None if i % 3 != 0 else -2.0] for i in range(4)], "Negate")
expr_case("UnaryComputers_EvaluationIntroducesNulls", UC + ":136-147", [F64, F64], [[1.0 - i, None if i > 1 else (1.0 - i) ** 0.5] for i in range(4)], "SqrtNulling", nullable=False)
expr_case("UnaryComputers_EvaluationWorksForSafe", UC + ":158-177", [F64, F64], [[4.0, 2.0], [9.0, 3.0], [-1.0, None]], "SqrtNulling", nullable=False)
expr_case("UnaryComputers_SqrtNulling_keeps_input_nulls", UC + ":120-134", [F64, F64], [[None if i % 3 != 0 else 1.0 - i, None if (i % 3 != 0 or 1 - i < 0) else (1.0 - i) ** 0.5] for i in range(4)], "SqrtNulling")

# ---- hybrid_aggregate_test.cc: DISTINCT next to plain aggregations, and DISTINCT over two different columns.  The operation
# under test there is HybridGroupAggregate, whose result contract is GroupAggregate's (aggregate.h: the hybrid form differs in
# how it spills, not in what it returns); restated here on GroupAggregate, which is what the device path implements.
HY = "supersonic/cursor/core/hybrid_aggregate_test.cc"
op_case("Hybrid_NoGroupByColumns", HY + ":591-616", cols([I32]), [[1], [1], [3], [3], [2], [3], [1]],
        ["GroupAggregate", ["CompoundSingleSourceProjector"], [["SUM", "col0", "sum"], ["COUNT", "col0", "cnt"], ["COUNT_DISTINCT", "col0", "dcnt"]], "INPUT"],
        [I32, U64, U64], [[14, 7, 3]])
op_case("Hybrid_Simple1", HY + ":618-650", cols([I32, I32]), [[1, 3], [1, 4], [3, -3], [2, 4], [3, -5]],
        ["GroupAggregate", ["ProjectNamedAttribute", "col0"],
         [["SUM", "col1", "sum"], ["SUM", "col0", "sum2"], ["COUNT", "col0", "cnt"], ["COUNT_DISTINCT", "col0", "dcnt"]], "INPUT"],
        [I32, I32, I32, U64, U64], [[1, 7, 2, 2, 1], [2, 4, 2, 1, 1], [3, -8, 6, 2, 1]], ordered=False)
op_case("Hybrid_DistinctAggregations_two_columns", HY + ":652-681", cols([I32, I32]), [[1, 3], [1, 4], [3, -1], [3, -2], [2, 4], [3, -3], [1, 3]],
        ["GroupAggregate", ["ProjectNamedAttribute", "col0"],
         [["SUM_DISTINCT", "col1", "sum"], ["COUNT_DISTINCT", "col1", "cnt"], ["COUNT_DISTINCT", "col0", "cnt2"]], "INPUT"],
        [I32, I32, U64, U64], [[1, 7, 2, 1], [2, 4, 1, 1], [3, -6, 3, 1]], ordered=False)
op_case("Hybrid_NonDistinctAndDistinctAggregations", HY + ":683-716", cols([I32, I32]),
        [[1, 3], [1, 4], [3, -1], [3, -2], [2, 4], [3, -3], [1, 3], [1, None]],
        ["GroupAggregate", ["ProjectNamedAttribute", "col0"],
         [["SUM_DISTINCT", "col1", "sum"], ["COUNT_DISTINCT", "col1", "cnt"], ["SUM", "col1", "sum2"], ["COUNT", "col1", "cnt2"], ["COUNT", "", "cnt3"]], "INPUT"],
        [I32, I32, U64, I32, U64, U64], [[1, 7, 2, 10, 3, 4], [2, 4, 1, 4, 1, 1], [3, -6, 3, -6, 3, 3]], ordered=False)

_hy_rows1 = [[1], [1], [3], [3], [2], [3], [1]]
_hy_spec = [["SUM", "col0", "sum"], ["COUNT", "col0", "cnt"], ["COUNT_DISTINCT", "col0", "dcnt"]]
op_case("Hybrid_GroupByColumnPosition1", HY + ":894-919", cols([I32, I32]), [[r[0], 0] for r in _hy_rows1],
        ["GroupAggregate", ["ProjectAttributeAt", 1], _hy_spec, "INPUT"], [I32, I32, U64, U64], [[0, 14, 7, 3]], exp_names=["col1", "sum", "cnt", "dcnt"])
op_case("Hybrid_GroupByAllColumns", HY + ":921-948", cols([I32]), _hy_rows1,
        ["GroupAggregate", ["ProjectAllAttributes"], _hy_spec, "INPUT"], [I32, I32, U64, U64], [[1, 3, 3, 1], [2, 2, 1, 1], [3, 9, 3, 1]], ordered=False)
op_case("Hybrid_GroupByColumnRenamed", HY + ":950-977", cols([I32, I32]), [[r[0], 0] for r in _hy_rows1],
        ["GroupAggregate", ["ProjectNamedAttributeAs", "col1", "key"], _hy_spec, "INPUT"], [I32, I32, U64, U64], [[0, 14, 7, 3]],
        exp_names=["key", "sum", "cnt", "dcnt"])

# ---- base/infrastructure/projector_test.cc: the projectors the operations take, bound against a schema (restated as a Project over two rows)
PJ = "supersonic/base/infrastructure/projector_test.cc"
_pj_schema = [["schema 0 attribute 0", I64, True], ["schema 0 attribute 1", STR, True]]
_pj_rows = [[7, "x"], [None, None]]
op_case("Projector_AllAttributes", PJ + ":68-75", _pj_schema, _pj_rows, ["Project", ["ProjectAllAttributes"], "INPUT"], [I64, STR], _pj_rows,
        exp_names=["schema 0 attribute 0", "schema 0 attribute 1"], exp_nullable=[True, True])
op_case("Projector_AllAttributesWithPrefix", PJ + ":76-85", _pj_schema, _pj_rows, ["Project", ["ProjectAllAttributes", "prefix "], "INPUT"], [I64, STR], _pj_rows,
        exp_names=["prefix schema 0 attribute 0", "prefix schema 0 attribute 1"], exp_nullable=[True, True])
op_case("Projector_AttributeAtPosition", PJ + ":87-99", _pj_schema, _pj_rows, ["Project", ["ProjectAttributeAt", 1], "INPUT"], [STR], [["x"], [None]],
        exp_names=["schema 0 attribute 1"], exp_nullable=[True])

# ---- base/infrastructure/operators_test.cc: Equal / Less across signed and unsigned integers (the operators behind the
# comparison expressions and ThreeWayCompare).  Each EXPECT is restated as the comparison expression over two columns of
# the operand types; the negative operands are the tests' static_cast<unsigned>(-5) values.
OT = "supersonic/base/infrastructure/operators_test.cc"
_INTS = [I32, U32, I64, U64]


def _neg5(t):
    return {I32: -5, I64: -5, U32: (1 << 32) - 5, U64: (1 << 64) - 5}[t]


_eq_rows = {}
for ta in _INTS:
    for tb in _INTS:
        rows = [[3, 3, True]]                                        # :41-59 eq(k*_3, k*_3)
        if (ta in (I32, I64)) != (tb in (I32, I64)):                 # :61-69 a negative number never equals a huge unsigned one
            rows.append([_neg5(ta), _neg5(tb), False])
        expr_case("Operators_Equal_MixedNumerics_%s_%s" % (ta, tb), OT + ":40-72", [ta, tb, BOOL], rows, "Equal")
expr_case("Operators_Equal_MixedNumerics_3_vs_5", OT + ":71", [I32, I64, BOOL], [[3, 5, False]], "Equal")
expr_case("Operators_Equal_Bool", OT + ":74-80", [BOOL, BOOL, BOOL], [[False, False, True], [False, True, False], [True, False, False], [True, True, True]], "Equal")
expr_case("Operators_Equal_String", OT + ":82-90", [STR, STR, BOOL], [["a", "a", True], ["a", "aa", False], ["aa", "a", False], ["aa", "aa", True]], "Equal")
expr_case("Operators_Less_Trivial", OT + ":94-99", [I32, I32, BOOL], [[3, 5, True], [-5, 3, True], [5, -5, False]], "Less")
for ts in (I32, I64):                                                # :101-124 signed vs unsigned, both ways
    for tu in (U32, U64):
        expr_case("Operators_Less_MixedNumerics_%s_%s" % (ts, tu), OT + ":101-124", [ts, tu, BOOL], [[-5, 5, True], [3, 5, True]], "Less")
        expr_case("Operators_Less_MixedNumerics_%s_%s" % (tu, ts), OT + ":101-124", [tu, ts, BOOL], [[5, -5, False], [5, 3, False]], "Less")
expr_case("Operators_Less_Bool", OT + ":126-132", [BOOL, BOOL, BOOL], [[False, False, False], [False, True, True], [True, False, False], [True, True, False]], "Less")
expr_case("Operators_Less_String", OT + ":134-142", [STR, STR, BOOL], [["a", "a", False], ["a", "aa", True], ["aa", "a", False], ["aa", "aa", False]], "Less")
_cmp3 = [[5, 3], [5, 5], [3, 5]]
for _f, _want, _src in (("Greater", [True, False, False], ":146-151"), ("LessOrEqual", [False, True, True], ":153-158"),
                        ("GreaterOrEqual", [True, True, False], ":160-165"), ("NotEqual", [True, False, True], ":167-172")):
    expr_case("Operators_Complements_" + _f, OT + _src, [I32, I32, BOOL], [r + [w] for r, w in zip(_cmp3, _want)], _f)


# =====================================================================================================================================
# DERIVED cases (kind "derived"): combinations the reference's own tests hold no vector of.  The expected rows are worked out BY HAND
# from the cited reference lines -- each case carries its derivation -- not taken from the oracle or the device: a misreading shared by
# oracle and kernels (written by the same hand) would show here as a disagreement with the derivation.
# =====================================================================================================================================
def derived_case(name, source, derivation, schema, rows, plan, exp_types, exp_rows, ordered=True):
    CASES.append({"name": name, "source": source, "kind": "derived", "derivation": derivation,
                  "input": {"schema": schema, "rows": rows}, "plan": plan,
                  "expected": {"types": exp_types, "rows": exp_rows, "names": None, "nullable": None},
                  "ordered": ordered, "expect_error": None})


AGG = "supersonic/cursor/core/aggregator.cc:88-101"
CLC = "supersonic/cursor/core/aggregate_clusters.cc:436-520"
DST = "supersonic/cursor/core/column_aggregator.cc:308-376"
RHS = "supersonic/cursor/infrastructure/row_hash_set.cc:500-511"
AOP = "supersonic/base/infrastructure/aggregation_operators.h:290-320"
D_CLUSTER = ("Aggregator::Init makes a DistinctAggregator for every aggregation with is_distinct, whatever cursor owns the Aggregator (" + AGG + "); "
             "AggregateClusters numbers the clusters of a block 0, 1, 2 ... as result indices and a cluster is a run of ADJACENT equal keys (" + CLC + "); "
             "DistinctAggregator keeps one value set PER RESULT INDEX, skips NULL inputs and feeds the inner aggregator only the first occurrence of a value (" + DST + "). ")

# ---- A: DISTINCT aggregates inside AggregateClusters --------------------------------------------------------------------------------
derived_case("Derived_Clusters_DistinctPerCluster", AGG + "; " + CLC + "; " + DST,
             D_CLUSTER + "Keys 1,1,1,2,2,1 = three clusters (the last 1 is not adjacent to the first run).  Cluster 0 holds 5,5,7: distinct {5,7} -> SUM 12, "
             "COUNT 2; plain COUNT 3, SUM 17.  Cluster 1 holds 5,NULL: distinct {5} -> 5, 1; COUNT 1, SUM 5.  Cluster 2 holds 5 -> 5, 1, 1, 5 "
             "(its set is its own: the 5 of cluster 0 does not hide it).",
             cols([I32, I32]), [[1, 5], [1, 5], [1, 7], [2, 5], [2, None], [1, 5]],
             ["AggregateClusters", ["ProjectNamedAttribute", "col0"],
              [["SUM_DISTINCT", "col1", "sd"], ["COUNT_DISTINCT", "col1", "cd"], ["COUNT", "col1", "c"], ["SUM", "col1", "s"]], "INPUT"],
             [I32, I32, U64, U64, I32], [[1, 12, 2, 3, 17], [2, 5, 1, 1, 5], [1, 5, 1, 1, 5]])
derived_case("Derived_Clusters_DistinctAllNullCluster", AGG + "; " + DST,
             D_CLUSTER + "A cluster whose values are all NULL never calls the inner aggregator: SUM DISTINCT stays NULL (the aggregate's initial state, "
             "column_aggregator.cc:108-124), COUNT DISTINCT 0.  The next cluster (9,9) gives 9 and 1.",
             cols([I32, I32]), [[4, None], [4, None], [6, 9], [6, 9]],
             ["AggregateClusters", ["ProjectNamedAttribute", "col0"], [["SUM_DISTINCT", "col1", "sd"], ["COUNT_DISTINCT", "col1", "cd"]], "INPUT"],
             [I32, I32, U64], [[4, None, 0], [6, 9, 1]])
derived_case("Derived_Clusters_DistinctWithoutClusteredColumn", AGG + "; " + CLC + "; " + DST,
             D_CLUSTER + "No clustering column: the whole input is one cluster (aggregate_clusters_test.cc:105-121).  Values 3,1,3,NULL,1 -> set {3,1}: "
             "SUM DISTINCT 4, COUNT DISTINCT 2, MAX DISTINCT 3 (MAX of the distinct values is the MAX), COUNT(*) 5.",
             cols([I32]), [[3], [1], [3], [None], [1]],
             ["AggregateClusters", ["CompoundSingleSourceProjector"],
              [["SUM_DISTINCT", "col0", "sd"], ["COUNT_DISTINCT", "col0", "cd"], ["MAX_DISTINCT", "col0", "mx"], ["COUNT", "", "n"]], "INPUT"],
             [I32, U64, I32, U64], [[4, 2, 3, 5]])
derived_case("Derived_Clusters_TwoDistinctColumns", AGG + "; " + DST,
             D_CLUSTER + "Each DISTINCT aggregation has its own DistinctAggregator, so two columns keep independent sets.  Cluster 1: a = 1,1 -> {1}: COUNT 1; "
             "b = 9,8 -> {9,8}: COUNT 2, SUM 17.  Cluster 2: a = 1,2,2 -> {1,2}: COUNT 2, SUM 3; b = 9,9,9 -> {9}: COUNT 1, SUM 9.",
             cols([I32, I32, I32]), [[1, 1, 9], [1, 1, 8], [2, 1, 9], [2, 2, 9], [2, 2, 9]],
             ["AggregateClusters", ["ProjectNamedAttribute", "col0"],
              [["COUNT_DISTINCT", "col1", "ca"], ["SUM_DISTINCT", "col1", "sa"], ["COUNT_DISTINCT", "col2", "cb"], ["SUM_DISTINCT", "col2", "sb"]], "INPUT"],
             [I32, U64, I32, U64, I32], [[1, 1, 1, 2, 17], [2, 2, 3, 1, 9]])
derived_case("Derived_Clusters_DistinctDoubles", AGG + "; " + DST,
             D_CLUSTER + "DOUBLE values compare by value in the set: 0.5,0.5,0.25 -> {0.5,0.25}: SUM DISTINCT 0.75 (exact), COUNT DISTINCT 2; cluster 8: -1.5 alone.",
             cols([I32, F64]), [[7, 0.5], [7, 0.5], [7, 0.25], [8, -1.5]],
             ["AggregateClusters", ["ProjectNamedAttribute", "col0"], [["SUM_DISTINCT", "col1", "sd"], ["COUNT_DISTINCT", "col1", "cd"]], "INPUT"],
             [I32, F64, U64], [[7, 0.75, 2], [8, -1.5, 1]])
_big = [[i // 1500, i % 7] for i in range(3000)]
derived_case("Derived_Clusters_DistinctAcrossInputBlocks", CLC + "; " + DST,
             D_CLUSTER + "Two clusters of 1500 rows: longer than the 1024-row blocks the cursor pulls, so each cluster meets the aggregator in more than one "
             "ProcessInput call; the trailing partial cluster of a block is re-processed with the next one (aggregate_clusters.cc:470-520), so its set still "
             "sees every value once.  Values i mod 7: every cluster holds 0..6 -> COUNT DISTINCT 7, SUM DISTINCT 21; COUNT(*) 1500 each.",
             cols([I32, I32]), _big,
             ["AggregateClusters", ["ProjectNamedAttribute", "col0"], [["COUNT_DISTINCT", "col1", "cd"], ["SUM_DISTINCT", "col1", "sd"], ["COUNT", "", "n"]], "INPUT"],
             [I32, U64, I32, U64], [[0, 7, 21, 1500], [1, 7, 21, 1500]])

# ---- B: FIRST / LAST next to DISTINCT aggregates -------------------------------------------------------------------------------------
D_FL = ("FIRST assigns a group's first non-NULL value and never changes it, LAST assigns every non-NULL value (" + AOP + "; NULL inputs are skipped by "
        "ColumnAggregatorImpl::UpdateAggregation, column_aggregator.cc:108-124): both follow the INPUT order, whatever the DISTINCT aggregates next to them do (" + DST + "). ")
derived_case("Derived_Group_FirstLastNextToDistinct", AOP + "; " + DST,
             D_FL + "Group 1 = rows 0,2,4: v = NULL,7,9 -> FIRST 7, LAST 9; w = 3,5,3 -> {3,5}: COUNT DISTINCT 2, SUM DISTINCT 8.  Group 2 = rows 1,3: v = 4,4 -> "
             "FIRST 4, LAST 4; w = 3,NULL -> {3}: 1, 3.",
             cols([I32, I32, I32]), [[1, None, 3], [2, 4, 3], [1, 7, 5], [2, 4, None], [1, 9, 3]],
             ["GroupAggregate", ["ProjectNamedAttribute", "col0"],
              [["FIRST", "col1", "f"], ["LAST", "col1", "l"], ["COUNT_DISTINCT", "col2", "cd"], ["SUM_DISTINCT", "col2", "sd"]], "INPUT"],
             [I32, I32, I32, U64, I32], [[1, 7, 9, 2, 8], [2, 4, 4, 1, 3]], ordered=False)
derived_case("Derived_Scalar_FirstLastNextToDistinct", AOP + "; " + DST,
             D_FL + "One group of NULL,2,2,5,NULL: FIRST 2, LAST 5, set {2,5}: COUNT DISTINCT 2, SUM DISTINCT 7; COUNT 3.",
             cols([I32]), [[None], [2], [2], [5], [None]],
             ["ScalarAggregate", [["FIRST", "col0", "f"], ["LAST", "col0", "l"], ["COUNT_DISTINCT", "col0", "cd"], ["SUM_DISTINCT", "col0", "sd"], ["COUNT", "col0", "c"]], "INPUT"],
             [I32, I32, U64, I32, U64], [[2, 5, 2, 7, 3]])
derived_case("Derived_Group_FirstLastOfAnAllNullGroupNextToDistinct", AOP + "; " + DST,
             D_FL + "Group 5 has only NULL values of v: FIRST and LAST stay NULL; its w = 1,1 -> COUNT DISTINCT 1.  Group 6: v = 8 -> FIRST = LAST = 8; w = 2,1 -> 2.",
             cols([I32, I32, I32]), [[5, None, 1], [6, 8, 2], [5, None, 1], [6, None, 1]],
             ["GroupAggregate", ["ProjectNamedAttribute", "col0"], [["FIRST", "col1", "f"], ["LAST", "col1", "l"], ["COUNT_DISTINCT", "col2", "cd"]], "INPUT"],
             [I32, I32, I32, U64], [[5, None, None, 1], [6, 8, 8, 2]], ordered=False)
derived_case("Derived_Group_LastFollowsInputOrderNotValueOrder", AOP + "; " + DST,
             D_FL + "The DISTINCT aggregate's value order must not leak into LAST: group 1's v arrives as 9,1,5 -> FIRST 9, LAST 5 (not the largest, not the smallest); "
             "SUM DISTINCT of v = 15; group 2: v = 2,2 -> FIRST 2, LAST 2, SUM DISTINCT 2.",
             cols([I32, I32]), [[1, 9], [2, 2], [1, 1], [1, 5], [2, 2]],
             ["GroupAggregate", ["ProjectNamedAttribute", "col0"], [["FIRST", "col1", "f"], ["LAST", "col1", "l"], ["SUM_DISTINCT", "col1", "sd"]], "INPUT"],
             [I32, I32, I32, I32], [[1, 9, 5, 15], [2, 2, 2, 2]], ordered=False)

# ---- C: max_unique_keys_in_result with FIRST / LAST and with keys wider than 64 bits -----------------------------------------------------
D_LIM = ("RowHashSet::Insert appends an unseen key while the index holds <= max_unique_keys_in_result rows and answers every LATER unseen key with the index's "
         "last row (" + RHS + "): rows 0 .. limit are the first limit + 1 keys in first-seen order, and row `limit` also receives every row of every other key, "
         "in input order.  The result keeps the index's order (aggregate_groups.cc:404-433). ")
_lim_keys = [7, 8, 9, 8, 10, 7]
derived_case("Derived_Limit_FirstLastOfTheFoldedRow", RHS + "; " + AOP,
             D_LIM + D_FL + "Limit 1: 7 -> row 0, 8 -> row 1 (the index held 1 <= 1 rows), 9 -> unseen, index holds 2 > 1 -> row 1; 8 -> row 1; 10 -> row 1; 7 -> row 0.  "
             "Row 0 = inputs 0,5: SUM 1 + 6 = 7, FIRST 1, LAST 6, COUNT 2.  Row 1 = inputs 1,2,3,4 (values 2,3,4,5): SUM 14, FIRST 2, LAST 5 -- the LAST of "
             "row 1 comes from key 10 -- COUNT 4.",
             cols([I32, I32], nullable=False), [[k, v] for k, v in zip(_lim_keys, [1, 2, 3, 4, 5, 6])],
             ["GroupAggregate", ["ProjectNamedAttribute", "col0"], [["SUM", "col1", "s"], ["FIRST", "col1", "f"], ["LAST", "col1", "l"], ["COUNT", "", "n"]], "INPUT",
              {"max_unique_keys_in_result": 1}],
             [I32, I32, I32, I32, U64], [[7, 7, 1, 6, 2], [8, 14, 2, 5, 4]])
derived_case("Derived_Limit_FirstLastSkipNullsOfTheFoldedRow", RHS + "; " + AOP,
             D_LIM + D_FL + "Same keys, values 1,NULL,3,NULL,NULL,6.  Row 1 = inputs 1..4 = NULL,3,NULL,NULL: its own key 8 only brings NULLs, the one value 3 comes "
             "from key 9: FIRST 3, LAST 3, COUNT(v) 1, SUM 3.  Row 0: 1 and 6.",
             cols([I32, I32]), [[k, v] for k, v in zip(_lim_keys, [1, None, 3, None, None, 6])],
             ["GroupAggregate", ["ProjectNamedAttribute", "col0"], [["FIRST", "col1", "f"], ["LAST", "col1", "l"], ["COUNT", "col1", "c"], ["SUM", "col1", "s"]], "INPUT",
              {"max_unique_keys_in_result": 1}],
             [I32, I32, I32, U64, I32], [[7, 1, 6, 2, 7], [8, 3, 3, 1, 3]])
derived_case("Derived_Limit_ZeroKeepsOneRow", RHS,
             D_LIM + "Limit 0: the first key is appended (the index held 0 <= 0 rows), every other key folds into it: ONE row, key 7, SUM 21, MIN 1, MAX 6, COUNT 6, "
             "FIRST 1, LAST 6.",
             cols([I32, I32], nullable=False), [[k, v] for k, v in zip(_lim_keys, [1, 2, 3, 4, 5, 6])],
             ["GroupAggregate", ["ProjectNamedAttribute", "col0"],
              [["SUM", "col1", "s"], ["MIN", "col1", "mn"], ["MAX", "col1", "mx"], ["COUNT", "", "n"], ["FIRST", "col1", "f"], ["LAST", "col1", "l"]], "INPUT",
              {"max_unique_keys_in_result": 0}],
             [I32, I32, I32, I32, U64, I32, I32], [[7, 21, 1, 6, 6, 1, 6]])
derived_case("Derived_Limit_NotReachedChangesNothing", RHS,
             D_LIM + "Limit 5 with four distinct keys: nothing folds.  First-seen order 7, 8, 9, 10: SUMs 7, 6, 3, 5; LASTs 6, 4, 3, 5.",
             cols([I32, I32], nullable=False), [[k, v] for k, v in zip(_lim_keys, [1, 2, 3, 4, 5, 6])],
             ["GroupAggregate", ["ProjectNamedAttribute", "col0"], [["SUM", "col1", "s"], ["LAST", "col1", "l"]], "INPUT", {"max_unique_keys_in_result": 5}],
             [I32, I32, I32], [[7, 7, 6], [8, 6, 4], [9, 3, 3], [10, 5, 5]])
derived_case("Derived_Limit_WideKeys", RHS + "; " + AOP,
             D_LIM + D_FL + "Two INT64 keys (128 packed bits: the device's sorted shape).  Limit 1: (1,1) -> row 0; (1,2) -> row 1; (2,1) unseen, index holds 2 -> row 1; "
             "(1,2) -> row 1; (1,1) -> row 0.  Row 0: 10 + 50 = 60, FIRST 10, LAST 50.  Row 1: 20 + 30 + 40 = 90, FIRST 20, LAST 40, and it keeps the key (1,2).",
             cols([I64, I64, I32], nullable=False), [[1, 1, 10], [1, 2, 20], [2, 1, 30], [1, 2, 40], [1, 1, 50]],
             ["GroupAggregate", ["ProjectNamedAttributes", ["col0", "col1"]], [["SUM", "col2", "s"], ["FIRST", "col2", "f"], ["LAST", "col2", "l"]], "INPUT",
              {"max_unique_keys_in_result": 1}],
             [I64, I64, I32, I32, I32], [[1, 1, 60, 10, 50], [1, 2, 90, 20, 40]])
derived_case("Derived_Limit_NullKeyIsAKey", RHS + "; supersonic/cursor/infrastructure/row_hash_set.cc:143-210",
             D_LIM + "NULL keys compare equal to each other (RowComparator, row_hash_set.cc:143-210), so NULL is one key like any other.  Limit 1: NULL -> row 0, "
             "5 -> row 1, NULL -> row 0, 6 -> folds into row 1.  Row 0 (key NULL): 1 + 3 = 4; row 1 (key 5): 2 + 4 = 6, LAST 4.",
             cols([I32, I32]), [[None, 1], [5, 2], [None, 3], [6, 4]],
             ["GroupAggregate", ["ProjectNamedAttribute", "col0"], [["SUM", "col1", "s"], ["LAST", "col1", "l"], ["COUNT", "", "n"]], "INPUT",
              {"max_unique_keys_in_result": 1}],
             [I32, I32, I32, U64], [[None, 4, 3, 2], [5, 6, 4, 2]])
derived_case("Derived_Limit_WideKeysMinMaxCount", RHS,
             D_LIM + "Three INT32 keys (96 packed bits).  Limit 2: (1,1,1) -> 0, (2,2,2) -> 1, (3,3,3) -> 2, (4,4,4) -> folds into row 2, (2,2,2) -> 1, (5,5,5) -> row 2.  "
             "Row 2 = values 30, 40, 60: MIN 30, MAX 60, COUNT 3, SUM 130; row 1 = 20, 50.",
             cols([I32, I32, I32, I32], nullable=False), [[1, 1, 1, 10], [2, 2, 2, 20], [3, 3, 3, 30], [4, 4, 4, 40], [2, 2, 2, 50], [5, 5, 5, 60]],
             ["GroupAggregate", ["ProjectNamedAttributes", ["col0", "col1", "col2"]],
              [["MIN", "col3", "mn"], ["MAX", "col3", "mx"], ["COUNT", "", "n"], ["SUM", "col3", "s"]], "INPUT", {"max_unique_keys_in_result": 2}],
             [I32, I32, I32, I32, I32, U64, I32], [[1, 1, 1, 10, 10, 1, 10], [2, 2, 2, 20, 50, 2, 70], [3, 3, 3, 30, 60, 3, 130]])

# ---- C2: DISTINCT aggregates under max_unique_keys_in_result: one seen-value set per RESULT row ------------------------------------------------
D_LIMD = (D_LIM + "DistinctAggregator::UpdateAggregation looks a row's value up in distinct_values_[result_index_map[i]] (" + DST + "): the set belongs to the RESULT "
          "ROW the hash set answered, so every key folded into row `limit` shares that row's one set; NULL inputs are skipped. ")
derived_case("Derived_Limit_DistinctSharesTheFoldedRowsSet", RHS + "; " + DST,
             D_LIMD + "Keys 1,2,3,4,3,5,4,1,2 under limit 2: rows 0, 1 = keys 1, 2; row 2 = key 3 and, folded, 4 and 5.  Row 2 sees 7 (key 3), 7 (key 4: already in "
             "the row's set), 8 (key 3), 8 (key 5: seen), 9 (key 4): distinct {7, 8, 9} -> COUNT 3, SUM 24 -- per-key answers would add up to 2 + 2 + 1 = 5 and 46; plain "
             "COUNT(*) 5.  Row 0: 5, 5 -> {5}: 1, 5, 2 rows.  Row 1: 6, 1 -> 2, 7, 2 rows.",
             cols([I32, I64], nullable=False), [[1, 5], [2, 6], [3, 7], [4, 7], [3, 8], [5, 8], [4, 9], [1, 5], [2, 1]],
             ["GroupAggregate", ["ProjectNamedAttribute", "col0"], [["COUNT_DISTINCT", "col1", "c"], ["SUM_DISTINCT", "col1", "s"], ["COUNT", "", "n"]], "INPUT",
              {"max_unique_keys_in_result": 2}],
             [I32, U64, I64, U64], [[1, 1, 5, 2], [2, 2, 7, 2], [3, 3, 24, 5]])
derived_case("Derived_Limit_DistinctZeroLimitIsAScalarDistinct", RHS + "; " + DST,
             D_LIMD + "Limit 0: every row lands in row 0, whose key is the first one seen (9).  Values 4, NULL, 4, 6, NULL, 6, 5: distinct {4, 6, 5} -> COUNT 3, SUM 15; "
             "COUNT(v) 5; LAST 5 (NULLs never count).",
             cols([I32, I32]), [[9, 4], [8, None], [7, 4], [9, 6], [None, None], [6, 6], [8, 5]],
             ["GroupAggregate", ["ProjectNamedAttribute", "col0"], [["COUNT_DISTINCT", "col1", "c"], ["SUM_DISTINCT", "col1", "s"], ["COUNT", "col1", "n"], ["LAST", "col1", "l"]],
              "INPUT", {"max_unique_keys_in_result": 0}],
             [I32, U64, I32, U64, I32], [[9, 3, 15, 5, 5]])
derived_case("Derived_Limit_DistinctNullKeyOwnsTheFoldedRow", RHS + "; " + DST + "; supersonic/cursor/infrastructure/row_hash_set.cc:143-210",
             D_LIMD + "NULL is a key (RowComparator, row_hash_set.cc:143-210).  Keys 5, NULL, 6, NULL, 7 under limit 1: row 0 = key 5; row 1 = key NULL (the index held 1 <= 1 "
             "rows), and 6, 7 fold into it -- the row shows the key NULL.  Row 1 sees 2, 3, 2, 3: distinct {2, 3} -> COUNT 2, SUM 5, FIRST 2; row 0: {1} -> 1, 1, 1.",
             cols([I32, I32]), [[5, 1], [None, 2], [6, 3], [None, 2], [7, 3]],
             ["GroupAggregate", ["ProjectNamedAttribute", "col0"], [["COUNT_DISTINCT", "col1", "c"], ["SUM_DISTINCT", "col1", "s"], ["FIRST", "col1", "f"]], "INPUT",
              {"max_unique_keys_in_result": 1}],
             [I32, U64, I32, I32], [[5, 1, 1, 1], [None, 2, 5, 2]])

derived_case("Derived_Limit_ConcatJoinsTheFoldedRowsValuesInInputOrder", RHS + "; supersonic/cursor/core/column_aggregator.cc:108-124",
             D_LIM + "CONCAT appends a row's printed value to the string of its RESULT row (column_aggregator.cc:108-124 walks result_index_map like every other "
             "aggregator; NULL inputs are skipped, ',' between values).  Keys 1,2,3,4,3,5,4,1,2 under limit 2: row 2 = keys 3, 4, 5 = inputs 2,3,4,5,6 = 7, 7, 8, NULL, 9 "
             "-> '7,7,8,9' in INPUT order (not key after key); row 0 = '5,5'; row 1 = '6,1'.",
             cols([I32, I64]), [[1, 5], [2, 6], [3, 7], [4, 7], [3, 8], [5, None], [4, 9], [1, 5], [2, 1]],
             ["GroupAggregate", ["ProjectNamedAttribute", "col0"], [["CONCAT", "col1", "c"], ["COUNT", "", "n"]], "INPUT", {"max_unique_keys_in_result": 2}],
             [I32, STR, U64], [[1, "5,5", 2], [2, "6,1", 2], [3, "7,7,8,9", 5]])

derived_case("Derived_DistinctConcatPrintsFirstOccurrences", DST + "; supersonic/cursor/core/column_aggregator.cc:108-124,568-590",
             "CreateDistinctAggregator wraps the CONCAT column aggregator in a DistinctAggregator (column_aggregator.cc:568-590), which hands it only the rows whose value "
             "is not yet in the set of their result index (" + DST + "), NULLs never; the CONCAT then appends those, ',' between values, in input order.  Group 1 "
             "sees 5, 7, 5, NULL, 7, 9 -> '5,7,9' (plain CONCAT next to it: '5,7,5,7,9'); group 2 sees 5, 5 -> '5' (its set is its own); group 3 only NULL -> NULL.",
             cols([I32, I32]), [[1, 5], [2, 5], [1, 7], [1, 5], [3, None], [1, None], [2, 5], [1, 7], [1, 9]],
             ["GroupAggregate", ["ProjectNamedAttribute", "col0"], [["CONCAT_DISTINCT", "col1", "d"], ["CONCAT", "col1", "c"]], "INPUT"],
             [I32, STR, STR], [[1, "5,7,9", "5,7,5,7,9"], [2, "5", "5,5"], [3, None, None]])

# ---- D: SUM of a floating input into an integer result (the reference's row-after-row arithmetic) --------------------------------------------
D_SEQ = ("AddAggregationWithDefinedOutputType(SUM, DOUBLE column, INT result): AggregationOperator<SUM>::Update is `*result += val` on an integer result and a "
         "floating val (supersonic/base/infrastructure/aggregation_operators.h:173-185): C++ converts *result to the floating type, adds, and truncates the sum "
         "back toward zero -- after EVERY row; the first value is assigned (truncated). ")
derived_case("Derived_SumOfDoublesIntoInt32TruncatesEveryRow", "supersonic/base/infrastructure/aggregation_operators.h:173-185",
             D_SEQ + "0.6, 0.6, 0.6, 0.6: assigned 0 (trunc 0.6), then 0 + 0.6 = 0.6 -> 0, again 0, again 0: result 0, where a sum first, truncate last reading gives 2.  "
             "1.5, 1.5: assigned 1, then 1 + 1.5 = 2.5 -> 2.  -0.9, -0.9, -0.3: assigned 0 (trunc toward zero), 0 - 0.9 -> 0, 0 - 0.3 -> 0.",
             cols([I32, F64]), [[1, 0.6], [1, 0.6], [1, 0.6], [1, 0.6], [2, 1.5], [2, 1.5], [3, -0.9], [3, -0.9], [3, -0.3]],
             ["AggregateClusters", ["ProjectNamedAttribute", "col0"], [["SUM", "col1", "s", I32]], "INPUT"],
             [I32, I32], [[1, 0], [2, 2], [3, 0]])
derived_case("Derived_Limit_SumOfDoublesIntoInt64FoldsTheFoldedRowInInputOrder", RHS + "; supersonic/base/infrastructure/aggregation_operators.h:173-185",
             D_LIM + D_SEQ + "Keys 1, 2, 3, 2 under limit 1: row 0 = key 1 = 5.5 -> 5.  Row 1 = key 2 and, folded, key 3 = inputs 1, 2, 3 IN INPUT ORDER: 1.6 -> "
             "assigned 1; 1 - 1.9 = -0.9 -> 0; 0 + 1.6 = 1.6 -> 1: result 1.  (Key after key -- 1.6, 1.6, then -1.9 -- would give 1, 2.6 -> 2, 0.1 -> 0.)  CONCAT next to it "
             "joins the same rows in the same order.",
             cols([I32, F64], nullable=False), [[1, 5.5], [2, 1.6], [3, -1.9], [2, 1.6]],
             ["GroupAggregate", ["ProjectNamedAttribute", "col0"], [["SUM", "col1", "s", I64], ["CONCAT", "col1", "c"]], "INPUT", {"max_unique_keys_in_result": 1}],
             [I32, I64, STR], [[1, 5, "5.5"], [2, 1, "1.6,-1.9,1.6"]])
derived_case("Derived_SumOfDoublesIntoInt64NextToDistinctKeepsInputOrder", "supersonic/base/infrastructure/aggregation_operators.h:173-185; " + DST,
             D_SEQ + "Every aggregation of a specification is fed the same rows in the same (input) order -- a DISTINCT aggregate next to the sum changes nothing for "
             "the sum (" + DST + " filters the rows of ITS aggregator only).  Group 1 holds 2.5, -0.75, 0.5 in this order: assigned 2; 2 - 0.75 = 1.25 -> 1; 1 + 0.5 = 1.5 -> 1: "
             "result 1 (folded in VALUE order -- -0.75, 0.5, 2.5, the order a DISTINCT implementation may sort into -- it would be 0, 0, 2: result 2).  COUNT DISTINCT 3, SUM "
             "DISTINCT 2.25.  Group 2 holds 0.5, 0.5: SUM 0, COUNT DISTINCT 1, SUM DISTINCT 0.5.",
             cols([I32, F64], nullable=False), [[1, 2.5], [2, 0.5], [1, -0.75], [1, 0.5], [2, 0.5]],
             ["GroupAggregate", ["ProjectNamedAttribute", "col0"], [["SUM", "col1", "s", I64], ["COUNT_DISTINCT", "col1", "c"], ["SUM_DISTINCT", "col1", "d"]], "INPUT"],
             [I32, I64, U64, F64], [[1, 1, 3, 2.25], [2, 0, 1, 0.5]], ordered=False)
derived_case("Derived_ConcatNextToDistinctKeepsInputOrder", "supersonic/cursor/core/column_aggregator.cc:108-124; " + DST,
             "CONCAT appends the printed values of its result row in input order (column_aggregator.cc:108-124); the DistinctAggregator next to it keeps its own set and "
             "filters only its own aggregator's rows (" + DST + ").  Group 7 holds 9, 3, 9, NULL, 3, 1: CONCAT '9,3,9,3,1' (input order, repeats kept, NULL skipped) next to COUNT "
             "DISTINCT 3 and SUM DISTINCT 13; DISTINCT CONCAT of the same column '9,3,1'.  Group 8 holds NULL only: both strings NULL, COUNT DISTINCT 0, SUM DISTINCT NULL.",
             cols([I32, I32]), [[7, 9], [7, 3], [8, None], [7, 9], [7, None], [7, 3], [7, 1]],
             ["GroupAggregate", ["ProjectNamedAttribute", "col0"],
              [["CONCAT", "col1", "c"], ["COUNT_DISTINCT", "col1", "n"], ["SUM_DISTINCT", "col1", "s"], ["CONCAT_DISTINCT", "col1", "dc"]], "INPUT"],
             [I32, STR, U64, I32, STR], [[7, "9,3,9,3,1", 3, 13, "9,3,1"], [8, None, 0, None, None]], ordered=False)
derived_case("Derived_SumOfDoublesIntoInt64OrderMatters", "supersonic/base/infrastructure/aggregation_operators.h:173-185",
             D_SEQ + "2.75, 0.5, 0.5: assigned 2, 2 + 0.5 = 2.5 -> 2, 2 + 0.5 -> 2: result 2 (the real sum 3.75 would truncate to 3).  NULLs are skipped: "
             "NULL, 7.9, NULL, 0.2 -> assigned 7, 7 + 0.2 = 7.2 -> 7.",
             cols([F64]), [[2.75], [0.5], [0.5]],
             ["ScalarAggregate", [["SUM", "col0", "s", I64]], "INPUT"], [I64], [[2]])

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_tests.json")
    with open(out, "w") as f:
        json.dump(CASES, f, indent=0, separators=(",", ":"))
    print("wrote %d cases to %s" % (len(CASES), out))
