"""The reference-side binding, read as text and held to the C headers.

Julia is not installed here or on the GPU box, so integration/julia/HybridNLPHIP.jl (and the Julia snippets of
INTEGRATION.md) cannot be executed.  What CAN be checked without Julia, and is checked here:

  * every `ccall((:sym, LIB), Ret, (argtypes...), args...)` names a function that include/qln_evaluator.h (or
    include/qln_multi.h) declares, with the same number of arguments, the same return class, and for every argument the
    same class -- pointer-to-double / pointer-to-int32 / pointer-to-int64 / pointer-to-struct X / opaque pointer /
    pointer-to-pointer / int32 / int64 / uint32 / double -- and the same number of VALUES passed as types declared;
  * the field lists of QlnModel, QlnBatchDesc and QlnSolveOptions equal the header's structs: same names, same order,
    same widths (so a field added to the header breaks this test instead of silently corrupting the Julia call);
  * the binding defines a `solve` METHOD for its own type -- the reference's is `solve(x0, prob::HybridNLP; ...)`
    (src/moi.jl:46) on a concrete struct (src/nlp.jl:13) and cannot dispatch on HybridNLPHIP -- with the reference's
    keyword names and defaults, and every MOI callback the reference defines (src/moi.jl:1-33) has a method here;
  * qln_variable_bounds (what that method asks the library for) gives solve()'s bounds incl. quirk Q6 (no GPU needed).
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "integration", "julia", "HybridNLPHIP.jl")


# ------------------------------------------------------------------------------------------------ C side
def _strip_c(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def _c_arg_class(arg, structs, opaque):
    a = " ".join(arg.replace("const", " ").split())
    a = re.sub(r"\[\d*\]", "*", a)                      # `int32_t cinds[14]` is a pointer
    stars = a.count("*")
    base = a.replace("*", " ").split()
    ty = base[0] if base[0] != "struct" else base[1]
    if ty == "unsigned":
        ty = "unsigned " + base[1]
    scalar = {"double": "f64", "float": "f32", "int32_t": "i32", "int": "i32", "int64_t": "i64", "uint32_t": "u32",
              "uint64_t": "u64", "unsigned char": "u8", "char": "char", "void": "void", "size_t": "u64"}
    if ty in scalar:
        cls = scalar[ty]
    elif ty in structs:
        cls = "struct " + ty
    elif ty in opaque:
        cls = "void"                                    # an opaque handle type: only ever passed by pointer
    else:
        raise AssertionError(f"unknown C type in {arg!r}")
    return "ptr " * stars + cls if stars else cls


def _parse_header(name):
    src = _strip_c(open(os.path.join(ROOT, "include", name)).read())
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            ty, names = decl.rsplit(" ", 1)[0], decl.rsplit(" ", 1)[1]
            # `double g, mb, mf;` and `uint64_t pcg_state[2], pcg_inc[2];`
            head = decl.split(",")[0]
            ty = head.rsplit(" ", 1)[0]
            for nm in [head.rsplit(" ", 1)[1]] + [n.strip() for n in decl.split(",")[1:]]:
                fields.append((nm, ty))
        structs[m.group(3)] = fields
    opaque = set(re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", src))
    funcs = {}
    for m in re.finditer(r"(?:^|\n)\s*((?:const\s+)?[\w]+\s*\*?)\s*\b(qln_\w+)\s*\(([^;{}]*?)\)\s*;", src):
        ret, fn, args = m.group(1), m.group(2), m.group(3)
        args = [] if args.strip() in ("", "void") else [a.strip() for a in args.split(",")]
        funcs[fn] = (ret, args)
    return structs, opaque, funcs


def _all_headers():
    structs, opaque, funcs = {}, set(), {}
    for h in ("qln_evaluator.h", "qln_multi.h"):
        s, o, f = _parse_header(h)
        structs.update(s), opaque.update(o), funcs.update(f)
    classes = {}
    for fn, (ret, args) in funcs.items():
        classes[fn] = (_c_arg_class(ret + " x" if "*" not in ret else ret.replace("*", "* x"), structs, opaque),
                       [_c_arg_class(a, structs, opaque) for a in args])
    return structs, classes


# ------------------------------------------------------------------------------------------------ Julia side
JL_STRUCT_OF = {"QlnModel": "qln_model", "QlnBatchDesc": "qln_batch_desc", "QlnSolveOptions": "qln_solve_options"}
JL_SCALAR = {"Cdouble": "f64", "Float64": "f64", "Cfloat": "f32", "Float32": "f32", "Cint": "i32", "Int32": "i32", "Int64": "i64",
             "Clonglong": "i64", "UInt32": "u32", "Cuint": "u32", "UInt64": "u64", "UInt8": "u8", "Cvoid": "void", "Cchar": "char"}


def _jl_type_class(t):
    t = t.strip()
    m = re.fullmatch(r"(?:Ptr|Ref)\{(.+)\}", t)
    if m:
        return "ptr " + _jl_type_class(m.group(1))
    if t == "Cstring":
        return "ptr char"
    if t in JL_SCALAR:
        return JL_SCALAR[t]
    if t in JL_STRUCT_OF:
        return "struct " + JL_STRUCT_OF[t]
    raise AssertionError(f"unknown Julia type {t!r}")


def _split_top(s):
    """split on commas that are not inside (), [] or {}"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


def _balanced(text, start):
    """text[start] == '(' -> index one past its matching ')'"""
    depth = 0
    for i in range(start, len(text)):
        if text[i] == "(":
            depth += 1
        elif text[i] == ")":
            depth -= 1
            if depth == 0:
                return i + 1
    raise AssertionError("unbalanced parentheses")


def _jl_code(text):
    return "\n".join(line.split("#")[0] if '"' not in line.split("#")[0] or line.split("#")[0].count('"') % 2 == 0 else line
                     for line in text.splitlines())


def _ccalls(text):
    text = _jl_code(text)
    out = []
    for m in re.finditer(r"\bccall\s*\(", text):
        end = _balanced(text, m.end() - 1)
        parts = _split_top(text[m.end(): end - 1])
        sym = re.fullmatch(r"\(\s*:(\w+)\s*,\s*(\w+)\s*\)", parts[0])
        assert sym, f"ccall target not of the form (:sym, LIB): {parts[0]!r}"
        argt = parts[2].strip()
        assert argt.startswith("(") and argt.endswith(")"), parts[2]
        types = _split_top(argt[1:-1])
        out.append(dict(sym=sym.group(1), lib=sym.group(2), ret=parts[1], types=types, values=parts[3:]))
    return out


def _jl_structs(text):
    text = _jl_code(text)
    out = {}
    for m in re.finditer(r"(?:^|\n)\s*struct\s+(\w+)\s*[;\n](.*?)\bend\b", text, flags=re.S):
        fields = re.findall(r"(\w+)::((?:Ptr|Ref)\{[^}]+\}|\w+)", m.group(2))
        out[m.group(1)] = fields
    return out


def _check_ccalls(calls, classes, where):
    assert calls, f"no ccall found in {where}"
    for c in calls:
        assert c["sym"] in classes, f"{where}: ccall names {c['sym']}, which no header declares"
        ret_c, args_c = classes[c["sym"]]
        assert _jl_type_class(c["ret"]) == ret_c, (where, c["sym"], "return", c["ret"], ret_c)
        got = [_jl_type_class(t) for t in c["types"]]
        assert len(got) == len(args_c), f"{where}: {c['sym']} takes {len(args_c)} arguments, the ccall declares {len(got)}"
        for i, (g, w) in enumerate(zip(got, args_c)):
            if w == "ptr void" and re.fullmatch(r"ptr (u8|void)", g):
                continue  # an untyped byte buffer (`void* id`): a Ptr{UInt8} is the same thing to the ABI
            assert g == w, f"{where}: {c['sym']} argument {i + 1}: Julia {c['types'][i]} ({g}) vs C ({w})"
        assert len(c["values"]) == len(got), f"{where}: {c['sym']}: {len(got)} types declared, {len(c['values'])} values passed"


C_WIDTH = {"double": 8, "float": 4, "int32_t": 4, "int": 4, "int64_t": 8, "uint32_t": 4, "uint64_t": 8}


def test_every_ccall_of_the_julia_file_matches_the_header():
    structs, classes = _all_headers()
    calls = _ccalls(open(JL).read())
    _check_ccalls(calls, classes, "HybridNLPHIP.jl")
    used = {c["sym"] for c in calls}
    # the seven MOI callbacks + construction + the two solves bind exactly these entry points
    for need in ("qln_create", "qln_destroy", "qln_problem_dims", "qln_constraint_bounds", "qln_eval_objective_host",
                 "qln_eval_objective_gradient_host", "qln_eval_constraint_host", "qln_eval_constraint_jacobian_host",
                 "qln_eval_constraint_jacobian_dense_host", "qln_jacobian_structure", "qln_last_error",
                 "qln_variable_bounds", "qln_solve_host", "qln_solve_default_options"):
        assert need in used, need


def test_julia_snippets_of_the_integration_guide_match_the_headers():
    structs, classes = _all_headers()
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```julia\n(.*?)```", md, flags=re.S)
    calls = [c for b in blocks for c in _ccalls(b)]
    _check_ccalls(calls, classes, "INTEGRATION.md")
    assert any(c["sym"].startswith("qln_multi_") for c in calls) and any(c["sym"].startswith("qln_comm_") for c in calls)


def test_julia_struct_mirrors_equal_the_header_structs_field_by_field():
    structs, _ = _all_headers()
    jl = _jl_structs(open(JL).read())
    for jname, cname in JL_STRUCT_OF.items():
        assert jname in jl, jname
        cf, jf = structs[cname], jl[jname]
        assert [n for n, _ in jf] == [re.sub(r"\[\d+\]", "", n) for n, _ in cf], f"{jname}: field names / order differ from {cname}"
        for (jn, jt), (cn, ct) in zip(jf, cf):
            ct = " ".join(ct.replace("const", " ").split())
            if "*" in ct or cn.endswith("]"):
                want = "ptr"
            elif ct in C_WIDTH:
                want = {"double": "f64", "float": "f32", "int32_t": "i32", "int": "i32", "int64_t": "i64", "uint32_t": "u32",
                        "uint64_t": "u64"}[ct]
            else:
                want = "struct " + ct
            got = _jl_type_class(jt)
            got = "ptr" if got.startswith("ptr ") else got
            assert got == want, f"{jname}.{jn}: Julia {jt} vs C `{ct} {cn}`"
    # and the Python mirror the GPU tests execute has the same three layouts (ctypes computes C's natural alignment)
    from quadruped_landing_amd import _lib
    for jname, py in (("QlnModel", _lib.QlnModel), ("QlnBatchDesc", _lib.QlnBatchDesc), ("QlnSolveOptions", _lib.QlnSolveOptions)):
        assert [n for n, _ in jl[jname]] == [n for n, _ in py._fields_]
        for (jn, jt), (pn, pt) in zip(jl[jname], py._fields_):
            cls = _jl_type_class(jt)
            width = 8 if cls.startswith("ptr ") else {"f64": 8, "i32": 4, "i64": 8, "u32": 4}.get(cls) or C.sizeof(pt)
            assert C.sizeof(pt) == width, (jname, jn)


def test_the_binding_defines_solve_for_its_own_type_with_the_references_keywords():
    """src/moi.jl:46 is `solve(x0, prob::HybridNLP; tol=1.0e-6, c_tol=1.0e-6, max_iter=2000)`: a method for the concrete
    reference type.  The binding must add its own method (or the documented usage line is a MethodError)."""
    src = _jl_code(open(JL).read())
    m = re.search(r"function\s+solve\s*\(\s*x0\s*,\s*prob::HybridNLPHIP\s*;([^)]*)\)", src)
    assert m, "no solve(x0, prob::HybridNLPHIP; ...) method in the binding"
    kw = dict((k.strip(), float(v)) for k, v in (p.split("=") for p in m.group(1).split(",")))
    assert kw == {"tol": 1.0e-6, "c_tol": 1.0e-6, "max_iter": 2000.0}
    body = src[m.end():]
    for needle in ('"max_iter"', '"tol"', '"constr_viol_tol"', "MOI.NLPBlockData", "MOI.NLPBoundsPair", "prob.lb", "prob.ub",
                   "Ipopt.Optimizer()", "MOI.LessThan", "MOI.GreaterThan", "MOI.VariablePrimalStart", "MOI.MIN_SENSE",
                   "MOI.optimize!", "MOI.VariablePrimal", ":qln_variable_bounds"):
        assert needle in body, needle
    # the evaluator handed to Ipopt is the argument, never a global (quirk Q4 of the reference, src/moi.jl:22)
    assert re.search(r"MOI\.NLPBlockData\([^\n]*\bprob\b", body)
    # every MOI method the reference defines on HybridNLP (src/moi.jl:1-33) exists for HybridNLPHIP
    for meth in ("eval_objective", "eval_objective_gradient", "eval_constraint", "eval_constraint_jacobian",
                 "features_available", "initialize", "jacobian_structure"):
        assert re.search(rf"MOI\.{meth}\s*\(\s*\w+::HybridNLPHIP", src), meth
    for fn in ("num_primals", "num_duals"):
        assert re.search(rf"{fn}\s*\(\s*\w+::HybridNLPHIP\s*\)", src), fn
    # INTEGRATION.md no longer claims the reference's own method is reused "unchanged"
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "src/moi.jl:46, unchanged" not in md and "MethodError" in md


@pytest.mark.parametrize("N", [2, 3, 40, 61])
def test_variable_bounds_from_the_library_are_solves_bounds(N):
    """qln_variable_bounds (no handle, no GPU) against the Python restatement of src/moi.jl:51-67 that KA6 pins
    (tests/test_oracle_known_answers.py: the iteration-0 objective 1.8380701 comes out only with quirk Q6's indices)."""
    from quadruped_landing_amd import _lib, nlp as NL

    L = _lib.lib()
    n = 20 * N - 5
    xl, xu = np.empty(n), np.empty(n)
    _lib.check(L.qln_variable_bounds(N, None, xl.ctypes.data, xu.ctypes.data))
    wl, wu = NL.variable_bounds(N)
    assert np.array_equal(xl, wl) and np.array_equal(xu, wu)
    if N == 61:
        only_lower = np.isfinite(xl) & ~np.isfinite(xu)
        both = np.isfinite(xl) & np.isfinite(xu)
        assert only_lower.sum() == 120 and both.sum() == 121   # the Ipopt header of the shipped run, src/main.ipynb:222-223
    opt = _lib.QlnSolveOptions()
    _lib.check(L.qln_solve_default_options(C.byref(opt)))
    opt.q6_bounds = 0
    _lib.check(L.qln_variable_bounds(N, C.byref(opt), xl.ctypes.data, xu.ctypes.data))
    assert np.isfinite(xl).sum() == N + (N - 1)
    assert L.qln_variable_bounds(1, None, xl.ctypes.data, xu.ctypes.data) == _lib.QLN_ERR_INVALID_ARGUMENT
    assert L.qln_variable_bounds(N, None, None, xu.ctypes.data) == _lib.QLN_ERR_INVALID_ARGUMENT
