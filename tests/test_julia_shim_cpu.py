"""The Julia shim (julia/TotalLeastSquaresHIP.jl) cannot be executed in this image - no Julia.  What CAN be checked without
one: every `ccall` in it names a function include/tlsq.h declares, with an argument-type tuple that matches the header's
prototype and the ctypes binding the tests run (totalleastsquares.jl_amd/_lib.py) argument by argument, and its mirrors of
the option / report structs list the header's fields in the header's order with the header's types."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "julia", "TotalLeastSquaresHIP.jl")
HDR = os.path.join(ROOT, "include", "tlsq.h")


def _strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", "", s, flags=re.S)


def _balanced(s, i):
    """s[i] == '(' -> index just past its matching ')'"""
    depth = 0
    for j in range(i, len(s)):
        if s[j] == "(":
            depth += 1
        elif s[j] == ")":
            depth -= 1
            if depth == 0:
                return j + 1
    raise ValueError("unbalanced")


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def julia_ccalls():
    """[(name or None for an interpolated symbol, return type, [argument types], number of call arguments)]"""
    src = open(JL).read()
    src = re.sub(r"#.*", "", src)
    calls = []
    for m in re.finditer(r"\bccall\(", src):
        end = _balanced(src, m.end() - 1)
        parts = _split_top(src[m.end():end - 1])
        fn = parts[0]
        nm = re.match(r"\(\s*:(\w+)\s*,", fn)
        name = nm.group(1) if nm else None          # `$(QuoteNode(sym))` in the @eval loops: resolved by the caller below
        ret = parts[1]
        types = _split_top(parts[2].strip()[1:-1]) if parts[2].strip() != "()" else []
        types = [t for t in types if t]
        calls.append((name, ret, types, len(parts) - 3, src[max(0, m.start() - 1500):m.start()]))
    return calls


# what a Julia ccall type says about the C argument: (kind, pointee) with kind in {"ptr", "i32", "i64", "f64", "f32", "cstr"}
def julia_kind(t, T=None):
    t = t.strip().replace("$T", T or "T")
    if t in ("Cint", "Int32"):
        return "i32"
    if t in ("Int64", "Clonglong"):
        return "i64"
    if t in ("UInt64",):
        return "u64"
    if t in ("Cdouble", "Float64"):
        return "f64"
    if t in ("Cfloat", "Float32"):
        return "f32"
    if t == "Cstring":
        return "ptr"
    if t.startswith("Ptr{") or t.startswith("Ref{"):
        return "ptr"
    if t == "T":
        return "scalarT"
    raise AssertionError(f"unknown Julia ccall type {t!r}")


def c_kind(t):
    t = t.strip()
    if "*" in t or t in ("tlsq_handle",) or t.startswith("tlsq_on_iter") or t.endswith("_cb"):
        return "ptr"
    t = re.sub(r"\bconst\b", "", t).strip()
    base = t.split()[0] if t else t
    return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "double": "f64", "float": "f32",
            "unsigned": "i32"}[base]


def header_prototypes():
    src = _strip_c_comments(open(HDR).read())
    protos = {}
    for m in re.finditer(r"\b(?:int|void|const char\*|void\*)\s+(tlsq_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        if args in ("", "void"):
            protos[m.group(1)] = []
            continue
        kinds = []
        for a in _split_top(args):
            a = re.sub(r"\[[^\]]*\]", "*", a)                      # `unsigned char id[128]` decays to a pointer
            typ = a.rsplit(None, 1)[0] if not a.rstrip().endswith("*") and " " in a else a
            typ = re.sub(r"\b[a-zA-Z_][a-zA-Z0-9_]*$", "", a) if "*" not in a.split()[-1] else a   # drop the name
            kinds.append(c_kind(typ if typ.strip() else a))
        protos[m.group(1)] = kinds
    return protos


def ctypes_kind(t):
    if t in (C.c_void_p, C.c_char_p) or (isinstance(t, type) and issubclass(t, C._Pointer)):
        return "ptr"
    return {C.c_int: "i32", C.c_int32: "i32", C.c_int64: "i64", C.c_uint64: "u64", C.c_double: "f64", C.c_float: "f32"}[t]


def test_every_ccall_matches_header_and_ctypes():
    from tlsq_amd import _lib as L
    lib = L.load()
    protos = header_prototypes()
    assert len(protos) >= 40
    calls = julia_ccalls()
    assert len(calls) >= 20
    seen = set()
    for name, ret, types, nargs, before in calls:
        # the @eval loops: `for (T, sym) in ((Float64, :tlsq_x_f64), (Float32, :tlsq_x_f32))`
        variants = [(name, None)] if name else [(s, t) for t, s in re.findall(r"\((Float64|Float32),\s*:(tlsq_\w+)\)", before)[-2:]]
        assert variants, "a ccall with an interpolated symbol but no (T, sym) loop in front of it"
        for fname, T in variants:
            assert fname in protos, f"{fname}: ccall in the Julia shim, but include/tlsq.h declares no such function"
            assert len(types) == nargs, f"{fname}: {len(types)} types for {nargs} arguments"
            jk = [julia_kind(t, T) for t in types]
            want = protos[fname]
            assert len(jk) == len(want), f"{fname}: the shim passes {len(jk)} arguments, the header takes {len(want)}"
            suffix_f32 = fname.endswith("_f32")
            for i, (a, b) in enumerate(zip(jk, want)):
                if a == "scalarT":
                    a = "f32" if (T == "Float32" or suffix_f32) else "f64"
                assert a == b, f"{fname}: argument {i + 1} is {types[i]} in the shim, {b} in the header"
            at = getattr(lib, fname).argtypes
            if at is not None:
                ck = [ctypes_kind(t) for t in at]
                assert ck == want, f"{fname}: the ctypes binding {ck} disagrees with the header {want}"
            assert ret in ("Cint", "Cvoid", "Cstring", "Ptr{Cvoid}")
            seen.add(fname)
    # the shim binds every entry point of the reference's path (the tlsq_k_* kernel entries are test-only)
    for must in ("tlsq_create", "tlsq_create_multi", "tlsq_rpca_f64", "tlsq_rpca_f32", "tlsq_lowrankfilter_f64", "tlsq_hankel_f64",
                 "tlsq_unhankel_f64", "tlsq_tls_f64", "tlsq_rtls_f64", "tlsq_rpca_c64_svd", "tlsq_rpca_ga_f64",
                 "tlsq_rtls_batched_f64", "tlsq_rpca_batched_f32"):
        assert must in seen, must


def _julia_struct(name):
    src = re.sub(r"#.*", "", open(JL).read())
    m = re.search(r"mutable struct " + name + r"\b(.*?)\n\s*" + name + r"\(\) = new\(\)", src, flags=re.S)
    assert m, name
    fields = []
    for f in re.split(r"[;\n]", m.group(1)):
        f = f.strip()
        if f:
            nm, ty = f.split("::")
            fields.append((nm.strip(), ty.strip()))
    return fields


def _header_struct(name):
    src = _strip_c_comments(open(HDR).read())
    m = re.search(r"typedef struct " + name + r"\s*\{(.*?)\}\s*" + name + r"\s*;", src, flags=re.S)
    assert m, name
    fields = []
    body = re.sub(r"\(\*\s*(\w+)\s*\)\s*\([^)]*\)", r"FNPTR \1", m.group(1))     # `void (*on_iter)(...)` -> `void FNPTR on_iter`
    for f in body.split(";"):
        f = f.strip()
        if not f:
            continue
        names = [n.strip() for n in f.split(",")]                   # `double ms_total, ms_loop, ...;`
        typ, names[0] = names[0].rsplit(None, 1)
        for nm in names:
            if "FNPTR" in typ or typ.split()[0].endswith("_cb") or "*" in typ or "*" in nm:
                fields.append((nm.replace("*", ""), "ptr"))
            else:
                fields.append((nm, c_kind(typ)))
    return fields


@pytest.mark.parametrize("jl_name,c_name,py_name", [("RpcaOpts", "tlsq_rpca_opts", "RpcaOpts"), ("RpcaInfo", "tlsq_rpca_info", "RpcaInfo"),
                                                    ("GaOpts", "tlsq_ga_opts", "GaOpts")])
def test_struct_mirrors_have_the_headers_fields(jl_name, c_name, py_name):
    from tlsq_amd import _lib as L
    jf = _julia_struct(jl_name)
    hf = _header_struct(c_name)
    assert [n for n, _ in jf] == [n if n != "lambda" else "lambda" for n, _ in hf], (jf, hf)
    for (jn, jt), (hn, hk) in zip(jf, hf):
        jk = "ptr" if jt.startswith("Ptr{") else julia_kind(jt)
        assert jk == hk, f"{jl_name}.{jn}: {jt} in the shim, {hk} in the header"
    pf = getattr(L, py_name)._fields_
    assert [n.rstrip("_") for n, _ in pf] == [n for n, _ in hf]
    for (pn, pt), (hn, hk) in zip(pf, hf):
        pk = "ptr" if (pt in (C.c_void_p,) or not hasattr(pt, "_type_") or isinstance(getattr(pt, "_type_", None), type)) and pt not in (
            C.c_int32, C.c_int64, C.c_uint64, C.c_double, C.c_float, C.c_int) else ctypes_kind(pt)
        assert pk == hk, f"{py_name}.{pn}: ctypes {pt} vs header {hk}"
