#!/usr/bin/env python3
"""Generate the Rust `extern "C"` block for include/denovo_kmer.h (every exported function, 1:1).

    python tools/gen_rust_extern.py            # prints the block INTEGRATION.md embeds

There is no Rust toolchain in the build image, so the block cannot be compiled here; generating it from the
header (and tests/test_abi.py re-generating it and comparing it with INTEGRATION.md, name by name and argument by
argument) is the guard that keeps the documented binding and the header from drifting apart."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "denovo_kmer.h")

SCALARS = {"dk_status": "i32", "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64",
           "uint8_t": "u8", "float": "f32", "char": "c_char", "void": "c_void", "size_t": "usize"}


def declarations(text=None):
    """-> [(name, return C type, [(C type, arg name)])] for every function the header declares"""
    text = text if text is not None else open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    out = []
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(dk_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        if "typedef" in ret or not ret:
            continue
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a)
                ctype, pname = mm.group(1).strip(), mm.group(2)
                params.append((" ".join(ctype.replace("*", " * ").split()), pname))
        out.append((name, " ".join(ret.replace("*", " * ").split()), params))
    return out


def rust_type(ctype):
    toks = ctype.split()
    # pointer levels from the right: "const T * const *" etc.
    base_const = False
    i = 0
    if toks[i] == "const":
        base_const = True
        i += 1
    base = toks[i]
    i += 1
    rt = SCALARS.get(base, base)               # opaque handles and structs keep their C names
    const_here = base_const
    while i < len(toks):
        if toks[i] == "*":
            rt = ("*const " if const_here else "*mut ") + rt
            const_here = False
        elif toks[i] == "const":
            const_here = True
        i += 1
    return rt


def rust_block():
    lines = ['extern "C" {']
    for name, ret, params in declarations():
        args = ", ".join("%s: %s" % (("type_" if p == "type" else p), rust_type(t)) for t, p in params)
        r = "" if ret == "void" else " -> " + rust_type(ret)
        lines.append("    pub fn %s(%s)%s;" % (name, args, r))
    lines.append("}")
    return "\n".join(lines)


if __name__ == "__main__":
    sys.stdout.write(rust_block() + "\n")
