#!/usr/bin/env python3
"""Generates rust/ffi.rs — the `extern "C"` block, `#[repr(C)]` structs and constants a Rust maintainer binds
libthzgpu.so / libthzio.so with — from include/thzgpu.h and include/thzio.h, so that the Rust side cannot
drift from the C ABI: tests/test_abi_exports.py re-runs this generator and compares with the committed file,
and checks header <-> ffi symbol equality.

The headers are written in a regular style (one declaration per statement, typedef'd structs and enums,
no function pointers, no macros with arguments), which is all this parser understands.

    python scripts/gen_rust_ffi.py            # rewrites rust/ffi.rs
    python scripts/gen_rust_ffi.py --check    # exit 1 if rust/ffi.rs is stale
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, "include", "thzgpu.h"), os.path.join(ROOT, "include", "thzio.h")]
OUT = os.path.join(ROOT, "rust", "ffi.rs")

SCALARS = {"int": "c_int", "unsigned": "c_uint", "unsigned int": "c_uint", "long": "c_long", "size_t": "usize", "float": "f32",
           "double": "f64", "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64",
           "uint8_t": "u8", "char": "c_char", "void": "c_void"}
OPAQUE = []          # filled from `typedef struct X X;`
RUST_KEYWORDS = {"type", "in", "ref", "box", "move", "loop", "match", "fn", "mod", "use", "where", "self", "str"}


def camel(name):
    return "".join(p.capitalize() for p in name.split("_"))


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    return src


def rust_type(ctype):
    """'const float *const *' -> '*const *const f32' etc."""
    t = ctype.replace("volatile", " ").strip()
    t = re.sub(r"\s+", " ", t)
    # split pointer levels from the right
    levels = []     # each: True if that pointer level points to const
    while t.endswith("*") or t.endswith("* const") or t.endswith("*const"):
        t = re.sub(r"\*\s*const$", "*", t).rstrip()
        t = t[:-1].rstrip()
        levels.append(None)
    base_const = False
    words = t.split(" ")
    if "const" in words:
        base_const = True
        words = [w for w in words if w != "const"]
    words = [w for w in words if w not in ("struct", "enum")]
    base = " ".join(words)
    if base in SCALARS:
        r = SCALARS[base]
    elif base.startswith("thz_"):
        r = camel(base)
    else:
        raise ValueError(f"unknown C type '{ctype}'")
    # constness of inner pointer levels: reparse with a regex on the original text
    stars = re.findall(r"\*\s*(const)?", ctype.replace("volatile", " "))
    # stars[i] = 'const' if pointer level i (left to right) is itself const; pointee constness of level i is
    # base_const for i = 0, stars[i-1] for i > 0
    out = r
    for i in range(len(stars)):
        pointee_const = base_const if i == 0 else (stars[i - 1] == "const")
        out = ("*const " if pointee_const else "*mut ") + out
    return out


def ident(name):
    return name + "_" if name in RUST_KEYWORDS else name


def parse_params(params):
    params = params.strip()
    if params in ("", "void"):
        return []
    out = []
    for i, p in enumerate(split_top(params)):
        p = re.sub(r"\s+", " ", p.strip())
        m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", p)
        if not m or m.group(2) in SCALARS or m.group(1).strip() == "" or m.group(1).strip() in ("const", "unsigned"):
            ctype, name = p, f"arg{i}"
        else:
            ctype, name = m.group(1).strip(), m.group(2)
        out.append((ident(name), rust_type(ctype)))
    return out


def split_top(s):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            depth += ch in "([{"
            depth -= ch in ")]}"
            cur += ch
    if cur.strip():
        parts.append(cur)
    return parts


def parse_header(path):
    src = strip_comments(open(path).read())
    src = re.sub(r"#\s*(ifndef|ifdef|endif|include|if|else|define\s+\w+\s*$)[^\n]*", " ", src, flags=re.M)
    consts, structs, opaque, funcs = [], [], [], []
    for m in re.finditer(r"#\s*define\s+(\w+)\s+([^\n]+)", src):
        name, val = m.group(1), m.group(2).strip()
        if re.fullmatch(r"[0-9]+u?", val):
            consts.append((name, "u32" if val.endswith("u") else "c_int", val.rstrip("u")))
    src = re.sub(r"#[^\n]*", " ", src)
    src = src.replace('extern "C" {', " ")
    # opaque handles
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", src):
        opaque.append(m.group(2))
    src = re.sub(r"typedef\s+struct\s+\w+\s+\w+\s*;", " ", src)
    # enums (typedef'd or anonymous)
    for m in re.finditer(r"(?:typedef\s+)?enum\s*(\w*)\s*\{(.*?)\}\s*(\w*)\s*;", src, flags=re.S):
        nxt = 0
        for item in split_top(m.group(2)):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, val = (x.strip() for x in item.split("=", 1))
                nxt = int(val, 0)
            else:
                name = item
            consts.append((name, "c_int", str(nxt)))
            nxt += 1
    src = re.sub(r"(?:typedef\s+)?enum\s*\w*\s*\{.*?\}\s*\w*\s*;", " ", src, flags=re.S)
    # structs
    for m in re.finditer(r"typedef\s+struct\s*(\w*)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for stmt in m.group(2).split(";"):
            stmt = re.sub(r"\s+", " ", stmt.strip())
            if not stmt:
                continue
            # 'float base_a, base_b' / 'const float *knots' / 'float *a, *b' / 'float position[3]'
            first, *rest = split_top(stmt)
            mm = re.match(r"^(.*?)(\**\s*)([A-Za-z_]\w*)(\[\d+\])?$", first.strip())
            base = mm.group(1).strip()
            decls = [(mm.group(2).strip(), mm.group(3), mm.group(4))]
            for r in rest:
                r2 = re.match(r"^(\**\s*)([A-Za-z_]\w*)(\[\d+\])?$", r.strip())
                decls.append((r2.group(1).strip(), r2.group(2), r2.group(3)))
            for stars, name, arr in decls:
                rt = rust_type((base + " " + stars).strip())
                if arr:
                    rt = f"[{rt}; {arr[1:-1]}]"
                fields.append((ident(name), rt))
        structs.append((m.group(3), fields))
    src = re.sub(r"typedef\s+struct\s*\w*\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    # functions
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(thz_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3)
        funcs.append((name, None if ret == "void" else rust_type(ret), parse_params(params)))
    return consts, structs, opaque, funcs


def generate():
    lines = ["// rust/ffi.rs — GENERATED by scripts/gen_rust_ffi.py from include/thzgpu.h and include/thzio.h.",
             "// Do not edit: regenerate.  tests/test_abi_exports.py fails when this file and the headers disagree.",
             "//",
             "// The C ABI of libthzgpu.so (MI355X engine for the data_thread recompute path) and libthzio.so (dotTHz",
             "// reader / writer over the HDF5 C API) as a Rust maintainer binds it.  No Rust toolchain exists in the",
             "// image this repository is built in, so this file is unverified by rustc; every symbol in it is exercised",
             "// through the same ABI by the ctypes tests and the C++ host mirror.",
             "#![allow(non_camel_case_types, non_upper_case_globals, dead_code)]",
             "use std::os::raw::{c_char, c_int, c_long, c_uint, c_void};", ""]
    for path in HEADERS:
        consts, structs, opaque, funcs = parse_header(path)
        lib = "thzgpu" if path.endswith("thzgpu.h") else "thzio"
        lines.append(f"// ------------------------------------------------------------------ {os.path.basename(path)}")
        for name in opaque:
            lines.append(f"#[repr(C)] pub struct {camel(name)} {{ _private: [u8; 0] }}")
        lines.append("")
        for name, ty, val in consts:
            lines.append(f"pub const {name}: {ty} = {val};")
        lines.append("")
        for name, fields in structs:
            lines.append("#[repr(C)]")
            lines.append("#[derive(Clone, Copy)]")
            lines.append(f"pub struct {camel(name)} {{")
            for fname, ftype in fields:
                lines.append(f"    pub {fname}: {ftype},")
            lines.append("}")
            lines.append("")
        lines.append(f'#[link(name = "{lib}")]')
        lines.append('extern "C" {')
        for name, ret, params in funcs:
            args = ", ".join(f"{n}: {t}" for n, t in params)
            lines.append(f"    pub fn {name}({args})" + (f" -> {ret};" if ret else ";"))
        lines.append("}")
        lines.append("")
    return "\n".join(lines)


def main():
    text = generate()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            print("rust/ffi.rs is stale: run scripts/gen_rust_ffi.py", file=sys.stderr)
            sys.exit(1)
        return
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    print(f"wrote {OUT}")


if __name__ == "__main__":
    main()
