#!/usr/bin/env python3
"""Regenerates the downcall-handle block of java/ai/prophetizo/wavelet/gpu/VwGpuNative.java from include/vwgpu.h.

Every prototype of the header becomes one `static final MethodHandle <SYMBOL IN UPPER CASE>` whose
FunctionDescriptor lists one layout per C parameter: any pointer -> ADDRESS, int32_t / uint32_t -> JAVA_INT,
int64_t / size_t -> JAVA_LONG, double -> JAVA_DOUBLE.  The block between the BEGIN/END GENERATED HANDLES markers
is replaced; everything else in the file is hand-written.  tests/test_java_binding.py checks the result with a
parser of its own.

usage: python tools/gen_java_handles.py [--check]
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vwgpu.h")
JAVA = os.path.join(ROOT, "java", "ai", "prophetizo", "wavelet", "gpu", "VwGpuNative.java")
BEGIN = "    // ---- BEGIN GENERATED HANDLES (tools/gen_java_handles.py; do not edit by hand) ----\n"
END = "    // ---- END GENERATED HANDLES ----\n"


def layout(ctype):
    t = " ".join(ctype.replace("const", " ").split())
    if "*" in t:
        return "ADDRESS"
    return {"int32_t": "JAVA_INT", "uint32_t": "JAVA_INT", "int64_t": "JAVA_LONG", "size_t": "JAVA_LONG",
            "double": "JAVA_DOUBLE"}[t]


def prototypes():
    s = open(HEADER).read()
    s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    s = re.sub(r"//[^\n]*", "", s)
    s = re.sub(r"^\s*#.*$", "", s, flags=re.M)
    out = []
    for ret, name, args in re.findall(r"([A-Za-z_][\w\s\*]*?)\b(vw_\w+)\s*\(([^)]*)\)\s*;", s):
        params = []
        if args.strip() != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                m = re.match(r"(.*?)(\w+)$", a)
                params.append((m.group(1).strip(), m.group(2)))
        out.append((ret.strip(), name, params))
    return out


def block():
    lines = [BEGIN]
    for ret, name, params in prototypes():
        sig = ", ".join(p[1] for p in params)
        lines.append("    // %s %s(%s)\n" % (" ".join(ret.split()), name, sig))
        lay = [layout(ret)] + [layout(p[0]) for p in params]
        text = "    static final MethodHandle %s = h(\"%s\", FunctionDescriptor.of(%s));\n" % (name.upper(), name, ", ".join(lay))
        if len(text) > 128:  # wrap
            head = "    static final MethodHandle %s = h(\"%s\",\n        FunctionDescriptor.of(" % (name.upper(), name)
            cur, rows = head, []
            for i, l in enumerate(lay):
                piece = l + (", " if i + 1 < len(lay) else "));\n")
                if len(cur.split("\n")[-1]) + len(piece) > 124:
                    cur = cur.rstrip(" ") + "\n            "
                cur += piece
            text = cur
        lines.append(text)
    lines.append(END)
    return "".join(lines)


def main():
    src = open(JAVA).read()
    i, j = src.index(BEGIN), src.index(END) + len(END)
    new = src[:i] + block() + src[j:]
    if "--check" in sys.argv:
        sys.exit(0 if new == src else 1)
    open(JAVA, "w").write(new)


if __name__ == "__main__":
    main()
