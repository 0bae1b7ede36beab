"""The Go shim under go/ cannot be compiled here (no Go toolchain), so at least its C side is checked: every C
symbol, constant and struct field the cgo file uses must exist in include/pqdecode.h."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(*p):
    return open(os.path.join(ROOT, *p)).read()


def test_every_c_name_used_by_the_go_shim_is_declared_in_the_header():
    hdr = _read("include", "pqdecode.h")
    go = _read("go", "gpu", "pqdecode.go")
    used = set(re.findall(r"\bC\.((?:pqd|PQD)_\w+)", go))
    assert len(used) > 25, used
    for name in sorted(used):
        assert re.search(r"\b%s\b" % re.escape(name), hdr), "go/gpu/pqdecode.go uses C.%s, which include/pqdecode.h does not declare" % name


def test_struct_literals_of_the_go_shim_use_header_fields():
    hdr = _read("include", "pqdecode.h")
    go = _read("go", "gpu", "pqdecode.go")
    for struct in ("pqd_chunk_desc", "pqd_page_desc"):
        body = re.search(r"typedef struct \{([^}]*)\} %s;" % struct, hdr).group(1)
        fields = set(re.findall(r"\b(\w+);", body))
        for lit in re.findall(r"C\.%s\{(.*?)\}" % struct, go, re.S):
            for f in re.findall(r"(\w+):", lit):
                assert f in fields, (struct, f)
    info = re.search(r"typedef struct \{([^}]*)\} pqd_chunk_info;", hdr).group(1)
    for f in re.findall(r"\bci\.([a-z_]+)\b", go):  # the C struct's fields are lower case, the Go Info's are not
        assert re.search(r"\b%s\b" % f, info), f


def test_templates_name_only_shim_functions_that_exist():
    go = _read("go", "gpu", "pqdecode.go")
    tpl = _read("go", "gen", "template_gpu.go")
    exported = set(re.findall(r"^func (?:\([^)]*\) )?([A-Z]\w*)", go, re.M)) | set(re.findall(r"^\s+(\w+)\s+= (?:int32|Output)\(", go, re.M)) \
        | set(re.findall(r"^type (\w+)", go, re.M))
    for name in set(re.findall(r"\bgpu\.([A-Z]\w*)", tpl)):
        assert name in exported, "template_gpu.go uses gpu.%s, which go/gpu/pqdecode.go does not define" % name
    for m in set(re.findall(r"\bres\.(\w+)\(", tpl)):
        assert re.search(r"func \(r \*Result\) %s\(" % m, go), m
