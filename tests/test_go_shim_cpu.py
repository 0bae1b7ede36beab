"""The Go shim under go/ cannot be compiled here (no Go toolchain), so at least its C side is checked: every C
symbol, constant and struct field the cgo file uses must exist in include/pqdecode.h."""
import os
import re

import pytest

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


# ---- the templates against the reference's own declarations (skipped where the reference is not mounted)
REF = "/root/reference"


def _go_struct_fields(src, name):
    """{field: type} of `type <name> struct { ... }` in Go source text."""
    m = re.search(r"type %s struct \{(.*?)\n\}" % re.escape(name), src, re.S)
    assert m, name
    out = {}
    for line in m.group(1).splitlines():
        line = line.split("//")[0].strip()
        if not line:
            continue
        parts = line.split()
        if len(parts) >= 2:
            for f in parts[0].split(","):
                out[f.strip()] = parts[1]
    return out


def _go_func_params(src, name):
    """Parameter types of `func name(...)` / `func (r T) name(...)` in Go source text, in order."""
    m = re.search(r"func (?:\([^)]*\) )?%s\(([^)]*)\)" % re.escape(name), src)
    assert m, name
    types = []
    pending = 0
    for part in [p.strip() for p in m.group(1).split(",") if p.strip()]:
        bits = part.split()
        if len(bits) == 1:
            pending += 1  # `a, b T`: the type follows with a later name
        else:
            types += [bits[-1]] * (pending + 1)
            pending = 0
    return types


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference not mounted")
def test_templates_use_the_reference_types_as_declared():
    tpl = _read("go", "gen", "template_gpu.go")
    ref_parquet = open(os.path.join(REF, "parquet.go")).read()
    ref_schema = open(os.path.join(REF, "schema", "parquet.go")).read()
    ref_tpl = open(os.path.join(REF, "cmd", "parquetgen", "gen", "template.go")).read()
    page = _go_struct_fields(ref_parquet, "Page")
    # every pg.X the templates read is a field of parquet.Page
    used = set(re.findall(r"\bpg\.(\w+)", tpl))
    assert used and used <= set(page), (used, page)
    # pg.Codec goes to gpuCodec, whose parameter is the field's declared type (an int64 enum, not a string)
    assert page["Codec"] == "sch.CompressionCodec"
    assert re.search(r"type CompressionCodec int64", ref_schema)
    assert _go_func_params(tpl, "gpuCodec") == [page["Codec"]]
    assert re.search(r"func gpuCodec\([^)]*\) \(int32, error\)", tpl)
    for const in re.findall(r"sch\.(CompressionCodec_\w+)", tpl):
        assert re.search(r"\b%s CompressionCodec = \d" % const, ref_schema), const
    assert '"unsupported column chunk codec: %s"' in tpl  # the words of fields.go's default arm
    assert 'unsupported column chunk codec: %s' in open(os.path.join(REF, "fields.go")).read()
    # pg.N and pg.Size are compared with an int counter in walkPageHeaders
    assert page["N"] == "int" and page["Size"] == "int"
    assert re.search(r"var nRead int\b", tpl)
    # ph.X: fields of the thrift PageHeader / DataPageHeader the walk reads
    ph = _go_struct_fields(ref_schema, "PageHeader")
    dph = _go_struct_fields(ref_schema, "DataPageHeader")
    for f in set(re.findall(r"\bph\.(\w+)", tpl)):
        assert f in ph, f
    for f in set(re.findall(r"\bph\.DataPageHeader\.(\w+)", tpl)):
        assert f in dph, f
    assert ph["CompressedPageSize"] == "int32" and dph["NumValues"] == "int32"
    # parquet.X: functions and types of the reference package that the templates call
    for name in set(re.findall(r"\bparquet\.([A-Z]\w*)", tpl)):
        assert re.search(r"^(?:func|type) (?:\([^)]*\) )?%s\b" % name, ref_parquet, re.M), name
    assert _go_func_params(ref_parquet, "PageHeader") == ["io.Reader"]
    # p.X: fields of the generated ParquetReader that readRowGroup touches (the GPU ones are added by the support template)
    reader = _go_struct_fields(ref_tpl, "ParquetReader")
    added = {"device", "gpu", "file", "gpuInit"}
    for f in set(re.findall(r"\bp\.(\w+)", tpl)) - added:
        assert f in reader or re.search(r"func \(p \*ParquetReader\) %s\(" % f, ref_tpl + tpl), f
    # the struct the row group columns come from
    assert re.search(r"func \(r RowGroup\) Columns\(\)", ref_parquet) or re.search(r"Columns\(\)", ref_parquet)


def test_bench_file_tool_named_by_the_go_bench_exists():
    go = _read("go", "bench", "read_bench_test.go")
    for tool in re.findall(r"tools/(\w+\.py)", go):
        assert os.path.exists(os.path.join(ROOT, "tools", tool)), tool
