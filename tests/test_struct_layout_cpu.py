"""Every ctypes mirror of a struct of include/*.h against the C compiler's own layout of that struct: same member names in the same
order, same offsets, same size.  The member names are read from the header text, the offsets from a probe program compiled here
with gcc -- so a member appended to a header struct (an ABI change) that is not appended to its Python mirror fails this test
instead of shifting whatever the library reads behind it."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


def _mirrors():
    from quilt_amd.gibbs_nipt import GibbsOpts
    from quilt_amd.impute import BamRangeIo, ImputeNipt, ImputeParams, ImputeRareCommon, SampleSource, SampleView
    from quilt_amd.io import BamOpts
    from quilt_amd.native import FullpassOpts, PanelDesc
    return {
        "qa_impute_params_t": ImputeParams, "qa_impute_nipt_t": ImputeNipt, "qa_impute_rare_common_t": ImputeRareCommon,
        "qa_sample_view_t": SampleView, "qa_sample_source_t": SampleSource, "qa_gibbs_opts_t": GibbsOpts,
        "qa_fullpass_opts_t": FullpassOpts, "qa_panel_desc_t": PanelDesc, "qa_bam_opts_t": BamOpts, "qa_bam_range_io_t": BamRangeIo,
    }


def _header_text():
    text = ""
    for name in ("quilt_amd.h", "quilt_amd_io.h"):
        text += open(os.path.join(INC, name)).read() + "\n"
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _members(struct_name, text):
    """Member names of `typedef struct { ... } <struct_name>;` in declaration order."""
    m = re.search(r"typedef\s+struct\s*\{([^{}]*)\}\s*" + re.escape(struct_name) + r"\s*;", text, flags=re.S)
    assert m, f"{struct_name} not found in include/*.h"
    names = []
    for decl in m.group(1).split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        fp = re.search(r"\(\s*\*\s*(\w+)\s*\)\s*\(", decl)   # a function pointer member
        if fp:
            names.append(fp.group(1))
            continue
        for part in decl.split(","):                         # `const int32_t *a, *b` / `int32_t x, y` / `double z[4]`
            ident = re.search(r"(\w+)\s*(\[[^\]]*\])?\s*$", part.strip())
            assert ident, decl
            names.append(ident.group(1))
    return names


def test_python_mirrors_have_the_headers_layout(tmp_path):
    mirrors = _mirrors()
    text = _header_text()
    members = {s: _members(s, text) for s in mirrors}
    lines = ["#include <stddef.h>", "#include <stdio.h>", '#include "quilt_amd.h"', '#include "quilt_amd_io.h"', "int main(void) {"]
    for s, names in members.items():
        lines.append(f'    printf("{s} size %zu\\n", sizeof({s}));')
        for n in names:
            lines.append(f'    printf("{s} {n} %zu\\n", offsetof({s}, {n}));')
    lines += ["    return 0;", "}"]
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-std=c11", "-I", INC, str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    c_layout = {}
    for line in out.splitlines():
        s, n, v = line.split()
        c_layout.setdefault(s, {})[n] = int(v)
    for s, cls in mirrors.items():
        py_names = [f[0] for f in cls._fields_]
        assert py_names == members[s], f"{cls.__name__} vs {s}: member names / order"
        for n in py_names:
            assert getattr(cls, n).offset == c_layout[s][n], f"{cls.__name__}.{n}: offset {getattr(cls, n).offset} vs C {c_layout[s][n]}"
        assert C.sizeof(cls) == c_layout[s]["size"], f"{cls.__name__}: size {C.sizeof(cls)} vs C {c_layout[s]['size']}"


def test_the_member_parser_reads_the_shapes_the_headers_use():
    text = "typedef struct { const int32_t *a, *b; int32_t n; void (*cb)(void *ctx, int32_t x); double v[4]; qa_x_t inner; } t_t;"
    assert _members("t_t", text) == ["a", "b", "n", "cb", "v", "inner"]
    with pytest.raises(AssertionError):
        _members("missing_t", text)
