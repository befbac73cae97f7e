"""INTEGRATION.md §B shows a maintainer the ctypes binding for one entry point.  A document that is not run drifts (it
still described ABI v5 with a 6-field RatAttnParams while the library read a 7th field).  Here the snippet is EXTRACTED FROM THE
DOCUMENT and executed: its struct layouts are compared with include/rat_hip.h and with the plugin's own binding (CPU), and its
`intra_attention` is called on a reference-shaped module and compared with a torch restatement of PreNorm(Attention)(x) + x (GPU)."""
import ctypes
import os
import re
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "www24-rat_amd"))


def _snippet():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- snippet:B:begin -->\s*```python\n(.*?)```\s*<!-- snippet:B:end -->", text, re.S)
    assert m, "INTEGRATION.md lost its executable snippet markers"
    return m.group(1)


def _run_snippet():
    from rat_amd._lib import DEFAULT_LIB
    if not os.path.exists(DEFAULT_LIB):
        import importlib.util
        spec = importlib.util.spec_from_file_location("rat_build", os.path.join(ROOT, "www24-rat_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()
    os.environ.setdefault("RAT_HIP_LIBRARY", DEFAULT_LIB)
    ns = {}
    exec(compile(_snippet(), "INTEGRATION.md#B", "exec"), ns)
    return ns


def _header_struct_fields(name):
    header = open(os.path.join(ROOT, "include", "rat_hip.h")).read()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), header, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(",")
        first = names[0].split()[-1]
        fields += [first.lstrip("*")] + [n.strip().lstrip("*") for n in names[1:]]
    return fields


def test_doc_snippet_structs_match_the_header_and_the_plugin_binding():
    ns = _run_snippet()
    from rat_amd import _lib
    assert ns["lib"].rat_version() == _lib.ABI_VERSION
    for name in ("RatSeqMap", "RatAttnParams"):
        doc, own = ns[name], getattr(_lib, name)
        assert [f[0] for f in doc._fields_] == [f[0] for f in own._fields_] == _header_struct_fields(name), name
        assert [ctypes.sizeof(f[1]) for f in doc._fields_] == [ctypes.sizeof(f[1]) for f in own._fields_], name
        assert ctypes.sizeof(doc) == ctypes.sizeof(own)
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "ABI version %d: %d `extern" % (_lib.ABI_VERSION, len(_lib.EXPORTED_SYMBOLS)) in text, "INTEGRATION.md §B quotes a stale ABI"
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert "%d entry points (ABI v%d)" % (len(_lib.EXPORTED_SYMBOLS), _lib.ABI_VERSION) in design, "DESIGN.md §1 quotes a stale ABI"


@pytest.mark.gpu
def test_doc_snippet_call_matches_prenorm_attention_plus_residual():
    ns = _run_snippet()
    torch.manual_seed(0)
    dev = "cuda:0"
    B, T, S, d, heads, dh = 3, 4, 6, 16, 2, 8
    inner = heads * dh
    norm = torch.nn.LayerNorm(d).to(dev)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.2, 0.2)
    to_qkv = torch.nn.Linear(d, 3 * inner, bias=False).to(dev)
    to_out = torch.nn.Sequential(torch.nn.Linear(inner, d), torch.nn.Dropout(0.0)).to(dev)
    fn = types.SimpleNamespace(heads=heads, to_qkv=to_qkv, to_out=to_out)                 # the attributes of RAT_m2.py:176-190's Attention
    blk = types.SimpleNamespace(intra_attention=types.SimpleNamespace(norm=norm, fn=fn))
    x = torch.randn(B, T, S, d, device=dev)
    y = ns["intra_attention"](x, blk)
    torch.cuda.synchronize()
    with torch.no_grad():                                                                 # RAT_m2.py:192-202 on 'b t n d -> (b t) n d'
        xf = x.reshape(B * T, S, d)
        q, k, v = to_qkv(norm(xf)).chunk(3, dim=-1)
        sp = lambda t: t.reshape(B * T, S, heads, dh).permute(0, 2, 1, 3)                 # noqa: E731
        att = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * dh ** -0.5, dim=-1)
        o = (att @ sp(v)).permute(0, 2, 1, 3).reshape(B * T, S, inner)
        want = (to_out(o) + xf).reshape(B, T, S, d)
    assert float((y - want).abs().max()) < 2e-5
