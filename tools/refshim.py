"""Generator-side import shim for the read-only reference tree (/root/reference).

Used ONLY by tools/gen_golden.py in the build container to produce the fixtures
under tests/golden/.  It never travels into the product path: nothing under
diffphycon_amd/, bench.py or the gpu tests imports it, and /root/reference does not
exist on the GPU box.

What it does:
  * installs `sys.modules` stubs for packages the reference imports but the image
    lacks.  Two of them carry arithmetic and are restated from the upstream
    packages' published algorithm:
      - einops_exts.rearrange_many  (einops-exts 0.0.4)  = map(rearrange)
      - rotary_embedding_torch.RotaryEmbedding (0.8.4)   = interleaved-pair RoPE
        (PARITY UNPINNED: the wheel is not available offline)
  * makes vendored phi (PhiFlow 0.x) importable under NumPy 2 / Python 3.10 by
    AST-rewriting `a[list-with-slices]` to `a[tuple(...)]` at import time.
Nothing in /root/reference is modified or copied.
"""
import ast
import builtins
import collections
import collections.abc
import importlib.abc
import importlib.machinery
import sys
import types

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


def install():
    sys.dont_write_bytecode = True
    import accelerate  # noqa: F401  must be imported before the tensorboardX stub exists
    import einops
    import torch
    from torch import nn

    # ---- einops_exts 0.0.4
    def rearrange_many(tensors, pattern, **kw):
        return map(lambda t: einops.rearrange(t, pattern, **kw), tensors)

    def check_shape(t, pattern, **kw):
        return einops.rearrange(t, f"{pattern} -> {pattern}", **kw)

    _stub("einops_exts", rearrange_many=rearrange_many, check_shape=check_shape)

    # ---- rotary_embedding_torch 0.8.4 (assumed semantics)
    def rotate_half(x):
        x = einops.rearrange(x, "... (d r) -> ... d r", r=2)
        x1, x2 = x.unbind(dim=-1)
        x = torch.stack((-x2, x1), dim=-1)
        return einops.rearrange(x, "... d r -> ... (d r)")

    class RotaryEmbedding(nn.Module):
        def __init__(self, dim, theta=10000):
            super().__init__()
            self.freqs = nn.Parameter(
                1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim)), requires_grad=False
            )

        def rotate_queries_or_keys(self, t, seq_dim=-2):
            n = t.shape[seq_dim]
            pos = torch.arange(n, device=t.device).type_as(self.freqs)
            freqs = torch.einsum("i,j->ij", pos, self.freqs)
            freqs = einops.repeat(freqs, "n f -> n (f r)", r=2)
            return t * freqs.cos() + rotate_half(t) * freqs.sin()

    _stub("rotary_embedding_torch", RotaryEmbedding=RotaryEmbedding)

    # ---- no-op stubs
    class _EMA:
        def __init__(self, *a, **k):
            pass

    _stub("ema_pytorch", EMA=_EMA)
    _stub("IPython", embed=lambda *a, **k: None)
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms")
    tv.utils = _stub("torchvision.utils")
    _stub("tensorboardX", SummaryWriter=object)
    _stub("termcolor", colored=lambda s, *a, **k: s)
    ds = _stub("deepsnap")
    ds.batch = _stub("deepsnap.batch", Batch=object)
    _stub("h5py")
    _stub("imageio")
    tg = _stub("torch_geometric")
    tg.data = _stub("torch_geometric.data", Dataset=object, Data=object)
    _stub("pdb_stub")

    # ---- phi under NumPy 2 / Python 3.10
    collections.Iterable = collections.abc.Iterable

    def _fixidx(i):
        if isinstance(i, list) and any(isinstance(e, (slice, type(Ellipsis))) or e is None for e in i):
            return tuple(i)
        return i

    builtins._fixidx = _fixidx

    class T(ast.NodeTransformer):
        def visit_Subscript(self, node):
            self.generic_visit(node)
            if isinstance(node.slice, (ast.Constant, ast.Slice, ast.Tuple)):
                return node
            node.slice = ast.Call(func=ast.Name(id="_fixidx", ctx=ast.Load()), args=[node.slice], keywords=[])
            return node

    class Loader(importlib.machinery.SourceFileLoader):
        def source_to_code(self, data, path, *, _optimize=-1):
            tree = ast.parse(data, filename=path)
            tree = T().visit(tree)
            ast.fix_missing_locations(tree)
            return compile(tree, path, "exec", dont_inherit=True, optimize=_optimize)

    class Finder(importlib.abc.MetaPathFinder):
        def find_spec(self, fullname, path, target=None):
            if fullname != "phi" and not fullname.startswith("phi."):
                return None
            spec = importlib.machinery.PathFinder.find_spec(fullname, [REF] if path is None else path)
            if spec is None or spec.origin is None or not spec.origin.endswith(".py"):
                return spec
            spec.loader = Loader(fullname, spec.origin)
            return spec

    sys.meta_path.insert(0, Finder())
    if REF not in sys.path:
        sys.path.insert(0, REF)
