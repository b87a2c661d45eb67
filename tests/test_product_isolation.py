"""The oracle is test infrastructure: nothing under legommenders_amd/ (the product) may import, call or
execute anything under oracle/, and there is no CPU compute fallback to route through."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "legommenders_amd")


def _py_files():
    for d, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(".py"):
                yield os.path.join(d, f)


def test_product_never_imports_the_oracle():
    bad = []
    for path in _py_files():
        tree = ast.parse(open(path).read())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            if any(n == "oracle" or n.startswith("oracle.") for n in names):
                bad.append(path)
        if "lego_oracle" in open(path).read():
            bad.append(path)
    assert not bad, bad


def test_oracle_header_declares_itself_test_infrastructure():
    head = open(os.path.join(ROOT, "oracle", "lego_oracle.py")).read(1200)
    assert "TEST INFRASTRUCTURE ONLY" in head


def test_bench_uses_oracle_only_for_cpu_baseline():
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name != "cpu_baseline":
            body = ast.get_source_segment(src, node) or ""
            assert "lego_oracle" not in body and "from oracle" not in body, node.name
