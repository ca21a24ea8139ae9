"""The test suite checks itself: a test function that shares its name with a later one in the same module never runs
(Python keeps the last binding, pytest collects only that one) -- a hole round 3 had in tests/test_gpu_parity.py.
"""
import ast
import glob
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def _duplicates(path):
    tree = ast.parse(open(path).read(), path)
    dups = []

    def scan(body, where):
        seen = {}
        for node in body:
            if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                if node.name in seen and (node.name.startswith("test") or node.name.startswith("Test")):
                    dups.append(f"{where}{node.name} (lines {seen[node.name]} and {node.lineno})")
                seen[node.name] = node.lineno
                if isinstance(node, ast.ClassDef):
                    scan(node.body, f"{where}{node.name}.")
    scan(tree.body, os.path.basename(path) + "::")
    return dups


def test_no_test_function_is_shadowed_by_a_later_one_of_the_same_name():
    files = sorted(glob.glob(os.path.join(HERE, "*.py")))
    assert len(files) >= 6
    dups = [d for f in files for d in _duplicates(f)]
    assert not dups, "shadowed tests (only the last definition of a name is collected): " + "; ".join(dups)


def test_every_gpu_test_module_marks_its_tests():
    """a module whose name says gpu must carry the gpu marker (module-level pytestmark or per test), so the CPU run skips it"""
    for f in sorted(glob.glob(os.path.join(HERE, "test_gpu_*.py"))):
        src = open(f).read()
        assert "pytest.mark.gpu" in src, f
