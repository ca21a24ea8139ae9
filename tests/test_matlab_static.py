"""Static checks of matlab/*.m (no MATLAB or Octave on the image: the classes cannot be executed here).  A small lexer of the MATLAB subset the
package uses -- comments, '...' continuation, single- / double-quoted strings against the transpose operator, `end` as an index inside
brackets -- verifies that every block keyword has its `end`, that brackets balance statement by statement, that the subclasses implement the
base class' abstract methods, and that every command string the classes hand to the gateway exists in matlab/nd_dwt_hip_mex.c."""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MDIR = os.path.join(ROOT, "matlab")
OPENERS = {"classdef", "properties", "methods", "events", "enumeration", "function", "if", "for", "parfor", "while", "switch", "try"}
TOKEN = re.compile(r"\s+|\.\.\.|[A-Za-z_]\w*|\d+\.?\d*(?:[eE][-+]?\d+)?|\.\*|\./|\.\^|\.'|==|~=|<=|>=|&&|\|\||.")


def logical_lines(text):
    """[(first line number, [tokens])]: comments stripped, continuation lines joined, string literals as single tokens.  A single quote is
    the transpose operator when it follows an operand directly (identifier, number, closing bracket, another quote, `.`), else it opens a string."""
    out, cur, start = [], [], 1
    for no, line in enumerate(text.split("\n"), 1):
        toks, i = [], 0
        while i < len(line):
            ch = line[i]
            if ch == "%":
                break
            if ch == '"' or (ch == "'" and not (i > 0 and re.match(r"[\w\)\]\}'.]", line[i - 1]))):
                j = i + 1
                while True:
                    assert j < len(line), f"line {no}: unterminated string"
                    if line[j] == ch:
                        if j + 1 < len(line) and line[j + 1] == ch:      # a doubled quote inside the string
                            j += 2
                            continue
                        break
                    j += 1
                toks.append(line[i:j + 1])
                i = j + 1
                continue
            m = TOKEN.match(line, i)
            i = m.end()
            if not m.group(0).isspace():
                toks.append(m.group(0))
        if not cur:
            start = no
        if toks and toks[-1] == "...":
            cur += toks[:-1]
            continue
        cur += toks
        if cur:
            out.append((start, cur))
        cur = []
    assert not cur, "file ends in a continuation"
    return out


def check_blocks(path):
    text = open(path).read()
    stack = []
    is_class = False
    for no, toks in logical_lines(text):
        depth, at_start = 0, True
        for t in toks:
            if t in "([{":
                depth += 1
            elif t in ")]}":
                depth -= 1
                assert depth >= 0, f"{path}:{no}: closing bracket without an opening one"
            elif depth == 0:
                if t in OPENERS and at_start:
                    stack.append((t, no))
                    is_class = is_class or t == "classdef"
                elif t == "end":
                    assert stack, f"{path}:{no}: `end` without a block"
                    stack.pop()
            at_start = depth == 0 and t in {",", ";"}          # a block keyword opens a block at the start of a statement only
        assert depth == 0, f"{path}:{no}: unbalanced brackets in `{' '.join(toks)[:80]}`"
    if not is_class:                                   # function / script files may leave their functions without `end`
        stack = [s for s in stack if s[0] != "function"]
    assert not stack, f"{path}: blocks without `end`: {stack}"


M_FILES = sorted(glob.glob(os.path.join(MDIR, "*.m")))


@pytest.mark.parametrize("path", M_FILES, ids=[os.path.basename(p) for p in M_FILES])
def test_m_file_blocks_and_brackets_balance(path):
    check_blocks(path)


def test_the_lexer_rejects_broken_files(tmp_path):
    for body in ("function y = f(x)\n  if x\n    y = 1;\nend\n  y = a(end;\n", "classdef c\n  methods\n    function f(o)\n    end\n  end\n",
                 "x = [1 2 3;\n"):
        p = tmp_path / "bad.m"
        p.write_text(body)
        with pytest.raises(AssertionError):
            check_blocks(str(p))
    good = tmp_path / "good.m"
    good.write_text("classdef c\n  methods\n    function y = f(o, x)\n      y = x(end)'; s = 'it''s'; z = {x.' 'a'};  % 'comment\n      if y, y = [y ...\n 1]; end\n    end\n  end\nend\n")
    check_blocks(str(good))


def test_subclasses_implement_the_abstract_methods_of_the_base_class():
    base = open(os.path.join(MDIR, "nd_dwt_hip_base.m")).read()
    block = re.search(r"methods\s*\(Abstract[^)]*\)(.*?)\n\s*end", base, re.S).group(1)
    abstract = re.findall(r"=\s*(\w+)\s*\(", block)
    assert set(abstract) == {"ndim_", "size_error_", "wname_error_", "level_from_bands_"}
    for d in (1, 2, 3, 4):
        src = open(os.path.join(MDIR, f"nd_dwt_{d}D_hip.m")).read()
        assert re.search(rf"classdef\s+nd_dwt_{d}D_hip\s*<\s*nd_dwt_hip_base", src)
        for name in abstract:
            assert re.search(rf"function\s+[^\n]*\b{name}\s*\(", src), (d, name)
        assert re.search(rf"function\s+obj\s*=\s*nd_dwt_{d}D_hip\s*\(", src)          # the constructor


def test_gateway_commands_used_by_the_classes_exist_in_the_mex_file():
    csrc = open(os.path.join(MDIR, "nd_dwt_hip_mex.c")).read()
    known = set(re.findall(r'strcmp\s*\(\s*\w+\s*,\s*"(\w+)"\s*\)', csrc))
    used = set()
    for path in M_FILES:
        used |= set(re.findall(r"nd_dwt_hip_mex\(\s*'(\w+)'", open(path).read()))
    assert used and used <= known, (used - known)
    assert {"dec_keep", "rec_handle", "shrink", "fetch", "release", "denoise"} <= known
