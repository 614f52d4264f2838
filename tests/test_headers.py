"""Header tokeniser / per-field delta coder of the C++ shim (fqcomp28_amd/csrc/headers.hpp,
SURVEY.md 8(f) row 3) against the reference's known answers (test/headers_test.cpp:12-33), its
round-trip properties (:38-116) and the Python restatement oracle/headers_oracle.py.  CPU only."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import headers_oracle as HO  # noqa: E402


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    """headers_tool over headers.hpp, built with AddressSanitizer and UBSan when g++ has their runtime here (every
    test below then doubles as a memory / undefined-behaviour check of the tokeniser and the field coder: a report
    ends the tool with a code no test expects), plainly otherwise."""
    exe = str(tmp_path_factory.mktemp("hdr") / "headers_tool")
    src = os.path.join(ROOT, "tests", "cpp", "headers_tool.cpp")
    san = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Werror", "-fsanitize=address,undefined",
                          "-fno-sanitize-recover=undefined", "-o", exe, src], capture_output=True)
    if san.returncode != 0:
        subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-o", exe, src], check=True)
    os.environ.setdefault("ASAN_OPTIONS", "detect_leaks=0")
    return exe


def _fmt(tool, header):
    r = subprocess.run([tool, "fmt", header], capture_output=True, text=True)
    if r.returncode == 3:
        return None
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.splitlines()
    return lines[0].split()[1:], [int(x) for x in lines[1].split()[1:]]


def test_header_format_known_answers(tool):
    """the two examples of the reference's TEST_CASE("HeaderFormatSpec")"""
    t, s = _fmt(tool, "@SRR22543904.1 1 length=150")
    assert t == list("SNNSN") and bytes(s) == b".  ="
    t, s = _fmt(tool, "@SRR065390.1000 HWUSI-EAS687_61DAJ:8:1:1174:9158 length=100")
    assert t == list("SNSSSNNNNSN") and bytes(s) == b". -_:::: ="
    for h in ("@SRR22543904.1 1 length=150", "@a", "@12", "@x..y", "@SYN.7 7 length=150"):
        assert _fmt(tool, h) == tuple(map(list, HO.format_from_header(h.encode())))


def test_header_ending_in_separator_is_refused(tool):
    assert _fmt(tool, "@SRR1.1 length=150/") is None  # src/headers.cpp:64-66 throws invalid_argument
    with pytest.raises(ValueError):
        HO.format_from_header(b"@SRR1.1 length=150/")


def _code(tool, path):
    r = subprocess.run([tool, "code", path], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-400:] + r.stderr
    lines = r.stdout.splitlines()
    assert lines[-1].startswith("roundtrip ok")
    out = []
    for ln in lines[:-1]:
        p = ln.split()
        out.append((p[2], *[b"" if x == "-" else bytes.fromhex(x) for x in p[3:6]]))
    return out, int(lines[-1].split()[2])


def _headers_of(path):
    with open(path, "rb") as f:
        return f.read().split(b"\n")[0::4][: None]


@pytest.mark.parametrize("name", ["SRR065390_1_first5.fastq", "SRR065390_sub_1.fastq", "SRR065390_sub_2.fastq",
                                  "without_ns.fastq"])
def test_streams_match_the_oracle_on_the_reference_fixtures(tool, golden_dir, name):
    path = os.path.join(golden_dir, name)
    hdrs = [h for h in _headers_of(path) if h]
    got, n = _code(tool, path)
    assert n == len(hdrs)
    types, seps, streams = HO.encode_headers(hdrs)
    assert [g[0] for g in got] == types
    for g, s, t in zip(got, streams, types):
        assert g[1] == bytes(s.flags) and g[2] == bytes(s.content) and g[3] == bytes(s.lengths)
        if t == "S":  # sizes the reference's tests check (test/headers_test.cpp:50, 98-100)
            assert len(g[1]) == len(hdrs)
        else:
            assert g[1] == b"" and g[3] == b"" and len(g[2]) == 4 * len(hdrs)
    assert HO.decode_headers(len(hdrs), hdrs[0], streams) == hdrs


def test_synthetic_headers_strings_numbers_and_wraparound(tool, tmp_path):
    """string values that repeat and change (the reference's {store,load}String case), numbers that
    go down and jump by more than 2^31 (the difference wraps), fields shorter than their separator
    search (one character), a separator as a field's first character"""
    rng = np.random.default_rng(7)
    names = ["EAS687", "EAS688", "EAS688", "TIOBDUREN", "TIOBDUREN", "BEZNOGIM", "B", "B"]
    nums = [5, 4, 2147483647, -2147483648 + 2**32 - 2**32, 0, 33808546, 7, 7]
    hdrs = []
    for i in range(400):
        a = names[i % len(names)] if i < 16 else names[int(rng.integers(len(names)))]
        b = nums[i % len(nums)] if i < 16 else int(rng.integers(1, 33808546))
        if b < 0:
            b = 2147483647 - i  # from_chars parses no sign-less negative here; keep values non-negative
        hdrs.append(b"@%s.%d %d length=%d" % (a.encode(), i + 1, b, 50 + i % 251))
    path = tmp_path / "syn.fastq"
    path.write_bytes(b"".join(h + b"\nACGT\n+\nIIII\n" for h in hdrs))
    got, n = _code(tool, str(path))
    assert n == len(hdrs)
    types, seps, streams = HO.encode_headers(hdrs)
    assert types == list("SNNSN")
    for g, s in zip(got, streams):
        assert (g[1], g[2], g[3]) == (bytes(s.flags), bytes(s.content), bytes(s.lengths))
    assert HO.decode_headers(len(hdrs), hdrs[0], streams) == hdrs


def test_numeric_field_that_is_not_a_number_is_refused(tool, tmp_path):
    path = tmp_path / "bad.fastq"
    path.write_bytes(b"@r.1\nA\n+\nI\n@r.x2\nA\n+\nI\n")
    r = subprocess.run([tool, "code", str(path)], capture_output=True, text=True)
    assert r.returncode == 3 and "not an int32" in r.stdout
    path.write_bytes(b"@r.1\nA\n+\nI\n@r.99999999999\nA\n+\nI\n")
    r = subprocess.run([tool, "code", str(path)], capture_output=True, text=True)
    assert r.returncode == 3
