"""The streaming closed form behind the Levenshtein <= 2 scan (csrc/lev2_stream.inc - the
reference's default metric and threshold, count_well_duplicates.py:200, :258, :282-283) compiled
for the CPU and run against the textbook edit distance: every split into rounds, prefix
aliveness, the dword code path of the interleaved layout."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("lev2") / "lev2_stream_check")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", os.path.join(REPO, "tools", "lev2_stream_check.cpp"),
                           "-o", exe])
    return exe


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_streaming_closed_form_equals_textbook_distance(check, seed):
    out = subprocess.run([check, str(seed), "150000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and " 0 disagreements" in out.stdout, (out.stdout, out.stderr[-2000:])
