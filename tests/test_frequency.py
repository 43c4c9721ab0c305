"""CPU: scope row f4 — per-site frequency aggregation vs outputs of the reference script itself
(tests/golden/make_frequency_golden.py ran /root/reference/scripts/call_modification_frequency.py)."""
import json
import os

import pytest

from deepsignal_amd import call_modification_frequency as cmf

GOLD = os.path.join(os.path.dirname(__file__), "golden", "frequency_golden.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_matches_reference_script(gold, idx, tmp_path):
    case = gold["cases"][idx]
    inp, out = str(tmp_path / "calls.tsv"), str(tmp_path / "freq.tsv")
    with open(inp, "w") as f:
        f.write("\n".join(gold["input_rows"]) + "\n")
    assert cmf.main(["-i", inp, "-o", out] + case["flags"]) == 0
    assert open(out).read().splitlines() == case["output"]       # byte-identical, unsorted order included


def test_directory_input_and_uid(gold, tmp_path):
    d = tmp_path / "calls"
    d.mkdir()
    half = len(gold["input_rows"]) // 2
    (d / "part1.calls.tsv").write_text("\n".join(gold["input_rows"][:half]) + "\n")
    (d / "part2.calls.tsv").write_text("\n".join(gold["input_rows"][half:]) + "\n")
    (d / "notes.txt").write_text("ignored\n")
    files = cmf.collect_input_files([str(d)], "calls.tsv")
    assert len(files) == 2
    stats = cmf.calculate_mods_frequency(sorted(files))
    assert sum(s.coverage for s in stats.values()) == len(gold["input_rows"])
    with pytest.raises(ValueError):
        cmf.collect_input_files([str(tmp_path / "nope")])
