"""The drop-in entry point's command line (reference inference_streaming_longva_v2.py:935-975): what it accepts and what it refuses up front."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = ["--model_name", "m", "--video_dir", "v", "--memory_basic_dir", "mem", "--save_file", "s.json", "--annotations", "a.json", "--language", "en"]


def _entry():
    spec = importlib.util.spec_from_file_location("sc_entry", os.path.join(ROOT, "inference_streaming_longva_v2.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _args(m, extra):
    import argparse
    try:
        return m.parse_args(BASE + extra)
    except SystemExit as e:
        if e.code == 2 and "--num_beams" not in " ".join(extra):          # BASE does not name every required flag of this build: find out which
            raise AssertionError("BASE is missing a required flag") from e
        raise


def test_beam_sampling_is_refused_before_anything_is_loaded_and_beam_search_is_accepted(capsys):
    """the reference forwards --num_beams to HF generate (:254).  Deterministic beam search is built (streamchat_amd/beam.py:
    --num_beams N --temperature 0); beam SAMPLING (the reference's default temperature 0.2 with beams) is not, and says so at parse time (exit
    code 2, one line on stderr) instead of a NotImplementedError out of the first answer."""
    m = _entry()
    assert _args(m, []).num_beams == 1 and _args(m, ["--num_beams", "1"]).num_beams == 1
    assert _args(m, ["--num_beams", "3", "--temperature", "0"]).num_beams == 3
    with pytest.raises(SystemExit) as e:
        _args(m, ["--num_beams", "3"])
    assert e.value.code == 2
    assert "beam sampling is not implemented" in capsys.readouterr().err
