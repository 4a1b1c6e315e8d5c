"""The library's environment variables (round 5's verdict, item 6): ONE table of at most 20 documented knobs
(csrc/awfm_knobs.h), read through one function; INTEGRATION.md section 7 lists exactly those."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "avxwindowfmindex_amd", "csrc")


def _table():
    text = open(os.path.join(CSRC, "awfm_knobs.h")).read()
    body = text[text.index("awfmKnobTable[AWFM_KNOB_COUNT] = {"):]
    return re.findall(r'\{"(AWFM_[A-Z_]+)", "', body)


def test_one_table_of_at_most_twenty_knobs_and_one_getenv():
    names = _table()
    assert len(names) == len(set(names)) and 0 < len(names) <= 20
    enum = re.findall(r"^\s+AWFM_KNOB_([A-Z_]+)(?: = 0)?,?$", open(os.path.join(CSRC, "awfm_knobs.h")).read(), re.M)
    assert enum[-1] == "COUNT" and len(enum) - 1 == len(names)
    for e, n in zip(enum, names):  # the enumerators name the entries in the table's order
        assert n.endswith(e), (e, n)
    for path in glob.glob(os.path.join(CSRC, "*")):
        if os.path.isfile(path) and path.endswith((".hip", ".h", ".c", ".cpp")) and not path.endswith("awfm_knobs.h"):
            assert "getenv" not in open(path).read(), f"{os.path.basename(path)} reads the environment by itself"


def test_sources_name_no_variable_outside_the_table():
    names = set(_table())
    for path in glob.glob(os.path.join(CSRC, "*")) + glob.glob(os.path.join(ROOT, "include", "*.h")):
        if not (os.path.isfile(path) and path.endswith((".hip", ".h", ".c", ".cpp"))):
            continue
        for var in set(re.findall(r"\$(AWFM_[A-Z][A-Z_]+[A-Z])", open(path).read())):
            assert var in names, f"{os.path.basename(path)} documents ${var}, which the library does not read"


def test_integration_md_lists_exactly_the_table():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("## 7. Environment variables"):]
    rows = re.findall(r"^\| `(AWFM_[A-Z_]+)` \|", section, re.M)
    assert sorted(rows) == sorted(_table())
