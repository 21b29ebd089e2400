"""The evidence set under profiles/ must be what the documents say it is: no file that is really a Python traceback (round 3 committed two), every JSON
parses, and every profiles/<file> that DESIGN.md, README.md, INTEGRATION.md, profiles/README.md, bench.py or benchmarks/subrecords.py cites exists."""
import json
import os
import re

from conftest import ROOT

PROFILES = os.path.join(ROOT, "profiles")
CITING = ["DESIGN.md", "README.md", "INTEGRATION.md", os.path.join("profiles", "README.md"), os.path.join("profiles", "HISTORY.md"), "bench.py",
          os.path.join("benchmarks", "subrecords.py")]


def _text(path):
    with open(path, errors="replace") as f:
        return f.read()


def test_no_profile_file_is_a_traceback_and_every_json_parses():
    bad = []
    for name in sorted(os.listdir(PROFILES)):
        path = os.path.join(PROFILES, name)
        if not os.path.isfile(path):
            continue
        t = _text(path)
        if "Traceback (most recent call last)" in t or "AttributeError:" in t or "undefined symbol" in t:
            bad.append(name)
        if name.endswith(".json"):
            for line in ([t] if t.lstrip().startswith(("{", "[")) and "\n{" not in t.strip() else [x for x in t.splitlines() if x.strip()]):
                json.loads(line)
    assert not bad, f"profiles/ files that hold an error message instead of a measurement: {bad}"


def test_every_cited_profile_exists():
    have = set(os.listdir(PROFILES))
    missing = []
    for doc in CITING:
        path = os.path.join(ROOT, doc)
        if not os.path.exists(path):
            continue
        for m in re.finditer(r"profiles/([A-Za-z0-9_.\-]+)", _text(path)):
            name = m.group(1).rstrip(".")
            if not name or name in ("README.md", "HISTORY.md") or name.endswith(("_", "-")):
                continue                                    # a prefix written with a wildcard or placeholder after it (profiles/r04_*)
            if "." not in name:
                continue                                    # a directory-like mention or a stem followed by a placeholder
            if name not in have:
                missing.append((doc, name))
    assert not missing, f"cited but absent under profiles/: {missing}"
