"""The evidence the documents cite is in the tree: every `profiles/<file>` named in DESIGN.md / README.md / INTEGRATION.md / BASELINE.md exists, profiles/README.md has a
line for every file of profiles/, and the directory stays small enough to read (VERDICT of round 5, item 8: <= 150 files)."""
import os
import re

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def _cited(doc):
    text = open(os.path.join(ROOT, doc), encoding="utf-8").read()
    # `profiles/name.ext` with a plain file name (no glob, no placeholder)
    return sorted(set(m for m in re.findall(r"profiles/([A-Za-z0-9_.\-]+\.(?:txt|json|csv|md))", text)))


def test_cited_profile_files_exist():
    missing = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md", "BASELINE.md"):
        for name in _cited(doc):
            if not os.path.exists(os.path.join(ROOT, "profiles", name)):
                missing.append(f"{doc}: profiles/{name}")
    assert not missing, missing


def test_profiles_directory_is_listed_and_small():
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if os.path.isfile(os.path.join(ROOT, "profiles", f)))
    assert len(files) <= 150, len(files)
    listing = open(os.path.join(ROOT, "profiles", "README.md"), encoding="utf-8").read()
    unlisted = [f for f in files if f != "README.md" and f"`{f}`" not in listing]
    assert not unlisted, unlisted
