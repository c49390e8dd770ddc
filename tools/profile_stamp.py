#!/usr/bin/env python3
"""What the committed profiles were taken at: a hash of the kernel sources (dv-pari_amd/csrc, include/) that bench.py can recompute on
any box -- the GPU box has no .git -- and the git commit, when there is one.  `python tools/profile_stamp.py rNN` writes
profiles/rNN_profile_stamp.json (called by tools/digest_profiles.py); bench.py compares the stamp with the tree it runs from."""
import glob
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha16(root=ROOT):
    h = hashlib.sha256()
    files = []
    for pat in ("dv-pari_amd/csrc/*.hip", "dv-pari_amd/csrc/*.cuh", "dv-pari_amd/csrc/*.h", "dv-pari_amd/csrc/*.cpp", "include/*.h"):
        files += glob.glob(os.path.join(root, pat))
    for f in sorted(files):
        h.update(os.path.relpath(f, root).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def git_head(root=ROOT):
    try:
        return subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def write(tag, taken_at_sha16=None):
    d = {"tag": tag, "source_sha16": taken_at_sha16 or source_sha16(), "git_head_when_digested": git_head(),
         "digested_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
         "note": "source_sha16 = sha256 over dv-pari_amd/csrc/*.{hip,cuh,h,cpp} and include/*.h (names and bytes, sorted) of the tree the profiled "
                 "library was built from (tools/refresh_profiles.sh records it on the GPU box next to the raw profiles); bench.py prints its own "
                 "tree's value beside it (`profiles.sources_match`)"}
    json.dump(d, open(os.path.join(ROOT, "profiles", f"{tag}_profile_stamp.json"), "w"), indent=1)
    return d


if __name__ == "__main__":
    if len(sys.argv) > 1:
        print(write(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None))
    else:
        print(source_sha16())
