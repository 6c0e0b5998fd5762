#!/usr/bin/env python3
"""Ties the evidence under profiles/ to the code it was measured on.

    python tools/profile_manifest.py --write     here, before `gpurun ... tools/collect_profiles.sh`: refuses (exit 2) unless the product
                                                 sources are committed (no diff against HEAD in CODE_PATHS) and the built library is newer
                                                 than every source; writes tools/.profile_manifest.json (git-ignored, travels with the
                                                 snapshot): the code commit + sha256 of the library and of every source file
    python tools/profile_manifest.py --check     on the GPU box (no .git there), first thing in collect_profiles.sh / collect_traffic.sh:
                                                 refuses unless the files on the box hash to the manifest - the profiles are of THAT commit
    python tools/profile_manifest.py --stamp     here, in refresh_profiles.py: prints the manifest's code commit; refuses unless it still is
                                                 the last commit that touched CODE_PATHS and those paths are still clean

"Code commit" = the last commit that touched the product (kernels, C ABI, host package, bench.py): commits that only add
profiles or documentation do not move it."""
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MANIFEST = os.path.join(ROOT, "tools", ".profile_manifest.json")
CODE_PATHS = ["applied-image-processing_amd", "include", "bench.py"]
LIB = os.path.join(ROOT, "applied-image-processing_amd", "libadain_hip.so")


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def sources():
    out = []
    for pat in ("applied-image-processing_amd/csrc/*", "applied-image-processing_amd/*.py", "applied-image-processing_amd/AdaIN/*.py", "include/*.h",
                "bench.py"):
        out += sorted(glob.glob(os.path.join(ROOT, pat)))
    return [p for p in out if os.path.isfile(p)]


def git(*a):
    return subprocess.run(["git"] + list(a), cwd=ROOT, capture_output=True, text=True).stdout.strip()


def code_commit():
    return git("log", "-1", "--format=%h", "--abbrev=12", "--", *CODE_PATHS)


def dirty():
    return git("status", "--porcelain", "--", *CODE_PATHS)


def hashes():
    return {os.path.relpath(p, ROOT): sha(p) for p in sources() + [LIB]}


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else ""
    if mode == "--write":
        d = dirty()
        if d:
            sys.exit("profile_manifest: uncommitted changes in the product sources - commit first, the profiles must name a commit:\n" + d)
        stale = [os.path.relpath(p, ROOT) for p in sources() if p.endswith((".hip", ".h")) and os.path.getmtime(p) > os.path.getmtime(LIB)]
        if stale:
            sys.exit(f"profile_manifest: {LIB} is older than {stale}: rebuild (python __graft_entry__.py) first")
        m = {"code_commit": code_commit(), "head": git("rev-parse", "--short=12", "HEAD"), "files": hashes()}
        json.dump(m, open(MANIFEST, "w"), indent=1)
        print("profile manifest written for code commit", m["code_commit"])
    elif mode == "--check":
        if not os.path.exists(MANIFEST):
            sys.exit("profile_manifest: no tools/.profile_manifest.json - run `python tools/profile_manifest.py --write` before gpurun")
        m = json.load(open(MANIFEST))
        now = hashes()
        bad = [k for k, v in m["files"].items() if now.get(k) != v] + [k for k in now if k not in m["files"]]
        if bad:
            sys.exit(f"profile_manifest: these files differ from the manifest of commit {m['code_commit']}: {bad}")
        print("profile manifest ok: code commit", m["code_commit"])
    elif mode == "--stamp":
        if not os.path.exists(MANIFEST):
            sys.exit("profile_manifest: no manifest")
        m = json.load(open(MANIFEST))
        if dirty() or code_commit() != m["code_commit"]:
            sys.exit(f"profile_manifest: the profiles were collected at code commit {m['code_commit']} but the product sources are now at "
                     f"{code_commit()}{' + uncommitted changes' if dirty() else ''}: collect again")
        print(m["code_commit"])
    else:
        sys.exit(__doc__)


if __name__ == "__main__":
    main()
