"""What libmvmc_hip.so was built from: lib/BUILD_INFO.json, written by the build (csrc/Makefile runs this file after linking) and
checked by _cabi.load() -- a library whose kernel sources are not the tree's (a stale .so that travelled to the GPU box, a half-finished
rebuild) is refused instead of being measured.  No package imports: the Makefile runs it as a script."""
import hashlib
import json
import os
import socket
import subprocess
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INFO_PATH = os.path.join(_HERE, "lib", "BUILD_INFO.json")


def sources_sha(csrc: str = CSRC) -> str:
    """Hash of the HIP sources the library is built from (the stamp of every record under profiles/ too)."""
    h = hashlib.sha256()
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


def read():
    """BUILD_INFO.json as a dict, or None."""
    try:
        with open(INFO_PATH) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def write() -> dict:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    try:
        ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True, timeout=60).stdout.strip().splitlines()
        ver = next((l for l in ver if "HIP version" in l), ver[0] if ver else "?")
    except (OSError, subprocess.SubprocessError):
        ver = "?"
    info = {"kernel_sources_sha": sources_sha(), "hipcc": ver, "arch": os.environ.get("ARCH", "gfx950"), "host": socket.gethostname(),
            "built_at_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "extra_flags": os.environ.get("EXTRA", "")}
    os.makedirs(os.path.dirname(INFO_PATH), exist_ok=True)
    with open(INFO_PATH, "w") as f:
        json.dump(info, f, indent=1)
    return info


if __name__ == "__main__":
    print("BUILD_INFO", json.dumps(write()))
