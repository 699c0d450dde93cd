"""Fingerprint of the kernel sources, so that a counter profile can say which code it measured."""
import glob
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def kernel_source_sha():
    """sha256 (first 16 hex digits) over the HIP/C++ sources of libmpc_amd.so, in name order."""
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(_CSRC, "*"))):
        if path.endswith((".hip", ".h", ".cpp")):
            h.update(os.path.basename(path).encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]
