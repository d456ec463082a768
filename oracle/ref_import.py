"""Import the UNMODIFIED reference package from /root/reference (only exists in the build container).

``torch_rechub/__init__.py:3-6`` asks importlib.metadata for an installed distribution, so a stub
``torch_rechub-0.8.0.dist-info/METADATA`` is placed in a temp dir ahead of the reference on sys.path.
Nothing is copied from or written to /root/reference.
"""
import os
import sys
import tempfile

REFERENCE_ROOT = os.environ.get("RECHUB_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "torch_rechub"))


def import_reference():
    """Returns the imported ``torch_rechub`` module (raises RuntimeError when the checkout is absent)."""
    if "torch_rechub" in sys.modules:
        return sys.modules["torch_rechub"]
    if not available():
        raise RuntimeError(f"reference checkout not found at {REFERENCE_ROOT}")
    stub = os.path.join(tempfile.gettempdir(), "rechub_ref_stub")
    info = os.path.join(stub, "torch_rechub-0.8.0.dist-info")
    os.makedirs(info, exist_ok=True)
    meta = os.path.join(info, "METADATA")
    if not os.path.exists(meta):
        with open(meta, "w") as f:
            f.write("Metadata-Version: 2.1\nName: torch-rechub\nVersion: 0.8.0\n")
    for p in (REFERENCE_ROOT, stub):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch_rechub  # noqa: F401
    return torch_rechub
