"""Identity of the kernel sources a measurement was taken on: bench.py quotes HBM-traffic figures from a committed
rocprofv3 --pmc summary only when that summary was collected on the same kernel sources as the library it is running."""
import glob
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")


def source_sha16() -> str:
    """sha256 (first 16 hex digits) over the product kernel sources (csrc/*.hip, *.hpp, *.cpp, Makefile; not experiments/)."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(_CSRC, "*.hip")) + glob.glob(os.path.join(_CSRC, "*.hpp")) +
                    glob.glob(os.path.join(_CSRC, "*.cpp")) + [os.path.join(_CSRC, "Makefile")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_sha16())
