"""Convert the Planetoid ``ind.<dataset>.*`` files into the ``.npz`` containers
shipped under ``gcn_amd/data/`` (plain numpy arrays, loadable with
``allow_pickle=False``).  The source files are decoded by
``gcn_amd.planetoid.decode``, which executes nothing from them.

usage: python tools/make_planetoid_npz.py <dir with ind.* files> [out_dir]
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gcn_amd import planetoid  # noqa: E402


def main():
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(
        os.path.dirname(os.path.abspath(__file__)), "..", "gcn_amd", "data")
    for ds in ("cora", "citeseer", "pubmed"):
        parts = planetoid.read_ind_files(src, ds, missing_ok=True)
        out = os.path.join(dst, "ind.%s.npz" % ds)
        planetoid.save_npz(out, parts)
        print(ds, sorted(parts.keys()), os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
