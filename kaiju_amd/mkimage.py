"""Write the device image of a Kaiju index: ``python -m kaiju_amd.mkimage db.fmi db.kjimg``.

The image holds the arrays of the HBM layout (rank blocks, SA sample, taxon tables, k-mer table) as packed
by ``kaiju_gpu_index_load``; ``kaiju -f db.kjimg`` (or ``api.Index("db.kjimg")``) then uploads it without
parsing or packing the .fmi again.  Needs no GPU."""
import sys

from . import api


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 2:
        print(__doc__)
        return 2
    api.write_index_image(argv[0], argv[1])
    return 0


if __name__ == "__main__":
    sys.exit(main())
