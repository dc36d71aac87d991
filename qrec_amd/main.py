"""``python -m qrec_amd.main config/BPR.conf`` -- run a stock QRec conf file unchanged
(paths inside the conf are cwd-relative, as in the reference's main.py:51-58)."""
import sys
import time

from .QRec import QRec
from .util.config import ModelConf


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        print("usage: python -m qrec_amd.main <path/to/model.conf>")
        return 2
    from .dist import init_from_env
    init_from_env()          # one process per GPU under torch.distributed.run; nothing happens otherwise
    start = time.time()
    QRec(ModelConf(argv[0])).execute()
    print("Running time: %f s" % (time.time() - start))
    return 0


if __name__ == "__main__":
    sys.exit(main())
